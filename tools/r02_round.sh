#!/bin/bash
# round 2: everything under profiles/r02_* in one GPU call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
bash tools/profile_round.sh r02 > gpurun_out/r02_profile.log 2>&1; echo "profile_round rc=$?"
bash tools/other_configs.sh > gpurun_out/r02_other_configs.txt 2>&1; echo "other configs rc=$?"
python3 tools/config5_run.py 200000 /tmp/c5 > gpurun_out/r02_config5_cli.txt 2>&1; echo "config5 rc=$?"
python3 tools/config5_run.py 200000 /tmp/c5b --deviceList 0,0 > gpurun_out/r02_config5_cli_two_contexts.txt 2>&1; echo "config5 two contexts rc=$?"
python3 tools/config5_run.py 1000000 /tmp/c3 em > gpurun_out/r02_config3_cli.txt 2>&1; echo "config3 cli rc=$?"
timeout -k 10 600 python3 -m tests.fuzz_parity --n 1500 --seed 21 > gpurun_out/r02_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r02_fuzz_parity.txt
python3 -m tests.deviation_report > gpurun_out/r02_deviation_vs_fp64.txt 2>&1; echo "deviation rc=$?"
bash tools/r02_trace125k.sh > gpurun_out/r02_fixed_cost_trace.txt 2>&1
for n in 125000 50000; do python3 bench.py --nseq $n --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null; done > gpurun_out/r02_shard_sizes.jsonl
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --force-dist 2>/dev/null >> gpurun_out/r02_shard_sizes.jsonl
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --force-dist --torch-allreduce 2>/dev/null >> gpurun_out/r02_shard_sizes.jsonl
