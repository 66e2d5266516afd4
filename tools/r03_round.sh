#!/bin/bash
# round 3: everything under profiles/r03_* in one GPU call (BAMM_COMMIT=<hash> bash tools/r03_round.sh)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
bash tools/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1; echo "profile_round rc=$?"
bash tools/other_configs.sh > gpurun_out/r03_other_configs.txt 2>&1; echo "other configs rc=$?"
python3 tools/c4_cold_passes.py > gpurun_out/r03_c4_cold_passes.txt 2>&1; echo "c4 cold rc=$?"
python3 tools/config5_run.py 200000 /tmp/c5 > gpurun_out/r03_config5_cli.txt 2>&1; echo "config5 rc=$?"
python3 tools/config5_run.py 200000 /tmp/c5b --deviceList 0,0,0,0,0,0,0,0 > gpurun_out/r03_config5_cli_eight_contexts.txt 2>&1; echo "config5 eight contexts rc=$?"
python3 tools/config5_run.py 1000000 /tmp/c3 em > gpurun_out/r03_config3_cli.txt 2>&1; echo "config3 cli rc=$?"
timeout -k 10 900 python3 -m tests.fuzz_parity --n 1500 --seed 31 > gpurun_out/r03_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r03_fuzz_parity.txt
python3 -m tests.deviation_report > gpurun_out/r03_deviation_vs_fp64.txt 2>&1; echo "deviation rc=$?"
python3 tools/pass_times.py 1000000 120 > gpurun_out/r03_pass_times.txt 2>&1; echo "pass times rc=$?"
# the per-iteration cost at an eighth of the set (one GPU's shard of 8), with the model update fused into the next
# pass's kernel and as a launch of its own, without a collective and with the library's RCCL call on a 1-rank communicator
rm -f gpurun_out/r03_shard_sizes.jsonl
for n in 1000000 125000 50000; do
  for f in "" "--no-fused-update"; do
    python3 bench.py --nseq $n --no-cpu-baseline --no-extras --steps 200 --warmup 20 $f 2>/dev/null >> gpurun_out/r03_shard_sizes.jsonl
  done
done
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --force-dist 2>/dev/null >> gpurun_out/r03_shard_sizes.jsonl
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null >> gpurun_out/r03_shard_sizes.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r03_shard_sizes.jsonl"):
    j = json.loads(l)
    print(j["config"]["n_seqs"], j["n_gpus"], "ms_per_step %.4f" % j["ms_per_step"], "kernel %.4f" % j["roofline"]["avg_kernel_ms"], j["allreduce"][:40], j["launcher"])
PY
