#!/bin/bash
# round 4: everything under profiles/r04_* in one GPU call (BAMM_COMMIT=<hash> bash tools/r04_round.sh)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1; echo "profile_round rc=$?"
bash tools/other_configs.sh > gpurun_out/r04_other_configs.txt 2>&1; echo "other configs rc=$?"
python3 tools/c4_cold_passes.py > gpurun_out/r04_c4_cold_passes.txt 2>&1; echo "c4 cold rc=$?"
python3 tools/config5_run.py 200000 /tmp/c5b --deviceList 0,0,0,0,0,0,0,0 > gpurun_out/r04_config5_cli_eight_contexts.txt 2>&1; echo "config5 eight contexts rc=$?"
timeout -k 10 900 python3 -m tests.fuzz_parity --n 1500 --seed 41 > gpurun_out/r04_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r04_fuzz_parity.txt
python3 -m tests.deviation_report > gpurun_out/r04_deviation_vs_fp64.txt 2>&1; echo "deviation rc=$?"
python3 tools/pass_times.py 1000000 120 > gpurun_out/r04_pass_times.txt 2>&1; echo "pass times rc=$?"
# the per-iteration cost at an eighth of the set (one GPU's shard of 8), with the model update fused into the next
# pass's kernel and as a launch of its own, without a collective and with the library's RCCL call on a 1-rank communicator
rm -f gpurun_out/r04_shard_sizes.jsonl
for n in 1000000 125000 50000; do
  for f in "" "--no-fused-update"; do
    python3 bench.py --nseq $n --no-cpu-baseline --no-extras --steps 200 --warmup 20 $f 2>/dev/null >> gpurun_out/r04_shard_sizes.jsonl
  done
done
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --force-dist 2>/dev/null >> gpurun_out/r04_shard_sizes.jsonl
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null >> gpurun_out/r04_shard_sizes.jsonl
# two ranks on the one device with the host-staged collective and, as the extra, the in-kernel all-reduce (its first and only
# rehearsal before the driver's multi-GPU run): the whole line, attribution and ms_per_step_peer_allreduce included
python3 bench.py --nseq 250000 --no-cpu-baseline --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null > gpurun_out/r04_two_local_ranks_bench.json
python3 tools/prep_time.py 1000000 > gpurun_out/r04_prep_time.txt 2>&1; echo "prep time rc=$?"
for i in 1 2 3; do python3 tools/config5_run.py 1000000 /tmp/c3 em; done > gpurun_out/r04_config3_cli.txt 2>&1
for i in 1 2 3; do python3 tools/config5_run.py 200000 /tmp/c5; done > gpurun_out/r04_config5_cli.txt 2>&1
python3 tools/config5_run.py 200000 /tmp/c5h --hostSampler --hostPacking > gpurun_out/r04_config5_cli_host_paths.txt 2>&1
timeout -k 5 60 tools/ipc_probe 2000 > gpurun_out/r04_ipc_probe.txt 2>&1; echo "ipc probe rc=$?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_shard_sizes.jsonl"):
    j = json.loads(l)
    print(j["config"]["n_seqs"], j["n_gpus"], "ms_per_step %.4f" % j["ms_per_step"], "kernel %.4f" % j["roofline"]["avg_kernel_ms"], j["allreduce"][:40], j["launcher"])
PY
