#!/bin/bash
# round 5: everything under profiles/r05_* in three GPU calls (each within gpurun's 1200 s):
#   BAMM_COMMIT=<hash> bash tools/r05_round.sh a|b|c
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
case "$1" in
a)  # the judged files: bench line, rocprofv3 kernel stats of the same command, PMC passes + LDS ceilings, config 4
  bash tools/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1; echo "profile_round rc=$?"; tail -3 gpurun_out/r05_profile.log
  ;;
b)
  bash tools/other_configs.sh > gpurun_out/r05_other_configs.txt 2>&1; echo "other configs rc=$?"
  python3 tools/c4_cold_passes.py > gpurun_out/r05_c4_cold_passes.txt 2>&1; echo "c4 cold rc=$?"
  python3 tools/pass_times.py 1000000 120 > gpurun_out/r05_pass_times.txt 2>&1; echo "pass times rc=$?"
  for m in plain getr fused optimize; do python3 tools/first_passes.py $m 1000000 8; done > gpurun_out/r05_first_passes.txt 2>&1; echo "first passes rc=$?"
  rm -f gpurun_out/r05_shard_sizes.jsonl
  for n in 1000000 125000 50000; do
    for f in "" "--no-fused-update"; do
      python3 bench.py --nseq $n --no-cpu-baseline --no-extras --steps 200 --warmup 20 $f 2>/dev/null >> gpurun_out/r05_shard_sizes.jsonl
    done
  done
  python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --force-dist 2>/dev/null >> gpurun_out/r05_shard_sizes.jsonl
  python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null >> gpurun_out/r05_shard_sizes.jsonl
  for sh in "--order 1" "--ss"; do python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 $sh 2>/dev/null >> gpurun_out/r05_shard_sizes.jsonl; done
  python3 bench.py --nseq 250000 --no-cpu-baseline --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null > gpurun_out/r05_two_local_ranks_bench.json
  python3 bench.py --nseq 250000 --order 1 --no-cpu-baseline --steps 200 --warmup 20 --gpus 2 --local-ranks 2>/dev/null > gpurun_out/r05_two_local_ranks_k1_bench.json
  python3 - <<'PY'
import json
for l in open("gpurun_out/r05_shard_sizes.jsonl"):
    j = json.loads(l)
    print(j["config"]["workload"][:48], j["n_gpus"], "ms_per_step %.4f" % j["ms_per_step"], "kernel %.4f" % j["roofline"]["avg_kernel_ms"], j["allreduce"][:40], j["launcher"])
PY
  ;;
c)
  python3 tools/config5_run.py 200000 /tmp/c5b --deviceList 0,0,0,0,0,0,0,0 > gpurun_out/r05_config5_cli_eight_contexts.txt 2>&1; echo "config5 eight contexts rc=$?"
  for i in 1 2 3; do python3 tools/config5_run.py 1000000 /tmp/c3 em; done > gpurun_out/r05_config3_cli.txt 2>&1
  for i in 1 2 3; do python3 tools/config5_run.py 200000 /tmp/c5; done > gpurun_out/r05_config5_cli.txt 2>&1
  python3 tools/prep_time.py 1000000 > gpurun_out/r05_prep_time.txt 2>&1; echo "prep time rc=$?"
  timeout -k 10 500 python3 -m tests.fuzz_parity --n 1500 --seed 51 > gpurun_out/r05_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r05_fuzz_parity.txt
  python3 -m tests.deviation_report > gpurun_out/r05_deviation_vs_fp64.txt 2>&1; echo "deviation rc=$?"
  # the driver's launch form rehearsed on ONE device: two and four PROCESSES, the library's shared-memory communicator, the
  # communicator self-test in front of the timed region, topology, attribution, the in-kernel all-reduce through hipIpc handles
  HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 \
      --dist-backend gloo --shm-comm --nseq 250000 --blocks 112 --no-cpu-baseline 2> gpurun_out/r05_torchrun_two.err > gpurun_out/r05_torchrun_two_processes_one_device.json; echo "torchrun 2 rc=$?"
  HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 4 \
      --dist-backend gloo --shm-comm --nseq 200000 --blocks 56 --no-cpu-baseline 2> gpurun_out/r05_torchrun_four.err > gpurun_out/r05_torchrun_four_processes_one_device.json; echo "torchrun 4 rc=$?"
  timeout -k 10 400 python3 tools/phase_clock.py > gpurun_out/r05_phase_clock.txt 2>&1; echo "phase clock rc=$?"
  ;;
*) echo "usage: r05_round.sh a|b|c"; exit 2;;
esac
