#!/bin/bash
# round 3, first GPU call: the fused model update (tests + what it buys per iteration at 1M and at a 125k shard)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fused_update_gpu.py tests/test_bench_multirank_gpu.py tests/test_comm_gpu.py tests/test_parity_gpu.py -x -q > gpurun_out/r03_step1_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03_step1_tests.log
tail -5 gpurun_out/r03_step1_tests.log
for n in 1000000 125000 50000; do
  for f in "" "--no-fused-update"; do
    timeout -k 10 300 python bench.py --nseq $n --steps 200 --warmup 20 --no-cpu-baseline --no-extras $f 2>>gpurun_out/r03_step1_bench.err | tee -a gpurun_out/r03_step1_bench.jsonl | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$n', '$f', 'ms_per_step', round(j['ms_per_step'],4), 'kernel_ms', round(j['roofline']['avg_kernel_ms'],4))"
  done
done
