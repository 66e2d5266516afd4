#!/bin/bash
# full GPU suite + the per-iteration cost at three set sizes, fused update on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_step2_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03_step2_tests.log
tail -5 gpurun_out/r03_step2_tests.log
rm -f gpurun_out/r03_step2_bench.jsonl
for n in 1000000 125000 50000; do
  for f in "" "--no-fused-update"; do
    timeout -k 10 300 python bench.py --nseq $n --steps 200 --warmup 20 --no-cpu-baseline --no-extras $f 2>>gpurun_out/r03_step2_bench.err | tee -a gpurun_out/r03_step2_bench.jsonl | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$n', '$f', 'ms_per_step', round(j['ms_per_step'],4), 'kernel_ms', round(j['roofline']['avg_kernel_ms'],4))"
  done
done
