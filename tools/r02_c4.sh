#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_seed_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -15
C4="--order 4 --len 500 --width 30 --steps 12 --warmup 12 --no-cpu-baseline"
python3 bench.py $C4 > gpurun_out/r02d_c4_bench.json 2> gpurun_out/r02d_c4.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02d_c4_bench.json')); print('c4 ms/step', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), d['llh_last'])"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o stats --output-format csv -- python3 bench.py $C4 --no-extras > gpurun_out/prof/stats_c4.log 2>&1
cp "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/r02d_c4_kernel_stats.csv; head -6 gpurun_out/r02d_c4_kernel_stats.csv | cut -c1-200
