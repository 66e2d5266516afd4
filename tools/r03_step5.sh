#!/bin/bash
# the model update spread over blocks: parity tests of the K >= 3 paths, then config 4's kernel table
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_cli_gpu.py tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_fullsize_properties_gpu.py tests/test_fused_update_gpu.py -x -q -m gpu > gpurun_out/r03_step5_tests.txt 2>&1; rc=$?; tail -5 gpurun_out/r03_step5_tests.txt
[ $rc -eq 0 ] || exit $rc

rm -rf /tmp/c4prof
rocprofv3 --kernel-trace --stats -d /tmp/c4prof -o c4 --output-format csv -- python3 bench.py --nseq 1000000 --len 500 --width 30 --order 4 --no-cpu-baseline --no-extras --steps 20 --warmup 4 > gpurun_out/r03_step5_c4.json 2> gpurun_out/r03_step5_c4.err; echo "c4 rc=$?"
f=$(find /tmp/c4prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03_step5_c4_kernel_stats.csv; head -8 "$f"
