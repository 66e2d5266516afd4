#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_fullsize_properties_gpu.py tests/test_parity_gpu.py -x -q > gpurun_out/r03_step4_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03_step4_tests.log
for pct in 45 30 60; do
python - <<PY
import json, subprocess, sys
PY
done
timeout -k 10 600 python bench.py --len 500 --width 30 --order 4 --steps 12 --warmup 12 --no-cpu-baseline 2>>gpurun_out/r03_step4_bench.err | tee gpurun_out/r03_step4_c4.json | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('C4 ms_per_step', round(j['ms_per_step'],4), 'cold', j.get('ms_per_step_cold'), 'opt', j.get('ms_per_step_optimize_mode'))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_c4_prof -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --len 500 --width 30 --order 4 --steps 0 --warmup 0 --no-cpu-baseline > /dev/null 2>>$GRAFT_REPO_ROOT/gpurun_out/r03_step4_bench.err; cd $GRAFT_REPO_ROOT
find gpurun_out/r03_c4_prof -name "*kernel_stats*" | head -3
