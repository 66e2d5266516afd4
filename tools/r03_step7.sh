#!/bin/bash
# scratch blocks kept per context: config 4's from-seed figure (a second handle on the context), config 5 through the CLI
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline --order 4 --len 500 --width 30 --steps 12 --warmup 12 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); print("c4", j["ms_per_step"], j["from_seed"])'
python3 tools/c4_cold_passes.py > gpurun_out/r03_step7_c4_cold.txt 2>&1; cat gpurun_out/r03_step7_c4_cold.txt
python3 tools/config5_run.py 200000 /tmp/c5 > gpurun_out/r03_step7_c5.txt 2>&1; head -3 gpurun_out/r03_step7_c5.txt; grep -n "Runtime:" gpurun_out/r03_step7_c5.txt
