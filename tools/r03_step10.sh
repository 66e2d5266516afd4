#!/bin/bash
# the CLI with the negative set sampled beside the main run: its tests, then config 5 twice
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout -k 10 900 python3 -m pytest tests/test_cli_gpu.py tests/test_integration_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do python3 tools/config5_run.py 200000 /tmp/c5_$i > gpurun_out/r03_step10_c5_$i.txt 2>&1; sed -n 2p gpurun_out/r03_step10_c5_$i.txt; grep "Runtime:" gpurun_out/r03_step10_c5_$i.txt; done
grep -n "negative\|EM (\|fold" gpurun_out/r03_step10_c5_2.txt | head
python3 tools/config5_run.py 200000 /tmp/c5b --deviceList 0,0,0,0,0,0,0,0 > gpurun_out/r03_step10_c5_eight.txt 2>&1; sed -n 2p gpurun_out/r03_step10_c5_eight.txt
