#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6
python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/r02e_bench125k.json 2> gpurun_out/r02e.err
python3 -c "
import json; d=json.load(open('gpurun_out/r02e_bench125k.json')); print('125k ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'], d['llh_last'])"
bash tools/r02_trace125k.sh
