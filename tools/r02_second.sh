#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02c_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02c_gpu_tests.log
python3 bench.py --no-cpu-baseline > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; echo "bench rc=$?"
python3 bench.py --nseq 125000 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r02c_bench125k.json 2>> gpurun_out/r02c_bench.err
python3 bench.py --nseq 125000 --no-cpu-baseline --steps 200 --warmup 20 --force-dist > gpurun_out/r02c_bench125k_native.json 2>> gpurun_out/r02c_bench.err
python3 bench.py --nseq 125000 --no-cpu-baseline --steps 200 --warmup 20 --force-dist --torch-allreduce > gpurun_out/r02c_bench125k_torch.json 2>> gpurun_out/r02c_bench.err
tools/lds_mix_bench > gpurun_out/r02_lds_mix_bench.txt 2>&1
tail -5 gpurun_out/r02c_bench.err
