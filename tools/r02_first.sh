#!/bin/bash
# round 2, first GPU call: test-suite on the rebuilt library, bench line, 125k-sequence shard, atomic fold microbench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/atomic_bench > gpurun_out/r02_atomic_bench.txt 2>&1; echo "atomic bench rc=$?"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02a_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02a_gpu_tests.log
python3 bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"
python3 bench.py --nseq 125000 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r02a_bench125k.json 2>> gpurun_out/r02a_bench.err; echo "bench125k rc=$?"
