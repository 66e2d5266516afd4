#!/bin/bash
# One GPU call that refreshes everything under profiles/ for the bench workload:
#   bench line, rocprofv3 kernel stats of the same command, PMC passes (tools/pmc_run.sh).
# Usage (GPU box): bash tools/profile_round.sh rNN
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"; cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline > gpurun_out/prof/stats.log 2>&1
cp "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
echo "stats done"; head -5 gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf gpurun_out/pmc
bash tools/pmc_run.sh
