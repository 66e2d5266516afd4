#!/bin/bash
# One GPU call that refreshes the judged files for the bench workload (config 3) and for config 4:
#   bench line, rocprofv3 kernel stats of the same command, PMC passes + LDS ceiling (tools/pmc_run.sh).
# Usage (GPU box): BAMM_COMMIT=<hash> bash tools/profile_round.sh rNN     ->  gpurun_out/rNN_*  (copy into profiles/)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
# PMC first: bench.py picks the summary up from profiles/ (same commit) or gpurun_out/ (this call)
bash tools/pmc_run.sh gpurun_out/${tag}_hbm_traffic.json 401000000 2
cp gpurun_out/${tag}_hbm_traffic.json profiles/${tag}_hbm_traffic.json
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"; cat gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof/*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/prof/stats.log 2>&1
cp "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
echo "stats done"; head -5 gpurun_out/${tag}_bench_kernel_stats.csv
# the same trace, per dispatch: the mean over the launches of the timed call alone (what avg_kernel_ms covers)
python3 tools/trace_timed_region.py "$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)" k_em_ 40 > gpurun_out/${tag}_bench_kernel_trace_timed.txt 2>&1 || true
cat gpurun_out/${tag}_bench_kernel_trace_timed.txt
# config 4: 1M x 500 bp, W = 30, k = 4 (column-sliced path)
C4="--order 4 --len 500 --width 30 --steps 12 --warmup 12"
bash tools/pmc_run.sh gpurun_out/${tag}_c4_hbm_traffic.json 1001000000 4 $C4
cp gpurun_out/${tag}_c4_hbm_traffic.json profiles/${tag}_c4_hbm_traffic.json
python3 bench.py --no-cpu-baseline $C4 > gpurun_out/${tag}_c4_bench.json 2>> gpurun_out/${tag}_bench.err
echo "c4 bench done"; cat gpurun_out/${tag}_c4_bench.json
rm -rf gpurun_out/prof/*
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-extras $C4 > gpurun_out/prof/stats_c4.log 2>&1
cp "$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_c4_kernel_stats.csv
echo "c4 stats done"; head -6 gpurun_out/${tag}_c4_kernel_stats.csv
