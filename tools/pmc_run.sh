#!/bin/bash
# Steady-state PMC passes over a bench workload (separate rocprofv3 --pmc runs per counter group, no tracing
# flags), then the LDS ceiling for the fused kernel's mix, then the summary json.
#   bash tools/pmc_run.sh OUT.json POSITIONS ORDER [bench.py flags]
# e.g. bash tools/pmc_run.sh gpurun_out/r02_hbm_traffic.json 401000000 2
#      bash tools/pmc_run.sh gpurun_out/r02_c4_hbm_traffic.json 1001000000 4 --order 4 --len 500 --width 30
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
out=$1; positions=$2; order=$3; shift 3
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
i=0
# TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots) -> separate passes
for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d gpurun_out/pmc/g$i -o pmc --output-format csv -- \
      python3 bench.py --no-cpu-baseline --no-extras --warmup 25 --steps 6 "$@" > gpurun_out/pmc/g$i.log 2>&1
  echo "pmc group $i done"
done
# the same command once more under --kernel-trace --stats: the kernels' average durations over the SAME dispatches the
# counters were averaged over (rates per kernel = counter / duration)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc/stats -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-extras --warmup 25 --steps 6 "$@" > gpurun_out/pmc/stats.log 2>&1
stats=$(find gpurun_out/pmc/stats -name '*kernel_stats.csv' | head -1)
echo "kernel stats done: $stats"
tools/lds_mix_bench > gpurun_out/lds_mix_bench.txt 2>&1 || true
extra=""
if [ "$order" -ge 4 ]; then
  [ -x tools/lds_c4_bench ] || hipcc --offload-arch=gfx950 -O3 tools/lds_c4_bench.hip -o tools/lds_c4_bench
  tools/lds_c4_bench > gpurun_out/lds_c4_bench.txt 2>&1 || true
  extra="--lds-c4 gpurun_out/lds_c4_bench.txt"
fi
python3 tools/summarize_pmc.py gpurun_out/pmc $positions --order $order --out $out --lds-mix gpurun_out/lds_mix_bench.txt --kernel-stats "$stats" $extra
