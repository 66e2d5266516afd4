#!/bin/bash
# Steady-state PMC passes over the bench workload (separate rocprofv3 --pmc runs per counter group,
# no tracing flags).  Usage (on the GPU box): bash tools/pmc_run.sh [extra bench.py flags]
# PMC_SCRIPT="tools/time_e_only.py" profiles another driver script instead of bench.py.
# Results: gpurun_out/pmc/<group>/..., summary gpurun_out/hbm_traffic.json (tools/summarize_pmc.py).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
# TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots) -> separate passes
for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d gpurun_out/pmc/g$i -o pmc --output-format csv -- \
      python3 ${PMC_SCRIPT:-bench.py --no-cpu-baseline --warmup 25 --steps 6} "$@" > gpurun_out/pmc/g$i.log 2>&1
  echo "pmc group $i done"
done
python3 tools/summarize_pmc.py gpurun_out/pmc 401000000
