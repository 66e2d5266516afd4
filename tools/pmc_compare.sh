#!/bin/bash
# LDS / wait counters of the sequence kernel for several bench.py configurations side by side.
#   bash tools/pmc_compare.sh OUT.txt "flags of config 1" "flags of config 2" ...
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
out=$1; shift
rm -rf gpurun_out/pmcc; mkdir -p gpurun_out/pmcc
: > $out
c=0
for flags in "$@"; do
  c=$((c+1)); i=0
  for grp in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM" \
             "GRBM_GUI_ACTIVE TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp -d gpurun_out/pmcc/c${c}g$i -o pmc --output-format csv -- \
        python3 bench.py --no-cpu-baseline --no-extras --warmup 25 --steps 6 $flags > gpurun_out/pmcc/c${c}g$i.log 2>&1 || echo "group $i failed for: $flags" >> $out
  done
  python3 - "$flags" gpurun_out/pmcc/c${c}g* >> $out <<'PY'
import sys, glob, csv, os
from collections import defaultdict
flags = sys.argv[1]; acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== bench.py", flags)
for k, cs in acc.items():
    if not any(t in k for t in ("k_em_grp", "k_em_seq", "k_em_mix")): continue
    print("  ", k[:70])
    for c_, v in sorted(cs.items()):
        print("      %-26s %.4g  (%d dispatches)" % (c_, sum(v) / len(v), len(v)))
PY
done
cat $out
