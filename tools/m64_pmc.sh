#!/bin/bash
# the 64-positions-per-lane class (200k x 2000 bp): instruction-cache and scalar-cache counters of the sequence kernel, for the
# tree's library and for another commit's under tools/.v3/old (see tools/README.md)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
for side in new old; do
  dir=$ROOT; [ $side = old ] && dir=$ROOT/tools/.v3/old
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH"; do
    rm -rf /tmp/pm; (cd $dir && timeout -k 10 200 rocprofv3 --pmc $grp -d /tmp/pm -o p --output-format csv -- python3 $dir/bench.py --no-cpu-baseline --no-extras --nseq 200000 --len 2000 --steps 3 --warmup 2 > /dev/null 2>&1)
    python3 - "$side" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_em_grp" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
  done
done
