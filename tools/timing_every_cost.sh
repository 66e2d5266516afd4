#!/bin/bash
# What the HIP events around every pass cost by shard size: `bench.py --timing-every 1` (every pass bracketed: the default)
# against 8 (every eighth) and 0 (none), alternating on one box.
#   GPU box: gpurun -- 'bash tools/timing_every_cost.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
one() { python3 bench.py --no-cpu-baseline --no-extras "${@:2}" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step']*1e3,2), 'us per step; avg kernel', round((j['roofline'].get('avg_kernel_ms') or 0)*1e3,2))"; }
for rep in 1 2; do
  for n in 1000000 125000 50000; do
    for every in 1 8 0; do
      one "nseq $n timing-every $every" --nseq $n --steps 200 --warmup 20 --timing-every $every
    done
  done
done
