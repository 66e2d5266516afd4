#!/bin/bash
# A/B on ONE box, the tree against tools/.v3/old/ (tools/ab_old.sh): the column-sliced path (config 4) and the per-column kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
one() { (cd $1 && python3 bench.py --no-cpu-baseline --no-extras "${@:3}" 2>/dev/null) | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$2', round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_ms'],4))"; }
for rep in 1 2; do
  for side in new old; do
    dir=$ROOT; [ $side = old ] && dir=$ROOT/tools/.v3/old
    one $dir "$side config 4 (k=4 W=30 1Mx500)   " --order 4 --len 500 --width 30 --steps 12 --warmup 12
    one $dir "$side k=3 W=30 1Mx200 (k_em_seq)   " --order 3 --width 30 --steps 20 --warmup 10
    one $dir "$side k=5 W=12 200kx200 (sliced)   " --order 5 --width 12 --nseq 200000 --steps 12 --warmup 12
  done
done
