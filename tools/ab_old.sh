#!/bin/bash
# A/B on ONE box (boxes of the pool differ by +-3 %) without building there: `tools/.v3/old/` holds bench.py, the Python
# package and the library of another commit (built here in a git worktree: see tools/README.md), the tree holds the new
# ones; the two benches alternate.
#   GPU box: gpurun -- 'bash tools/ab_old.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
one() { (cd $1 && python3 bench.py --no-cpu-baseline --no-extras "${@:3}" 2>/dev/null) | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$2', round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_ms'],4))"; }
for rep in 1 2; do
  for side in new old; do
    dir=$ROOT; [ $side = old ] && dir=$ROOT/tools/.v3/old
    one $dir "$side 1M passes 6-25   " --nseq 1000000 --steps 20 --warmup 5
    one $dir "$side 1M passes 21-220 " --nseq 1000000 --steps 200 --warmup 20
    one $dir "$side 125k             " --nseq 125000 --steps 200 --warmup 20
    one $dir "$side 50k              " --nseq 50000 --steps 300 --warmup 30
    one $dir "$side k=1 1M           " --nseq 1000000 --order 1 --steps 100 --warmup 20
  done
done
