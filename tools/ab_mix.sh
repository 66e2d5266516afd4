#!/bin/bash
# A/B on ONE box (boxes of the pool differ by +-3 %): the bench with the tree's kernels, then with the csrc files
# saved under tools/.old/ (rebuilt on the box), then the tree's files are put back.
#   here:     bash tools/ab_mix.sh prepare        # tools/.old/ <- the HEAD version of every csrc file the tree changed
#   GPU box:  gpurun -- 'bash tools/ab_mix.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
if [ "$1" = prepare ]; then
  rm -rf tools/.old; mkdir -p tools/.old
  for f in $(git diff --name-only HEAD -- bammmotif2_amd/csrc); do git show HEAD:$f > tools/.old/$(basename $f); echo "old: $f"; done
  exit 0
fi
one() { python3 bench.py "${@:2}" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_ms'],4))"; }
run() {
  one "$1 1M passes 21-220" --nseq 1000000 --steps 200 --warmup 20; one "$1 1M passes 21-220" --nseq 1000000 --steps 200 --warmup 20
  one "$1 1M passes 6-25  " --nseq 1000000 --steps 20 --warmup 5; one "$1 1M passes 6-25  " --nseq 1000000 --steps 20 --warmup 5
  one "$1 125k            " --nseq 125000 --steps 200 --warmup 20; one "$1 125k            " --nseq 125000 --steps 200 --warmup 20
  one "$1 50k             " --nseq 50000 --steps 300 --warmup 30; one "$1 50k             " --nseq 50000 --steps 300 --warmup 30
  one "$1 k=1             " --nseq 1000000 --order 1 --steps 100 --warmup 20
  one "$1 k=1 125k        " --nseq 125000 --order 1 --steps 200 --warmup 20; one "$1 k=1 125k        " --nseq 125000 --order 1 --steps 200 --warmup 20
  one "$1 --ss            " --nseq 1000000 --ss --steps 100 --warmup 20
  one "$1 --ss 125k       " --nseq 125000 --ss --steps 200 --warmup 20; one "$1 --ss 125k       " --nseq 125000 --ss --steps 200 --warmup 20
  one "$1 k=3             " --nseq 1000000 --order 3 --steps 100 --warmup 20
}
timeout -k 10 900 python3 -m pytest tests/test_grouped_gpu.py tests/test_fuzz_gpu.py tests/test_fused_update_gpu.py tests/test_golden_gpu.py tests/test_peer_allreduce_gpu.py tests/test_partition_exact_gpu.py -x -q -m gpu 2>&1 | tail -3
run new
mkdir -p /tmp/new
# whatever ends the script (a timeout, Ctrl-C, a failed build): the tree's own files and library come back
restore() { for f in /tmp/new/*; do [ -f "$f" ] && cp "$f" bammmotif2_amd/csrc/$(basename "$f"); done
            python3 -c "from bammmotif2_amd import build as b; b.build_library()" > /dev/null 2>&1; }
trap restore EXIT
for f in tools/.old/*; do b=$(basename $f); cp bammmotif2_amd/csrc/$b /tmp/new/$b; cp $f bammmotif2_amd/csrc/$b; done
python3 -c "from bammmotif2_amd import build as b; b.build_library()" > /dev/null 2>&1
run old
