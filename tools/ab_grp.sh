#!/bin/bash
# A/B of the uniform-row kernels on ONE box (as tools/ab_mix.sh: `bash tools/ab_mix.sh prepare` first)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
one() { python3 bench.py "${@:2}" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_ms'],4))"; }
run() {
  one "$1 k=1 125k" --nseq 125000 --order 1 --steps 200 --warmup 20; one "$1 k=1 125k" --nseq 125000 --order 1 --steps 200 --warmup 20
  one "$1 k=0 125k" --nseq 125000 --order 0 --steps 200 --warmup 20
  one "$1 k=3 125k" --nseq 125000 --order 3 --steps 200 --warmup 20; one "$1 k=3 125k" --nseq 125000 --order 3 --steps 200 --warmup 20
  one "$1 ss  125k" --nseq 125000 --ss --steps 200 --warmup 20
  one "$1 k=2 W=12 125k" --nseq 125000 --width 12 --steps 200 --warmup 20
  one "$1 k=1 1M  " --nseq 1000000 --order 1 --steps 100 --warmup 20
}
timeout -k 10 600 python3 -m pytest tests/test_grouped_gpu.py tests/test_fuzz_gpu.py tests/test_partition_exact_gpu.py -x -q -m gpu 2>&1 | tail -3
run new
mkdir -p /tmp/new
# whatever ends the script (a timeout, Ctrl-C, a failed build): the tree's own files and library come back
restore() { for f in /tmp/new/*; do [ -f "$f" ] && cp "$f" bammmotif2_amd/csrc/$(basename "$f"); done
            python3 -c "from bammmotif2_amd import build as b; b.build_library()" > /dev/null 2>&1; }
trap restore EXIT
for f in tools/.old/*; do b=$(basename $f); cp bammmotif2_amd/csrc/$b /tmp/new/$b; cp $f bammmotif2_amd/csrc/$b; done
python3 -c "from bammmotif2_amd import build as b; b.build_library()" > /dev/null 2>&1
run old
