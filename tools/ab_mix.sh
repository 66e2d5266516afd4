#!/bin/bash
# A/B on ONE box: bench with the tree's kernels, then with the files under tools/.old/ (rebuilt on the box)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { for i in 1 2; do python3 bench.py --nseq 1000000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_ms'],4))"; done; python3 bench.py --nseq 125000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 125k', round(j['ms_per_step'],4))"; python3 bench.py --nseq 1000000 --order 1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 k=1', round(j['ms_per_step'],4))"; }
timeout 600 python3 -m pytest tests/test_grouped_gpu.py tests/test_fuzz_gpu.py tests/test_fused_update_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
run new
mkdir -p /tmp/new; for f in abi.cpp grouped_kernel.h mixed_kernel.h common.h; do cp bammmotif2_amd/csrc/$f /tmp/new/$f; cp tools/.old/$f bammmotif2_amd/csrc/$f; done
python3 -c "from bammmotif2_amd import build as b; b.build_library()" > /dev/null 2>&1
run old
for f in abi.cpp grouped_kernel.h mixed_kernel.h common.h; do cp /tmp/new/$f bammmotif2_amd/csrc/$f; done
