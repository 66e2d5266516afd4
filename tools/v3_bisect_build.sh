#!/bin/bash
# tools/.v3/bis_<commit>/: the package + library of <commit> with the fused-update prologue compiled into (and planned for)
# every length class of k_em_grp -- for tools/v3_probe.py on the GPU box:
#   BAMM_PROBE_PACKAGE=$PWD/tools/.v3/bis_<commit> python tools/v3_probe.py
set -e
c=$1
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/bis_$c
git -C $root worktree add -f /tmp/bis_$c $c -q
cd /tmp/bis_$c
sed -i 's/if constexpr (ACCUM \&\& !FIXG \&\& M <= 16) {/if constexpr (ACCUM \&\& !FIXG \&\& M <= 64) {/' bammmotif2_amd/csrc/grouped_kernel.h
sed -i 's/kMClasses\[em->ebuckets\[i\].mclass\] <= 16 \&\&/kMClasses[em->ebuckets[i].mclass] <= 64 \&\&/' bammmotif2_amd/csrc/abi.cpp
grep -c "M <= 64" bammmotif2_amd/csrc/grouped_kernel.h
python3 -c "
from bammmotif2_amd import build
build.build_library(verbose=False)" 2>&1 | tail -2
mkdir -p $root/tools/.v3/bis_$c/bammmotif2_amd
cp bammmotif2_amd/*.py bammmotif2_amd/libbamm_em.so $root/tools/.v3/bis_$c/bammmotif2_amd/
echo "built bis_$c"
