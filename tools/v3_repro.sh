#!/bin/bash
# The 56 / 64-positions-per-lane classes with the fused-update prologue compiled in: round 3's wrong counts.
# Every argument is a variant: a comma-separated list of OBJECT=REPLACEMENT (objects of bammmotif2_amd/build, replacements
# under tools/.v3/), linked into the library in place of the tree's, probed with tools/v3_probe.py, then the tree's library
# comes back.
#   here:    hipcc <build.py's flags> -DBAMM_FUSE_MAX_M=64 -c bammmotif2_amd/csrc/grouped_xl.hip -o tools/.v3/xl_fused.o
#            hipcc <build.py's flags> -DBAMM_FUSE_MAX_M=64 -c bammmotif2_amd/csrc/abi.cpp -o tools/.v3/abi_fused.o   (the planner then fuses those classes)
#   GPU box: gpurun -- 'bash tools/v3_repro.sh grouped_xl.o=xl_fused.o grouped_xl.o=xl_fused.o,abi.o=abi_fused.o'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B=bammmotif2_amd/build
mkdir -p gpurun_out
cp bammmotif2_amd/libbamm_em.so /tmp/orig.so
trap 'cp /tmp/orig.so bammmotif2_amd/libbamm_em.so' EXIT
echo "--- the tree's library" | tee gpurun_out/v3_repro.txt
timeout -k 10 200 python3 tools/v3_probe.py 2>&1 | tee -a gpurun_out/v3_repro.txt
for variant in "$@"; do
  objs=""
  for o in $B/*.o; do
    name=$(basename $o); use=$o
    for pair in ${variant//,/ }; do [ "${pair%%=*}" = "$name" ] && use=tools/.v3/${pair##*=}; done
    objs="$objs $use"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o bammmotif2_amd/libbamm_em.so || exit 1
  echo "--- $variant" | tee -a gpurun_out/v3_repro.txt
  timeout -k 10 200 python3 tools/v3_probe.py 2>&1 | tee -a gpurun_out/v3_repro.txt
done
