#!/bin/bash
# tools/.v3/plain/: a copy of this tree built with -DBAMM_PLAIN_LDS -- every hand-issued LDS instruction of the sequence
# kernels as the plain HIP statement it stands for (csrc/device_utils.h) -- to run the parity tests on when a toolchain change
# makes the hand-written part suspect:
#   here:     bash tools/plain_lds_build.sh                      (~6 min; the tree's own build is not touched)
#   GPU box:  cd tools/.v3/plain && python3 -m pytest tests -m gpu -x -q
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/tools/.v3/plain
rm -rf $dst && mkdir -p $dst
(cd $root && tar cf - --exclude=./tools --exclude=./.git --exclude=./gpurun_out --exclude=./profiles --exclude='*.o' --exclude='*.so' --exclude=./bammmotif2_amd/build --exclude=__pycache__ .) | tar xf - -C $dst
cd $dst
python3 - <<'PY'
from bammmotif2_amd import build
build.FLAGS.append("-DBAMM_PLAIN_LDS")
import __graft_entry__ as g
g.build()                                                    # library (with the flag), host side, oracle
PY
echo "built $dst"
