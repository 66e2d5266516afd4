#!/bin/bash
# The other BASELINE.json configurations on one GPU (bench.py flags), one summary line each.
# Usage (GPU box): bash tools/other_configs.sh > gpurun_out/r01_other_configs.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
  echo "cfg: $*"
  python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read())
print("%s | %.3f ms/iter | %.1f it/s | %.2e pos/s | %s" % (d["config"]["workload"], d["ms_per_step"], d["iterations_per_s"], d["value"], d["roofline"]["kernel"]))'
}
run --nseq 50000
run --nseq 1000000 --ss
run --nseq 200000
run --nseq 1000000 --len 500 --width 30 --order 4 --steps 12 --warmup 12
run --nseq 1000000 --order 1
run --nseq 1000000 --order 0
run --nseq 1000000 --order 3
run --nseq 1000000 --len 500
run --nseq 300000 --len 750 --steps 30 --warmup 10
run --nseq 300000 --len 1000 --steps 30 --warmup 10
run --nseq 4000000 --len 40 --width 12 --steps 30 --warmup 10
run --nseq 2000000 --len 90 --steps 30 --warmup 10
run --nseq 200000 --len 1250 --steps 20 --warmup 10
run --nseq 200000 --len 1500 --steps 20 --warmup 10
run --nseq 200000 --len 1500 --order 3 --steps 20 --warmup 10
run --nseq 200000 --len 2000 --steps 20 --warmup 10
