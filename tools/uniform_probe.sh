# pass time of the uniform grouped kernel on a few shapes (bench.py flags), one line each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("%-60s %.3f ms/iter  kernel %.3f ms  %s" % (d["config"]["workload"][:60], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["kernel"][:9]))'; }
run --group-layout 3
run --order 1
run --order 0
run --order 3
run --ss
run --width 19
run --nseq 300000 --len 750 --steps 30 --warmup 10
run --nseq 125000 --steps 200 --warmup 20 --group-layout 3
