#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof125; mkdir -p gpurun_out/prof125
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof125 -o t --output-format csv -- python3 bench.py --nseq 125000 --no-cpu-baseline --no-extras --steps 100 --warmup 20 > gpurun_out/prof125/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof125/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-200:]   # the timed region's tail
prev_end = None
dur = {}; gaps = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-30:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(name, []).append(e - s)
    if prev_end is not None: gaps.setdefault(name, []).append(s - prev_end)
    prev_end = e
for k in dur: print("%-32s n=%3d dur %.2f us   gap before it %.2f us" % (k, len(dur[k]), sum(dur[k])/len(dur[k])/1e3, sum(gaps.get(k,[0]))/max(1,len(gaps.get(k,[0])))/1e3))
period = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / (len(rows)/2) / 1e3
print("period per iteration %.2f us" % period)
PY
