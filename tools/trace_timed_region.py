#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `python bench.py ...`: the sequence kernel's launches in dispatch order, and the
mean duration over the launches of the TIMED call (the last `steps` launches of the kernel the first call of which was the
warm-up) -- the figure to hold against the bench line's `avg_kernel_ms`, which is one HIP event pair around the same launches
divided by their number (so it additionally holds the launch boundaries between them).

    trace_timed_region.py KERNEL_TRACE.csv KERNEL_SUBSTRING STEPS"""
import csv, sys
path, needle, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for r in csv.DictReader(open(path)):
    name = r.get("Kernel_Name") or r.get("kernel_name") or ""
    if needle in name:
        rows.append((int(r.get("Start_Timestamp") or r.get("start_timestamp")), int(r.get("End_Timestamp") or r.get("end_timestamp"))))
rows.sort()
d = [(e - s) / 1e3 for s, e in rows]
timed = rows[-steps:]
dt = [(e - s) / 1e3 for s, e in timed]
span = (timed[-1][1] - timed[0][0]) / 1e3
print(f"{len(rows)} launches of *{needle}*: all {sum(d) / len(d):.2f} us mean; the timed call's {len(timed)} (launches {len(rows) - steps + 1}-{len(rows)}): "
      f"{sum(dt) / len(dt):.2f} us mean, {min(dt):.2f} min, {max(dt):.2f} max; first start to last end {span / len(timed):.2f} us per launch "
      f"(= what one HIP event pair around them divides out to)")
