#!/usr/bin/env python3
"""Runs tools/init_probe (see there) and adds the two intervals only the parent sees: spawn -> main, last line -> reaped."""
import os, re, subprocess, sys, time
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "init_probe")
for args in (["free", "2048"], ["leave", "2048"], ["reset", "2048"], ["free_dev", "2048"], ["free_host", "2048"], ["free_stream", "2048"], ["leave", "64"], ["free", "2048"], ["leave", "2048"], ["free_dev", "2048"], ["free_host", "2048"], ["free_stream", "2048"]):
    t = time.time()
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    t1 = time.time()
    ent = float(re.search(r"entered ([\d.]+)", r.stdout).group(1)); left = float(re.search(r"left ([\d.]+)", r.stdout).group(1))
    print("init_probe", " ".join(args), "MB: wall %.3f s; spawn -> main %.3f s; last line -> reaped %.3f s" % (t1 - t, ent - t, t1 - left))
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("  ")))
