#!/usr/bin/env python3
"""Sequence::Sequence on the host (bamm_pack_codes_seeded, all granted cores) against the device (bamm_seqs_from_codes) on
the bench set: wall time of either, and that the arrays agree.  python tools/prep_time.py [n_seqs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bammmotif2_amd as bm
from bammmotif2_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
pwm = synth.make_pwm(20, 1234)
codes, off = synth.make_sequences(N, 200, pwm, 1234, 0.5)
ctx = bm.Context(0)
bm.abi.load().bamm_set_host_threads(len(os.sched_getaffinity(0)))
for rep in range(3):
    t0 = time.perf_counter(); host = bm.PackedSeqs.from_codes(codes, off, False, seed=42); t1 = time.perf_counter()
    sh = bm.SeqSet(ctx, host); t2 = time.perf_counter()
    vb_h = host.bg_model(2, np.array([1, 10, 10], np.float32)); t3 = time.perf_counter()
    tp = time.perf_counter(); only, _ = bm.SeqSet.from_codes(ctx, codes, off, False, seed=42, resident=False); tq = time.perf_counter()
    only.free()
    print(f"   device packing alone (upload of the codes, kernels, the draws on the host, the packed set brought back): {tq - tp:.3f} s", flush=True)
    t3 = time.perf_counter()
    dev, sd = bm.SeqSet.from_codes(ctx, codes, off, False, seed=42); t4 = time.perf_counter()
    vb_d = sd.bg_model(2, np.array([1, 10, 10], np.float32)); t5 = time.perf_counter()
    print(f"{N} x 200 bp ds  host: pack {t1 - t0:.3f} s + upload {t2 - t1:.3f} s + bg {t3 - t2:.3f} s = {t3 - t0:.3f} s   "
          f"device: pack + resident set {t4 - t3:.3f} s + bg {t5 - t4:.3f} s = {t5 - t3:.3f} s", flush=True)
    if rep == 0:
        a, b = host.arrays(), dev.arrays()
        print("   arrays equal:", all(np.array_equal(a[k], b[k]) for k in a), " bg equal:", np.array_equal(vb_h, vb_d), " exceptions:", len(a["exc_pos"]))
    sh.close(); sd.close(); host.free(); dev.free()
