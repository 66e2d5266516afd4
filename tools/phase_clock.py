#!/usr/bin/env python3
"""Where a launch of k_em_mix spends its fixed cost: rebuilds the library with -DBAMM_PHASE_CLOCK (thread 0 of every block
leaves the 100 MHz wall clock at its phase boundaries), runs the bench model at several sizes and prints, averaged over
the blocks of the LAST launch, microseconds from the first block's entry.  Run it on the GPU box only (the instrumented
library replaces the tree's on that box; nothing is committed).
    python tools/phase_clock.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bammmotif2_amd import build
import atexit
build.FLAGS.append("-DBAMM_PHASE_CLOCK")


def _restore():
    """Leave the tree as it was found: the instrumented library stores to global memory at every phase boundary, and
    whatever runs next on this box must not time it (build.py also keeps the flags a library was built with and treats
    another set as stale)."""
    build.FLAGS.remove("-DBAMM_PHASE_CLOCK")
    build.build_library()


atexit.register(_restore)
t0 = time.time(); build.build_library(force=True); print(f"instrumented build: {time.time() - t0:.0f} s", flush=True)
import bammmotif2_amd as bm
from bammmotif2_amd import synth

lib = C.CDLL(build.LIB)
L0, W, K = 200, 20, 2
pwm = synth.make_pwm(W, 1234)
ctx = bm.Context(0)
names = {0: "entry", 8: "  update: loads landed (thread 0)", 9: "  update: barrier A", 10: "  update: lower orders", 11: "  update: barrier B", 12: "  update: chains",
         1: "update done", 2: "tables built", 3: "wave 0 done", 4: "block done", 7: "  bins zeroed", 5: "log folded", 13: "  sums in the bins", 14: "  barrier", 6: "marginalised + atomics issued"}
for N in (4096, 125000):
    codes, off = synth.make_sequences(N, L0, pwm, 1234)
    pk = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    vbg = pk.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
    seqs = bm.SeqSet(ctx, pk)
    ctx.set_tuning(group_layout=8)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, max_iterations=100)
    ctx.set_tuning(group_layout=-1)
    em.iterate(30); ctx.sync()
    t0 = time.perf_counter(); em.iterate(50); ctx.sync(); dt = (time.perf_counter() - t0) / 50 * 1e6
    buf = np.zeros((256, 16), np.uint64)
    assert lib.bamm_debug_phase_clock(buf.ctypes.data_as(C.c_void_p)) == 0
    nb = min(256, max(1, (N + 15) // 16))
    t = buf[:nb].astype(np.int64)
    first = t[:, 0].min()
    us = (t - first) / 100.0
    print(f"N={N}: {dt:.1f} us per pass (instrumented); blocks {nb}; last block's entry at {us[:, 0].max():.2f} us")
    for i, nm in names.items():
        print(f"   {nm:24s} mean {us[:, i].mean():8.2f}  min {us[:, i].min():8.2f}  max {us[:, i].max():8.2f}")
    em.close(); seqs.close()
