#!/usr/bin/env python3
"""Round 3 left the 56 / 64-positions-per-lane classes of k_em_grp without the fused-update prologue because they
"produced wrong counts with the prologue compiled in, fused or not".  This probe localises such a failure: one pass of
the g_k2_m56_64 / g_k1_m64_ss shapes through whatever libbamm_em.so is in the tree (tools/v3_repro.sh links one with
-DBAMM_FUSE_MAX_M=64), compared against the fp64 restatement: which statistics, which cells (column j, y), and -- through
getR() against the oracle's r -- which sequences, lanes and positions per lane are off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("BAMM_PROBE_PACKAGE"):                   # another commit's package + library (tools/.v3/<name>/), this tree's oracle and cases
    sys.path.insert(0, os.environ["BAMM_PROBE_PACKAGE"])
import numpy as np
import bammmotif2_amd as bm
print("package:", os.path.dirname(bm.__file__))
import oracle
from tests.cases import Case

orc = oracle.Oracle(); orc.set_threads(1)
ctx = bm.Context(0)
bad_any = False
for spec in (dict(name="g_k2_m56_64", N=8, L0=1600, W=20, K=2, n_frac=0.0003, ragged=440),
             dict(name="g_k1_m64_ss", N=6, L0=3900, W=14, K=1, ss=True, ragged=190),
             dict(name="g_k2_m40_48", N=10, L0=1280, W=20, K=2, n_frac=0.0005, ragged=240)):
    c = Case(**spec)
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(ctx, pk)
    em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order)
    print(f"== {c.name}: plan (grouped, per-column, launches) = {em.plan()}")
    Kb = min(c.bg_order, c.K)
    em.EStep()
    s_o = orc.linear_s(c.v0, vbg, c.K, c.W, Kb)
    r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, c.q)
    r_g = em.getR()
    rel = np.abs(r_g - r_o) / np.maximum(np.abs(r_o), 1e-30)
    print(f"   getR (WRITE_R kernel) vs oracle: max rel {rel.max():.2e}; llh {em.getLLH():.6f} vs {llh_o:.6f}")
    em.MStep()
    n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
    n_g = em.getCounts()
    v64, *_ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    dv = np.abs(em.getV() - v64) / np.abs(v64)
    dn = np.abs(n_g - n_o) / np.maximum(np.abs(n_o), 1e-6)
    print(f"   counts (ACCUM kernel) vs oracle: max rel {dn.max():.2e};  v vs fp64: max rel {dv.max():.2e}")
    if dv.max() > 1e-5 or rel.max() > 1e-3:
        bad_any = True
        W = c.W
        top = (4 ** (c.K + 2) - 4) // 3 * W - 4 ** (c.K + 1) * W       # offset of the top order in the flat [k][y][j] layout
        nk_g, nk_o = n_g[top:].reshape(-1, W), n_o[top:].reshape(-1, W)
        d = np.abs(nk_g - nk_o)
        js = np.argsort(-d.sum(axis=0))[:6]
        print("   columns with the largest absolute count error:", [(int(j), float(d[:, j].sum())) for j in js])
        print(f"   total count, kernel {nk_g.sum():.4f} vs oracle {nk_o.sum():.4f}; per column (kernel - oracle):")
        print("   ", np.round(nk_g.sum(axis=0) - nk_o.sum(axis=0), 4))
        lens = np.diff(off.astype(np.int64))
        for n in range(c.N):
            seg = slice(int(off[n]), int(off[n + 1]))
            e = rel[seg]
            if e.max() > 1e-3:
                L = int(lens[n]); M = -(-L // 64)
                badp = (L - 1 - np.nonzero(e > 1e-3)[0])          # r[L-1-p] holds slot p
                print(f"   seq {n}: L={L} (M={M}) {len(badp)} bad r slots; lanes {sorted(set((badp // M).tolist()))[:20]} ; m in lane {sorted(set((badp % M).tolist()))[:20]}")
    em.close(); ss.close()
    # the fused update TAKEN (only where the library was built with -DBAMM_FUSE_MAX_M=64, abi.cpp included: the planner
    # then lets these classes carry it) against the update as a launch of its own: same bits or not?
    res = []
    for fused in (1, 0):
        ctx.set_tuning(fused_update=fused)
        ss = bm.SeqSet(ctx, pk)
        em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, max_iterations=8)
        em.iterate(4)
        res.append((em.getV(), em.trace()[0].copy()))
        em.close(); ss.close()
    ctx.set_tuning(fused_update=1)
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    dvf = np.abs(res[0][0] - res[1][0]).max()
    print(f"   iterate(4) with fused_update=1 vs 0: {'identical' if same else 'DIFFERENT (max |dv| %.3e)' % dvf}")
    if not same:
        bad_any = True
print("v3_probe:", "WRONG RESULTS" if bad_any else "all fine")
