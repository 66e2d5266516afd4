#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libbammref.so, i.e. the
reference's own EM / Motif / BackgroundModel / ScoreSeqSet translation units compiled in place
from /root/reference/src).  Runs only where that build exists (the development container):

    make -C oracle ref && python tests/golden/make_golden.py

Every expectation below is produced by reference code, single-threaded; inputs are stored too
(small cases) or regenerated from bammmotif2_amd.synth with a digest check (large cases).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tests.cases import SMALL_CASES, Case  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

LARGE = [
    dict(name="g4_1k", N=1000, L0=200, W=20, K=2, seed=1234),
    dict(name="g4_10k", N=10000, L0=200, W=20, K=2, seed=1234),
    dict(name="g5_k4", N=300, L0=500, W=30, K=4, seed=77, ss=True),
]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(R, c, store_inputs, n_iter=3, r_seqs=None, with_optimize=True):
    S = R.session(c.codes, c.in_off, c.ss, 42)
    out = dict(N=c.N, L0=c.L0, W=c.W, K=c.K, ss=int(c.ss), bg_order=c.bg_order, q=np.float32(c.q),
               alpha=c.alpha, alpha_bg=c.alpha_bg, A=c.A, v0=c.v0, pwm=c.pwm,
               codes_sha256=digest(c.codes), in_off_sha256=digest(c.in_off))
    if store_inputs:
        out.update(codes=c.codes, in_off=c.in_off)
    kmer = S.kmers()
    out["off"] = S.off
    out["kmer_sha256"] = digest(kmer)
    if store_inputs:
        out["seq"] = S.seq_codes()
        out["kmer"] = kmer
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    out["vbg"] = vbg
    Kb = min(c.bg_order, c.K)
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    e = S.em(m, bg, False, False)
    nr = S.N if r_seqs is None else min(r_seqs, S.N)
    rlen = int(S.off[nr])
    for it in range(n_iter):
        S.R.ref_em_estep(e)
        out[f"s_{it}"] = S.motif_s(m, c.K, c.W)
        out[f"r_{it}"] = S.em_r(e)[:rlen]
        out[f"llh_{it}"] = np.float32(S.R.ref_em_llh(e))
        S.R.ref_em_mstep(e)
        out[f"n_{it}"] = S.em_n(e, c.K, c.W)
        out[f"v_{it}"] = S.motif_v(m)
    out["r_seqs"] = nr
    S.R.ref_em_optimize_q(e)
    out["q_after_optimize_q"] = np.float32(S.R.ref_em_q(e))
    out["p_final"] = S.motif_p(m)
    S.R.ref_motif_log_s(m, bg, Kb)
    out["logs_final"] = S.motif_s(m, c.K, c.W)
    mops, zoops, z = S.logodds(m, bg, c.W)
    out["zoops"], out["z"] = zoops, z
    if store_inputs:
        out["mops"] = mops
    else:
        out["mops_sha256"] = digest(mops)
    ihbcp, ihbp = S.write_motif(m)
    hbcp, hbp = S.write_bg(bg)
    out["file_ihbcp"], out["file_ihbp"] = np.frombuffer(ihbcp, np.uint8), np.frombuffer(ihbp, np.uint8)
    out["file_hbcp"], out["file_hbp"] = np.frombuffer(hbcp, np.uint8), np.frombuffer(hbp, np.uint8)
    if with_optimize:
        for oq in (0, 1):
            m2 = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
            e2 = S.em(m2, bg, bool(oq), False)
            # iteration count: replay the stop rule next to the reference's own loop
            S.R.ref_em_optimize(e2)
            out[f"opt{oq}_v"] = S.motif_v(m2)
            out[f"opt{oq}_llh"] = np.float32(S.R.ref_em_llh(e2))
            out[f"opt{oq}_q"] = np.float32(S.R.ref_em_q(e2))
            out[f"opt{oq}_n"] = S.em_n(e2, c.K, c.W)
            # the reference does not expose its iteration count / per-pass trace; the C oracle is
            # bit-identical to it (asserted here), so its trace is the reference's trace
            O = oracle.Oracle()
            O.set_threads(1)
            res = O.optimize(kmer, S.off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=bool(oq))
            assert np.array_equal(res["v"], out[f"opt{oq}_v"]) and np.float32(res["llh"]) == out[f"opt{oq}_llh"]
            assert np.float32(res["q"]) == out[f"opt{oq}_q"]
            out[f"opt{oq}_iterations"] = res["iterations"]
            out[f"opt{oq}_trace_llh"] = res["trace_llh"]
            out[f"opt{oq}_trace_vdiff"] = res["trace_vdiff"]
    S.close()
    return out


def main():
    if not oracle.have_reference():
        raise SystemExit("oracle/_ref/libbammref.so missing: run `make -C oracle ref` in the dev container")
    R = oracle.Reference()
    R.set_threads(1)
    for spec in SMALL_CASES:
        c = Case(**spec)
        big = c.N * c.L0 > 20000
        out = run_case(R, c, store_inputs=True, r_seqs=16 if big else None)
        if big:
            out["mops_sha256"] = digest(out.pop("mops"))
            out.pop("kmer"); out.pop("seq")          # recomputable: digests stay
        np.savez_compressed(os.path.join(HERE, f"small_{c.name}.npz"), **out)
        print("wrote", c.name)
    for spec in LARGE:
        c = Case(**spec)
        out = run_case(R, c, store_inputs=False, n_iter=3 if c.K <= 2 else 2, r_seqs=8,
                       with_optimize=(c.N <= 1000 and c.K <= 2))
        if c.K > 2:                                  # 41k-element tables: keep the last pass only
            for key in ("s_0", "n_0", "v_0", "p_final", "logs_final"):
                out.pop(key, None)
        np.savez_compressed(os.path.join(HERE, f"large_{c.name}.npz"), **out)
        print("wrote", c.name)


if __name__ == "__main__":
    main()
