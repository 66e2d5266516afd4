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
    dict(name="c4_k4_ds", N=160, L0=500, W=30, K=4, seed=78),          # BASELINE config 4's shape: both strands, L = 1001
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


def run_eval_case(R):
    """Evaluation side (config 5 / the Travis smoke line): negative sampler, 4-fold CV with EM per
    fold, PR / p-value statistics, window p-values and the .occurrence writer -- all reference code."""
    c = Case("eval", N=120, L0=50, W=8, K=1, seed=11, n_frac=0.01)
    S = R.session(c.codes, c.in_off, c.ss, 42)
    out = dict(codes=c.codes, in_off=c.in_off, W=c.W, K=c.K, q=np.float32(c.q), v0=c.v0, alpha=c.alpha,
               alpha_bg=c.alpha_bg, A=c.A)
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    out["vbg"] = vbg
    for tag, generic in (("neg", False), ("gneg", True)):
        neg = S.negset(2, 2, generic)
        out[tag + "_codes"] = neg.seq_codes()
        out[tag + "_off"] = neg.off
        if not generic:
            keep = neg
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    files, scores, q = S.fdr(keep, m, bg, 4, True, True, em=True, optimizeQ=False, threads=1)
    for name, data in files.items():
        out["fdr_file_" + name.replace(".", "_")] = np.frombuffer(data, np.uint8)
    for name, sc in zip(("pos_max", "neg_max", "pos_all", "neg_all"), scores):
        out["fdr_" + name] = sc
    out["fdr_q"] = np.float32(q)
    # occurrences: positives scored with a model refined by 3 EM passes, negatives = sampled set
    e = S.em(m, bg, False, False)
    for _ in range(3):
        S.R.ref_em_estep(e); S.R.ref_em_mstep(e)
    out["occ_v"] = S.motif_v(m)
    neg_mops, _, _ = keep.logodds(m, bg, c.W)
    pos_mops, _, _ = S.logodds(m, bg, c.W)
    out["occ_neg_mops"], out["occ_pos_mops"] = neg_mops, pos_mops
    data, pv = S.occurrence(m, bg, c.W, neg_mops, 0.02, c.ss)
    out["occ_file"] = np.frombuffer(data, np.uint8)
    out["occ_pvalues"] = pv
    np.savez_compressed(os.path.join(HERE, "eval_small.npz"), **out)
    print("wrote eval_small")


def run_travis_case(R):
    """The reference's CI smoke line (.travis.yml:21): JunD.fasta + PWM_peng10.meme, --EM -k 0 --FDR
    --scoreSeqset --maxPWM 1.  FASTA parsing and the PWM seeding go through our pinned restatements
    (the reference's reader needs Boost); everything from there on is reference code."""
    from tests.test_host_io_cpu import FASTA, MEME, read_fasta_py
    codes, off = read_fasta_py(FASTA)
    O = oracle.Oracle()
    O.set_threads(1)
    S = R.session(codes, off, False, 42)
    kmer = S.kmers()
    alpha_bg = np.array([1, 10, 10], np.float32)
    bg, vbg = S.bg(2, alpha_bg)
    lines = open(MEME).read().split("\n")
    i = [k for k, l in enumerate(lines) if "letter-probability matrix" in l][0]
    W, K = 12, 0
    pwm = np.array([[float(x) for x in lines[i + 1 + j].split()] for j in range(W)], np.float32).T.copy()
    alpha = np.array([1.0], np.float32)
    v0 = O.init_from_pwm(pwm, W, K, np.repeat(alpha, W), vbg, kmer, S.off, 0.3)
    neg = S.negset(2, 5000 // 300 + 1, False)                  # mainBaMM.cpp:100-106
    m = S.motif(W, K, alpha, bg, 0.3, v0)
    e = S.em(m, bg, False, False)
    S.R.ref_em_optimize(e)
    out = dict(v0=v0, v_final=S.motif_v(m), vbg=vbg, neg_n=neg.N, neg_len=neg.L[:4])
    neg_mops, _, _ = neg.logodds(m, bg, W)
    occ, _ = S.occurrence(m, bg, W, neg_mops, 1e-4, False, base="JunD_motif_1")
    out["occurrence"] = np.frombuffer(occ, np.uint8)
    m2 = S.motif(W, K, alpha, bg, 0.3, v0)
    files, scores, q = S.fdr(neg, m2, bg, 4, False, True, em=True, optimizeQ=False, threads=1, save_pvalues=False)
    out["zoops_stats"] = np.frombuffer(files["zoops.stats"], np.uint8)
    np.savez_compressed(os.path.join(HERE, "travis_jund.npz"), **out)
    print("wrote travis_jund", neg.N, len(occ), len(files["zoops.stats"]))


def run_config5_case(R):
    """BASELINE config 5's command line on a reduced set: `--EM -k 2 --FDR -n 5 -m 10`, 600 x 200 bp, both
    strands, W = 20.  The model is seeded through --BaMMFile (Motif::initFromBaMM is reference code here, the
    PWM seeder is not buildable without Boost); mainBaMM.cpp:100-106 raises mFold to 5000/600+1 = 9 for a
    set this small, exactly as the drop-in CLI does.  Everything below the FASTA reader is reference code."""
    import tempfile
    c = Case("config5", N=600, L0=200, W=20, K=2, seed=1234)
    S = R.session(c.codes, c.in_off, c.ss, 42)
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    m0 = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    seed_ihbcp, _ = S.write_motif(m0, base="seed")             # Motif::write: %.3e text
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "seed.ihbcp")
        open(path, "wb").write(seed_ihbcp)
        m = S.R.ref_motif_from_bamm_file(c.W, c.K, np.ascontiguousarray(c.alpha, np.float32), bg, c.q, path.encode())
    minSeqN, posN = 5000, c.N
    mfold = minSeqN // posN + (1 if minSeqN % posN else 0)     # mainBaMM.cpp:100-106 overrides -m 10
    neg = S.negset(2, mfold, False)
    files, scores, q = S.fdr(neg, m, bg, 5, False, True, em=True, optimizeQ=False, threads=1, save_pvalues=True,
                             save_logodds=True)
    out = dict(codes=c.codes, in_off=c.in_off, W=c.W, K=c.K, seed_ihbcp=np.frombuffer(seed_ihbcp, np.uint8), vbg=vbg,
               v_seed=S.motif_v(m), neg_n=neg.N, mfold=mfold, fdr_q=np.float32(q),
               zoops_stats=np.frombuffer(files["zoops.stats"], np.uint8),
               zoops_pvalues=np.frombuffer(files["zoops.pvalues"], np.uint8),
               pos_max=scores[0], neg_max=scores[1],
               zoops_logodds=np.frombuffer(files["zoops.logOdds"], np.uint8))      # --saveLogOdds, FDR.cpp:416-433
    # the same statistics with MOPS scores and without p-values (vectors left in descending order) on a subset
    m2 = S.motif(c.W, c.K, c.alpha, bg, c.q, S.motif_v(m))
    files2, scores2, _ = S.fdr(neg, m2, bg, 5, True, True, em=False, threads=1, save_pvalues=False, save_logodds=True)
    out["noem_zoops_logodds"] = np.frombuffer(files2["zoops.logOdds"], np.uint8)
    out["noem_mops_logodds_sha256"] = digest(np.frombuffer(files2["mops.logOdds"], np.uint8))
    out["noem_mops_logodds_head"] = np.frombuffer(files2["mops.logOdds"][:4000], np.uint8)
    out["noem_pos_max"], out["noem_neg_max"] = scores2[0], scores2[1]
    out["noem_pos_all_sha256"], out["noem_neg_all_sha256"] = digest(scores2[2]), digest(scores2[3])
    # the full-set model of the same command line (mainBaMM.cpp:131-147 runs it before the folds)
    m_full = S.motif(c.W, c.K, c.alpha, bg, c.q, S.motif_v(m))
    e = S.em(m_full, bg, False, False)
    S.R.ref_em_optimize(e)
    out["v_full"] = S.motif_v(m_full)
    # --saveLogOdds with --scoreSeqset (mainBaMM.cpp:204-227): ScoreSeqSet::writeLogOdds for both sets
    out["pos_logoddszoops"] = np.frombuffer(S.write_logodds(m_full, bg, c.ss, base="p"), np.uint8)
    out["neg_logoddszoops"] = np.frombuffer(neg.write_logodds(m_full, bg, c.ss, base="n"), np.uint8)
    _, out["pos_zoops_full"], out["pos_z_full"] = S.logodds(m_full, bg, c.W)
    _, out["neg_zoops_full"], out["neg_z_full"] = neg.logodds(m_full, bg, c.W)
    out["neg_codes_sha256"] = digest(neg.seq_codes())        # the sampler itself is pinned by eval_small
    np.savez_compressed(os.path.join(HERE, "config5_small.npz"), **out)
    print("wrote config5_small", neg.N, len(files["zoops.stats"]))


def run_mask_cases(R):
    """EM::mask (--advanceEM, EM.cpp:261-503) run by the reference itself on the small cases; W=1 is
    left out (the reference reads pos_[n][L], one float past its allocation, EM.cpp:416)."""
    O = oracle.Oracle()
    O.set_threads(1)
    for spec in SMALL_CASES:
        c = Case(**spec)
        if c.W < 2:
            continue
        S = R.session(c.codes, c.in_off, c.ss, 42)
        kmer = S.kmers()
        bg, vbg = S.bg(c.bg_order, c.alpha_bg)
        out = dict(vbg=vbg, kmer_sha256=digest(kmer))
        nr = min(16, S.N)
        rlen = int(S.off[nr])
        for oq in (0, 1):
            for f in (0.05, 0.2):
                tag = f"oq{oq}_f{int(f * 100)}"
                m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
                e = S.em(m, bg, bool(oq), False, f)
                S.R.ref_em_mask(e)
                out[tag + "_v"] = S.motif_v(m)
                out[tag + "_p"] = S.motif_p(m)
                out[tag + "_q"] = np.float32(S.R.ref_em_q(e))
                out[tag + "_llh"] = np.float32(S.R.ref_em_llh(e))
                out[tag + "_n"] = S.em_n(e, c.K, c.W)
                r = S.em_r(e)
                out[tag + "_r"] = r[:rlen]
                out[tag + "_r_sha256"] = digest(r)
                # iteration count / cut-off are not exposed by the reference: taken from the C
                # restatement after checking that it reproduces every exposed value bit for bit
                res = O.mask(kmer, S.off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=bool(oq), f=f)
                assert np.array_equal(res["v"], out[tag + "_v"]) and np.array_equal(res["r"], r)
                assert np.float32(res["q"]) == out[tag + "_q"] and np.float32(res["llh"]) == out[tag + "_llh"]
                out[tag + "_iterations"] = res["iterations"]
                out[tag + "_cutoff"] = np.float32(res["cutoff"])
                out[tag + "_listed"] = res["listed"]
                out[tag + "_trace_llh"] = res["trace_llh"]
                out[tag + "_trace_vdiff"] = res["trace_vdiff"]
        S.close()
        np.savez_compressed(os.path.join(HERE, f"mask_{c.name}.npz"), **out)
        print("wrote mask", c.name)


def main():
    if not oracle.have_reference():
        raise SystemExit("oracle/_ref/libbammref.so missing: run `make -C oracle ref` in the dev container")
    R = oracle.Reference()
    R.set_threads(1)
    if sys.argv[1:] == ["mask"]:
        run_mask_cases(R)
        return
    if sys.argv[1:] == ["config5"]:
        run_config5_case(R)
        return
    if sys.argv[1:2] == ["large"]:                   # one large case by name
        for spec in LARGE:
            if spec["name"] in sys.argv[2:]:
                c = Case(**spec)
                out = run_case(R, c, store_inputs=False, n_iter=3 if c.K <= 2 else 2, r_seqs=8,
                               with_optimize=(c.N <= 1000 and c.K <= 2))
                if c.K > 2:
                    for key in ("s_0", "n_0", "v_0", "p_final", "logs_final"):
                        out.pop(key, None)
                np.savez_compressed(os.path.join(HERE, f"large_{c.name}.npz"), **out)
                print("wrote", c.name)
        return
    run_mask_cases(R)
    run_config5_case(R)
    run_eval_case(R)
    run_travis_case(R)
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
