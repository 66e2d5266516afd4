"""End-to-end: the C++ `BaMMmotif OUTDIR FASTA --PWMFile ... --EM` drop-in on the GPU against the
oracle pipeline and the known answer of BASELINE config 1 (13 iterations,
llh 454.708 -> 475.799 on example/JunD.fasta seeded from PWM_peng10.meme, -k 0)."""
import os
import re
import subprocess

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import build
from tests.test_host_io_cpu import FASTA, MEME, read_fasta_py

pytestmark = pytest.mark.gpu


def parse_ihbcp(path, K, W):
    v = np.zeros(bm.v_size(K, W), np.float32)
    blocks = open(path).read().split("\n\n")
    for j in range(W):
        lines = blocks[j].strip("\n").split("\n")
        for k in range(K + 1):
            vals = [float(x) for x in lines[k].split()]
            assert len(vals) == 4 ** (k + 1)
            v[bm.v_offset(k, W) + np.arange(4 ** (k + 1)) * W + j] = vals
    return v


def oracle_run(orc, K):
    codes, off = read_fasta_py(FASTA)
    _, kmer, o = orc.encode_set(codes, off, False, 42)
    vbg = orc.bg_model(kmer, o, 2, np.array([1, 10, 10], np.float32))
    lines = open(MEME).read().split("\n")
    i = [k for k, l in enumerate(lines) if "letter-probability matrix" in l][0]
    W = 12
    pwm = np.array([[float(x) for x in lines[i + 1 + j].split()] for j in range(W)], np.float32).T.copy()
    A = bm.synth.alpha_matrix(bm.synth.default_alpha(K), W)
    v0 = orc.init_from_pwm(pwm, W, K, A, vbg, kmer, o, 0.3)
    res = orc.optimize(kmer, o, K, W, 2, vbg, A, v0, 0.3)
    return res, kmer, o, W


@pytest.mark.parametrize("K", [0, 2])
def test_cli_em_on_jund(K, tmp_path, orc, gpu_ctx):
    build.build_host()
    out = tmp_path / "o"
    r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "--EM", "-k", str(K), "--maxPWM", "1",
                        "--verbose", "--saveBaMMs"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    res, kmer, off, W = oracle_run(orc, K)
    it_lines = re.findall(r"^(\d+) iter, llh=([-\d.e+]+), diff_llh=([-\d.e+]+), v_diff=([-\d.e+]+)$", r.stdout, re.M)
    llh = np.array([float(x[1]) for x in it_lines])
    assert abs(len(it_lines) - res["iterations"]) <= 1
    m = min(len(llh), res["iterations"])
    np.testing.assert_allclose(llh[:m], res["trace_llh"][:m], rtol=2e-5)
    if K == 0:                                               # the survey's probe of the real reference
        assert len(it_lines) == 13
        assert llh[0] == pytest.approx(454.708, abs=2e-3) and llh[-1] == pytest.approx(475.799, abs=2e-3)
    assert re.search(r"--- Runtime for EM: [\d.e+-]+ seconds ---", r.stdout)        # EM.cpp:134
    assert "optimized q = 0.3" in r.stdout                                           # mainBaMM.cpp:147
    v = parse_ihbcp(out / "JunD_motif_1.ihbcp", K, W)
    if len(it_lines) == res["iterations"]:
        np.testing.assert_allclose(v, res["v"], rtol=2e-3, atol=1e-6)               # %.3e files
    # --saveBaMMs: .counts are int-truncated n (EM.cpp:570), .positions lists windows with r >= 0.3
    cnt = [l for l in open(out / "JunD_motif_1.counts").read().split("\n") if l.strip()]
    assert len(cnt) == W * (K + 1)
    pos = open(out / "JunD_motif_1.positions").read().strip().split("\n")
    assert pos[0] == "seq\tlength\tstrand\tstart..end\tpattern"
    n_hits = 0
    for n in range(300):
        L = int(off[n + 1] - off[n])
        rr = res["r"][int(off[n]):int(off[n + 1])]
        n_hits += int((rr[: L - W + 1] >= 0.3).sum())
    assert abs((len(pos) - 1) - n_hits) <= 3                 # r right at the 0.3 cut-off may flip
    f = pos[1].split("\t")
    assert f[0].startswith(">") and f[1] == "205" and f[2] in "+-" and len(f[4]) == W


def test_cli_travis_smoke_line(tmp_path, gpu_ctx):
    """The reference's own CI command (.travis.yml:21) through the drop-in CLI:
    --EM -k 0 --FDR --scoreSeqset --maxPWM 1; outputs against what the reference's EM / FDR /
    ScoreSeqSet / SeqGenerator produced for the same inputs (tests/golden/travis_jund.npz)."""
    from tests import golden_util as gu
    build.build_host()
    g = dict(np.load(os.path.join(gu.GOLDEN_DIR, "travis_jund.npz")))
    out = tmp_path / "o"
    r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "--EM", "-k", "0", "--FDR", "--scoreSeqset",
                        "--maxPWM", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert sorted(os.listdir(out)) == ["JunD.hbcp", "JunD.hbp", "JunD_motif_1.ihbcp", "JunD_motif_1.ihbp",
                                       "JunD_motif_1.occurrence", "JunD_motif_1.zoops.stats"]
    assert r.stdout.count("--- Runtime for EM:") == 5                    # 1 full run + 4 folds
    v = parse_ihbcp(out / "JunD_motif_1.ihbcp", 0, 12)
    np.testing.assert_allclose(v, g["v_final"], rtol=2e-3, atol=1e-6)

    def rows(b):
        lines = b.decode().strip().split("\n")
        return lines[0], [l.rstrip("\t").split("\t") for l in lines[1:]]

    h_ref, ref = rows(g["zoops_stats"].tobytes())
    h_mine, mine = rows(open(out / "JunD_motif_1.zoops.stats", "rb").read())
    assert h_mine.split("\t")[:6] == h_ref.split("\t")[:6]               # TP FP FDR Recall p-value mFold=17
    assert float(h_mine.split("\t")[6]) == pytest.approx(float(h_ref.split("\t")[6]), abs=5e-3)   # occ_frac
    assert len(mine) == len(ref) == 300 + 5100
    a = np.array(mine, float)
    b = np.array(ref, float)
    assert np.mean(np.all(a[:, :2] == b[:, :2], axis=1)) > 0.98          # TP/FP ranks: ties may swap
    np.testing.assert_allclose(a[:, 4], b[:, 4], rtol=0.05, atol=2e-4)   # p-values along the ranking

    def hits(bts):                      # golden rows carry synthetic headers (seqN): compare strand + range
        return sorted(tuple(l.split("\t")[1:4]) for l in bts.decode().strip().split("\n")[1:])

    ref_hits = hits(g["occurrence"].tobytes())
    my_hits = hits(open(out / "JunD_motif_1.occurrence", "rb").read())
    assert abs(len(my_hits) - len(ref_hits)) <= max(3, 0.02 * len(ref_hits))    # p right at the cut-off may flip
    from collections import Counter
    common = sum((Counter(ref_hits) & Counter(my_hits)).values())
    assert common >= 0.97 * len(ref_hits)


@pytest.mark.parametrize("oq", [False, True], ids=["fixq", "optq"])
def test_cli_advance_em_on_jund(oq, tmp_path, orc, gpu_ctx):
    """`--EM --advanceEM -f 0.1` (mainBaMM.cpp:133-137 -> EM::mask) against the pinned restatement."""
    build.build_host()
    K, f = 1, 0.1
    out = tmp_path / "o"
    cmd = [build.CLI, str(out), FASTA, "--PWMFile", MEME, "--EM", "--advanceEM", "-f", str(f), "-k", str(K), "--maxPWM", "1",
           "--verbose", "--saveBaMMs"] + (["--optimizeQ"] if oq else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert "10% of the sequences are used for EM after masking." in r.stdout           # Global.cpp:370-372
    codes, off = read_fasta_py(FASTA)
    _, kmer, o = orc.encode_set(codes, off, False, 42)
    vbg = orc.bg_model(kmer, o, 2, np.array([1, 10, 10], np.float32))
    lines = open(MEME).read().split("\n")
    i = [k for k, l in enumerate(lines) if "letter-probability matrix" in l][0]
    W = 12
    pwm = np.array([[float(x) for x in lines[i + 1 + j].split()] for j in range(W)], np.float32).T.copy()
    A = bm.synth.alpha_matrix(bm.synth.default_alpha(K), W)
    v0 = orc.init_from_pwm(pwm, W, K, A, vbg, kmer, o, 0.3)
    res = orc.mask(kmer, o, K, W, 2, vbg, A, v0, 0.3, optimizeQ=oq, f=f)
    d = [float(x) for x in re.findall(r"^\d+th iteration, delta_llikelihood=([-\d.e+]+)$", r.stdout, re.M)]
    assert abs(len(d) - res["iterations"]) <= 1
    m = min(len(d), res["iterations"], 8)
    ref_d = np.diff(np.concatenate([[0.0], res["trace_llh"]]))
    np.testing.assert_allclose(d[:m], ref_d[:m], rtol=1e-3, atol=2e-3)
    q_line = re.search(r"optimized q = ([\d.e+-]+)", r.stdout)                           # mainBaMM.cpp:147
    assert float(q_line.group(1)) == pytest.approx(res["q"], rel=1e-5)
    if len(d) == res["iterations"]:
        v = parse_ihbcp(out / "JunD_motif_1.ihbcp", K, W)
        np.testing.assert_allclose(v, res["v"], rtol=5e-3, atol=1e-6)
        pos = open(out / "JunD_motif_1.positions").read().strip().split("\n")           # r_ after mask (EM.cpp:583-601)
        n_hits = 0
        for n in range(300):
            L = int(o[n + 1] - o[n])
            n_hits += int((res["r"][int(o[n]):int(o[n]) + L - W + 1] >= 0.3).sum())
        assert abs((len(pos) - 1) - n_hits) <= 3


def _write_config5_inputs(tmp_path, g):
    codes, off = g["codes"], g["in_off"]
    lut = np.frombuffer(b"NACGT", np.uint8)
    with open(tmp_path / "pos.fasta", "wb") as f:
        for n in range(len(off) - 1):
            f.write(b">seq%d\n" % n)
            f.write(lut[codes[int(off[n]):int(off[n + 1])]].tobytes() + b"\n")
    open(tmp_path / "seed.ihbcp", "wb").write(g["seed_ihbcp"].tobytes())
    return str(tmp_path / "pos.fasta"), str(tmp_path / "seed.ihbcp")


CONFIG5_FLAGS = ["--EM", "-k", "2", "--FDR", "-n", "5", "-m", "10", "--savePvalues", "--saveLogOdds"]


def test_cli_config5_fdr_line(tmp_path, gpu_ctx):
    """BASELINE config 5's command line, `--EM -k 2 --FDR -n 5 -m 10`, on the reduced set of
    tests/golden/config5_small.npz (600 x 200 bp, both strands, W = 20): 1 full EM + 5 fold EMs + scoring of
    the test folds and of the sampled negatives, against what the reference's EM / FDR / ScoreSeqSet /
    SeqGenerator produced for the same inputs (FDR.cpp:28-145, mainBaMM.cpp:97-116,243-265)."""
    from tests import golden_util as gu
    build.build_host()
    g = dict(np.load(os.path.join(gu.GOLDEN_DIR, "config5_small.npz")))
    fasta, seed = _write_config5_inputs(tmp_path, g)
    out = tmp_path / "o"
    r = subprocess.run([build.CLI, str(out), fasta, "--BaMMFile", seed] + CONFIG5_FLAGS, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert r.stdout.count("--- Runtime for EM:") == 6                    # 1 full run + 5 folds
    assert "Folds for cross-validation (FDR estimation): 5" in r.stdout
    K, W = 2, 20
    np.testing.assert_allclose(parse_ihbcp(out / "pos_motif_1.ihbcp", K, W), g["v_full"], rtol=2e-3, atol=1e-6)   # %.3e files

    def rows(b):
        lines = b.decode().strip().split("\n")
        return lines[0], [l.rstrip("\t").split("\t") for l in lines[1:]]

    h_ref, ref = rows(g["zoops_stats"].tobytes())
    h_mine, mine = rows(open(out / "pos_motif_1.zoops.stats", "rb").read())
    assert h_mine.split("\t")[:6] == h_ref.split("\t")[:6]               # TP FP FDR Recall p-value mFold (= 9, mainBaMM.cpp:100-106)
    assert float(h_mine.split("\t")[5]) == int(g["mfold"]) == 9
    assert float(h_mine.split("\t")[6]) == pytest.approx(float(h_ref.split("\t")[6]), abs=5e-3)   # occ_frac
    assert len(mine) == len(ref) == 600 + int(g["neg_n"])
    a, b = np.array(mine, float), np.array(ref, float)
    assert np.mean(np.all(a[:, :2] == b[:, :2], axis=1)) > 0.98          # TP/FP along the ranking: near-ties may swap
    np.testing.assert_allclose(a[:, 4], b[:, 4], rtol=0.05, atol=2e-4)   # p-values along the ranking
    pv_ref = np.array(g["zoops_pvalues"].tobytes().decode().split(), float)
    pv = np.array(open(out / "pos_motif_1.zoops.pvalues").read().split(), float)
    assert len(pv) == len(pv_ref) == 600
    np.testing.assert_allclose(pv, pv_ref, rtol=0.02, atol=2e-4)
    lo_ref = np.array([l.split("\t") for l in g["zoops_logodds"].tobytes().decode().strip().split("\n")[1:]], float)
    lo = np.array([l.split("\t") for l in open(out / "pos_motif_1.zoops.logOdds").read().strip().split("\n")[1:]], float)
    assert lo.shape == lo_ref.shape == (600, 2)
    np.testing.assert_allclose(lo, lo_ref, rtol=2e-4, atol=2e-4)         # scores of models that agree to ~1e-5


def test_cli_fold_replicas_and_native_comm_do_not_change_the_files(tmp_path, gpu_ctx):
    """--gpus N: fold f trains on GPU f mod N on a host thread of its own, scores are merged in fold order
    (FDR.cpp:37-127).  A 1-GPU box runs the thread-per-fold path with two (three) contexts on device 0; every
    output file must equal the single-context run byte for byte.  --forceComm runs the sharded-EM code path
    (handle per GPU + RCCL all-reduce inside libbamm_em) with a 1-rank communicator: same bytes again."""
    from tests import golden_util as gu
    build.build_host()
    g = dict(np.load(os.path.join(gu.GOLDEN_DIR, "config5_small.npz")))
    fasta, seed = _write_config5_inputs(tmp_path, g)
    flags = CONFIG5_FLAGS + ["--scoreSeqset", "--saveBaMMs"]
    outs, plans = {}, {}
    for tag, extra in (("one", []), ("gpus1", ["--gpus", "1"]), ("two", ["--deviceList", "0,0"]),
                       ("three", ["--deviceList", "0,0,0"]), ("comm", ["--forceComm"]),
                       # 8 slots, 5 folds: the main run sharded over 3 contexts (host-staged all-reduce: one device) WHILE
                       # the folds train on the other 5 -- the plan an 8-GPU node gets
                       ("eight", ["--deviceList", "0,0,0,0,0,0,0,0", "--timing"])):
        out = tmp_path / tag
        r = subprocess.run([build.CLI, str(out), fasta, "--BaMMFile", seed] + flags + extra, capture_output=True, text=True)
        assert r.returncode == 0, tag + ": " + r.stderr + r.stdout[-2000:]
        assert r.stdout.count("--- Runtime for EM:") == 6
        outs[tag] = {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))}
        plans[tag] = r.stderr
    names = sorted(outs["one"])
    assert {"pos_motif_1.zoops.stats", "pos_motif_1.zoops.pvalues", "pos_motif_1.zoops.logOdds", "pos_motif_1.ihbcp",
            "pos_motif_1.occurrence", "pos_motif_1.logOddsZoops", "pos.negSet.logOddsZoops", "pos_motif_1.positions",
            "pos_motif_1.counts"} <= set(names)
    for tag in ("gpus1", "two", "three", "comm", "eight"):
        assert sorted(outs[tag]) == names, tag
        for f in names:
            assert outs[tag][f] == outs["one"][f], f"{tag}: {f} differs"
    assert "main EM on slot(s) 0..2 (sharded, host-staged all-reduce" in plans["eight"] and "(while the main run trains)" in plans["eight"]
    assert "fold -> slot 0->3 1->4 2->5 3->6 4->7" in plans["eight"]
    # without --scoreSeqset only the folds' negatives (every cvFold-th) are sampled, packed and uploaded: same statistics
    for tag, extra in (("fdr_only", []), ("fdr_only_eight", ["--deviceList", "0,0,0,0,0,0,0,0"])):
        out = tmp_path / tag
        r = subprocess.run([build.CLI, str(out), fasta, "--BaMMFile", seed] + CONFIG5_FLAGS + extra, capture_output=True, text=True)
        assert r.returncode == 0, tag + ": " + r.stderr + r.stdout[-2000:]
        for f in ("pos_motif_1.zoops.stats", "pos_motif_1.zoops.pvalues", "pos_motif_1.zoops.logOdds", "pos_motif_1.ihbcp"):
            if f in outs["one"]:
                assert open(out / f, "rb").read() == outs["one"][f], f"{tag}: {f} differs"


def test_cli_device_packing_and_host_packing_write_the_same_files(tmp_path, gpu_ctx):
    """Sequence::Sequence and BackgroundModel's counts on the device (the default with a GPU stage; csrc/prep.hip) against
    --hostPacking (csrc/pack.cpp): every output file byte for byte, on a FASTA with N inside, at the ends and lower case."""
    build.build_host()
    import random
    rnd = random.Random(3)
    fa = tmp_path / "n.fasta"
    with open(fa, "w") as f:
        for i in range(300):
            L = rnd.randint(40, 160)
            s = "".join(rnd.choice("ACGT") for _ in range(L))
            if i % 7 == 0:
                s = "N" + s[1:]
            if i % 11 == 0:
                s = s[:-2] + "NN"
            if i % 5 == 0:
                k = rnd.randint(5, L - 5)
                s = s[:k] + "n" * rnd.randint(1, 14) + s[k:]
            if i % 13 == 0:
                s = s.lower()
            f.write(f">seq{i}\n{s[:70]}\n{s[70:]}\n" if len(s) > 70 else f">seq{i}\n{s}\n")
    outs = []
    for extra in ([], ["--hostPacking"]):
        out = tmp_path / ("dev" if not extra else "host")
        r = subprocess.run([build.CLI, str(out), str(fa), "--PWMFile", MEME, "--EM", "-k", "2", "--maxPWM", "1", "--saveBaMMs",
                            "--saveInitialBaMMs", "--FDR", "-m", "3", "-n", "3"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout[-2000:]
        outs.append({p.name: p.read_bytes() for p in sorted(out.iterdir())})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) >= 6
    for name in outs[0]:
        assert outs[0][name] == outs[1][name], name


@pytest.mark.parametrize("flags", [["--FDR", "-m", "3", "-n", "3", "--saveBaMMs", "--saveInitialBaMMs"], ["--scoreSeqset", "--saveLogOdds"], ["--advanceEM", "--saveBaMMs"]],
                         ids=["fdr", "score", "mask"])
def test_cli_fast_exit_leaves_the_same_files_as_the_orderly_teardown(flags, tmp_path, gpu_ctx):
    """The command leaves through _exit(0) once everything is written (host/main.cpp: no destructors, no atexit handlers);
    --debug takes the orderly way out.  Every output file byte for byte: a writer still holding a buffer at the fast exit
    would show here (round-4 advice)."""
    build.build_host()
    outs = []
    for extra in ([], ["--debug"]):
        out = tmp_path / ("fast" if not extra else "orderly")
        r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "--EM", "-k", "2", "--maxPWM", "1"] + flags + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout[-2000:]
        outs.append({p.name: p.read_bytes() for p in sorted(out.iterdir())})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) >= 3
    for name in outs[0]:
        assert outs[0][name] == outs[1][name], name


@pytest.mark.parametrize("flags", [["--FDR", "-m", "3", "-n", "3"], ["--scoreSeqset"], ["--FDR", "-m", "2", "-n", "4", "--genericNeg"]],
                         ids=["fdr", "score", "fdr_generic"])
def test_cli_device_sampler_and_host_sampler_write_the_same_files(flags, tmp_path, gpu_ctx):
    """SeqGenerator's negatives from the device (csrc/negs.hip, the default where the CLI wants them as a set of their own)
    against --hostSampler (host/fdr.cpp): every output file byte for byte."""
    build.build_host()
    outs = []
    for extra in ([], ["--hostSampler"]):
        out = tmp_path / ("dev" if not extra else "host")
        r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "--EM", "-k", "2", "--maxPWM", "1", "--timing"] + flags + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout[-2000:]
        assert ("sample (device" in r.stderr) == (not extra), r.stderr[-1500:]
        outs.append({p.name: p.read_bytes() for p in sorted(out.iterdir())})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) >= 4
    for name in outs[0]:
        assert outs[0][name] == outs[1][name], name


def test_cli_long_records(tmp_path, gpu_ctx):
    """Records beyond 8192 positions (both strands of 4 100+ bases) and beyond 65 535 (33 000+ bases): the window-by-window
    EM path, initFromPWM's pass with its per-wave arrays in the global scratch region (device seeding == --hostSeeding,
    byte for byte) and --advanceEM (EM::mask, 32-bit window lists there) run through the command line."""
    build.build_host()
    import random
    rnd = random.Random(11)
    fa = tmp_path / "long.fasta"
    with open(fa, "w") as f:
        for i, L in enumerate([300, 5000, 9000, 120, 34000, 2500, 800, 12000, 60, 450]):
            s = "".join(rnd.choice("ACGT") for _ in range(L))
            if i % 3 == 0:
                k = rnd.randint(5, L - 5)
                s = s[:k] + "N" + s[k + 1:]
            f.write(f">rec{i}\n")
            for a in range(0, L, 80):
                f.write(s[a:a + 80] + "\n")
    outs = []
    for extra in ([], ["--hostSeeding"]):
        out = tmp_path / ("dev" if not extra else "host")
        r = subprocess.run([build.CLI, str(out), str(fa), "--PWMFile", MEME, "--EM", "-k", "1", "--maxPWM", "1", "--maxEMIterations", "4",
                            "--saveBaMMs", "--saveInitialBaMMs"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout[-2000:]
        outs.append({p.name: p.read_bytes() for p in sorted(out.iterdir())})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) >= 4
    for name in outs[0]:
        assert outs[0][name] == outs[1][name], name
    out = tmp_path / "adv"
    r = subprocess.run([build.CLI, str(out), str(fa), "--PWMFile", MEME, "--EM", "-k", "1", "--maxPWM", "1", "--maxEMIterations", "3",
                        "--advanceEM", "--verbose"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert "3th iteration" in r.stdout or "2th iteration" in r.stdout
    assert any(p.name.endswith(".ihbcp") for p in out.iterdir())
    # ... and sharded over two contexts (the ranks' plans differ: one shard holds the 68 001-position record): same files
    out2 = tmp_path / "adv2"
    r = subprocess.run([build.CLI, str(out2), str(fa), "--PWMFile", MEME, "--EM", "-k", "1", "--maxPWM", "1", "--maxEMIterations", "3",
                        "--advanceEM", "--deviceList", "0,0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    a = {p.name: p.read_bytes() for p in sorted(out.iterdir())}
    b = {p.name: p.read_bytes() for p in sorted(out2.iterdir())}
    assert a.keys() == b.keys()
    for name in a:
        assert a[name] == b[name], name
