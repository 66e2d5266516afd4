"""C++ host side of the drop-in CLI (bammmotif2_amd/host): FASTA reader, seeders and model-file
writers against the reference's own outputs (tests/golden) and the pinned oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import abi, build
from tests import golden_util as gu

EX = os.path.join(gu.GOLDEN_DIR, "example")
FASTA, MEME = os.path.join(EX, "JunD.fasta"), os.path.join(EX, "PWM_peng10.meme")


@pytest.fixture(scope="module")
def host(lib):
    build.build_host()
    H = C.CDLL(build.HOST_LIB)
    H.bh_last_error.restype = C.c_char_p
    H.bh_base_name.restype = C.c_char_p
    return H


def read_fasta_py(path):
    seqs, cur, have = [], [], False
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        if line.startswith(">"):
            if have and cur:
                seqs.append("".join(cur))
            cur, have = [], True
        else:
            cur.append(line)
    if have and cur:
        seqs.append("".join(cur))
    lut = np.zeros(256, np.uint8)
    for i, c in enumerate("ACGT"):
        lut[ord(c)] = lut[ord(c.lower())] = i + 1
    codes = np.concatenate([lut[np.frombuffer(s.encode(), np.uint8)] for s in seqs])
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return codes, off


def host_fasta(H, path):
    n, m = C.c_uint64(), C.c_uint64()
    assert H.bh_read_fasta(path.encode(), C.byref(n), C.byref(m), None, None, None) == 0, H.bh_last_error()
    codes = np.zeros(m.value, np.uint8)
    off = np.zeros(n.value + 1, np.uint64)
    bf = np.zeros(4, np.float32)
    assert H.bh_read_fasta(path.encode(), C.byref(n), C.byref(m), codes.ctypes.data_as(C.c_void_p),
                           off.ctypes.data_as(C.c_void_p), bf.ctypes.data_as(C.c_void_p)) == 0
    return codes, off, bf


def test_fasta_reader(host, tmp_path):
    codes, off, bf = host_fasta(host, FASTA)
    c2, o2 = read_fasta_py(FASTA)
    assert np.array_equal(codes, c2) and np.array_equal(off, o2) and len(off) == 301
    np.testing.assert_allclose(bf, np.bincount(codes, minlength=5)[1:] / (codes > 0).sum(), rtol=1e-6)
    # multi-line records, lower case, unknown letters, CR in headers, blank lines, header-only record
    p = tmp_path / "x.fa"
    p.write_text(">a\tcomment\r\nACGT\nacgtn\n\n>empty\n>b\nRYAC\n")
    codes, off, _ = host_fasta(host, str(p))
    assert off.tolist() == [0, 9, 13]
    assert codes.tolist() == [1, 2, 3, 4, 1, 2, 3, 4, 0, 0, 0, 1, 2]
    p.write_text(">a\nAC GT\n")
    n, m = C.c_uint64(), C.c_uint64()
    assert host.bh_read_fasta(str(p).encode(), C.byref(n), C.byref(m), None, None, None) == 1
    assert b"space character" in host.bh_last_error()      # SequenceSet.cpp:146-150
    assert host.bh_base_name(b"/a/b/JunD.fasta") == b"JunD"


def test_fasta_reader_cut_over_threads(host, tmp_path):
    """Files beyond a few MB are cut at header lines and parsed on every granted core (host/io.cpp): the joined set
    equals a single pass -- multi-line records, blank lines, CR / TAB in headers, records without a sequence, and an
    error in the middle of the file is still THE error (SequenceSet.cpp:67-225)."""
    rng = np.random.default_rng(3)
    recs = []
    for n in range(60000):
        L = int(rng.integers(30, 400))
        seq = "".join(rng.choice(list("ACGTacgtNR"), L))
        if n % 7 == 0:
            seq = seq[:L // 2] + "\n" + seq[L // 2:]            # two lines
        head = f">s{n}" + ("\tdescr" if n % 5 == 0 else "") + ("\r" if n % 11 == 0 else "")
        recs.append(head + "\n" + seq + "\n" + ("\n" if n % 13 == 0 else "") + (">nothing\n" if n % 1001 == 0 else ""))
    p = tmp_path / "big.fa"
    p.write_text("".join(recs))
    assert p.stat().st_size > 12 << 20                           # several ranges
    codes, off, bf = host_fasta(host, str(p))
    c2, o2 = read_fasta_py(str(p))
    assert np.array_equal(off, o2) and np.array_equal(codes, c2) and len(off) == 60001
    np.testing.assert_allclose(bf, np.bincount(codes, minlength=5)[1:] / (codes > 0).sum(), rtol=1e-6)
    recs[41234] = ">bad\nAC GT\n"
    p.write_text("".join(recs))
    n, m = C.c_uint64(), C.c_uint64()
    assert host.bh_read_fasta(str(p).encode(), C.byref(n), C.byref(m), None, None, None) == 1
    assert b"space character" in host.bh_last_error()


@pytest.mark.parametrize("name", ["small_k2_ds_N", "small_k0_ss", "small_k3_ds"])
def test_model_writers_match_reference_bytes(name, host, tmp_path):
    c, g = gu.load(name)
    d = str(tmp_path).encode()
    v = np.ascontiguousarray(g["v_2"], np.float32)
    vbg = np.ascontiguousarray(g["vbg"], np.float32)
    assert host.bh_write_motif(d, b"m", c.W, c.K, v.ctypes.data_as(C.c_void_p), c.bg_order,
                               vbg.ctypes.data_as(C.c_void_p)) == 0
    assert open(tmp_path / "m.ihbcp", "rb").read() == g["file_ihbcp"].tobytes()    # Motif.cpp:515-547
    assert open(tmp_path / "m.ihbp", "rb").read() == g["file_ihbp"].tobytes()
    a = np.ascontiguousarray(c.alpha_bg, np.float32)
    assert host.bh_write_bg(d, b"bg", c.bg_order, a.ctypes.data_as(C.c_void_p), vbg.ctypes.data_as(C.c_void_p)) == 0
    assert open(tmp_path / "bg.hbcp", "rb").read() == g["file_hbcp"].tobytes()     # BackgroundModel.cpp:359-377
    assert open(tmp_path / "bg.hbp", "rb").read() == g["file_hbp"].tobytes()       # :407-426
    # read it back (BackgroundModel.cpp:48-129): 7 significant digits survive
    K, al, vv = C.c_uint32(), np.zeros(8, np.float32), np.zeros(bm.bg_size(4), np.float32)
    assert host.bh_read_bg(str(tmp_path / "bg.hbcp").encode(), C.byref(K), al.ctypes.data_as(C.c_void_p),
                           vv.ctypes.data_as(C.c_void_p), 4) == 0
    assert K.value == c.bg_order and np.array_equal(al[: c.bg_order + 1], c.alpha_bg)
    np.testing.assert_allclose(vv[: len(vbg)], vbg, rtol=1e-6)


def load_seed(H, path, tag, K, alpha, vbg, packed, index=0, max_pwm=2 ** 62, q=0.3):
    n, w, qq = C.c_uint32(), C.c_uint32(), C.c_float()
    v = np.zeros(bm.v_size(K, 64), np.float32)
    alpha = np.ascontiguousarray(alpha, np.float32)
    vbg = np.ascontiguousarray(vbg, np.float32)
    rc = H.bh_load_seed(path.encode(), tag.encode(), 0, 0, K, alpha.ctypes.data_as(C.c_void_p), C.c_uint64(max_pwm),
                        C.c_float(q), 2, vbg.ctypes.data_as(C.c_void_p), packed._p, index, C.byref(n), C.byref(w),
                        C.byref(qq), v.ctypes.data_as(C.c_void_p), C.c_uint64(len(v)))
    assert rc == 0, H.bh_last_error()
    return n.value, w.value, qq.value, v[: bm.v_size(K, w.value)]


@pytest.mark.parametrize("K", [0, 2])
def test_pwm_seeding_matches_oracle_restatement(K, host, orc):
    """MEME parser + Motif::initFromPWM (Motif.cpp:192-333) in the product's C++ (std::mt19937 +
    std::discrete_distribution) against the oracle's hand restatement of the same algorithm."""
    codes, off = read_fasta_py(FASTA)
    packed = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    _, kmer, o = orc.encode_set(codes, off, False, 42)
    vbg = packed.bg_model(2, np.array([1, 10, 10], np.float32))
    alpha = bm.synth.default_alpha(K)
    n, W, q, v = load_seed(host, MEME, "PWM", K, alpha, vbg, packed, index=0)
    assert (n, W) == (6, 12) and q == pytest.approx(0.3)
    lines = open(MEME).read().split("\n")
    i = [k for k, l in enumerate(lines) if "letter-probability matrix" in l][0]
    pwm = np.array([[float(x) for x in lines[i + 1 + j].split()] for j in range(W)], np.float32).T.copy()
    v_o = orc.init_from_pwm(pwm, W, K, bm.synth.alpha_matrix(alpha, W), vbg, kmer, o, 0.3)
    assert np.array_equal(v, v_o)
    n1, _, _, _ = load_seed(host, MEME, "PWM", K, alpha, vbg, packed, index=0, max_pwm=1)
    assert n1 == 1                                           # --maxPWM (MotifSet.cpp:145-147)


def pwm_sites(H, pwm, W, K, alpha, vbg, packed, q=0.3, ctx=None, seqs=None):
    """Motif::initFromPWM through the product's host C++ (or, with ctx/seqs, the device path): (v, z)."""
    v = np.zeros(bm.v_size(K, W), np.float32)
    z = np.zeros(max(packed.n_seqs, 1), np.uint32)
    pwm, alpha, vbg = (np.ascontiguousarray(a, np.float32) for a in (pwm, alpha, vbg))
    H.bh_last_error.restype = C.c_char_p
    rc = H.bh_pwm_sites(pwm.ctypes.data_as(C.c_void_p), W, K, alpha.ctypes.data_as(C.c_void_p), 2, vbg.ctypes.data_as(C.c_void_p),
                        packed._p, C.c_float(q), v.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                        ctx.h if ctx else None, seqs.h if seqs else None)
    assert rc == 0, H.bh_last_error()
    return v, z[: packed.n_seqs]


SITE_CASES = [dict(name="s_k2", N=300, L0=120, W=11, K=2, n_frac=0.01, ragged=40),
              dict(name="s_k1_ss", N=200, L0=90, W=7, K=1, ss=True, ragged=85),      # some sequences shorter than W: no draw
              dict(name="s_k0", N=150, L0=70, W=9, K=0, n_frac=0.03, ragged=20)]


@pytest.mark.parametrize("spec", SITE_CASES, ids=[d["name"] for d in SITE_CASES])
def test_pwm_seeding_samples_the_same_sites_as_the_oracle(spec, host, orc):
    """Sequence by sequence: the site the product's host path draws with the real std::mt19937 +
    std::discrete_distribution is the site the oracle's hand restatement of libstdc++'s algorithm draws
    (Motif.cpp:296-299) -- the two implementations are pinned to each other through z, not only through the
    model they lead to.  (Against the reference itself this row stays unpinned: initFromPWM needs the Boost
    FASTA reader)"""
    from tests.cases import Case
    c = Case(**spec)
    _, kmer, off = orc.encode_set(c.codes, c.in_off, c.ss, 42)
    packed = bm.PackedSeqs.from_codes(c.codes, c.in_off, c.ss, seed=42)
    vbg = packed.bg_model(2, np.array([1, 10, 10], np.float32))
    v, z = pwm_sites(host, c.pwm, c.W, c.K, c.alpha, vbg, packed, q=c.q)
    v_o, z_o, cnt_o = orc.init_from_pwm_sites(c.pwm, c.W, c.K, c.A, vbg, kmer, off, c.q)
    assert np.array_equal(z, z_o)
    assert np.array_equal(v, v_o)
    lens = np.diff(off.astype(np.int64))
    assert np.all(z[lens < c.W] == 0) and (z > 0).sum() > 0.2 * c.N and (z == 0).sum() > 0
    assert cnt_o[: 4 * c.W].reshape(4, c.W).sum(axis=0).tolist() == [int((z > 0).sum())] * c.W


def test_bamm_file_roundtrip(host, tmp_path, orc):
    c, g = gu.load("small_k2_ds_N")
    open(tmp_path / "m.ihbcp", "wb").write(g["file_ihbcp"].tobytes())
    packed = bm.PackedSeqs.from_codes(c.codes, c.in_off, c.ss, seed=42)
    n, W, q, v = load_seed(host, str(tmp_path / "m.ihbcp"), "BaMM", c.K, c.alpha, g["vbg"], packed)
    assert (n, W) == (1, c.W)
    np.testing.assert_allclose(v, g["v_2"], rtol=6e-4)        # files carry 4 significant digits (Motif.cpp:536)


def test_cli_without_em_and_option_errors(host, tmp_path):
    out = tmp_path / "o"
    r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "-k", "1", "--maxPWM", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Note: the model is not optimized!" in r.stdout and "------ Runtime:" in r.stdout
    assert sorted(os.listdir(out)) == ["JunD.hbcp", "JunD.hbp", "JunD_motif_1.ihbcp", "JunD_motif_1.ihbp",
                                       "JunD_motif_2.ihbcp", "JunD_motif_2.ihbp"]
    assert open(out / "JunD.hbcp").readline() == "# K = 2\n"
    r = subprocess.run([build.CLI, str(out), FASTA, "--PWMFile", MEME, "--noSuchFlag"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown option(s) remaining" in r.stderr          # Global.cpp:337-341
    r = subprocess.run([build.CLI, str(out), FASTA], capture_output=True, text=True)
    assert r.returncode == 1 and "No initial model is provided" in r.stderr          # Global.cpp:194-197
    r = subprocess.run([build.CLI, str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "Arguments are missing" in r.stderr                 # Global.cpp:127-131


def test_row_formatter_is_printf_g(host):
    """The .stats / .pvalues writers print floats as `ostream << float` does (printf %g at precision 6 / 3); the fast
    formatter must give those bytes on every float: random bit patterns, counts, tenths (FP = in / mFold), unit-interval
    values, tiny p-values, exact half-way cases, powers of ten, zeros, infinities, NaN, denormals."""
    host.bh_format_g_check.restype = C.c_uint64
    host.bh_format_g_check.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    rng = np.random.default_rng(5)
    sets = {
        "bits": rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        "counts": rng.integers(0, 3_000_000, 500_000).astype(np.float32),
        "tenths": rng.integers(0, 30_000_000, 500_000).astype(np.float32) / np.float32(10),
        "unit": rng.random(500_000).astype(np.float32),
        "pvalues": np.exp(-rng.random(500_000) * 60).astype(np.float32),
        "halves": ((rng.integers(0, 2_000_000, 300_000).astype(np.float64) + 0.5) / 10.0 ** rng.integers(0, 8, 300_000)).astype(np.float32),
        "pow10": (10.0 ** rng.integers(-20, 20, 20_000)).astype(np.float32),
        "edge": np.array([0.0, -0.0, 1.0, 9.999995, 9.9999949, 999999.5, 999999.4, 1e6, 1e-5, 9.9999e-5, 1e-4, 0.000123456,
                          123456.5, 1234565, 0.5, 0.25, np.inf, -np.inf, np.nan, 1e-38, 1e-45, 3.4e38, 1e5, 999999, -2.5, -1e-7],
                         dtype=np.float32),
    }
    for name, x in sets.items():
        x = np.ascontiguousarray(x, dtype=np.float32)
        for precision in (6, 3, 1, 9):
            bits = C.c_uint32(0)
            bad = host.bh_format_g_check(x.ctypes.data_as(C.c_void_p), len(x), precision, C.byref(bits))
            assert bad == 0, (name, precision, bad, hex(bits.value))
