"""Motif::initFromPWM's pass over the sequences on the device (bamm_seed_from_pwm, csrc/seed.hip)
against the host C++ (std::mt19937 + std::discrete_distribution, host/io.cpp) and the oracle's hand
restatement (oracle/bamm_oracle.c: orc_init_from_pwm) -- identical models, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import build
from tests.cases import Case
from tests.test_host_io_cpu import FASTA, MEME, SITE_CASES, load_seed, pwm_sites, read_fasta_py

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    build.build_host()
    return C.CDLL(build.HOST_LIB)


def load_seed_dev(H, ctx, seqs, path, K, alpha, vbg, packed, index=0, q=0.3):
    n, w, qq = C.c_uint32(), C.c_uint32(), C.c_float()
    v = np.zeros(bm.v_size(K, 64), np.float32)
    alpha = np.ascontiguousarray(alpha, np.float32)
    vbg = np.ascontiguousarray(vbg, np.float32)
    H.bh_last_error.restype = C.c_char_p
    rc = H.bh_load_seed_dev(path.encode(), b"PWM", 0, 0, K, alpha.ctypes.data_as(C.c_void_p), C.c_uint64(2 ** 62),
                            C.c_float(q), 2, vbg.ctypes.data_as(C.c_void_p), packed._p, index, C.byref(n), C.byref(w),
                            C.byref(qq), v.ctypes.data_as(C.c_void_p), C.c_uint64(len(v)), ctx.h, seqs.h)
    assert rc == 0, H.bh_last_error()
    return n.value, w.value, qq.value, v[: bm.v_size(K, w.value)]


@pytest.mark.parametrize("K", [0, 2, 4])
def test_device_seeding_equals_host_seeding_on_jund(K, host, gpu_ctx, orc):
    codes, off = read_fasta_py(FASTA)
    packed = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    ss = bm.SeqSet(gpu_ctx, packed)
    vbg = packed.bg_model(2, np.array([1, 10, 10], np.float32))
    alpha = bm.synth.default_alpha(K)
    for index in (0, 3):
        _, W, q, v_host = load_seed(host, MEME, "PWM", K, alpha, vbg, packed, index=index)
        _, Wd, qd, v_dev = load_seed_dev(host, gpu_ctx, ss, MEME, K, alpha, vbg, packed, index=index)
        assert (W, q) == (Wd, qd)
        assert np.array_equal(v_host, v_dev)
    ss.close()


def mt19937_canonical(n):
    """std::generate_canonical<double,53> over a default-seeded std::mt19937 (seed 5489): two 32-bit
    draws per variate, (a + b * 2^32) / 2^64 in double."""
    raw = np.random.RandomState(5489)._bit_generator.random_raw(2 * n).astype(np.float64)
    u = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
    return np.where(u >= 1.0, np.nextafter(1.0, 0.0), u)


@pytest.mark.parametrize("spec", [dict(name="s_k2", N=300, L0=120, W=11, K=2, n_frac=0.01, ragged=40),
                                  dict(name="s_k1_ss", N=200, L0=90, W=7, K=1, ss=True, ragged=85),
                                  dict(name="s_long", N=7, L0=3000, W=15, K=2, ragged=1090, n_frac=0.0005),
                                  # beyond the per-wave LDS arrays (~10 000 positions): the arrays live in global scratch
                                  dict(name="s_xlong", N=9, L0=9000, W=12, K=2, ragged=4000, n_frac=0.0002),
                                  dict(name="s_xlong_ss", N=5, L0=40000, W=9, K=1, ss=True, ragged=25000)],
                         ids=["k2_ds_N", "k1_ss_short", "k2_ds_L8181", "k2_ds_L26001", "k1_ss_L65000"])
def test_seed_from_pwm_matches_oracle(spec, gpu_ctx, orc):
    """The ABI entry on its own: uniform variates from the documented recipe, counts turned into a
    model by the oracle-side formulas, against orc_init_from_pwm."""
    c = Case(**spec)
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    pwm = np.maximum(c.pwm.astype(np.float32), np.float32(1e-8))
    pwm = (pwm / pwm.sum(axis=0, dtype=np.float32)).astype(np.float32)      # W = 4 terms: order-free here
    lens = np.diff(off.astype(np.int64))
    valid = lens >= c.W
    u = np.zeros(c.N)
    u[valid] = mt19937_canonical(int(valid.sum()))
    score = (pwm / vbg[:4, None]).astype(np.float32)
    counts, z = bm.seed_from_pwm(gpu_ctx, ss, c.K, c.W, score.ravel(), c.q, u)
    # against orc_init_from_pwm directly: the sampled site of every sequence and the site counts of all orders
    _, z_o, cnt_o = orc.init_from_pwm_sites(c.pwm, c.W, c.K, c.A, vbg, kmer, off, c.q)
    assert np.array_equal(z, z_o)
    assert np.array_equal(counts, cnt_o)
    assert np.all(z[~valid] == 0) and np.all(z <= np.maximum(lens - c.W + 1, 0))
    # counts of order 0 = sampled sites per column
    assert np.all(counts[: 4 * c.W].reshape(4, c.W).sum(axis=0) == int((z > 0).sum()))
    # the same sites through the oracle's restatement give the same counts
    exp = np.zeros_like(counts)
    for n in np.flatnonzero(z > 0):
        base = int(off[n]) + int(z[n]) - 1
        for k in range(c.K + 1):
            o = bm.v_offset(k, c.W)
            for j in range(c.W):
                exp[o + int(kmer[base + j] % 4 ** (k + 1)) * c.W + j] += 1
    assert np.array_equal(counts, exp)
    ss.close()


def test_device_seeding_equals_host_seeding_on_ragged_set(host, gpu_ctx, orc, tmp_path):
    """Sequences shorter than the motif (skipped without consuming a draw, Motif.cpp:240-248), N inside
    sequences, both strands: host C++ and device give the same model."""
    c = Case(name="s_rag", N=400, L0=60, W=14, K=2, n_frac=0.02, ragged=55)
    packed = bm.PackedSeqs.from_codes(c.codes, c.in_off, False, seed=42)
    ss = bm.SeqSet(gpu_ctx, packed)
    vbg = packed.bg_model(2, np.array([1, 10, 10], np.float32))
    meme = tmp_path / "m.meme"
    with open(meme, "w") as f:
        f.write("MEME version 4\n\nALPHABET= ACGT\n\nMOTIF m\nletter-probability matrix: alength= 4 w= %d nsites= 9\n" % c.W)
        for j in range(c.W):
            f.write(" ".join("%.6f" % c.pwm[y, j] for y in range(4)) + "\n")
    _, W, q, v_host = load_seed(host, str(meme), "PWM", c.K, c.alpha, vbg, packed)
    _, Wd, qd, v_dev = load_seed_dev(host, gpu_ctx, ss, str(meme), c.K, c.alpha, vbg, packed)
    assert W == Wd == c.W
    assert np.array_equal(v_host, v_dev)
    ss.close()


@pytest.mark.parametrize("spec", SITE_CASES, ids=[d["name"] for d in SITE_CASES])
def test_three_seeders_sample_the_same_sites(spec, host, gpu_ctx, orc):
    """Device kernel, host C++ (std::discrete_distribution) and the oracle restatement, sequence by sequence."""
    c = Case(**spec)
    _, kmer, off = orc.encode_set(c.codes, c.in_off, c.ss, 42)
    packed = bm.PackedSeqs.from_codes(c.codes, c.in_off, c.ss, seed=42)
    ss = bm.SeqSet(gpu_ctx, packed)
    vbg = packed.bg_model(2, np.array([1, 10, 10], np.float32))
    v_h, z_h = pwm_sites(host, c.pwm, c.W, c.K, c.alpha, vbg, packed, q=c.q)
    v_d, z_d = pwm_sites(host, c.pwm, c.W, c.K, c.alpha, vbg, packed, q=c.q, ctx=gpu_ctx, seqs=ss)
    v_o, z_o, _ = orc.init_from_pwm_sites(c.pwm, c.W, c.K, c.A, vbg, kmer, off, c.q)
    assert np.array_equal(z_d, z_o) and np.array_equal(z_h, z_o)
    assert np.array_equal(v_d, v_o) and np.array_equal(v_h, v_o)
    ss.close()
