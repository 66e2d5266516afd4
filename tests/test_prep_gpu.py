"""Sequence::Sequence and BackgroundModel's counting pass on the device (csrc/prep.hip; include/bamm_em.h:
bamm_seqs_from_codes, bamm_seqs_bg_model) against the host restatement (csrc/pack.cpp), which is pinned bit for bit to the
reference's own kmer_ arrays (tests/test_golden_cpu.py::test_product_host_path_*).  Integer work: every array must be
equal -- the 2-bit stream, the offsets, and the exception list with the reference's rand() draws in it
(/root/reference/src/init/Sequence.cpp:4-43, init/Alphabet.cpp:46-55, init/BackgroundModel.cpp:26-42)."""
import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import synth

pytestmark = pytest.mark.gpu


def _sets():
    rs = np.random.RandomState(11)
    out = []
    pwm = synth.make_pwm(8, 3)
    for name, N, L0, n_frac, ragged in (("clean", 300, 200, 0.0, 0), ("few_N", 300, 120, 0.01, 40), ("many_N", 200, 90, 0.2, 30),
                                        ("long", 12, 5000, 0.001, 2000), ("short", 400, 14, 0.05, 6)):
        codes, off = synth.make_sequences(N, L0, pwm, 5, 0.5, n_frac, ragged)
        out.append((name, codes, off))
    # hand-made corners: N at both ends, runs of N longer than 11, all N, one base, records of length 1..12
    recs = [[0, 1, 2, 3, 4, 0], [0] * 15 + [1, 2, 3], [1, 2, 3] + [0] * 15, [0] * 30, [3], [0], [1, 0], [0, 2]]
    recs += [list(rs.randint(0, 5, size=k)) for k in range(1, 13)]
    recs += [list(rs.randint(1, 5, size=40)) + [0] + list(rs.randint(1, 5, size=9)) + [0, 0] + list(rs.randint(1, 5, size=3))]
    codes = np.array([c for r in recs for c in r], np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    out.append(("corners", codes, off))
    return out


@pytest.mark.parametrize("ss", [False, True], ids=["ds", "ss"])
@pytest.mark.parametrize("name,codes,off", _sets(), ids=[s[0] for s in _sets()])
def test_device_packing_equals_the_host_packing(name, codes, off, ss, gpu_ctx):
    host = bm.PackedSeqs.from_codes(codes, off, ss, seed=42)
    dev, seqs = bm.SeqSet.from_codes(gpu_ctx, codes, off, ss, seed=42)
    a, b = host.arrays(), dev.arrays()
    for k in a:
        assert np.array_equal(a[k], b[k]), (name, k)
    # the background model's counts from the resident set (orders whose tables take the LDS histogram and beyond)
    for K in (0, 2, 4, 6):
        alpha = np.array([1.0] + [10.0] * K, np.float32)
        assert np.array_equal(seqs.bg_model(K, alpha), host.bg_model(K, alpha)), (name, K)
    seqs.close()


def test_another_seed_and_an_offset_into_the_codes(gpu_ctx):
    pwm = synth.make_pwm(8, 3)
    codes, off = synth.make_sequences(100, 80, pwm, 9, 0.5, 0.03, 20)
    host = bm.PackedSeqs.from_codes(codes, off, False, seed=7)
    dev, seqs = bm.SeqSet.from_codes(gpu_ctx, codes, off, False, seed=7)
    assert all(np.array_equal(v, dev.arrays()[k]) for k, v in host.arrays().items())
    assert not np.array_equal(host.arrays()["exc_kmer"], bm.PackedSeqs.from_codes(codes, off, False, seed=42).arrays()["exc_kmer"])
    seqs.close()


def test_em_on_a_device_packed_set_is_the_same_em(gpu_ctx):
    W, K = 12, 2
    pwm = synth.make_pwm(W, 21)
    codes, off = synth.make_sequences(600, 150, pwm, 21, 0.5, 0.004, 30)
    host = bm.PackedSeqs.from_codes(codes, off, False, seed=42)
    dev, sd = bm.SeqSet.from_codes(gpu_ctx, codes, off, False, seed=42)
    sh = bm.SeqSet(gpu_ctx, host)
    vbg = sd.bg_model(2, np.array([1, 10, 10], np.float32))
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    res = []
    for s in (sd, sh):
        em = bm.EM(gpu_ctx, s, K, W, vbg, A, v0, 0.3)
        em.iterate(3)
        res.append(em.getV())
        em.close()
    assert np.array_equal(res[0], res[1])
    sd.close(); sh.close()


def test_random_shapes(gpu_ctx):
    """Forty seeded random sets: 1..3000 records of 1..700 bases (now and then one of 20 000), N fractions from none to all,
    both strand modes -- every array of the packed set equal to the host packing's, and the background counts with it."""
    rs = np.random.RandomState(2024)
    for trial in range(40):
        N = int(rs.choice([1, 2, 7, 60, 500, 3000]))
        top = int(rs.choice([1, 3, 12, 40, 200, 700]))
        lens = rs.randint(1, top + 1, size=N)
        if trial % 9 == 0:
            lens[rs.randint(N)] = 20000
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        codes = rs.randint(1, 5, size=int(off[-1])).astype(np.uint8)
        codes[rs.random_sample(len(codes)) < rs.choice([0.0, 0.001, 0.05, 0.5, 1.0])] = 0
        ss = bool(rs.randint(2))
        seed = int(rs.choice([42, 1, 12345]))
        host = bm.PackedSeqs.from_codes(codes, off, ss, seed=seed)
        dev, seqs = bm.SeqSet.from_codes(gpu_ctx, codes, off, ss, seed=seed)
        a, b = host.arrays(), dev.arrays()
        for k in a:
            assert np.array_equal(a[k], b[k]), (trial, N, top, ss, k)
        K = int(rs.choice([0, 1, 2, 3, 5]))
        alpha = np.array([1.0] + [10.0] * K, np.float32)
        assert np.array_equal(seqs.bg_model(K, alpha), host.bg_model(K, alpha)), (trial, K)
        seqs.close()
