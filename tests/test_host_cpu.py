"""CPU-only checks of the host side of the C ABI: packing, sharding, helpers, symbol export,
and that the HIP path refuses to run without a GPU instead of falling back."""
import ctypes as C
import os

import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import abi
from tests.cases import SMALL_CASES, Case


def test_library_exports_every_declared_symbol(lib):
    import re
    hdr = open(os.path.join(os.path.dirname(abi.HERE), "include", "bamm_em.h")).read()
    declared = set(re.findall(r"\b(bamm_[a-z0-9_]+)\s*\(", hdr)) - {"bamm_allreduce_fn"}
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.bamm_version()


@pytest.mark.parametrize("spec", SMALL_CASES, ids=[d["name"] for d in SMALL_CASES])
def test_pack_roundtrip_matches_reference_kmers(spec, orc, lib):
    c = Case(**spec)
    seq, kmer, off, _ = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    assert pk.n_seqs == c.N and pk.total_len == int(off[-1])
    for K in (0, 1, 2, 3, 5, 10):
        y = pk.unpack_y(K)
        assert np.array_equal(y.astype(np.uint64), kmer % np.uint64(4 ** (K + 1))), K
    # the same packing straight from the alphabet codes, with the reference's rand() protocol
    pk2 = bm.PackedSeqs.from_codes(c.codes, c.in_off, c.ss, seed=42)
    assert np.array_equal(pk2.words, pk.words)
    assert pk2.n_exceptions == pk.n_exceptions
    assert np.array_equal(pk2.unpack_y(10), pk.unpack_y(10))
    if not c.ss:
        # double-strand: the strand separator is an N in every sequence (Sequence.cpp:10-13)
        assert pk.n_exceptions >= c.N


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_packing_does_not_depend_on_host_threads(threads, orc, lib):
    """bamm_pack_codes takes the rand() draws serially in the reference's order (Sequence.cpp:35-41) and
    encodes on several threads: same words, same exceptions, equal to the reference's kmer_."""
    c = Case(name="pk_thr", N=257, L0=70, W=8, K=2, n_frac=0.03, ragged=60)
    _, kmer, off, _ = c.encode(orc)
    lib.bamm_set_host_threads(threads)
    try:
        pk = bm.PackedSeqs.from_codes(c.codes, c.in_off, False, seed=42)
    finally:
        lib.bamm_set_host_threads(0)
    assert np.array_equal(pk.unpack_y(10).astype(np.uint64), kmer % np.uint64(4 ** 11))
    ref = bm.PackedSeqs.from_kmers(kmer, off)
    assert np.array_equal(pk.words, ref.words) and pk.n_exceptions == ref.n_exceptions


@pytest.mark.parametrize("spec", SMALL_CASES[:4], ids=[d["name"] for d in SMALL_CASES[:4]])
def test_bg_model_matches_oracle(spec, orc, lib):
    c = Case(**spec)
    _, kmer, off, _ = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    for K, alpha in ((0, [1.0]), (2, [1.0, 10.0, 10.0]), (4, [1.0, 10.0, 10.0, 10.0, 10.0])):
        alpha = np.array(alpha, np.float32)
        assert np.array_equal(pk.bg_model(K, alpha), orc.bg_model(kmer, off, K, alpha)), K


def test_pack_kmer_ptrs_equals_flat(orc, lib):
    c = Case(**SMALL_CASES[0])
    _, kmer, off, _ = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ptrs = (C.c_void_p * c.N)()
    lens = np.diff(off.astype(np.int64)).astype(np.uint64)
    kmer = np.ascontiguousarray(kmer)
    for n in range(c.N):
        ptrs[n] = kmer.ctypes.data + int(off[n]) * 8
    out = C.POINTER(abi.Packed)()
    abi.check(lib.bamm_pack_kmer_ptrs(ptrs, lens, c.N, C.byref(out)))
    pk2 = bm.PackedSeqs(out)
    assert np.array_equal(pk2.words, pk.words) and np.array_equal(pk2.unpack_y(10), pk.unpack_y(10))


def test_pack_empty_and_tiny(lib):
    pk = bm.PackedSeqs.from_kmers(np.zeros(0, np.uint64), np.zeros(1, np.uint64))
    assert pk.n_seqs == 0 and pk.total_len == 0
    # one sequence of one base (T = 3)
    pk = bm.PackedSeqs.from_kmers(np.array([3], np.uint64), np.array([0, 1], np.uint64))
    assert pk.words[0] == 3 << 30 and pk.n_exceptions == 0
    # 17 bases: second word starts with position 16
    km = np.zeros(17, np.uint64)
    roll = 0
    for i in range(17):
        roll = ((roll << 2) | (i % 4)) & (4 ** 11 - 1)
        km[i] = roll
    pk = bm.PackedSeqs.from_kmers(km, np.array([0, 17], np.uint64))
    assert len(pk.words) == 2 and pk.n_exceptions == 0
    assert np.array_equal(pk.unpack_y(10).astype(np.uint64), km)


def test_shard_ranges_partition_and_balance(orc, lib):
    c = Case("rag", N=1000, L0=120, W=12, K=2, ragged=60)
    _, kmer, off, _ = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    for world in (1, 2, 3, 8):
        cuts = [pk.shard_range(c.W, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == c.N
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0]
        work = [int((pk.lengths[b:e].astype(np.int64) - c.W + 1).sum()) for b, e in cuts]
        assert max(work) - min(work) <= 2 * (2 * (c.L0 + c.ragged) + 1)


def test_calculate_p_matches_oracle(orc, lib):
    rng = np.random.RandomState(3)
    for K, W, bgK in ((0, 5, 2), (2, 8, 2), (3, 6, 1), (4, 9, 2)):
        v = rng.random_sample(bm.v_size(K, W)).astype(np.float32)
        vbg = rng.random_sample(bm.bg_size(bgK)).astype(np.float32)
        assert np.array_equal(bm.calculate_p(v, vbg, bgK, K, W), orc.calculate_p(v, vbg, bgK, K, W))


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.BammError) as e:
        bm.Context(0)
    assert e.value.code == abi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(os.path.dirname(abi.HERE), "bammmotif2_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "bamm_oracle" not in txt, f


def test_argument_errors_do_not_crash(lib):
    """Bad arguments come back as BAMM_ERR_ARG with a message; nothing exits or segfaults."""
    import ctypes as C
    out = C.POINTER(abi.Packed)()
    assert lib.bamm_pack_kmers(np.zeros(1, np.uint64), np.zeros(2, np.uint64), 1, None) == abi.ERR_ARG
    assert b"null" in lib.bamm_last_error()
    b, e = C.c_uint64(), C.c_uint64()
    assert lib.bamm_shard_range(np.zeros(1, np.uint32), 1, 4, 3, 2, C.byref(b), C.byref(e)) == abi.ERR_ARG
    p = np.zeros(4, np.float32)
    assert lib.bamm_calculate_p(p, p, 0, 99, 1, p) == abi.ERR_ARG
    for name in ("bamm_em_estep", "bamm_em_mstep", "bamm_em_optimize_q", "bamm_em_accumulate", "bamm_em_update"):
        assert getattr(lib, name)(None) == abi.ERR_ARG, name
    assert lib.bamm_em_iterate(None, 1) == abi.ERR_ARG
    assert lib.bamm_em_destroy(None) == abi.OK and lib.bamm_seqs_destroy(None) == abi.OK and lib.bamm_ctx_destroy(None) == abi.OK
    assert lib.bamm_seqs_upload(None, None, 0, 0, None) == abi.ERR_ARG
    prm = abi.EmParams()
    lib.bamm_em_default_params(C.byref(prm))
    assert (prm.K, prm.bg_order, prm.max_iterations) == (2, 2, 1000) and abs(prm.q - 0.3) < 1e-7 and abs(prm.epsilon - 0.01) < 1e-9
    assert lib.bamm_v_size(2, 20) == 1680 and lib.bamm_v_offset(2, 20) == 400 and lib.bamm_bg_size(2) == 84


def test_integration_sources_compile_against_the_headers(tmp_path):
    """integration/ is code, not prose: the multi-GPU caller (sharded_em.cpp) compiles against include/bamm_em.h
    and links with libbamm_em.so; EM_hip.cpp / ScoreSeqSet_hip.cpp compile against the reference's UNCHANGED
    EM.h / ScoreSeqSet.h where that tree is present (development container; `make -C oracle ref_hip` builds the
    library tests/test_integration_gpu.py runs on the GPU box)."""
    import shutil, subprocess
    from bammmotif2_amd import build
    build.build_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++") or "g++"
    obj = tmp_path / "sharded_em.o"
    subprocess.check_call([cxx, "-std=c++17", "-pthread", "-Wall", "-Werror", "-fPIC", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(root, "integration", "sharded_em.cpp"), "-o", str(obj)])
    subprocess.check_call([cxx, "-shared", "-pthread", str(obj), "-L", os.path.dirname(build.LIB), "-lbamm_em",
                           "-Wl,--no-undefined", "-o", str(tmp_path / "libsharded.so")])
    ref = "/root/reference/src"
    if os.path.isdir(ref):
        for f in ("EM_hip.cpp", "ScoreSeqSet_hip.cpp"):
            subprocess.check_call([cxx, "-std=c++11", "-fopenmp", "-w", "-fPIC", "-I", ref, "-I", os.path.join(root, "include"),
                                   "-c", os.path.join(root, "integration", f), "-o", str(tmp_path / (f + ".o"))])


def test_seeded_packing_equals_the_libc_stream():
    """bamm_pack_codes_seeded (glibc's rand() restated, every host thread jumping to its share of the draws) against
    srand(seed) + bamm_pack_codes (one thread stepping through libc's rand()): the same packed set, N-rich sequences,
    both strand modes, several seeds (Sequence.cpp:35-41)."""
    import ctypes as C
    from bammmotif2_amd import abi
    lib = abi.load()
    rng = np.random.default_rng(17)
    N = 40000
    lens = rng.integers(12, 90, N)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = rng.integers(1, 5, int(off[-1])).astype(np.uint8)
    codes[rng.random(len(codes)) < 0.03] = 0
    for ss in (0, 1):
        for seed in (42, 7):
            bm.libc_srand(seed)
            a = C.POINTER(abi.Packed)()
            abi.check(lib.bamm_pack_codes(codes, off, N, ss, C.byref(a)))
            b = C.POINTER(abi.Packed)()
            abi.check(lib.bamm_pack_codes_seeded(codes, off, N, ss, seed, C.byref(b)))
            pa, pb = bm.PackedSeqs(a), bm.PackedSeqs(b)
            assert pa.n_exceptions == pb.n_exceptions and pa.n_exceptions > 1000
            assert np.array_equal(pa.words, pb.words)
            assert np.array_equal(pa.unpack_y(10), pb.unpack_y(10))
            pa.free(); pb.free()


def test_rand_stream_jump_ahead_equals_stepping_and_libc(lib):
    """csrc/glibc_rand.h: the negative sampler and the N draws enter glibc's one rand() stream in the middle
    (GlibcRandStream::jump multiplies the state by t^n mod (t^31 - t^28 - 1)).  jump(n) against n single steps and against
    libc's own srand()/rand() for several n and seeds; beyond 2^32, where stepping is out of reach, jumps must compose:
    jump(a) then b steps == jump(a + b)."""
    import ctypes as C

    def draws(seed, skip, mode, count=16):
        out = (C.c_int32 * count)()
        m = C.c_int()
        assert lib.bamm_rand_stream_draws(seed, skip, mode, count, out, C.byref(m)) == 0
        return list(out), bool(m.value)

    first, same = draws(42, 0, 1)
    assert same, "libc's rand() is not the restated generator on this host (the host paths then draw from libc, serially)"
    assert first == draws(42, 0, -1)[0]
    for seed in (42, 1, 20260101):
        for n in (1, 2, 30, 31, 33, 34, 35, 343, 344, 1000, 65537, 1_000_003):
            j, _ = draws(seed, n, 1)
            assert j == draws(seed, n, 0)[0], (seed, n)
            assert j == draws(seed, n, -1)[0], (seed, n)
    # beyond 2^32: the stream entered at a + b directly and at a, then b steps further (b small enough to step through)
    for a, b in ((2 ** 32 + 12345, 100_000), (2 ** 40 + 7, 31), (3 * 2 ** 33, 1)):
        direct, _ = draws(42, a + b, 1, 24)
        chained, _ = draws(42, a, 1, 24 + b if b <= 64 else 24)
        if b <= 64:
            assert direct == chained[b:b + 24], (a, b)
        # jump(a + b) against jump(b) applied to the state jump(a) left: the library's own composition
        lo, _ = draws(42, a + b - 8, 1, 32)
        assert lo[8:] == direct, (a, b)
