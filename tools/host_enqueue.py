import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
import numpy as np, torch, torch.distributed as dist
import bammmotif2_amd as bm
from bammmotif2_amd import synth
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, K, L0, N = 20, 2, 200, 125000
pwm = synth.make_pwm(W, 1234)
codes, in_off = synth.make_sequences(N, L0, pwm, 1234, plant_frac=0.5)
packed = bm.PackedSeqs.from_codes(codes, in_off, False, seed=42)
A = synth.alpha_matrix(synth.default_alpha(K), W)
vbg = packed.bg_model(2, np.array([1.0, 10.0, 10.0], np.float32))
v0 = synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K)
ts = torch.cuda.Stream(device=0)
ctx = bm.Context(0, ts.cuda_stream)
seqs = bm.SeqSet(ctx, packed)
em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, bg_order=2, max_iterations=1000)
_, n = em.reduce_buffer()
red = torch.zeros(n, dtype=torch.int64, device="cuda:0"); torch.cuda.synchronize()
em.set_reduce_buffer(red.data_ptr(), n)
def cb(_p, _n, _s):
    dist.all_reduce(red); return 0
for use_cb in (False, True):
    em.set_allreduce(cb if use_cb else None)
    with torch.cuda.stream(ts):
        em.iterate(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); em.iterate(300); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("callback" if use_cb else "plain", "host enqueue us/iter %.1f" % ((t1 - t0) / 300 * 1e6), "total us/iter %.1f" % ((t2 - t0) / 300 * 1e6))
dist.destroy_process_group()
