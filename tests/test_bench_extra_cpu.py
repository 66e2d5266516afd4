"""bench.py's in-kernel all-reduce extra on N > 1: whatever goes wrong on ONE rank, every rank walks the same sequence of
collectives (a rank that returned early would leave its peers in the next barrier and cost the run its headline) and all of
them report the same thing.  Fakes instead of a GPU: the control flow is what is tested."""
import threading
import types

import numpy as np
import pytest

import bench


class FakeEM:
    def __init__(self, rank, fail_at):
        self.rank, self.fail_at = rank, fail_at
        self._maybe("create")

    def _maybe(self, what):
        if self.fail_at == (self.rank, what):
            raise RuntimeError(f"rank {self.rank} fails at {what}")

    def set_comm(self, comm): pass
    def comm_mode(self): self._maybe("vote"); return (2, None)
    def set_kernel_timing(self, every): pass
    def iterate(self, n): self._maybe("iterate")
    def kernel_time(self): return (1.0, 10)
    def getV(self): self._maybe("read-back"); return np.arange(8, dtype=np.float32)
    def close(self): pass


@pytest.mark.parametrize("fail_at", [None, (1, "create"), (0, "vote"), (1, "iterate"), (0, "read-back")],
                         ids=["fine", "create", "vote", "iterate", "read-back"])
def test_peer_extra_keeps_the_ranks_in_step(fail_at):
    N = 2
    gate = threading.Barrier(N, timeout=20)
    tmp = [0.0] * N
    calls = [[] for _ in range(N)]
    out = [None] * N
    args = types.SimpleNamespace(steps=4, warmup=2, nseq=100, timing_every=8)
    wl = dict(W=4, K=1, vbg=None, A=None, v0=None, q=0.3)

    def worker(r):
        def barrier():
            calls[r].append("barrier"); gate.wait()

        def rmax(x):
            calls[r].append("max"); tmp[r] = x; gate.wait(); m = max(tmp); gate.wait(); return m

        bm = types.SimpleNamespace(EM=lambda *a, **k: FakeEM(r, fail_at))
        ctx = types.SimpleNamespace(set_tuning=lambda **k: None)
        out[r] = bench.peer_allreduce_extra(bm, ctx, None, None, wl, args, barrier, lambda: None, rmax)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in ts: t.start()
    for t in ts: t.join(30)
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective its peer never entered"
    assert calls[0] == calls[1]                                   # the same collectives in the same order
    if fail_at is None:
        assert all(isinstance(o["ms_per_step_peer_allreduce"], float) and "peer_allreduce" in o for o in out)
    else:
        assert all(isinstance(o["ms_per_step_peer_allreduce"], str) and o["ms_per_step_peer_allreduce"].startswith("unavailable") for o in out)
        assert f"rank {fail_at[0]} fails at {fail_at[1]}" in out[fail_at[0]]["ms_per_step_peer_allreduce"]
