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


def test_watchdog_ends_a_phase_that_never_returns(tmp_path):
    """bench.py's per-phase wall-clock cap: a child process started before the GPU is touched kills the bench when a phase
    (a collective that never returns) outlives its cap, and says which phase it was."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog('rank 0 of 1', 1.0)\n"
            "wd.phase('a quick phase', 30); wd.phase('the collective that hangs', 0.5)\n"
            "time.sleep(60)\n") % root
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert r.returncode == -9 and time.time() - t0 < 30                     # SIGKILL from the watchdog, long before the sleep ends
    assert "phase 'the collective that hangs' did not end within its wall-clock cap" in r.stderr and "rank 0 of 1" in r.stderr
    # a bench that ends by itself takes its watchdog with it, silently
    code_ok = ("import sys; sys.path.insert(0, %r); import bench\n"
               "wd = bench.Watchdog('x', 1.0); wd.phase('p', 5); wd.close(); print('done')\n") % root
    r = subprocess.run([sys.executable, "-c", code_ok], capture_output=True, text=True, timeout=50)
    assert r.returncode == 0 and r.stdout.strip() == "done" and "watchdog" not in r.stderr


@pytest.mark.parametrize("fail_at,differ", [(None, False), ((1, "create"), False), ((0, "iterate"), False), (None, True)],
                         ids=["fine", "create", "iterate", "models_differ"])
def test_selftest_comm_keeps_the_ranks_in_step_and_names_the_failure(fail_at, differ):
    N = 2
    gate = threading.Barrier(N, timeout=20)
    tmp = [None] * N
    calls = [[] for _ in range(N)]
    out = [None] * N
    args = types.SimpleNamespace(nseq=100)
    wl = dict(W=4, K=1, vbg=None, A=None, v0=None, q=0.3)

    class EM(FakeEM):
        def getV(self):
            v = super().getV()
            return v + (self.rank if differ else 0)

    def worker(r):
        def gather(x):
            calls[r].append("gather"); tmp[r] = x; gate.wait(); got = list(tmp); gate.wait(); return got

        bm = types.SimpleNamespace(EM=lambda *a, **k: EM(r, fail_at))
        ctx = types.SimpleNamespace(set_tuning=lambda **k: None)
        out[r] = bench.selftest_comm(bm, ctx, None, None, wl, args, gather, lambda: None)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in ts: t.start()
    for t in ts: t.join(30)
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective its peer never entered"
    assert calls[0] == calls[1] and out[0] == out[1]              # the same collectives, the same verdict on every rank
    for kind in ("rccl", "peer"):
        if fail_at is None and not differ:
            assert out[0][kind]["ok"] and len(out[0][kind]["model_sha"]) == 16
        elif differ:
            assert not out[0][kind]["ok"] and "models differ" in out[0][kind]["why"]
        else:
            assert not out[0][kind]["ok"] and f"rank {fail_at[0]}" in out[0][kind]["why"] and fail_at[1].split()[0] in out[0][kind]["why"]


def test_choose_headline_takes_the_faster_region_only_when_the_in_kernel_one_is_sound():
    args = types.SimpleNamespace(steps=10, headline_collective="auto")
    peer_ok = {"ms_per_step_peer_allreduce": 0.12, "peer_allreduce": {"kernel_us": 118.0, "dt_s": 0.0012, "model_sha": "aa"}}
    # faster and on the collective's model: the in-kernel region is the headline, the collective's timing stays beside it
    ex = dict(peer_ok)
    dt, kms, n, kind = bench.choose_headline(args, ex, 0.0015, 1.4, 10, "rccl (...)", "aa")
    assert ex["headline_collective"] == "peer" and dt == 0.0012 and abs(kms - 1.18) < 1e-9 and n == 10 and "in-kernel" in kind and "rccl (...)" in kind
    assert ex["ms_per_step_rccl"] == pytest.approx(0.15)
    # slower: the collective's region stays
    ex = dict(peer_ok, ms_per_step_peer_allreduce=0.2)
    assert bench.choose_headline(args, ex, 0.0015, 1.4, 10, "rccl (...)", "aa")[0] == 0.0015 and ex["headline_collective"] == "rccl"
    # another model than the collective's: never, and the line says why
    ex = dict(peer_ok)
    assert bench.choose_headline(args, ex, 0.0015, 1.4, 10, "rccl (...)", "bb")[0] == 0.0015
    assert ex["headline_collective"] == "rccl" and "another model" in ex["ms_per_step_peer_allreduce"]
    # unavailable, or switched off
    ex = {"ms_per_step_peer_allreduce": "unavailable: x"}
    assert bench.choose_headline(args, ex, 0.0015, 1.4, 10, "k", "aa") == (0.0015, 1.4, 10, "k") and ex["headline_collective"] == "rccl"
    ex = dict(peer_ok)
    assert bench.choose_headline(types.SimpleNamespace(steps=10, headline_collective="rccl"), ex, 0.0015, 1.4, 10, "k", "aa")[0] == 0.0015
    ex = dict(peer_ok, ms_per_step_peer_allreduce=0.2)
    assert bench.choose_headline(types.SimpleNamespace(steps=10, headline_collective="peer"), ex, 0.0015, 1.4, 10, "k", "aa")[0] == 0.0012
