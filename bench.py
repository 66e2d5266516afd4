#!/usr/bin/env python3
"""EM-refinement throughput on MI355X: BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W

A "step" is one EM iteration (EStep + MStep + updateV, /root/reference/src/refinement/EM.cpp:
81-128) over the whole synthetic set: 1M x 200 bp, W = 20, k = 2, K_bg = 2, double-stranded
(L = 401), seeded from the planted PWM.  The set is resident in HBM
(2-bit packed) before the timed region.  With N > 1 the SAME 1M sequences are sharded over the
ranks ("strong" scaling, the shape BASELINE.json names) and the fused count buffer is all-reduced
over RCCL once per iteration.  Two launch forms, same library path (csrc/comm.cpp):
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE in the
    environment): one process per GPU, ncclCommInitRank, torch.distributed only carries the id and
    the barriers;
  * plain `python bench.py --gpus N`: ONE process, one context + host thread per GPU,
    ncclCommInitAll (what `BaMMmotif --gpus N` and integration/sharded_em.cpp do).  Fewer visible
    devices than N is an error, never a silent single-GPU run.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     algorithmic HBM bytes of the sequence kernel / its HIP-event duration vs 8 TB/s
  cpu_baseline the reference's own EM (oracle/_ref, OpenMP) timed on the host cores on a
               bounded sample of the same workload (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def algorithmic_bytes(positions: int, windows: int, sliced: bool) -> float:
    """SURVEY.md 8(d).  Fused path (tables fit one CU's LDS, k <= 3): the 2-bit base is the only
    per-position HBM read, 0.25 B/position.  Split path (k >= 4): the sequence is read by the E pass and
    by the M pass (2 x 0.25 B/position) and the responsibilities make one round trip as f32 (8 B/window)."""
    return (0.5 * positions + 8.0 * windows) if sliced else 0.25 * positions


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nseq", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=200, dest="L0")
    ap.add_argument("--width", type=int, default=20)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="a named BASELINE.json configuration instead of --nseq/--len/--width/--order: c2 = 50k x 200 bp W=20 k=2, "
                         "c3 = 1M x 200 bp W=20 k=2 (the default, the headline), c4 = 1M x 500 bp W=30 k=4 (12 warm-up + 12 timed passes "
                         "unless --steps/--warmup are given), c5 = the shape of config 5's runs, 200k x 200 bp W=20 k=2")
    ap.add_argument("--ss", action="store_true", help="single strand (default: both strands, L = 2*L0+1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=200000, help="sequences in the CPU-baseline sample (~15 s of host work)")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--group-layout", type=int, default=-1,
                    help="table layout of the grouped kernel (bamm_ctx_set_tuning group_layout: 0..3 uniform rows, "
                         "8 mixed rows); default: the planner's choice")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the RCCL all-reduce path even with one rank (self-test)")
    ap.add_argument("--torch-allreduce", action="store_true",
                    help="all-reduce through torch.distributed.all_reduce from a Python callback instead of the "
                         "library's own RCCL call (bamm_em_set_comm), which is the default with the nccl backend")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold-start / optimize()-mode figures")
    ap.add_argument("--local-ranks", action="store_true",
                    help="in-process form only: --gpus N ranks as N contexts on device 0 with the host-staged sum of "
                         "bamm_comm_init_local instead of RCCL (self-test of the N>1 logic on a 1-GPU box; never "
                         "used for reported numbers)")
    ap.add_argument("--shm-comm", action="store_true",
                    help="with --dist-backend gloo under a launcher: the library's host-staged communicator between processes "
                         "(bamm_comm_init_shm) instead of RCCL -- a rehearsal of the one-process-per-rank form (the in-kernel "
                         "all-reduce through hipIpc handles included) with all ranks on ONE device; never used for reported numbers")
    ap.add_argument("--timing-every", type=int, default=None,
                    help="HIP events of the timed call: -1 = ONE pair around all its passes (every pass covered, launch gaps "
                         "included, nothing added between passes: the default on 1 GPU); n >= 1 = a pair around every n-th pass "
                         "(a pair takes 7-8 us of stream time, profiles/r04_timing_every_cost.txt; 8 is the default with "
                         "N > 1, where the interval of -1 would include the collective); 0 = none")
    ap.add_argument("--headline-collective", default="auto", choices=["auto", "rccl", "peer"],
                    help="N > 1: which of the two timed regions the line's value / ms_per_step come from.  Both regions are the same "
                         "(W warm-up passes, exactly K timed passes between barriers, max over ranks) and end on the same model bit for "
                         "bit; one sums over the ranks with the RCCL collective behind every pass, the other inside the sequence kernel's "
                         "tail over peer-mapped inboxes (no collective launch).  auto = the faster one, provided the in-kernel path passed "
                         "the self-test and its own region; both timings are always reported (ms_per_step_rccl, ms_per_step_peer_allreduce)")
    ap.add_argument("--no-selftest-comm", action="store_true",
                    help="N > 1: skip the communicator self-test in front of the timed region (3 passes through the RCCL collective and "
                         "3 through the in-kernel all-reduce on handles of their own, model hashes compared across the ranks)")
    ap.add_argument("--phase-cap-scale", type=float, default=1.0,
                    help="scales the wall-clock caps of the watchdog (a child process started before anything touches the GPU; it "
                         "kills this process when a phase -- set-up, communicator, self-test, warm-up, timed region, extras -- "
                         "does not end: a collective that never returns costs one line, not the launcher's own time-out); 0 = no watchdog")
    ap.add_argument("--allreduce-iters", type=int, default=200, help="N > 1: bare all-reduces of the accumulator timed for `attribution`")
    ap.add_argument("--no-fused-update", action="store_true",
                    help="a k_update launch after every pass instead of the update fused into the next pass's kernel")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the fused buffer through the host: lets 2 ranks share ONE GPU (self-test of "
                         "the N>1 logic on a 1-GPU box; never used for reported numbers)")
    args = ap.parse_args()
    if args.config:
        given = set(a.split("=")[0] for a in sys.argv[1:] if a.startswith("--"))
        shape = {"c2": (50_000, 200, 20, 2), "c3": (1_000_000, 200, 20, 2), "c4": (1_000_000, 500, 30, 4), "c5": (200_000, 200, 20, 2)}[args.config]
        args.nseq, args.L0, args.width, args.order = shape
        if args.config == "c4":                              # profiles/r04_c4_bench.json's timed region
            if "--steps" not in given:
                args.steps = 12
            if "--warmup" not in given:
                args.warmup = 12
    if args.timing_every is None:
        args.timing_every = -1 if args.gpus <= 1 else 8
    return args


def usable_cpus():
    """CPUs this process may actually burn: the affinity mask capped by the cgroup CPU quota (a
    container can see 256 logical CPUs and be throttled to 16; OpenMP on all 256 then crawls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, round(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(codes, in_off, W, K, v0, alpha, q, sample, iters, ss):
    """The reference's EM::EStep/MStep (compiled in place into oracle/_ref) on the host cores."""
    import oracle
    n = min(sample, len(in_off) - 1)
    sub_off = np.ascontiguousarray(in_off[: n + 1])
    sub_codes = np.ascontiguousarray(codes[: int(sub_off[-1])])
    cores = usable_cpus()
    alpha_bg = np.array([1.0, 10.0, 10.0], np.float32)
    if oracle.have_reference():
        R = oracle.Reference()
        R.set_threads(cores)
        S = R.session(sub_codes, sub_off, ss, 42)
        bg, _ = S.bg(2, alpha_bg)
        m = S.motif(W, K, alpha, bg, q, v0)
        em = S.em(m, bg, False, False)
        S.R.ref_em_estep(em); S.R.ref_em_mstep(em)          # warm-up pass
        t0 = time.perf_counter()
        for _ in range(iters):
            S.R.ref_em_estep(em); S.R.ref_em_mstep(em)
        dt = time.perf_counter() - t0
        positions = int(S.off[-1])
        kind = "reference"
        # the reference's own default is --threads 4 (Global.cpp:96); its CAS float adds on a
        # 5 KB table contend badly on many-core hosts, so that figure is reported as well
        R.set_threads(4)
        t1 = time.perf_counter()
        S.R.ref_em_estep(em); S.R.ref_em_mstep(em)
        extra = {"positions_per_s_at_reference_default_4_threads": positions / (time.perf_counter() - t1)}
    else:
        O = oracle.Oracle()
        O.set_threads(cores)
        _, kmer, off = O.encode_set(sub_codes, sub_off, ss, 42)
        vbg = O.bg_model(kmer, off, 2, alpha_bg)
        from bammmotif2_amd import synth
        A = synth.alpha_matrix(alpha, W)
        O.optimize(kmer, off, K, W, 2, vbg, A, v0, q, epsilon=0.0, max_iter=1)
        t0 = time.perf_counter()
        O.optimize(kmer, off, K, W, 2, vbg, A, v0, q, epsilon=0.0, max_iter=iters)
        dt = time.perf_counter() - t0
        positions = int(off[-1])
        kind = "port"
        extra = {}
    return {**extra, "value": positions * iters / dt, "unit": "positions/s", "cores": cores, "kind": kind,
            "iterations_per_s_on_sample": iters / dt,
            "sample": f"first {n} sequences of the same set, {iters} timed EM iterations "
                      f"(EStep+MStep), OpenMP on {cores} host threads, -O2"}


def lds_roofline(tj, avg_kernel_s):
    """the sequence kernel's LDS wave-instructions per second against the rate an
    LDS-only loop of the same instruction mix reaches (tools/lds_mix_bench.hip, run on the same box by
    tools/pmc_run.sh; its result travels in the PMC summary).  None without a matching profile."""
    d = tj.get("derived") or {}
    insts = d.get("lds_wave_instr_per_launch")
    if not insts or not avg_kernel_s:
        return None
    out = {"wave_instr_per_s": insts / avg_kernel_s,
           "wave_instr_per_launch": insts,
           "lds_busy_frac": d.get("lds_busy_frac"),
           "bank_conflict_frac": d.get("lds_bank_conflict_frac"),
           "valu_busy_frac": d.get("valu_busy_frac")}
    peak = (tj.get("lds_mix_bench") or {}).get("wave_instr_per_s")
    if peak:
        out.update({"peak_from_lds_bench": peak, "frac": out["wave_instr_per_s"] / peak,
                    "peak_is": (tj.get("lds_mix_bench") or {}).get("what")})
    return out


def lds_roofline_sliced(tj):
    """Column-sliced path (k >= 4): the E pass (k_em_seq) and the M slices (k_m_list) each against the rate an LDS-only loop
    of their own instruction mix reaches (tools/lds_c4_bench.hip), from the PMC summary of the same command: LDS
    wave-instructions per launch (SQ_INSTS_LDS) over the kernel's average duration (rocprofv3 --kernel-trace --stats; both
    averages run over the same dispatches, the pass's idle flavour included on both sides).  None without such a profile."""
    per, ceil = tj.get("per_kernel") or {}, tj.get("lds_c4_bench") or {}
    out = {}
    for key, tag, peak in (("e_pass", "k_em_seq", ceil.get("e_pass_wave_instr_per_s")), ("m_list", "k_m_list", ceil.get("m_list_wave_instr_per_s"))):
        k = next((v for n, v in per.items() if tag in n), None)
        if not k or not k.get("lds_wave_instr_per_launch") or not k.get("avg_duration_us") or not peak:
            continue
        rate = k["lds_wave_instr_per_launch"] / (k["avg_duration_us"] * 1e-6)
        out[key] = {"wave_instr_per_s": rate, "wave_instr_per_launch": k["lds_wave_instr_per_launch"], "avg_duration_us": k["avg_duration_us"],
                    "peak_from_lds_bench": peak, "frac": rate / peak}
    if not out:
        return None
    out["peak_is"] = ceil.get("what")
    out["frac"] = min(v["frac"] for v in out.values() if isinstance(v, dict))      # the kernel furthest from its ceiling
    return out


def workload(args):
    """The synthetic set of SURVEY.md 8(d) (seed 1234) and the seed model, packed once on the host."""
    import bammmotif2_amd as bm
    from bammmotif2_amd import synth
    W, K, L0 = args.width, args.order, args.L0
    pwm = synth.make_pwm(W, 1234)
    codes, in_off = synth.make_sequences(args.nseq, L0, pwm, 1234, plant_frac=0.5)
    packed = bm.PackedSeqs.from_codes(codes, in_off, args.ss, seed=42)
    alpha = synth.default_alpha(K)
    return dict(W=W, K=K, L0=L0, pwm=pwm, codes=codes, in_off=in_off, packed=packed, alpha=alpha,
                A=synth.alpha_matrix(alpha, W), vbg=packed.bg_model(2, np.array([1.0, 10.0, 10.0], np.float32)),
                v0=synth.bamm_from_pwm((0.7 * pwm + 0.3 * 0.25).astype(np.float32), K), q=0.3)


def kernel_label(em, K):
    g_seqs, o_seqs, _ = em.plan()
    mixed_seqs = em.plan_mixed()
    name = ("k_em_seq (E pass, compacted lists of the non-zero windows) + k_m_list x slices (column-sliced M pass)" if K >= 4 else
            "k_em_mix (fused E+M, grouped columns: 5-mer rows, the last groups on 6-mer rows)" if o_seqs == 0 and mixed_seqs == g_seqs else
            "k_em_grp (fused E+M, grouped columns)" if o_seqs == 0 else
            "k_em_seq (fused E+M)" if g_seqs == 0 else "k_em_grp + k_em_seq (fused E+M)")
    return name, bool(mixed_seqs)


def git_head():
    if os.environ.get("BAMM_COMMIT"):                # the GPU box has no .git: the launcher passes the hash
        return os.environ["BAMM_COMMIT"]
    try:
        import subprocess
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def pmc_summary(local_positions, K, avg_kernel_s):
    """PMC figures of the same command from an EARLIER run (tools/pmc_run.sh -> tools/summarize_pmc.py: FETCH_SIZE / WRITE_SIZE
    and the LDS counters need rocprofv3 passes of their own, they cannot be taken inside this process), matched by
    workload.  The result says which committed file they come from."""
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json")), reverse=True):   # newest round first
        try:
            tj = json.load(open(tpath))
            if tj.get("positions_per_launch") == local_positions and tj.get("order", 2) == K:
                src = {"file": os.path.relpath(tpath, ROOT), "measured_at_commit": tj.get("commit"),
                       "how": "rocprofv3 --pmc passes of `python bench.py` (tools/pmc_run.sh), not this run"}
                lds = lds_roofline_sliced(tj) if K >= 4 else lds_roofline(tj, avg_kernel_s)
                return tj.get("hbm_bytes_per_launch"), lds, src
        except Exception:
            pass
    return None, None, None


def report(args, wl, world, dt, kernel_ms, launches, local_positions, local_windows, kernel_name, mixed, llh_last,
           allreduce_kind, extras, ranks=None, launcher="single process"):
    packed, W, K, L0 = wl["packed"], wl["W"], wl["K"], wl["L0"]
    total_positions = packed.total_len
    total_windows = int((packed.lengths.astype(np.int64) - W + 1).sum())
    its = args.steps / dt
    avg_kernel_s = kernel_ms / launches * 1e-3 if launches else float('nan')
    sliced = K >= 4
    alg_bytes = algorithmic_bytes(local_positions, local_windows, sliced)
    achieved = alg_bytes / avg_kernel_s / 1e9
    traffic, lds, pmc_src = pmc_summary(local_positions, K, avg_kernel_s)
    L = int(packed.lengths[0]) if packed.n_seqs else 0
    shape = f"{args.nseq // 1000000}M" if args.nseq % 1000000 == 0 else (f"{args.nseq // 1000}k" if args.nseq % 1000 == 0 else str(args.nseq))
    out = {
        "metric": f"EM seq-positions/sec (and iterations/sec), {shape}x{L0}bp k={K} W={W}",
        "value": total_positions * its,
        "unit": "positions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "iterations_per_s": its,
        "windows_per_s": total_windows * its,
        "timed_region": f"passes {args.warmup + 1}-{args.warmup + args.steps} of one handle through iterate() (fixed budget, SURVEY H5); "
                        "`from_seed` beside it: the first passes from the seed model, as iterate() and as optimize()",
        "config": {"workload": f"{args.nseq}x{L0}bp {'single' if args.ss else 'double'}-strand "
                               f"(L={L}), W={W}, k={K}, K_bg=2, --EM, fixed iteration budget",
                   "n_seqs": args.nseq, "seq_len": L0, "W": W, "k": K,
                   "parallelism": f"sequences sharded over {world} GPU(s), 1 all-reduce of "
                                  f"{4 ** (K + 1) * W + 3} int64 words per iteration" if world > 1 else "1 GPU"},
        # fused path: SURVEY's algorithmic bytes over the kernel time.  Sliced path (k >= 4): the bytes that MOVED (PMC) over
        # the kernel time -- SURVEY's 8.3 GB there is a dense round trip of r the path no longer makes (lists between E and
        # M), dividing it by the time would flatter; the algorithmic figure stays beside it under its own name
        "roofline": {"bound": "hbm",
                     "achieved": (traffic / avg_kernel_s / 1e9) if (sliced and traffic) else achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ((traffic / avg_kernel_s / 1e9) if (sliced and traffic) else achieved) / HBM_PEAK_GBS,
                     "frac_is": ("moved bytes (PMC traffic of the committed profile) / this run's kernel time / peak" if (sliced and traffic) else
                                 "algorithmic bytes / kernel time / peak"),
                     "algorithmic_achieved": achieved, "algorithmic_frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": pmc_src,
                     "kernel": kernel_name, "avg_kernel_ms": avg_kernel_s * 1e3,
                     "avg_kernel_ms_is": extras.pop("avg_kernel_ms_is", None) or ("one pair of HIP events around ALL passes of the timed call, divided by their number: every "
                                          "pass covered, the launch gaps between passes included (<= ms_per_step by construction)"
                                          if args.timing_every < 0 else
                                          "no pass timed" if args.timing_every == 0 else
                                          f"mean over every {args.timing_every}-th pass of the timed call, a pair of HIP events around each "
                                          "(7-8 us of stream time per pair)"),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "algorithmic_bytes_rule": ("0.5 B/position + 8 B/window (sequence read by the E and the M pass, "
                                                "r written and read once as f32)" if sliced else
                                                "0.25 B/position (2-bit base, read once)"),
                     "moved_bytes_frac": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "lds": lds, "lds_source": pmc_src if lds else None,
                     "note": ("E pass + M slices of one iteration, timed together; `moved_bytes_frac` = PMC traffic / time / peak "
                              "(the lists between E and M move fewer bytes than SURVEY's dense round trip of r)" if sliced else
                              "the fused kernel is LDS / VALU issue bound (DESIGN.md section 4): "
                              "`lds` prices it against a measured LDS ceiling" +
                              ("; `traffic` = the 2-bit stream and records (~1.5 x the algorithmic bytes) plus the "
                               "fix lanes' log of virtual-row sums, written during the pass and read back once per "
                               "block (mixed rows and K = 3 have no LDS left for all of those bins)"
                               if (mixed or K == 3) else ""))},
        "allreduce": allreduce_kind,
        "launcher": launcher,
        "ranks": ranks,
        "commit": git_head(),
        "parity": "learned v within 1e-5 of the reference on its own fixtures up to ~2k sequences; beyond that the "
                  "reference's fp32 CAS accumulation is itself 4e-5 (10k) to 4e-4 (200k) off exact arithmetic, while "
                  "this path stays within 3e-7 of the fp64 restatement at every size (integer accumulation; "
                  "profiles/r04_deviation_vs_fp64.txt, profiles/r04_parity_margins.txt, tests/test_golden_gpu.py, tests/test_fullsize_parity_gpu.py)",
        "llh_last": llh_last,
        **extras,
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(wl["codes"], wl["in_off"], W, K, wl["v0"], wl["alpha"], wl["q"], args.cpu_sample,
                                               args.cpu_iters, args.ss)
            # a ratio of the GPU's whole-set rate to the CPU's rate on the SAMPLE (positions/s does not depend
            # on the set size for the reference's loops): a stated baseline, not a kernel-quality figure
            out["cpu_baseline"]["gpu_over_cpu_sample_rate"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # the baseline is a reported number, never a reason to lose the line
            out["cpu_baseline"] = {"value": None, "unit": "positions/s", "cores": usable_cpus(),
                                   "kind": "unavailable", "sample": repr(e)}
    return out


def attribution(step_us, kernel_us, allreduce_us, args):
    """Where a sharded iteration's time goes, so that a scaling shortfall can be read off the first multi-GPU run: the
    sequence kernel (HIP events around it, every pass of the timed call), the bare collective (a loop of all-reduces of
    the accumulator alone, timed after the run), and the rest -- launch gaps, the fused update's prologue, waiting for the
    slowest rank.  Per rank and the max over ranks; `fixed_us` = step - kernel - allreduce from the maxima."""
    k = max(kernel_us)
    ar = [a for a in allreduce_us if a is not None]
    a = max(ar) if ar else None
    return {"step_us": step_us, "kernel_us": k, "allreduce_us": a,
            "fixed_us": step_us - k - (a or 0.0),
            "kernel_us_per_rank": list(kernel_us), "allreduce_us_per_rank": list(allreduce_us),
            "allreduce_is": f"{args.allreduce_iters} back-to-back all-reduces of the accumulator on the kernels' stream, HIP events "
                            "around the loop (bamm_comm_time_allreduce); inside an iteration the collective also waits for the slowest rank's kernel",
            "kernel_is": (f"HIP events around the sequence kernel(s) of every {args.timing_every}-th pass of the timed call" if args.timing_every > 0 else
                          "one HIP event pair around all passes of the timed call (includes the collective)") + " (the fused model update is its prologue)"}


def peer_allreduce_extra(bm, ctx, seqs, comm, wl, args, barrier, sync, reduce_max):
    """After the headline (which always runs the RCCL collective): the same timed region with the pass's all-reduce INSIDE
    the sequence kernels (include/bamm_em.h: bamm_em_comm_mode 2, default off) -- a number beside the headline, or the
    reason there is none.  barrier(): all ranks; sync(): this rank's stream; reduce_max(x): max over ranks."""
    W, K = wl["W"], wl["K"]
    em = None
    why = []                                                 # this rank's failures; every rank walks the SAME sequence of collectives

    def step(what, f):
        """A local step that may fail: the failure is kept, never raised -- a rank that left early would leave its peers in
        the next barrier."""
        if why:
            return None
        try:
            return f()
        except Exception as e:
            why.append(f"{what}: {e!r}")
            return None

    def agreed():
        """Collective: has any rank failed so far?"""
        return reduce_max(1.0 if why else 0.0) > 0.0

    def make():
        ctx.set_tuning(peer_allreduce=1)
        e = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=args.steps + args.warmup + 8,
                  n_seqs_bound=args.nseq)
        e.set_comm(comm)
        return e

    try:
        em = step("create", make)
        if agreed():                                         # before the library's own vote: all ranks take part in it, or none
            return {"ms_per_step_peer_allreduce": "unavailable: " + (why[0] if why else "another rank could not set it up")}
        res = step("vote", lambda: em.comm_mode())           # collective: the ranks vote (same answer everywhere)
        mode, note = res if res else (0, None)
        if agreed() or mode != 2:
            return {"ms_per_step_peer_allreduce": "unavailable: " + (why[0] if why else (note or "mode %d" % mode))}
        step("timing", lambda: em.set_kernel_timing(-1))     # ONE pair of events around all passes: no collective sits between them here
        step("warm-up", lambda: em.iterate(args.warmup))     # a block that waited in vain ends its launch with an error: bounded
        step("sync", sync); barrier()
        t0 = time.perf_counter()
        step("timed call", lambda: em.iterate(args.steps))
        step("sync", sync); barrier()
        dt = reduce_max(time.perf_counter() - t0)
        kt = step("kernel time", lambda: em.kernel_time()) or (0.0, 0)
        v = step("read-back", lambda: em.getV())              # raises BAMM_ERR_COMM if a block waited in vain
        k_us = reduce_max(kt[0] / max(kt[1], 1) * 1e3)
        if agreed():
            return {"ms_per_step_peer_allreduce": "unavailable: " + (why[0] if why else "another rank failed")}
        return {"ms_per_step_peer_allreduce": dt / args.steps * 1e3,
                "peer_allreduce": {"kernel_us": k_us, "dt_s": dt, "model_sha": hashlib.sha256(v.tobytes()).hexdigest()[:16],
                                   "what": "every pass but the call's last hands its sums to the peers from the kernel's own epilogue (last block, "
                                           "system-scope stores into peer-mapped inboxes) and collects theirs there; kernel_us then includes the "
                                           "wait for the slowest rank"}}
    finally:
        try:
            ctx.set_tuning(peer_allreduce=0)
            if em is not None:
                em.close()
        except Exception:
            pass


WATCHDOG_SRC = r"""
import os, select, signal, sys, time
pid, tag = int(sys.argv[1]), sys.argv[2]
deadline, phase, buf = None, "start", b""
while True:
    timeout = None if deadline is None else max(0.0, deadline - time.time())
    ready, _, _ = select.select([0], [], [], timeout)
    if ready:
        chunk = os.read(0, 4096)             # unbuffered: select() must see everything that is still unread
        if not chunk:
            sys.exit(0)                      # the pipe closed: the bench ended (or died) by itself
        buf += chunk
        while b"\n" in buf:
            line, buf = buf.split(b"\n", 1)
            cap, phase = line.decode().split(" ", 1)
            deadline = None if float(cap) <= 0 else time.time() + float(cap)
    elif deadline is not None and time.time() >= deadline:
        sys.stderr.write("[bench watchdog] %s: phase '%s' did not end within its wall-clock cap -- killing pid %d\n" % (tag, phase, pid))
        sys.stderr.flush()
        try:
            os.kill(pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        sys.exit(3)
"""


class Watchdog:
    """A hard wall-clock cap per phase, enforced from OUTSIDE the process that may be stuck: a child started before this
    process touches the GPU (a fresh interpreter, never an exec of a GPU process) reads `cap phase` lines from a pipe and
    kills this process (SIGKILL: the launcher sees a failed rank and ends the others) when a phase outlives its cap."""
    def __init__(self, tag, scale=1.0):
        import subprocess
        self.scale = scale
        self.proc = None
        if scale > 0:
            try:
                self.proc = subprocess.Popen([sys.executable, "-c", WATCHDOG_SRC, str(os.getpid()), tag], stdin=subprocess.PIPE, text=True)
            except Exception as e:                           # no watchdog is not a reason to lose the run
                print(f"[bench] no watchdog: {e!r}", file=sys.stderr)
                self.proc = None

    def phase(self, name, cap_s):
        if self.proc is not None and self.proc.poll() is None:
            try:
                self.proc.stdin.write("%g %s\n" % (cap_s * self.scale, name))
                self.proc.stdin.flush()
            except (BrokenPipeError, OSError):
                pass

    def close(self):
        if self.proc is not None:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=5)
            except Exception:
                pass
            self.proc = None


def topology(bm, devices):
    """What a first multi-GPU run wants on record beside its numbers: where each rank's device sits and who reaches whom."""
    try:
        n = bm.device_count()
        return {"visible_devices": n, "pci_bus_id": [bm.device_pci_bus_id(d) for d in range(n)],
                "can_access_peer": bm.peer_access_matrix(), "ranks_on_devices": list(devices)}
    except Exception as e:                                   # a record, never a reason to lose the line
        return {"error": repr(e)}


def selftest_comm(bm, ctx, seqs, comm, wl, args, gather, sync):
    """In front of the timed region, N > 1: three passes from the seed through the RCCL collective and three through the
    in-kernel all-reduce, each on a handle of its own, and the models' hashes compared across the ranks -- a collective
    that sums nothing, inboxes that map but do not deliver, ranks that disagree: one line that says which and why.
    gather(x): list of every rank's x (a collective).  Every rank walks the same sequence of collectives whatever fails
    locally (a failure is kept, not raised)."""
    W, K = wl["W"], wl["K"]
    out = {}
    for kind in ("rccl", "peer"):
        why, em, sha = [], None, None

        def step(what, f):
            if why:
                return None
            try:
                return f()
            except Exception as e:
                why.append(f"{what}: {e!r}")
                return None

        def make():
            ctx.set_tuning(peer_allreduce=1 if kind == "peer" else 0)
            e = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=8, n_seqs_bound=args.nseq)
            e.set_comm(comm)
            return e

        try:
            em = step("create", make)
            fails = gather(why[0] if why else None)
            if any(fails):
                out[kind] = {"ok": False, "why": "; ".join(f"rank {r}: {w}" for r, w in enumerate(fails) if w)}
                continue
            res = step("vote", lambda: em.comm_mode())       # collective on first use
            modes = gather((why[0] if why else None, res[0] if res else None, res[1] if res else None))
            if any(m[0] for m in modes):
                out[kind] = {"ok": False, "why": "; ".join(f"rank {r}: {m[0]}" for r, m in enumerate(modes) if m[0])}
                continue
            if kind == "peer" and any(m[1] != 2 for m in modes):
                out[kind] = {"ok": False, "why": "not agreed: " + "; ".join(sorted({str(m[2]) for m in modes if m[1] != 2}))}
                continue
            step("3 passes", lambda: em.iterate(3))
            step("sync", sync)
            v = step("read-back", lambda: em.getV())
            sha = hashlib.sha256(v.tobytes()).hexdigest()[:16] if v is not None else None
            got = gather((why[0] if why else None, sha))
            if any(g[0] for g in got):
                out[kind] = {"ok": False, "why": "; ".join(f"rank {r}: {g[0]}" for r, g in enumerate(got) if g[0])}
            elif len({g[1] for g in got}) != 1:
                out[kind] = {"ok": False, "why": "the ranks' models differ after 3 passes: " + ", ".join(f"rank {r}: {g[1]}" for r, g in enumerate(got))}
            else:
                out[kind] = {"ok": True, "model_sha": got[0][1]}
        finally:
            try:
                ctx.set_tuning(peer_allreduce=0)
                if em is not None:
                    em.close()
            except Exception:
                pass
    if out.get("rccl", {}).get("ok") and out.get("peer", {}).get("ok") and out["rccl"]["model_sha"] != out["peer"]["model_sha"]:
        out["peer"] = {"ok": False, "why": "the in-kernel all-reduce and the collective end on different models (%s / %s)" % (out["peer"]["model_sha"], out["rccl"]["model_sha"])}
    return out


def first_call(bm, ctx, seqs, wl, args, sync):
    """The FIRST handle of the process, nothing warmed up (no kernel of the library has run yet): what bamm_em_create costs
    and what the first optimize() call costs -- code-object loads, first-use allocations and the slow early passes included.
    Runs before the timed handle is even created; single rank only."""
    n_cold = 20
    W, K = wl["W"], wl["K"]
    sync()
    t0 = time.perf_counter()
    e = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=n_cold, epsilon=0.0, n_seqs_bound=args.nseq)
    sync()
    t1 = time.perf_counter()
    it = e.optimize()
    sync()
    t2 = time.perf_counter()
    e.close()
    return {"first_create_ms": (t1 - t0) * 1e3, "first_call_ms": (t2 - t1) * 1e3, "first_call_passes": it,
            "first_call_is": "bamm_em_create, then optimize() (epsilon 0, %d passes) on the first handle of the process before anything "
                             "else of the library has run on the device; wall clock, stream synchronised on both sides" % n_cold}


def choose_headline(args, extras, dt, kernel_ms, launches, allreduce_kind, model_sha_rccl=None):
    """N > 1: two identical timed regions were run, one per way of summing over the ranks; the line's value comes from the one
    --headline-collective names (auto: the faster, if the in-kernel region completed on every rank and ended on the collective's
    model).  Returns (dt, kernel_ms, launches, allreduce_kind) of the chosen region; both timings stay in the line."""
    extras["ms_per_step_rccl"] = dt / args.steps * 1e3
    extras["headline_collective"] = "rccl"
    peer_ms, peer = extras.get("ms_per_step_peer_allreduce"), extras.get("peer_allreduce") or {}
    if not isinstance(peer_ms, float) or args.headline_collective == "rccl":
        return dt, kernel_ms, launches, allreduce_kind
    if model_sha_rccl is not None and peer.get("model_sha") != model_sha_rccl:
        extras["ms_per_step_peer_allreduce"] = "unavailable: its region ended on another model than the collective's (%s / %s)" % (peer.get("model_sha"), model_sha_rccl)
        return dt, kernel_ms, launches, allreduce_kind
    if args.headline_collective == "peer" or peer_ms < extras["ms_per_step_rccl"]:
        extras["headline_collective"] = "peer"
        extras["avg_kernel_ms_is"] = ("one pair of HIP events around ALL passes of the timed call, divided by their number (max over ranks): with the "
                                      "all-reduce inside the kernel nothing else sits between two passes; it includes the wait for the slowest rank")
        kind = ("in-kernel all-reduce: the last block of every pass exchanges the GPU's totals with the peers through inboxes mapped over "
                "xGMI and leaves the sum in the accumulator -- no collective launch (bamm_em_comm_mode 2; same model as the collective bit "
                "for bit: selftest_comm, model_sha).  The RCCL region of the same run: ms_per_step_rccl; set-up votes over: " + allreduce_kind)
        return peer["dt_s"], peer["kernel_us"] * 1e-3 * max(launches, 1), max(launches, 1), kind
    return dt, kernel_ms, launches, allreduce_kind


def from_seed_extras(bm, ctx, seqs, wl, args, sync):
    """Beside the steady-state figure: what a run from the seed pays (the first passes are slower: few
    responsibilities are exactly zero yet), as iterate() and as optimize() (EM.cpp:81-128: the stopping rule looks
    at (llh, v_diff) of every pass); single rank only, outside the timed region."""
    n_cold = 20
    W, K = wl["W"], wl["K"]
    out = {}
    # (set-sized scratch -- 10 GB per handle at config 4 -- is allocated on a handle's first pass and goes back to the context's
    # pool when the handle closes: a throw-away handle takes that allocation, which `first_call_ms` prices, out of the pass times)
    e1 = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=n_cold, n_seqs_bound=args.nseq)
    e1.iterate(1); sync(); e1.close()
    e2 = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=n_cold, n_seqs_bound=args.nseq)
    sync()
    t1 = time.perf_counter(); e2.iterate(n_cold); sync()
    out["ms_per_step_cold"] = (time.perf_counter() - t1) / n_cold * 1e3
    e2.close()
    e3 = bm.EM(ctx, seqs, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=n_cold, epsilon=0.0, n_seqs_bound=args.nseq)
    sync()
    t1 = time.perf_counter(); it3 = e3.optimize(); sync()
    out["ms_per_step_optimize_mode"] = (time.perf_counter() - t1) / max(it3, 1) * 1e3
    out["cold_passes"] = n_cold
    e3.close()
    out["from_seed"] = {"passes": n_cold, "ms_per_step_iterate": out["ms_per_step_cold"], "ms_per_step_optimize": out["ms_per_step_optimize_mode"],
                        "positions_per_s_iterate": wl["packed"].total_len / (out["ms_per_step_cold"] * 1e-3)}
    return out


def merge_first_call(extras, first):
    """`from_seed` carries the first-call figures beside the per-step ones."""
    if first:
        extras.setdefault("from_seed", {}).update(first)
    return extras


def main_inprocess(args, result_fd):
    """`python bench.py --gpus N` without a launcher: ONE process, a context + host thread per GPU, the library's
    own RCCL communicator (ncclCommInitAll).  ctypes releases the GIL during the calls, so the N iterate() calls
    run side by side.  Nothing here touches torch."""
    import threading
    wd = Watchdog("in-process ranks", args.phase_cap_scale)  # before anything touches the GPU
    wd.phase("set-up (build, synthetic set, packing, resident shards)", 900)
    import bammmotif2_amd as bm
    from bammmotif2_amd import build

    build.build_library()
    N = max(args.gpus, 1)
    ndev = bm.device_count()                  # raises without a GPU
    if args.local_ranks:
        devices = [0] * N
    else:
        if ndev < N:
            raise SystemExit(f"bench.py --gpus {N}: only {ndev} HIP device(s) visible -- refusing to report a {N}-GPU number "
                             f"(use --local-ranks for a self-test of the {N}-rank logic on one device)")
        devices = list(range(N))
    wl = workload(args)
    packed, W, K = wl["packed"], wl["W"], wl["K"]
    ctxs, seqs, ems = [], [], []
    first = None
    for r in range(N):
        ctx = bm.Context(devices[r])
        if args.blocks or args.threads:
            ctx.set_launch(args.blocks, args.threads)
        elif args.local_ranks and N > 1:
            ctx.set_launch(max(1, 240 // N), 0)              # the ranks share one device's CUs
        if args.group_layout >= 0:
            ctx.set_tuning(group_layout=args.group_layout)
        if args.no_fused_update:
            ctx.set_tuning(fused_update=0)
        b, e = packed.shard_range(W, r, N)
        ss = bm.SeqSet(ctx, packed, b, e)
        if N == 1 and not args.force_dist and not args.no_extras:
            first = first_call(bm, ctx, ss, wl, args, ctx.sync)
        em = bm.EM(ctx, ss, K, W, wl["vbg"], wl["A"], wl["v0"], wl["q"], bg_order=2, max_iterations=args.steps + args.warmup + 8,
                   n_seqs_bound=args.nseq)
        em.set_kernel_timing(args.timing_every)
        ctxs.append(ctx); seqs.append(ss); ems.append(em)
    comms = []
    allreduce_kind = "none"
    wd.phase("communicator (ncclCommInitAll)", 180)
    if N > 1 or args.force_dist:
        if args.local_ranks:
            comms = bm.Comm.init_local(ctxs, 4 ** (K + 1) * W + 3)
            allreduce_kind = "host-staged sum inside the process (bamm_comm_init_local; self-test, not a reported configuration)"
        else:
            comms = bm.Comm.init_all(ctxs)
            allreduce_kind = "rccl (libbamm_em, ncclCommInitAll + ncclAllReduce int64 on the kernels' stream, one host thread per GPU)"
        for em, c in zip(ems, comms):
            em.set_comm(c)
    ranks = []
    for r in range(N):
        info = comms[r].info() if comms else dict(rank=0, world=1, rccl_version=None)
        if info["world"] != N or info["rank"] != r:
            raise SystemExit(f"rank {r}: communicator reports rank {info['rank']} of {info['world']}")
        ranks.append({**info, "device": devices[r], "device_name": ctxs[r].device_name(), "n_seqs": seqs[r].n_seqs})

    gate = threading.Barrier(N)
    dts, errors, ar_us = [0.0] * N, [None] * N, [None] * N
    peer_out, peer_tmp = [None] * N, [0.0] * N
    selftest, gather_tmp = [None] * N, [None] * N

    def worker(r):
        try:
            if comms and N > 1 and not args.no_selftest_comm:
                def gather(x, r=r):
                    gather_tmp[r] = x
                    gate.wait()
                    got = list(gather_tmp)
                    gate.wait()
                    return got
                if r == 0:
                    wd.phase("communicator self-test (3 passes over RCCL, 3 over the in-kernel all-reduce)", 180)
                selftest[r] = selftest_comm(bm, ctxs[r], seqs[r], comms[r], wl, args, gather, ctxs[r].sync)
                if not selftest[r]["rccl"]["ok"]:
                    raise RuntimeError("communicator self-test failed: " + selftest[r]["rccl"]["why"])
            if r == 0:
                wd.phase("warm-up passes", 180)
            ems[r].iterate(args.warmup)
            ctxs[r].sync()
            gate.wait()                                      # barrier + synchronize on both sides of the timed region
            if r == 0:
                wd.phase("timed region", 120 + 0.1 * args.steps)
            t0 = time.perf_counter()
            ems[r].iterate(args.steps)
            ctxs[r].sync()
            dts[r] = time.perf_counter() - t0
            gate.wait()
            if r == 0:
                wd.phase("after the timed region (bare all-reduce timing, in-kernel all-reduce extra)", 600)
            if comms and not args.local_ranks:               # outside the timed region: what the bare collective costs
                ar_us[r] = comms[r].time_allreduce(4 ** (K + 1) * W + 3, args.allreduce_iters)
                gate.wait()
            if comms and N > 1 and not args.no_extras and not (selftest[r] and not selftest[r]["peer"]["ok"]):
                def rmax(x, r=r):
                    peer_tmp[r] = x
                    gate.wait()
                    m = max(peer_tmp)
                    gate.wait()
                    return m
                peer_out[r] = peer_allreduce_extra(bm, ctxs[r], seqs[r], comms[r], wl, args, gate.wait, ctxs[r].sync, rmax)
        except BaseException as e:                           # a rank that fails alone would leave the others in the collective
            errors[r] = e
            gate.abort()
            for c in comms:
                c.abort()

    threads = [threading.Thread(target=worker, args=(r,), name=f"rank{r}") for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    failed = [(r, e) for r, e in enumerate(errors) if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if failed or any(errors):
        raise SystemExit("bench.py: " + "; ".join(f"rank {r}: {e!r}" for r, e in (failed or list(enumerate(errors)))))
    dt = max(dts)                                            # MAX over ranks
    # every rank holds the same model, bit for bit (integer all-reduce, redundant update)
    v0_ = ems[0].getV()
    agree = all(np.array_equal(v0_, em.getV()) for em in ems[1:])
    if not agree:
        raise SystemExit("bench.py: the ranks' models differ after the run")
    kernel = [em.kernel_time() for em in ems]
    kernel_ms, launches = max(kernel)                        # the slowest rank's sequence kernel
    llh, _, _ = ems[0].trace()
    name, mixed = kernel_label(ems[0], K)
    extras = {"ranks_agree_bitwise": agree, "ms_per_step_per_rank": [d / args.steps * 1e3 for d in dts],
              "topology": topology(bm, devices)}
    for rk in ranks:
        rk["pci_bus_id"] = (extras["topology"].get("pci_bus_id") or [None] * (rk["device"] + 1))[rk["device"]]
    if selftest[0] is not None:
        extras["selftest_comm"] = selftest[0]
        if not selftest[0]["peer"]["ok"]:
            extras["ms_per_step_peer_allreduce"] = "unavailable: " + selftest[0]["peer"]["why"]
    if comms:
        extras["attribution"] = attribution(dt / args.steps * 1e6, [k[0] / max(k[1], 1) * 1e3 for k in kernel], ar_us, args)
        extras["attribution"]["of"] = "the timed region that sums over the ranks with the collective (ms_per_step_rccl)"
    if peer_out[0] is not None:
        extras.update(peer_out[0])
        shas = {(p.get("peer_allreduce") or {}).get("model_sha") for p in peer_out}
        if len(shas) > 1:
            extras["ms_per_step_peer_allreduce"] = "unavailable: the ranks' models differ after the in-kernel all-reduce"
    if comms and N > 1:
        dt, kernel_ms, launches, allreduce_kind = choose_headline(args, extras, dt, kernel_ms, launches, allreduce_kind,
                                                                   hashlib.sha256(v0_.tobytes()).hexdigest()[:16])
    wd.phase("from-seed figures, CPU baseline, report", 1200)
    if N == 1 and not comms and not args.no_extras:
        extras.update(from_seed_extras(bm, ctxs[0], seqs[0], wl, args, ctxs[0].sync))
    merge_first_call(extras, first)
    lp = int(seqs[0].off[-1])
    lw = int((seqs[0].lengths.astype(np.int64) - W + 1).sum())
    out = report(args, wl, N, dt, kernel_ms, launches, lp, lw, name, mixed, float(llh[-1]) if len(llh) else None,
                 allreduce_kind, extras, ranks, "one process, one host thread per GPU")
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    for em in ems:
        em.close()
    for ss in seqs:
        ss.close()
    for c in comms:
        c.close()
    for ctx in ctxs:
        ctx.close()
    wd.close()


def main():
    args = parse()
    # stdout carries ONE JSON line (rank 0).  RCCL prints a version banner on stdout when a communicator
    # is created: everything written to fd 1 before the result goes to stderr instead.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.local_ranks):
        return main_inprocess(args, result_fd)               # no launcher: N ranks inside this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (torch.distributed.run "
                         f"--nproc-per-node {args.gpus}) or drop the launcher and let bench.py run them in-process")
    wd = Watchdog(f"rank {rank} of {world}", args.phase_cap_scale)     # before anything touches the GPU
    wd.phase("set-up (imports, build, process group, synthetic set, packing, resident shard)", 900)

    import torch
    import torch.distributed as dist
    import bammmotif2_amd as bm
    from bammmotif2_amd import build

    build.build_library()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    wl = workload(args)
    packed, W, K, A, vbg, v0, q = wl["packed"], wl["W"], wl["K"], wl["A"], wl["vbg"], wl["v0"], wl["q"]

    # one explicit HIP stream shared by the kernels and (through torch) the RCCL all-reduce:
    # torch's default stream has handle 0, which the C ABI would read as "create your own"
    tstream = torch.cuda.Stream(device=local_rank)
    ctx = bm.Context(local_rank, tstream.cuda_stream)
    if args.blocks or args.threads:
        ctx.set_launch(args.blocks, args.threads)
    if args.group_layout >= 0:
        ctx.set_tuning(group_layout=args.group_layout)
    if args.no_fused_update:
        ctx.set_tuning(fused_update=0)
    begin, end = packed.shard_range(W, rank, world)
    seqs = bm.SeqSet(ctx, packed, begin, end)
    first = None
    if world == 1 and not use_dist and not args.no_extras:
        with torch.cuda.stream(tstream):
            first = first_call(bm, ctx, seqs, wl, args, torch.cuda.synchronize)
    em = bm.EM(ctx, seqs, K, W, vbg, A, v0, q, bg_order=2, max_iterations=args.steps + args.warmup + 8,
               n_seqs_bound=args.nseq)      # same unit of the integer count accumulator whatever the number of ranks

    keep = []
    allreduce_kind = "none"
    rank_info = dict(rank=rank, world=world, rccl_version=None)
    wd.phase("communicator (unique id over the process group, ncclCommInitRank)", 180)
    if use_dist and args.dist_backend == "nccl" and not args.torch_allreduce:
        # the library's own collective: ncclAllReduce(int64, sum) on the context's stream, no Python in the
        # loop.  Rank 0's unique id travels over the torch process group that also serves the barriers.
        comm = None
        try:
            try:
                uid = [bm.Comm.unique_id() if rank == 0 else None]
            except Exception as e:                       # every rank must still take part in the broadcast
                uid = [None]
                print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            dist.broadcast_object_list(uid, src=0)
            if uid[0] is None:
                raise RuntimeError("rank 0 could not create a communicator id")
            comm = bm.Comm.init_rank(ctx, uid[0], rank, world)
        except Exception as e:   # report it, then fall back to the torch path below
            print(f"[bench] rank {rank}: native RCCL path unavailable ({e})", file=sys.stderr)
        # all ranks take the same path: one that failed pulls everybody to the torch collective
        agreed = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=torch.device("cuda", local_rank))
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            em.set_comm(comm)
            keep.append(comm)
            rank_info = comm.info()
            allreduce_kind = "rccl (libbamm_em, ncclCommInitRank + ncclAllReduce int64 on the kernels' stream)"
        else:
            if comm is not None:
                comm.close()
            print("[bench] using torch.distributed.all_reduce on every rank", file=sys.stderr)
    if use_dist and args.dist_backend == "gloo" and args.shm_comm:
        name = [f"/bamm_bench_{os.getpid()}" if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        comm = bm.Comm.init_shm(ctx, name[0], rank, world, 4 ** (K + 1) * W + 3)
        em.set_comm(comm)
        keep.append(comm)
        rank_info = comm.info()
        allreduce_kind = "host-staged sum between the processes (bamm_comm_init_shm): a rehearsal, all ranks on one device"
    if use_dist and allreduce_kind == "none":
        allreduce_kind = "torch.distributed.all_reduce (%s) from a callback" % args.dist_backend
        _, n = em.reduce_buffer()
        # the fused [n_K | llh | sum_r | N] accumulator (64-bit integers, fixed point) lives in a torch
        # tensor, so RCCL sees ordinary torch memory of this rank's device; int64 sums are exact
        red = torch.zeros(n, dtype=torch.int64, device=torch.device("cuda", local_rank))
        torch.cuda.synchronize()
        em.set_reduce_buffer(red.data_ptr(), n)
        keep.append(red)

        def allreduce(_ptr, _n, _stream):
            # called with `tstream` current (see iterate() below): RCCL waits for the count kernels,
            # the update waits for RCCL
            if args.dist_backend == "nccl":
                dist.all_reduce(red)
            else:                              # self-test path: through the host
                host = red.cpu()
                dist.all_reduce(host)
                red.copy_(host)
            return 0

        em.set_allreduce(allreduce)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(x):
        got = [None] * world
        dist.all_gather_object(got, x)
        return got

    selftest = None
    if use_dist and world > 1 and keep and isinstance(keep[0], bm.Comm) and not args.no_selftest_comm:
        wd.phase("communicator self-test (3 passes over the collective, 3 over the in-kernel all-reduce)", 180)
        with torch.cuda.stream(tstream):
            selftest = selftest_comm(bm, ctx, seqs, keep[0], wl, args, gather, torch.cuda.synchronize)
        if not selftest["rccl"]["ok"]:                       # the same verdict on every rank (it came out of a gather)
            if rank == 0:
                print("[bench] communicator self-test failed: " + selftest["rccl"]["why"], file=sys.stderr)
            raise SystemExit(4)

    # one HIP event pair around all passes of the timed call (1 GPU), or a pair around every 8th pass (N > 1): --timing-every
    em.set_kernel_timing(args.timing_every)
    with torch.cuda.stream(tstream):       # the context's stream is torch's current one for the callback
        wd.phase("warm-up passes", 180)
        em.iterate(args.warmup)
        barrier()
        wd.phase("timed region", 120 + 0.1 * args.steps)
        t0 = time.perf_counter()
        em.iterate(args.steps)
        barrier()
        dt = time.perf_counter() - t0
    wd.phase("after the timed region (reductions, bare all-reduce timing, extras, CPU baseline, report)", 1800)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernel_ms, launches = em.kernel_time()
    extras = {}
    my_ar_us = None
    if use_dist and keep and isinstance(keep[0], bm.Comm):
        my_ar_us = keep[0].time_allreduce(4 ** (K + 1) * W + 3, args.allreduce_iters)      # collective, outside the timed region
    if world == 1 and not use_dist and not args.no_extras:
        with torch.cuda.stream(tstream):
            extras = from_seed_extras(bm, ctx, seqs, wl, args, torch.cuda.synchronize)
    merge_first_call(extras, first)
    local_positions = int(seqs.off[-1])
    local_windows = int((seqs.lengths.astype(np.int64) - W + 1).sum())
    llh, vdiff, _ = em.trace()
    kernel_name, mixed = kernel_label(em, K)
    # what proves the N ranks: every rank's communicator view, gathered on rank 0
    topo = topology(bm, [local_rank])
    me = {**rank_info, "device": local_rank, "device_name": ctx.device_name(), "n_seqs": seqs.n_seqs,
          "pci_bus_id": (topo.get("pci_bus_id") or [None] * (local_rank + 1))[local_rank],
          "kernel_us": kernel_ms / max(launches, 1) * 1e3, "allreduce_us": my_ar_us}
    ranks = [me]
    if use_dist and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks = gathered
        if sorted(g["rank"] for g in ranks) != list(range(world)) or any(g["world"] != world for g in ranks):
            raise SystemExit(f"bench.py: the ranks do not form a world of {world}: {ranks}")
    if selftest is not None:
        extras["selftest_comm"] = selftest
        if not selftest["peer"]["ok"]:
            extras["ms_per_step_peer_allreduce"] = "unavailable: " + selftest["peer"]["why"]
    if use_dist and world > 1 and keep and isinstance(keep[0], bm.Comm) and not args.no_extras and not (selftest and not selftest["peer"]["ok"]):
        def rmax(x):
            t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        with torch.cuda.stream(tstream):
            extras.update(peer_allreduce_extra(bm, ctx, seqs, keep[0], wl, args, barrier, torch.cuda.synchronize, rmax))
        shas = [None] * world
        dist.all_gather_object(shas, (extras.get("peer_allreduce") or {}).get("model_sha"))
        if len(set(shas)) > 1:
            extras["ms_per_step_peer_allreduce"] = "unavailable: the ranks' models differ after the in-kernel all-reduce"
    if use_dist:
        topo["ranks_on_devices"] = [g["device"] for g in ranks]
        extras["topology"] = topo
        extras["attribution"] = attribution(dt / args.steps * 1e6, [g["kernel_us"] for g in ranks], [g["allreduce_us"] for g in ranks], args)
        slow = max(ranks, key=lambda g: g["kernel_us"])                # the roofline line prices the slowest rank's kernel
        kernel_ms, launches = slow["kernel_us"] * 1e-3 * max(launches, 1), max(launches, 1)
        extras["attribution"]["of"] = "the timed region that sums over the ranks with the collective (ms_per_step_rccl)"
    if use_dist and world > 1:
        sha_rccl = hashlib.sha256(em.getV().tobytes()).hexdigest()[:16]
        dt, kernel_ms, launches, allreduce_kind = choose_headline(args, extras, dt, kernel_ms, launches, allreduce_kind, sha_rccl)
    if rank == 0:
        out = report(args, wl, world, dt, kernel_ms, launches, local_positions, local_windows, kernel_name, mixed,
                     float(llh[-1]) if len(llh) else None, allreduce_kind, extras, ranks,
                     "torch.distributed.run, one process per GPU" if "WORLD_SIZE" in os.environ else "single process")
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    em.close(); seqs.close()
    for obj in keep:
        if isinstance(obj, bm.Comm):
            obj.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    wd.close()


if __name__ == "__main__":
    main()
