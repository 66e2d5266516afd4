"""Build the gfx950 shared library in-tree (``bammmotif2_amd/libbamm_em.so``).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import contextlib
import fcntl
import json
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbamm_em.so")
SOURCES = ["model.hip", "kernels.hip", "grouped.hip", "grouped_long.hip", "grouped_xl.hip", "grouped_mix.hip", "grouped_mix1.hip", "mask.hip", "seed.hip", "long_seq.hip", "prep.hip", "negs.hip", "abi.cpp", "comm.cpp", "pack.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "device_utils.h"), os.path.join(CSRC, "grouped_kernel.h"), os.path.join(CSRC, "mixed_kernel.h"), os.path.join(CSRC, "update_kernel.h"), os.path.join(CSRC, "phase_clock.h"),
           os.path.join(HERE, "..", "include", "bamm_em.h")]
# -Rpass-analysis=kernel-resource-usage: registers / scratch / spills of every kernel go to the compiler's
# stderr, which is kept per translation unit (build/<tu>.remarks) and checked by check_resources().
# No -Wno-pass-failed: a `#pragma unroll` the compiler gave up on is reported (and, for the kernels with
# hand-issued LDS reads, refused) instead of silently turning register arrays into scratch memory.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
# Translation units whose kernels exceed the 64 KB instruction cache (20 and more positions per lane: 72-153 KB of code each):
# their kernels start on 16 KB boundaries.  Where such a kernel lands inside its code object decides how its loop maps onto
# the cache -- the 64-per-lane class ran 5.75 ms per pass at one 256-byte offset and 4.93 ms 1280 bytes earlier, the same
# instructions, three times the instruction-cache misses (SQC_ICACHE_MISSES; profiles/r05_icache_alignment.txt) -- and every
# unrelated change to the unit moves it.  Aligned, the placement is the same in every build (4 KB: 5.23 ms, 16 KB: 4.88, 64 KB: 4.90).
TU_FLAGS = {"grouped_long.hip": ["-falign-functions=16384"], "grouped_xl.hip": ["-falign-functions=16384"], "kernels.hip": ["-falign-functions=16384"]}
OBJDIR = os.path.join(HERE, "build")
RESOURCES = os.path.join(OBJDIR, "resources.json")


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _remarks(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".remarks")


def _diagnostics(text: str) -> str:
    """The compiler's warnings / errors without the kernel-resource-usage remarks."""
    keep, skip = [], 0
    for line in text.splitlines():
        if "[-Rpass-analysis=kernel-resource-usage]" in line:
            skip = 2 if "Function Name" in line else 0       # the remark's source excerpt (2 lines) follows its first line
            continue
        if skip:
            skip -= 1
            continue
        keep.append(line)
    return "\n".join(keep).strip()


class KernelResourceError(RuntimeError):
    pass


def parse_resources(text: str) -> dict:
    """{mangled kernel name: {vgprs, agprs, sgprs, scratch, vgpr_spill, sgpr_spill, occupancy}} from the
    -Rpass-analysis=kernel-resource-usage remarks of one translation unit."""
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    for m in re.finditer(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis=kernel-resource-usage\]", text):
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    return out


def check_resources(sources, failed_unrolls_fatal: bool = True) -> dict:
    """Refuse a build in which some instruction touches the destination of an LDS read that is still in
    flight (kernel_audit.py: the hand-issued `ds_read_b128` gathers are not valid until their
    `s_waitcnt`; a compiler that spills or copies them in between stores garbage -- round 1: the grouped
    kernel at 56 / 64 positions per lane), or in which the compiler reports an unroll it did not perform
    in the translation units of the grouped kernel (-Wpass-failed: register arrays become scratch memory).
    Writes build/resources.json: per kernel the compiler's register / scratch / spill figures and the
    number of scratch instructions actually emitted."""
    from . import kernel_audit
    table, bad = {}, []
    for s in sources:
        text = open(_remarks(s)).read()
        res = parse_resources(text)
        violations, stats = kernel_audit.audit_object(_obj(s), os.path.join(OBJDIR, "audit"))
        for name, st in stats.items():
            res.setdefault(name, {}).update(st)
        table.update(res)
        for kernel, lineno, ins, regs in violations[:20]:
            bad.append(f"{kernel}: `{ins}` (disassembly line {lineno}) touches v{regs} while the LDS read that writes "
                       f"them is in flight ({s})")
        if len(violations) > 20:
            bad.append(f"... and {len(violations) - 20} more in {s}")
        if failed_unrolls_fatal and s.startswith("grouped"):
            bad += [f"{s}: {line.strip()}" for line in text.splitlines() if "-Wpass-failed" in line]
    os.makedirs(OBJDIR, exist_ok=True)
    with open(RESOURCES, "w") as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    if bad:
        raise KernelResourceError("kernel audit failed:\n  " + "\n  ".join(bad))
    return table


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# headers only some translation units include (a change there does not rebuild the sequence kernels)
EXTRA_DEPS = {"prep.hip": [os.path.join(CSRC, "prep.h")], "negs.hip": [os.path.join(CSRC, "negs.h")],
              "abi.cpp": [os.path.join(CSRC, "prep.h"), os.path.join(CSRC, "negs.h"), os.path.join(CSRC, "glibc_rand.h")],
              "pack.cpp": [os.path.join(CSRC, "glibc_rand.h"), os.path.join(CSRC, "prep.h")]}
FLAGS_STAMP = os.path.join(OBJDIR, "flags.txt")


def _flags_stamp() -> str:
    return " ".join(FLAGS) + " | " + " ".join(f"{k}: {' '.join(v)}" for k, v in sorted(TU_FLAGS.items()))


def _flags_changed() -> bool:
    """The library in the tree was built with other compiler flags than this process would use (tools/phase_clock.py
    appends -DBAMM_PHASE_CLOCK): as stale as a changed source -- nobody benchmarks an instrumented build by accident."""
    try:
        return open(FLAGS_STAMP).read() != _flags_stamp()
    except OSError:
        return os.path.exists(LIB) and os.path.isdir(OBJDIR) and bool(os.listdir(OBJDIR))   # built before the stamp existed


def is_stale() -> bool:
    """The library is older than a source or header, was built with other flags, or -- where the objects are at hand -- some
    object is older than its source (a source edited WHILE a build ran: the link that followed is newer than the edit)."""
    if _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [h for v in EXTRA_DEPS.values() for h in v]) or _flags_changed():
        return True
    return any((os.path.exists(_obj(s)) and _stale(_obj(s), [os.path.join(CSRC, s)] + HEADERS + EXTRA_DEPS.get(s, []))) or
               (os.path.isdir(OBJDIR) and os.path.exists(RESOURCES) and not os.path.exists(_obj(s))) for s in SOURCES)


@contextlib.contextmanager
def _build_lock():
    """One builder at a time: the ranks of a multi-GPU run all pass through here at start-up."""
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """One object per source (rebuilt only when it or a header changed, in parallel), then link."""
    if not force and not is_stale():
        return LIB
    with _build_lock():
        if not force and not is_stale():                     # another rank built it while this one waited
            return LIB
        return _build_library_locked(force, verbose)


def _build_library_locked(force: bool, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 extension")
    os.makedirs(OBJDIR, exist_ok=True)
    force = force or _flags_changed()                        # every object again, not only the ones whose source moved
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), _obj(s)
        if force or _stale(obj, [src] + HEADERS + EXTRA_DEPS.get(s, [])) or not os.path.exists(_remarks(s)):
            cmd = [hipcc] + FLAGS + TU_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            log = open(_remarks(s) + ".tmp", "w")
            jobs.append((s, cmd, subprocess.Popen(cmd, stderr=log), log))
    for s, cmd, proc, log in jobs:
        rc = proc.wait()
        log.close()
        text = open(_remarks(s) + ".tmp").read()
        diag = _diagnostics(text)
        if rc != 0 or "warning:" in diag or "error:" in diag:
            print(diag)
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
        os.replace(_remarks(s) + ".tmp", _remarks(s))
    check_resources([s for s in SOURCES if s.endswith(".hip")])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + ["-ldl", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(FLAGS_STAMP, "w") as fh:
        fh.write(_flags_stamp())
    return LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libbamm_host.so")
CLI = os.path.join(HERE, "BaMMmotif")
HOST_SOURCES = ["io.cpp", "fdr.cpp", "hooks.cpp"]


def build_host(force: bool = False, verbose: bool = False):
    """C++17 host code: libbamm_host.so (test hooks) and the `BaMMmotif` drop-in CLI."""
    build_library(force=False, verbose=verbose)
    deps = [os.path.join(HOST, f) for f in HOST_SOURCES + ["main.cpp", "bamm_host.h"]] + [LIB]
    outs = [HOST_LIB, CLI]
    if not force and all(os.path.exists(o) for o in outs) and \
            min(os.path.getmtime(o) for o in outs) >= max(os.path.getmtime(d) for d in deps):
        return outs
    cxx = shutil.which("g++") or "g++"
    common = [cxx, "-std=c++17", "-O2", "-fopenmp", "-Wall", "-fPIC", "-L" + HERE, "-Wl,-rpath,$ORIGIN"]
    cmd = common + ["-shared"] + [os.path.join(HOST, f) for f in HOST_SOURCES] + ["-lbamm_em", "-o", HOST_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    cmd = common + [os.path.join(HOST, "main.cpp"), os.path.join(HOST, "io.cpp"), os.path.join(HOST, "fdr.cpp"), "-lbamm_em", "-o", CLI]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return outs


if __name__ == "__main__":
    build_host(force=True, verbose=True)
    print(build_library(force=True, verbose=True))
