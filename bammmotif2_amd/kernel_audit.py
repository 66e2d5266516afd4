"""Build-time audit of the gfx950 code objects (used by ``build.py``; no GPU needed).

Why: the sequence kernels issue their LDS gathers by hand (``ds_read_b128`` in inline asm, the
``s_waitcnt lgkmcnt(0)`` a few instructions later, csrc/device_utils.h ``lds_read_b128`` /
csrc/grouped_kernel.h ``lds_read_b128_off``).  The compiler does not know that the destination
registers are not valid until that wait: if register pressure makes it spill or copy one of them in
between, the kernel silently computes garbage -- that is what happened to the grouped kernel at 56 / 64
positions per lane in round 1.  A compiler bump can do the same to a class that is fine today.

What is checked, on the disassembly of every kernel in the code object:

* ``in-flight`` audit: the LGKM counter is modelled instruction by instruction (DS and scalar-memory
  operations enter a queue, ``s_waitcnt lgkmcnt(N)`` retires all but the N youngest); any instruction that
  names a destination register of a ``ds_read`` that has not been retired yet is a violation.
* ``spill holders``: a VGPR whose lanes hold spilled SGPRs may only be moved (to an AGPR and back) in whole-wave mode
  (``audit_spill_holders``).
* resources: scratch instructions per kernel and the compiler's own figures (registers, spills,
  scratch bytes) from ``-Rpass-analysis=kernel-resource-usage`` go to ``build/resources.json``.
"""
from __future__ import annotations

import os
import re
import subprocess

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def extract_code_object(obj: str, workdir: str) -> str:
    """The gfx950 code object bundled into a hipcc host object -> path of the extracted ELF."""
    os.makedirs(workdir, exist_ok=True)
    local = os.path.join(workdir, os.path.basename(obj))
    if os.path.abspath(local) != os.path.abspath(obj):
        if os.path.lexists(local):
            os.remove(local)
        os.symlink(os.path.abspath(obj), local)
    subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = f"{local}.0.{TARGET}"
    if not os.path.exists(out):
        raise RuntimeError(f"no {TARGET} bundle in {obj}")
    return out


def disassemble(code_object: str) -> str:
    return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", code_object], text=True)


def _vregs(text: str):
    regs = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def audit_disassembly(text: str):
    """Returns (violations, per-kernel stats).  A violation is (kernel, line number, instruction, registers)."""
    violations, stats = [], {}
    kernel = None
    queue = []          # outstanding LGKM operations, oldest first: sets of destination VGPRs (empty for writes / SMEM)
    inflight = set()
    since_valu_exec = 99    # instructions since a VALU instruction wrote EXEC (v_cmpx*): a DPP needs 5 wait states
    for lineno, raw in enumerate(text.splitlines(), 1):
        m = _FUNC.match(raw)
        if m:
            kernel = m.group(1)
            stats[kernel] = {"scratch_instructions": 0, "ds_reads": 0, "instructions": 0}
            queue, inflight = [], set()
            since_valu_exec = 99
            continue
        if kernel is None or not raw.startswith("\t"):
            continue
        ins = raw.split("//")[0].strip()
        if not ins:
            continue
        op, _, rest = ins.partition(" ")
        st = stats[kernel]
        st["instructions"] += 1
        # The hand-written DPP multiply (device_utils.h: mul_wave_shr1) carries its own s_nop 1 for the VGPR-write
        # hazard; the compiler's hazard recogniser does not see it, so the other DPP hazard -- a VALU write of EXEC
        # within 5 wait states -- is checked here (hipcc uses v_cmp + s_and_saveexec, never v_cmpx, today)
        if op.startswith("v_cmpx"):
            since_valu_exec = 0
        else:
            since_valu_exec += (int(rest) + 1) if op == "s_nop" and rest.strip().isdigit() else 1
        if op == "v_mul_f32_dpp" and since_valu_exec <= 5:
            violations.append((kernel, lineno, ins + "   <- DPP within 5 wait states of a VALU write of EXEC", []))
        if op.startswith("scratch_"):
            st["scratch_instructions"] += 1
        if op == "s_waitcnt":
            w = _LGKM.search(rest)
            if w:
                keep = int(w.group(1))
                while len(queue) > keep:
                    inflight -= queue.pop(0)
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            queue, inflight = [], set()          # what follows is reached from elsewhere: state unknown, assume drained
            continue
        used = _vregs(rest)
        if inflight and used & inflight:
            violations.append((kernel, lineno, ins, sorted(used & inflight)))
        if op.startswith("ds_"):
            dest = set()
            if op.startswith("ds_read") or "_rtn" in op or op.startswith("ds_bpermute") or op.startswith("ds_permute") \
                    or op.startswith("ds_swizzle") or op.startswith("ds_consume") or op.startswith("ds_append"):
                first = rest.split(",")[0]
                dest = _vregs(first)
                if op.startswith("ds_read"):
                    st["ds_reads"] += 1
            queue.append(dest)
            inflight |= dest
        elif op.startswith(("s_load_", "s_buffer_load_", "s_store_", "s_buffer_store_", "s_atomic", "s_memtime",
                            "s_memrealtime", "s_dcache", "s_sendmsg")):
            queue.append(set())                  # scalar memory returns out of order: the compiler waits for 0 on it
    return violations, stats


def audit_spill_holders(text: str):
    """SGPRs the register allocator spills live in LANES of holder VGPRs (v_writelane / v_readlane, which ignore EXEC).  When
    VGPRs run out as well -- the 56 / 64-positions-per-lane classes: 256 VGPRs + 600-800 spilled SGPRs -- a holder is itself
    copied to an AGPR and back.  Those copies are ordinary VALU moves: under a partial EXEC they move some lanes only, i.e.
    they drop spilled SGPRs.  The compiler brackets them with `s_or_saveexec_b64 sX, -1` / `s_mov_b64 exec, -1` (whole-wave
    mode); the rule here is that it always does: any instruction other than v_writelane / v_readlane that names a holder
    must directly follow such an EXEC = -1.  (Round 3's wrong counts at those classes -- reproduced at 1b24a07 and gone
    with 1e80103, tools/v3_bisect_build.sh -- were looked for here first; both objects satisfy the rule.)
    Returns violations like audit_disassembly's."""
    violations = []
    kernel, body = None, []

    def flush():
        holders = set()
        for _, ins in body:
            m = re.match(r"v_writelane_b32\s+v(\d+)\b", ins)
            if m:
                holders.add(int(m.group(1)))
        if not holders:
            return
        prev = ""
        for lineno, ins in body:
            op = ins.split(" ", 1)[0]
            if op not in ("v_writelane_b32", "v_readlane_b32") and (_vregs(ins.partition(" ")[2]) & holders):
                wwm = ("exec, -1" in prev) or (prev.startswith("s_or_saveexec_b64") and prev.rstrip().endswith("-1"))
                if not wwm:
                    violations.append((kernel, lineno, ins + "   <- a VGPR that holds spilled SGPRs in its lanes, moved outside whole-wave mode",
                                       sorted(_vregs(ins.partition(" ")[2]) & holders)))
            prev = ins

    for lineno, raw in enumerate(text.splitlines(), 1):
        m = _FUNC.match(raw)
        if m:
            if kernel is not None:
                flush()
            kernel, body = m.group(1), []
            continue
        if kernel is None or not raw.startswith("\t"):
            continue
        ins = raw.split("//")[0].strip()
        if ins:
            body.append((lineno, ins))
    if kernel is not None:
        flush()
    return violations


def audit_object(obj: str, workdir: str):
    co = extract_code_object(obj, workdir)
    try:
        text = disassemble(co)
        violations, stats = audit_disassembly(text)
        return violations + audit_spill_holders(text), stats
    finally:
        for suffix in (f".0.{TARGET}", ".0.host-x86_64-unknown-linux-gnu-"):
            try:
                os.remove(os.path.join(workdir, os.path.basename(obj)) + suffix)
            except OSError:
                pass
