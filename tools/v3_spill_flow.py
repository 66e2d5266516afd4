#!/usr/bin/env python3
"""Where do the SGPRs a kernel spills come back from?  hipcc keeps spilled SGPRs in LANES of holder VGPRs (v_writelane /
v_readlane).  This walks the control-flow graph of one kernel's disassembly and computes, for every v_readlane, the set of
v_writelane instructions whose value can reach it (reaching definitions per (holder, lane) slot, holder copies through AGPRs
followed).  A read reached by two writes is legitimate for a loop-carried variable (initial value + update) and suspicious for
anything else; reads with no reaching write at all are reported too.

    tools/v3_spill_flow.py OBJECT.o KERNEL_SUBSTRING [--loop-only]
"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bammmotif2_amd import kernel_audit as ka


def load(obj, needle):
    co = ka.extract_code_object(obj, "/tmp/v3flow")
    txt = ka.disassemble(co)
    cur, body, start = None, [], {}
    for l in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", l)
        if m:
            if cur and needle in cur:
                break
            cur, body = m.group(2), []
            start[cur] = int(m.group(1), 16)
            continue
        if cur and l.startswith("\t"):
            ins, _, cm = l.partition("//")
            if not ins.strip() or not cm.strip():
                continue
            addr = int(cm.strip().split(":")[0], 16)
            body.append((addr, ins.strip(), cm))
    assert cur and needle in cur, "kernel not found"
    return cur, start[cur], body


def main():
    obj, needle = sys.argv[1], sys.argv[2]
    name, base, body = load(obj, needle)
    addr_index = {a: i for i, (a, _, _) in enumerate(body)}
    n = len(body)
    succ = [[] for _ in range(n)]
    for i, (a, ins, cm) in enumerate(body):
        op = ins.split(" ", 1)[0]
        tgt = None
        m = re.search(r"<[^>]+\+0x([0-9a-f]+)>", cm)
        if op.startswith("s_cbranch") or op == "s_branch":
            if m:
                tgt = addr_index.get(base + int(m.group(1), 16))
            if tgt is not None:
                succ[i].append(tgt)
            if op != "s_branch" and i + 1 < n:
                succ[i].append(i + 1)
        elif op in ("s_endpgm",):
            pass
        elif i + 1 < n:
            succ[i].append(i + 1)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in succ[i]:
            pred[j].append(i)
    # effects on slots
    W = re.compile(r"v_writelane_b32\s+v(\d+),\s*(s\d+|vcc_lo|vcc_hi|exec_lo|exec_hi|m0|[-\d]+|0x[0-9a-f]+),\s*(\d+)")
    R = re.compile(r"v_readlane_b32\s+(s\d+|vcc_lo|vcc_hi|exec_lo|exec_hi|m0),\s*v(\d+),\s*(\d+)")
    AW = re.compile(r"v_accvgpr_write_b32\s+a(\d+),\s*v(\d+)")
    AR = re.compile(r"v_accvgpr_read_b32\s+v(\d+),\s*a(\d+)")
    holders = {int(W.match(ins).group(1)) for _, ins, _ in body if W.match(ins)}
    # state: dict slot -> frozenset(def ids); slot = ("v"|"a", reg, lane)
    lanes = range(64)
    def transfer(i, st):
        a, ins, _ = body[i]
        m = W.match(ins)
        if m:
            st = dict(st); st[("v", int(m.group(1)), int(m.group(3)))] = frozenset([i]); return st
        m = AW.match(ins)
        if m and int(m.group(2)) in holders:
            st = dict(st)
            for L in lanes:
                k = ("v", int(m.group(2)), L)
                if k in st: st[("a", int(m.group(1)), L)] = st[k]
                else: st.pop(("a", int(m.group(1)), L), None)
            return st
        m = AR.match(ins)
        if m and int(m.group(1)) in holders:
            st = dict(st)
            for L in lanes:
                k = ("a", int(m.group(2)), L)
                if k in st: st[("v", int(m.group(1)), L)] = st[k]
                else: st.pop(("v", int(m.group(1)), L), None)
            return st
        return st
    IN = [None] * n
    IN[0] = {}
    work = [0]
    while work:
        i = work.pop()
        out = transfer(i, IN[i])
        for j in succ[i]:
            if IN[j] is None:
                IN[j] = dict(out); work.append(j)
            else:
                changed = False
                for k, v in out.items():
                    if k in IN[j]:
                        u = IN[j][k] | v
                        if u != IN[j][k]: IN[j][k] = u; changed = True
                    else:
                        IN[j][k] = v; changed = True
                if changed: work.append(j)
    # loops: instruction i is in a loop if it can reach itself -- approximate by back edges (succ with smaller index)
    back = [(i, j) for i in range(n) for j in succ[i] if j <= i]
    in_loop = [False] * n
    for i, j in back:
        for k in range(j, i + 1): in_loop[k] = True
    print(f"{name}: {n} instructions, {len(holders)} holder VGPRs {sorted(holders)}, {len(back)} back edges")

    def producer(d):
        """the nearest earlier instruction (straight line) that writes the SGPR a v_writelane stores"""
        src = body[d][1].split(",")[1].strip()
        for k in range(d - 1, max(d - 400, -1), -1):
            t = body[k][1]
            op, _, rest = t.partition(" ")
            first = rest.split(",")[0].strip()
            if first == src or re.match(r"s\[(\d+):(\d+)\]", first) and src.startswith("s") and src[1:].isdigit() and \
                    int(re.match(r"s\[(\d+):(\d+)\]", first).group(1)) <= int(src[1:]) <= int(re.match(r"s\[(\d+):(\d+)\]", first).group(2)):
                return f"+0x{body[k][0] - base:x} {t}"
        return "?"

    if "--suspects" in sys.argv:
        cnt = 0
        for i, (a, ins, _) in enumerate(body):
            m = R.match(ins)
            if not m or int(m.group(2)) not in holders or not in_loop[i]: continue
            defs = (IN[i] or {}).get(("v", int(m.group(2)), int(m.group(3))), frozenset())
            outside = [d for d in defs if not in_loop[d]]
            inside = [d for d in defs if in_loop[d]]
            if outside and inside:
                cnt += 1
                print(f"  +0x{a - base:x} {ins}")
                for d in sorted(defs):
                    print(f"      <- +0x{body[d][0] - base:x} {body[d][1]} {'[loop]' if in_loop[d] else '[before the loops]'}   produced by {producer(d)}")
        print(f"reads inside a loop reached by a write from before the loops AND one from inside: {cnt}")
        return
    multi = nodef = reads = 0
    for i, (a, ins, _) in enumerate(body):
        m = R.match(ins)
        if not m or int(m.group(2)) not in holders: continue
        reads += 1
        defs = (IN[i] or {}).get(("v", int(m.group(2)), int(m.group(3))), frozenset())
        if not defs:
            nodef += 1
            print(f"  +0x{a - base:x} {ins}   <- NO reaching write")
        elif len(defs) > 1 and (in_loop[i] or "--loop-only" not in sys.argv):
            multi += 1
            srcs = sorted((d, body[d][1]) for d in defs)
            print(f"  +0x{a - base:x} {ins}   {'[loop]' if in_loop[i] else ''} <- {len(defs)} writes: " + "; ".join(f"+0x{body[d][0] - base:x} {t.split(',')[1].strip()}{' [loop]' if in_loop[d] else ''}" for d, t in srcs))
    print(f"reads {reads}, reached by more than one write {multi}, by none {nodef}")


if __name__ == "__main__":
    main()
