"""The build refuses kernels in which an instruction touches the destination of an LDS read that is still
in flight (bammmotif2_amd/kernel_audit.py, wired into build.build_library).  Round 1 lost the grouped kernel
at 56 / 64 positions per lane to exactly that: hipcc parked the destinations of the hand-issued
`ds_read_b128` in AGPRs before the `s_waitcnt`."""
import os
import shutil
import subprocess

import pytest

from bammmotif2_amd import build, kernel_audit

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_audit_model_on_a_handwritten_listing():
    ok = """
0000000000001000 <k_ok>:
\tds_read_b128 v[4:7], v1                                    // 000000001000: D9FE0000
\tds_read_b128 v[8:11], v2 offset:16                         // 000000001008: D9FE0010
\tv_add_u32_e32 v1, 16, v1                                   // 000000001010: 68020290
\ts_waitcnt lgkmcnt(1)                                       // 000000001014: BF8CC17F
\tv_mul_f32_e32 v12, v4, v5                                  // 000000001018: 0A180B04
\ts_waitcnt lgkmcnt(0)                                       // 00000000101C: BF8CC07F
\tv_mul_f32_e32 v12, v8, v12                                 // 000000001020: 0A181908
\ts_endpgm                                                   // 000000001024: BF810000
"""
    v, st = kernel_audit.audit_disassembly(ok)
    assert v == [] and st["k_ok"]["ds_reads"] == 2
    bad = ok.replace("s_waitcnt lgkmcnt(1)", "s_nop 0").replace("k_ok", "k_bad")
    v, _ = kernel_audit.audit_disassembly(bad)
    assert len(v) == 1 and v[0][0] == "k_bad" and v[0][3] == [4, 5]
    spill = ok.replace("v_add_u32_e32 v1, 16, v1", "scratch_store_dword off, v9, off offset:16")
    v, st = kernel_audit.audit_disassembly(spill)
    assert len(v) == 1 and v[0][3] == [9] and st["k_ok"]["scratch_instructions"] == 1


def test_shipped_kernels_pass_the_audit():
    build.build_library()
    if not all(os.path.exists(build._obj(s)) for s in build.SOURCES):      # a prebuilt .so without its objects
        pytest.skip("object files not present")
    table = build.check_resources([s for s in build.SOURCES if s.endswith(".hip")])
    grp = [k for k in table if "k_em_grp" in k]
    assert len(grp) > 100 and all(table[k]["ds_reads"] > 0 for k in grp)
    # the bench kernel (7 positions per lane, 3 columns per row, 5-mer rows) keeps everything in registers
    bench = [k for k in grp if "k_em_grpILi7ELi3ELi5ELb1ELb0" in k]
    assert bench and table[bench[0]]["scratch_instructions"] == 0 and table[bench[0]]["vgpr_spill"] == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_guard_trips_on_a_kernel_that_uses_a_destination_in_flight(tmp_path):
    """Compile a kernel that multiplies the destination of a hand-issued ds_read_b128 before its s_waitcnt (what
    hipcc did by itself to the grouped kernel at 56 / 64 positions per lane in round 1, by parking such registers
    in AGPRs -- those classes read one slot at a time now and pass): the audit of the object must see it."""
    src = tmp_path / "hazard.hip"
    src.write_text('#include "%s"\n'
                   "namespace bamm {\n"
                   "void set_error(const char*, ...) {}\n"
                   "__global__ void k_hazard(float* out) {\n"
                   "    extern __shared__ float l[];\n"
                   "    f32x4 v = lds_read_b128(lds_offset(l) + threadIdx.x * 16u);\n"
                   "    const float early = v.x * 2.0f;                       // the read has not landed yet\n"
                   '    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));\n'
                   "    out[threadIdx.x] = early + v.y;\n"
                   "}\n}\n" % os.path.join(build.CSRC, "grouped_kernel.h"))
    obj = tmp_path / "hazard.o"
    flags = [f for f in build.FLAGS if not f.startswith("-Rpass")]
    subprocess.check_call([HIPCC] + flags + ["-c", str(src), "-o", str(obj)], stderr=subprocess.DEVNULL)
    violations, stats = kernel_audit.audit_object(str(obj), str(tmp_path / "audit"))
    assert any("k_hazard" in v[0] for v in violations), "the audit no longer sees a destination used in flight"


def test_audit_sees_a_dpp_behind_a_valu_write_of_exec():
    """The hand-written v_mul_f32_dpp (device_utils.h) carries its own s_nop for the VGPR-write hazard; the other
    DPP hazard, a VALU write of EXEC within 5 wait states, is the audit's to catch."""
    listing = """
0000000000001000 <k_dpp>:
\tv_cmpx_lt_u32_e64 exec, v1, v2                             // 000000001000: D0C9007E
\ts_nop 1                                                    // 000000001008: BF800001
\tv_mul_f32_dpp v3, v4, v3 wave_shr:1 row_mask:0xf bank_mask:0xf  // 00000000100C: 0A0606FA
\ts_endpgm                                                   // 000000001014: BF810000
"""
    v, _ = kernel_audit.audit_disassembly(listing)
    assert len(v) == 1 and "DPP" in v[0][2]
    v, _ = kernel_audit.audit_disassembly(listing.replace("s_nop 1", "s_nop 4"))
    assert v == []


def test_audit_sees_a_spill_holder_moved_outside_whole_wave_mode():
    """A VGPR whose lanes hold spilled SGPRs (v_writelane) copied to an AGPR under whatever EXEC happens to be: the lanes
    that are switched off lose their SGPRs.  The compiler's own copies sit behind `s_or_saveexec_b64 sX, -1`."""
    listing = """
0000000000002000 <k_spill>:
\tv_writelane_b32 v247, s0, 12                               // 000000002000: D28A00F7
\ts_and_saveexec_b64 s[2:3], s[4:5]                          // 000000002008: BE822004
\tv_accvgpr_write_b32 a0, v247                               // 00000000200C: D3D94000
\ts_mov_b64 exec, s[2:3]                                     // 000000002014: BEFE0102
\tv_accvgpr_read_b32 v247, a0                                // 000000002018: D3D840F7
\tv_readlane_b32 s0, v247, 12                                // 000000002020: D28900F7
\ts_endpgm                                                   // 000000002028: BF810000
"""
    v = kernel_audit.audit_spill_holders(listing)
    assert len(v) == 2 and all(x[3] == [247] for x in v)
    good = listing.replace("s_and_saveexec_b64 s[2:3], s[4:5]", "s_or_saveexec_b64 s[2:3], -1")
    good = good.replace("\tv_accvgpr_read_b32 v247, a0", "\ts_mov_b64 exec, -1\n\tv_accvgpr_read_b32 v247, a0")
    assert kernel_audit.audit_spill_holders(good) == []
