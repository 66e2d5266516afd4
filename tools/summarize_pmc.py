#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSVs (one pass per counter group) into a PMC summary json.

    summarize_pmc.py PMC_DIR POSITIONS [--order K] [--out FILE] [--lds-mix FILE]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB and on gfx950
FETCH_SIZE reads half of the bytes a coalesced stream fetches (/opt/skills/guides/MI355X_MICROARCH.md,
section HBM).  Fused path: the one sequence kernel (k_em_grp, else k_em_seq).  Column-sliced path (k >= 4):
the E pass and the M slices of one iteration, summed -- per iteration = per launch x launches per iteration,
the iteration count taken from k_update's dispatches.  --lds-mix: the json line of tools/lds_mix_bench
(the LDS ceiling for the kernel's instruction mix), carried along for bench.py's `roofline.lds`."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("pmc_dir")
ap.add_argument("positions", type=int)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "hbm_traffic.json"))
ap.add_argument("--lds-mix", default=None)
ap.add_argument("--lds-c4", default=None, help="output of tools/lds_c4_bench: the LDS ceilings of the sliced path's two kernels")
ap.add_argument("--kernel-stats", default=None, help="rocprofv3 --kernel-trace --stats csv of the same command: average duration per kernel")
ap.add_argument("--commit", default=os.environ.get("BAMM_COMMIT"), help="git commit the counters were taken at (the GPU box has no .git)")
args = ap.parse_args()

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(args.pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, cs in acc.items():
    summary[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summary[k]["dispatches"] = max(len(v) for v in cs.values())
res = {"positions_per_launch": args.positions, "order": args.order, "per_kernel_mean_counters": summary}


def hbm_bytes(m):
    return (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0


XCDS, CUS = 8.0, 256.0
seq_kernels = [k for k in summary if any(t in k for t in ("k_em_grp", "k_em_mix", "k_em_seq", "k_e_slice", "k_m_slice", "k_m_list", "k_long_em"))]
updates = next((summary[k]["dispatches"] for k in summary if "k_update" in k), None)
if args.order >= 4 and seq_kernels and updates:
    per_iter, parts = 0.0, {}
    cyc = 0.0
    for k in seq_kernels:
        m = summary[k]
        if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        launches = m["dispatches"] / updates
        parts[k] = {"launches_per_iteration": launches, "hbm_bytes_per_launch": hbm_bytes(m),
                    "fetch_size_kib": m["FETCH_SIZE"], "write_size_kib": m["WRITE_SIZE"],
                    "lds_wave_instr_per_launch": m.get("SQ_INSTS_LDS"),
                    "lds_busy_cycles_per_cu": (m["SQ_LDS_IDX_ACTIVE"] / CUS) if "SQ_LDS_IDX_ACTIVE" in m else None,
                    "lds_bank_conflict_cycles_per_cu": (m["SQ_LDS_BANK_CONFLICT"] / CUS) if "SQ_LDS_BANK_CONFLICT" in m else None}
        if "GRBM_GUI_ACTIVE" in m:
            parts[k]["gpu_cycles_per_launch"] = m["GRBM_GUI_ACTIVE"] / XCDS
            cyc += launches * m["GRBM_GUI_ACTIVE"] / XCDS
        per_iter += launches * hbm_bytes(m)
    res.update(kernel="E pass + M slices of one iteration", hbm_bytes_per_launch=per_iter, per_kernel=parts,
               gpu_cycles_per_iteration=cyc)
else:
    main = next((k for k in summary if "k_em_mix" in k and "true, false" in k), None) or \
        next((k for k in summary if "k_em_grp" in k and "true, false" in k), None) or \
        next((k for k in summary if "k_em_seq" in k), None)
    m = summary.get(main, {}) if main else {}
    if main and "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        res.update(kernel=main, fetch_size_kib=m["FETCH_SIZE"], write_size_kib=m["WRITE_SIZE"],
                   hbm_bytes_per_launch=hbm_bytes(m), hbm_bytes_per_launch_uncorrected=(m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0)
    if all(c in m for c in ("GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU")):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ_* counters over the 256 CUs (cycles of the CU's
        # LDS / VALU pipes being busy): busy fraction = per-CU busy cycles / kernel cycles
        cyc = m["GRBM_GUI_ACTIVE"] / XCDS
        res["derived"] = {"gpu_cycles_per_launch": cyc,
                          "lds_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / CUS / cyc,
                          "lds_bank_conflict_frac": m["SQ_LDS_BANK_CONFLICT"] / CUS / cyc,
                          "valu_busy_frac": m["SQ_ACTIVE_INST_VALU"] / CUS / cyc,
                          "lds_wave_instr_per_launch": m.get("SQ_INSTS_LDS"),
                          "valu_wave_instr_per_launch": m.get("SQ_INSTS_VALU"),
                          "salu_wave_instr_per_launch": m.get("SQ_INSTS_SALU")}
if args.lds_mix and os.path.exists(args.lds_mix):
    for line in open(args.lds_mix):
        if line.startswith("{"):
            res["lds_mix_bench"] = json.loads(line)
            if "k_em_mix" in str(res.get("kernel", "")) and "mixed_rows_wave_instr_per_s" in res["lds_mix_bench"]:
                # the ceiling that belongs to the kernel that ran: the mixed-row instruction mix
                res["lds_mix_bench"]["uniform_rows_wave_instr_per_s"] = res["lds_mix_bench"]["wave_instr_per_s"]
                res["lds_mix_bench"]["wave_instr_per_s"] = res["lds_mix_bench"]["mixed_rows_wave_instr_per_s"]
                res["lds_mix_bench"]["what"] = res["lds_mix_bench"]["mixed_rows_what"]
if args.lds_c4 and os.path.exists(args.lds_c4):
    for line in open(args.lds_c4):
        if line.startswith("{"):
            res["lds_c4_bench"] = json.loads(line)
if args.kernel_stats and os.path.exists(args.kernel_stats):
    # "Name","Calls","TotalDurationNs","AverageNs",...: the average launch duration of every kernel of the same command
    dur = {}
    for row in csv.DictReader(open(args.kernel_stats)):
        name = row.get("Name") or row.get("Kernel_Name") or ""
        avg = row.get("AverageNs") or row.get("Average(ns)") or row.get("AverageDurationNs")
        if name and avg:
            dur[name] = float(avg) * 1e-3
    res["avg_duration_us"] = dur
    for k, part in (res.get("per_kernel") or {}).items():
        base = k.split("::")[-1].split("(")[0]                # "k_m_list<16, 768>"
        hit = dur.get(k, next((v for n, v in dur.items() if base in n), None))
        if hit is not None:
            part["avg_duration_us"] = hit
res["commit"] = args.commit
json.dump(res, open(args.out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_kernel_mean_counters"}))
