#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSVs (one pass per counter group) into profiles/hbm_traffic.json.

HBM bytes per launch of the fused sequence kernel (k_em_grp, else k_em_seq) = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both
in KiB and on gfx950 FETCH_SIZE reads half of the bytes a coalesced stream fetches
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, positions = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, cs in acc.items():
    summary[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summary[k]["dispatches"] = max(len(v) for v in cs.values())
main = next((k for k in summary if "k_em_grp" in k and "true, false" in k), None) or \
    next((k for k in summary if "k_em_seq" in k), None)
res = {"positions_per_launch": positions, "per_kernel_mean_counters": summary}
if main and "FETCH_SIZE" in summary[main] and "WRITE_SIZE" in summary[main]:
    f, w = summary[main]["FETCH_SIZE"], summary[main]["WRITE_SIZE"]
    res.update(kernel=main, fetch_size_kib=f, write_size_kib=w,
               hbm_bytes_per_launch=(2.0 * f + w) * 1024.0,
               hbm_bytes_per_launch_uncorrected=(f + w) * 1024.0)
m = summary.get(main, {}) if main else {}
if all(c in m for c in ("GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU")):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ_* counters over the 256 CUs (cycles of the CU's
    # LDS / VALU pipes being busy): busy fraction = per-CU busy cycles / kernel cycles
    XCDS, CUS = 8.0, 256.0
    cyc = m["GRBM_GUI_ACTIVE"] / XCDS
    res["derived"] = {"gpu_cycles_per_launch": cyc,
                      "lds_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / CUS / cyc,
                      "lds_bank_conflict_frac": m["SQ_LDS_BANK_CONFLICT"] / CUS / cyc,
                      "valu_busy_frac": m["SQ_ACTIVE_INST_VALU"] / CUS / cyc}
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_kernel_mean_counters"}))
