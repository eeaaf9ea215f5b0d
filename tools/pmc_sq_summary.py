#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc SQ_* counter_collection CSV, with the usual ratios:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
        --output-format csv -d <dir> -- python3 tools/bench_configs.py
    python tools/pmc_sq_summary.py <dir>/*/*_counter_collection.csv > profiles/<round>_pmc_sq_all_configs.json

frac_active_valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of wave-cycles issuing VALU);
simd_cycles_per_valu_inst / valu_busy_floor: see below."""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
meta = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "hh::" not in k:
            continue
        acc[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
        meta[k] = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], wg=r["Workgroup_Size"],
                       grid=r["Grid_Size"], lds=r.get("LDS_Block_Size", ""))
out = {}
for k, ctrs in acc.items():
    o = dict(meta[k])
    o["launches"] = len(next(iter(ctrs.values())))
    for name, vals in ctrs.items():
        o[name] = sum(v for _, v in vals) / len(vals)
    wc = o.get("SQ_WAVE_CYCLES")
    if wc:
        for name, key in (("SQ_ACTIVE_INST_ANY", "frac_active_inst_any"),
                          ("SQ_ACTIVE_INST_VALU", "frac_active_valu"),
                          ("SQ_WAIT_ANY", "frac_wait_any"), ("SQ_WAIT_INST_ANY", "frac_wait_inst_any")):
            if name in o:
                o[key] = o[name] / wc
    # elapsed SIMD-cycles per VALU instruction issued on that SIMD: GRBM_GUI_ACTIVE is summed over
    # the 8 XCDs, SQ_INSTS_VALU over the 1024 SIMDs; v_fma_f64 occupies the pipe for 4 cycles
    # (tools/ubench/valu_rates.hip), so 4 / this figure is a floor of the VALU-busy fraction.
    if o.get("GRBM_GUI_ACTIVE") and o.get("SQ_INSTS_VALU"):
        o["simd_cycles_per_valu_inst"] = o["GRBM_GUI_ACTIVE"] / 8 * 1024 / o["SQ_INSTS_VALU"]
        o["valu_busy_floor"] = min(1.0, 4.0 / o["simd_cycles_per_valu_inst"])
    out[k] = o
json.dump(out, sys.stdout, indent=1)
