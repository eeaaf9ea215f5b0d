#!/usr/bin/env python3
"""Merges the SQ_INSTS_VALU counts of a `rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/pmc_round5.py` pass into
profiles/valu_insts.json (the rows bench.py's VALU rooflines read).
usage: valu_insts_r5.py <counter csv of tools/pmc_round5.py> <tag> [<counter csv of tools/bk_only.py grid>]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M = 1_000_000, 252
k = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        k[r["Kernel_Name"]].append(float(r["Counter_Value"]))
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = f"rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/pmc_round5.py, round {tag}"


def mean_of(pick):
    hits = [v for n, v in k.items() if pick(n)]
    assert hits, [n[:90] for n in k]
    return sum(sum(v) / len(v) for v in hits)


rows = {
    "heston_euler_generate": (mean_of(lambda n: "euler_kernel<hh::HestonModel<0, true>, 0, false, false" in n) / (N * M), "path-step"),
    "heston_euler_replay": (mean_of(lambda n: "euler_kernel<hh::HestonModel<0, true>, 0, true, false" in n) / (N * M), "path-step"),
    "heston_euler_generate_multi2": (mean_of(lambda n: "euler_multi_kernel<hh::HestonModel<0, true>, false, false, 2" in n) / (N * M),
                                     "path-step of the PASS (two models stepped on it)"),
    "heston_euler_replay_multi2": (mean_of(lambda n: "euler_multi_kernel<hh::HestonModel<0, true>, true, false, 2" in n) / (N * M),
                                   "path-step of the PASS (two models stepped on it)"),
    "lognormal_exact": (mean_of(lambda n: "exact_gbm_kernel<0, false, false, 4>" in n) / N, "path (four pairs per lane: 10^6 paths)"),
    "lognormal_exact_1e8": (mean_of(lambda n: "exact_gbm_kernel<0, false, false, 64>" in n) / (100 * N), "path (64 pairs per lane: 10^8 paths)"),
    "broadie_kaya": (mean_of(lambda n: "::bk_" in n) / N, "path (bk_cf_kernel: draws, series, inversion, ladder + bk_tail_kernel)"),
}
if len(sys.argv) > 3:  # the exact grid 2·10^5 x 12: EVERY kernel of the chain — draws + keys, the three counting-sort kernels, CF, tail, spots
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[3])):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and "fill_rows" not in r["Kernel_Name"] and "bk_tables" not in r["Kernel_Name"]:
            g[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    total = sum(sum(v) / len(v) for v in g.values())
    rows["heston_exact_grid"] = (total / (200_000 * 12),
                                 "transition (variance rows + sort by V0·V_T + the chain over all (date, trajectory) pairs + spot rows)")
dst = os.path.join(ROOT, "profiles", "valu_insts.json")
out = json.load(open(dst))
for key, (v, unit) in rows.items():
    out[key] = {"valu_insts_per_unit": v, "unit": unit, "source": src}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({a: out[a]["valu_insts_per_unit"] for a in rows}, indent=1))
