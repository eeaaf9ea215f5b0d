#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — collected in SEPARATE runs, with
--kernel-trace only) into profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

gfx950 corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so the read side is doubled (calibrated for
this kernel: 8 B per lane, a wave covers 4 full 128-B lines; raw count identical to the 16 B/lane
form on the same 4.032 GB buffer); WRITE_SIZE is exact for streaming stores (checked here
against wiener_fill_kernel, whose store volume is known exactly).

usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [tag]
"""
import collections
import csv
import json
import os
import sys

KERNEL = "euler_kernel<hh::HestonModel<0, true>, 0, true, false, 1, 0, false>"  # price-only REPLAY (256 threads)


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


PM_KERNEL = "euler_pm_kernel<hh::HestonModel<0, true>, 0, false>"  # path-major REPLAY, price-only


def path_major(fetch_csv, write_csv, tag):
    """usage: tools/pmc_traffic.py pm <fetch csv> <write csv> [tag] — adds the path-major kernel's
    traffic (PMC passes of `tools/tune_pm.py run 1` with HH_VARIANTS='{"main": []}') to the JSON.
    Its reads are 16 B per lane LDS-DMA of whole aligned 128-B lines: the x2 of the guide's gfx950 note."""
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    fk = next(k for k in fetch if PM_KERNEL in k)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                       "pmc_traffic.json")
    out = json.load(open(dst))
    f_kib, n = fetch[fk]
    w_kib = write[fk][0]
    tile = next((k for k in fetch if KERNEL in k), None)
    out.update({
        "path_major_kernel": fk, "path_major_launches": n,
        "path_major_FETCH_SIZE_KiB_raw": f_kib, "path_major_WRITE_SIZE_KiB_raw": w_kib,
        "path_major_hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
        "path_major_same_pass_tile_major_FETCH_SIZE_KiB_raw": fetch[tile][0] if tile else None,
        "path_major_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 "
                             "tools/tune_pm.py run 1 (HH_VARIANTS={\"main\": []}); round " + tag})
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.startswith("path_major")}, indent=1))


def main():
    if sys.argv[1] == "pm":
        return path_major(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "r03")
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
    fk = next(k for k in fetch if KERNEL in k)
    f_kib, n = fetch[fk]
    w_kib, _ = write[fk]
    fill = next((k for k in write if "wiener_fill_kernel" in k), None)
    out = {
        "kernel": fk,
        "launches": n,
        "FETCH_SIZE_KiB_raw": f_kib,
        "WRITE_SIZE_KiB_raw": w_kib,
        "read_bytes_corrected": 2.0 * f_kib * 1024.0,
        "write_bytes": w_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
        "correction": "FETCH_SIZE x2 (gfx950: 128-B requests of a coalesced streaming read tallied at 64 B; the 8 B/lane loads of this kernel give the same raw count as the 16 B/lane kernel it replaced on the same buffer), KiB -> bytes",
        "calibration": {"wiener_fill_kernel_WRITE_SIZE_bytes": write[fill][0] * 1024.0 if fill else None,
                        "expected_bytes": 3907 * 252 * 2 * 256 * 8},
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                  "--steps 3 --warmup 1 --no-cpu-baseline; round " + tag,
    }
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                       "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
