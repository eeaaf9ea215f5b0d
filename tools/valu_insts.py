#!/usr/bin/env python3
"""profiles/valu_insts.json — VALU wave-instructions per unit of work of the VALU-bound kernels, from
rocprofv3 PMC passes (SQ_INSTS_VALU summed over the chip, per dispatch).  bench.py multiplies them
by 4 SIMD-cycles (an fp64 FMA holds a SIMD's vector pipe that long) and divides by the live kernel
time and by 1024 SIMDs x 2.4 GHz: the fp64-VALU issue fraction of its roofline entries.

    rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d A -- python3 tools/bench_configs.py
    rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d B -- python3 tools/bk_only.py grid
    rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d L -- python3 tools/lsm_latency.py 1000000
    python tools/valu_insts.py A/*/*_counter_collection.csv B/*/*_counter_collection.csv [L/*/*_counter_collection.csv] [tag]
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    """kernel name -> [SQ_INSTS_VALU of each dispatch]"""
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def mean_of(k, names, pick):
    hits = [v for n, v in k.items() if pick(n)]
    assert hits, (names, list(k)[:5])
    return sum(sum(v) / len(v) for v in hits)


def main():
    a, b = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    lsm_csv = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".csv") else None
    tag = sys.argv[-1] if not sys.argv[-1].endswith(".csv") else "r02"
    src = f"rocprofv3 --pmc SQ_INSTS_VALU, round {tag}"
    N, M = 1_000_000, 252
    gen = mean_of(a, "generate", lambda n: "euler_kernel<hh::HestonModel<0, true>, 0, false, false" in n)
    exact = mean_of(a, "exact", lambda n: "exact_gbm_kernel<0, false, false>" in n)
    bk = mean_of(a, "bk", lambda n: "::bk_" in n)          # the five kernels of one chain, per launch each
    grid = mean_of(b, "grid", lambda n: "::bk_" in n)      # one chain over the 2e5 x 12 (date, trajectory) pairs
    out = {
        "heston_euler_generate": {"valu_insts_per_unit": gen / (N * M), "unit": "path-step", "source": src},
        "lognormal_exact": {"valu_insts_per_unit": exact / N, "unit": "path", "source": src},
        "broadie_kaya": {"valu_insts_per_unit": bk / N, "unit": "path (draw + cf (series, inversion) + scan + ladder + fall-back kernels)",
                         "source": src},
        "heston_exact_grid": {"valu_insts_per_unit": grid / (200_000 * 12),
                              "unit": "transition (variance rows + the chain over all (date, trajectory) pairs + spot rows)",
                              "source": src},
        "_what": "VALU wave-instructions (64 lanes each) per unit, SQ_INSTS_VALU averaged over the dispatches of a kernel",
    }
    if lsm_csv:  # tools/lsm_latency.py 1000000: 2*10^6 trajectories x 100 dates, both forms; the one-launch chain
        l = per_kernel(lsm_csv)
        ind = mean_of(l, "lsm", lambda n: "lsm_persistent_kernel<5, 16>" in n)
        grid = mean_of(l, "grid", lambda n: "gbm_grid_kernel<true>" in n)
        fin = mean_of(l, "final", lambda n: "lsm_final_kernel" in n)
        out["lsm_chain"] = {"valu_insts_per_unit": (ind + grid + fin) / (2_000_000 * 100),
                            "unit": "(trajectory, date): gbm_grid_kernel + lsm_persistent_kernel + lsm_final_kernel",
                            "induction_share": ind / (ind + grid + fin), "source": src}
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_insts.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
