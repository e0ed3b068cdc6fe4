#!/usr/bin/env python3
"""Folds the two rocprofv3 --pmc passes (gpurun_out/pmc_fetch, gpurun_out/pmc_write; tools/pmc_bench.sh) into
profiles/<tag>_pmc_1Brows_16cols.json, the file bench.py reads `roofline.traffic` from.

    python tools/make_pmc_json.py r01 1000000000
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def per_kernel_summary(path):
    """the per-kernel means tools/pmc_bench.sh printed (it clears its directory): `name COUNTER dispatches N mean V`"""
    out = {}
    for line in open(path):
        parts = line.split()
        if "tgx::" not in line or "dispatches" not in parts:
            continue
        i = parts.index("dispatches")
        name = "tgx::" + " ".join(parts[:i - 1]).split("tgx::")[1]
        out[name] = (int(parts[i + 1]), float(parts[i + 3]))
    return out


def per_kernel(d, tag=None):
    found = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True)
    if not found:
        return per_kernel_summary(os.path.join(ROOT, "gpurun_out", "%s_%s.txt" % (tag, d)))
    f = found[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "tgx::" not in name:
            continue
        name = "tgx::" + name.split("tgx::")[1].split("(")[0]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    tag, rows = sys.argv[1], int(sys.argv[2])
    fetch, write = per_kernel("pmc_fetch", tag), per_kernel("pmc_write", tag)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        kernels[k] = {"dispatches": fetch.get(k, write.get(k))[0],
                      "FETCH_SIZE_KB_mean": fetch.get(k, (0, None))[1],
                      "WRITE_SIZE_KB_mean": write.get(k, (0, None))[1]}
    scan = next(v for k, v in kernels.items() if "scan_kernel<" in k)
    sys.path.insert(0, ROOT)
    from term_amd import synth
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
                   "   (and a second pass with --pmc WRITE_SIZE); tools/pmc_bench.sh + tools/make_pmc_json.py",
        "rows_total": rows, "n_gpus": 1,
        "note": "FETCH_SIZE / WRITE_SIZE are in KiB per dispatch. On gfx950 FETCH_SIZE reports half the bytes of a "
                "16 B/lane coalesced stream (MI355X_MICROARCH.md, HBM section): scan_kernel reads only with "
                "global_load_dwordx4, so its HBM read bytes = 2 x FETCH_SIZE x 1024. WRITE_SIZE is exact for wide stores.",
        "kernels": kernels,
        "scan_kernel_traffic_bytes_per_launch": 2 * scan["FETCH_SIZE_KB_mean"] * 1024 + scan["WRITE_SIZE_KB_mean"] * 1024,
        # the 14 columns scan_kernel reads: the two unique-key columns ride on their DISTINCT pass (partition_kernel)
        "scan_kernel_algorithmic_bytes_per_launch": synth.algorithmic_bytes(
            [c for i, c in enumerate(synth.COLUMNS_16) if i not in synth.UNIQUE_COLUMNS_16], rows),
        "suite_algorithmic_bytes": synth.algorithmic_bytes(synth.COLUMNS_16, rows),
        "suite_traffic_bytes_per_step": sum(
            (2 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]) * 1024 * v["dispatches"] for v in kernels.values()
            if v["FETCH_SIZE_KB_mean"] is not None and v["WRITE_SIZE_KB_mean"] is not None) / max(1, scan["dispatches"]),
    }
    path = os.path.join(ROOT, "profiles", "%s_pmc_1Brows_16cols.json" % tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, out["scan_kernel_traffic_bytes_per_launch"], out["scan_kernel_algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main()
