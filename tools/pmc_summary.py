#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by `tools/gpu_round.sh pmc` (gpurun_out/pmc1..4).

Per (kernel, grid size): every counter averaged over the launches of that kernel, plus the derived figures
DESIGN.md quotes.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
  * FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1 kB = 1024 B per count;
  * gfx950 reports HALF of the bytes of wide coalesced reads in FETCH_SIZE -> doubled here.

    python tools/pmc_summary.py gpurun_out profiles/r01_v6_pmc_per_kernel.json \
        --dominant "gemm_et_big_kernel<1, false, true" profiles/r01_dominant_kernel_pmc.json
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("out")
    ap.add_argument("--dominant", nargs=2, metavar=("KERNEL_PREFIX", "OUT"))
    args = ap.parse_args()

    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(args.root, "pmc[0-9]*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    key = f"{short(r['Kernel_Name'])} grid={r['Grid_Size']}"
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for key, ctr in sorted(acc.items()):
        row = {c: sum(v) / len(v) for c, v in sorted(ctr.items())}
        row["launches"] = max(len(v) for v in ctr.values())
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes_per_launch"] = (2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024.0
        if row.get("TCC_HIT_sum", 0) + row.get("TCC_MISS_sum", 0) > 0:
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
        if row.get("GRBM_GUI_ACTIVE") and row.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            # MFMA-busy is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (each = kernel duration in cycles)
            row["mfma_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * row["GRBM_GUI_ACTIVE"])
        if row.get("SQ_WAVE_CYCLES"):
            row["wait_frac"] = row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"]
        out[key] = row
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{len(out)} kernel/grid rows -> {args.out}")

    if args.dominant:
        prefix, dout = args.dominant
        cands = {k: v for k, v in out.items() if k.startswith(prefix) and "hbm_bytes_per_launch" in v}
        if not cands:
            raise SystemExit(f"no kernel starting with {prefix!r} has FETCH/WRITE counters")
        key = max(cands, key=lambda k: cands[k]["hbm_bytes_per_launch"] * cands[k]["launches"])
        v = cands[key]
        dom = {
            "kernel": key,
            "fetch_size_kb": v["FETCH_SIZE"],
            "write_size_kb": v["WRITE_SIZE"],
            "traffic_bytes_per_launch": v["hbm_bytes_per_launch"],
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_round.sh pmc); FETCH_SIZE doubled "
                    "per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); averaged over launches of this "
                    "kernel / grid size",
            "mfma_busy_frac": v.get("mfma_busy_frac"),
            "l2_hit_rate": v.get("l2_hit_rate"),
            "launches": v["launches"],
            # bench.py reports `traffic` only while this hash matches the gemm.hip it runs (a stale measurement is dropped)
            "gemm_hip_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                               "samrs_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16],
            "source": "profiles/dominant_kernel_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, tools/gpu_round.sh pmc)",
        }
        with open(dout, "w") as fh:
            json.dump(dom, fh, indent=1)
        print(f"dominant: {key}: {dom['traffic_bytes_per_launch'] / 1e9:.3f} GB / launch -> {dout}")


if __name__ == "__main__":
    main()
