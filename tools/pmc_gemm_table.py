#!/usr/bin/env python3
"""Side-by-side table of the per-shape PMC passes written by tools/pmc_gemm.sh (gpurun_out/pmcg_<shape>_<pass>/):
our kernel next to the vendor kernel torch.matmul picked, per launch.  usage: pmc_gemm_table.py <shape-prefix> ..."""
import collections
import csv
import glob
import re
import sys


def label(k):
    m = re.search(r"(x64[a-z0-9]*_kernel<[^>]*>|gemm_et_[a-z0-9]+_kernel<[^>]*>|MT[0-9x]+[A-Za-z0-9_]*)", k)
    return (m.group(1) if m else k)[:34]


for shape in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(f"gpurun_out/pmcg_{shape}_[0-9]")):
        for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for r in csv.DictReader(open(d + "/p_kernel_trace.csv")):
            acc[r["Kernel_Name"]]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    ks = [k for k in acc if ("gemm_et" in k or "Cijk" in k)]
    names = sorted(set(n for k in ks for n in acc[k]))
    print("=====", shape)
    print(f"{'counter (average per launch)':34s}" + "".join(f"{label(k):>36s}" for k in ks))
    for n in names:
        print(f"{n:34s}" + "".join(f"{(sum(acc[k][n]) / len(acc[k][n]) if acc[k][n] else float('nan')):36.1f}" for k in ks))
    print(f"{'launches (all passes)':34s}" + "".join(f"{len(acc[k]['duration_us']):36d}" for k in ks))
