#!/usr/bin/env python
"""Per-(kernel, role, grid) summary of a rocprofv3 kernel trace -- the evidence DESIGN.md's per-kernel tables quote.

    rocprofv3 --kernel-trace --stats -d gpurun_out/trace -o r06 -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline ...
    python tools/trace_summary.py gpurun_out/trace/r06_results.db --steps 8 --out profiles/r06_trace_summary

`--stats` merges every launch of a kernel into one row: proj and lin2 (the same gemm_et_x64_kernel instantiation), the encoder's
and the decoder's LayerNorm, the windowed and the global qkv GEMM cannot be told apart there (VERDICT r05 "what's weak" 7).  This
script reads the dispatch table of the rocpd database (or the kernel-trace CSV) and names the ROLE of each launch from what
runs next to it on the same queue:

    gemm_et_x64 (fp32 residual output)   after an attention kernel -> proj;  after the lin1 kernel -> lin2;  else by grid
    layernorm                            grid = rows / 4: 8 tiles x 4096 tokens -> encoder;  anything else -> decoder / neck
    gemm_et_x64p / w4x (ET output)       with GELU -> lin1;  followed by window_attention -> qkv (windowed);  by vt_pack -> qkv (global)

Writes <out>.csv and <out>.md: role, kernel, grid, calls, avg / min / max us, total ms, ms per step, share of all kernel time.
"""
from __future__ import annotations

import argparse
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    m = re.match(r"^\d*([A-Za-z_0-9]+?)(I[LbE0-9_a-z]*E*)?v?P", name)          # mangled: <len>name I<targs>E ...
    if name.startswith(tuple("0123456789")) and m:
        return m.group(1) + (("<" + m.group(2) + ">") if m.group(2) else "")
    return name.split("(")[0]


def load_db(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, d.start, d.end, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z, d.queue_id, d.stream_id "
         f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start")
    return [dict(name=r[0], start=r[1], end=r[2], grid=r[3] // max(1, r[4]), queue=(r[5], r[6])) for r in db.execute(q)]


def load_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        geti = lambda k, d: int(r.get(k, d) or d)
        wg = geti("Workgroup_Size_X", geti("Workgroup_Size", 1)) * geti("Workgroup_Size_Y", 1) * geti("Workgroup_Size_Z", 1)
        gx = geti("Grid_Size_X", geti("Grid_Size", 0)) * geti("Grid_Size_Y", 1) * geti("Grid_Size_Z", 1)      # work items, all three axes
        rows.append(dict(name=r["Kernel_Name"], start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"]), grid=gx // max(1, wg),
                         queue=(r.get("Queue_Id", "0"), r.get("Stream_Id", "0"))))
    rows.sort(key=lambda x: x["start"])
    return rows


def roles(rows, enc_rows):
    """Label every launch; neighbours are taken on the launch's own queue / stream."""
    by_q = defaultdict(list)
    for i, r in enumerate(rows):
        by_q[r["queue"]].append(i)
    out = [""] * len(rows)
    for idx in by_q.values():
        names = [rows[i]["name"] for i in idx]
        for k, i in enumerate(idx):
            n = names[k]
            prev = names[k - 1] if k else ""
            nxt = names[k + 1] if k + 1 < len(idx) else ""
            role = ""
            if "gemm_et_x64_kernel" in n and ("Lb1ELb0E" in n or "<1, true, false" in n or "true, false, 5" in n):
                if "attention_kernel" in prev or "outlier_gather" in prev:
                    role = "proj + residual"
                elif "w4x" in prev or "x64p_kernel" in prev or "outlier_side_gemm" in prev:
                    role = "lin2 + residual"
                else:
                    role = "fp32-output GEMM (neck / patch / decoder)"
            elif "gemm_et_w4x_kernel" in n:
                role = "lin1 + GELU"
            elif "gemm_et_x64p_kernel" in n:
                m3 = re.search(r"x64p_kernelILi\d+ELb[01]ELi(\d)E", n) or re.search(r"x64p_kernel<\d+, (?:true|false), (\d)", n)
                if m3 and m3.group(1) != "0":
                    role = "lin1 + GELU"
                elif "window_attention" in nxt:
                    role = "qkv (windowed block)"
                elif "vt_pack" in nxt or "global_attention" in nxt:
                    role = "qkv (global block)"
                else:
                    role = "ET-output GEMM"
            elif "layernorm_kernel" in n:
                role = "LayerNorm (encoder block)" if rows[i]["grid"] == enc_rows // 4 else "LayerNorm (decoder / neck)"
            elif "gemm_f32_kernel" in n:
                role = "decoder token side (fp32)"
            elif "window_attention_kernel" in n:
                role = "windowed attention"
            elif "global_attention_kernel" in n or "vt_pack" in n:
                role = "global attention"
            elif any(k in n for k in ("t2i_", "i2t_", "upscaler", "token_self_attn", "k256", "prompt_tokens", "make_keys", "postprocess", "paint_area",
                                      "class_stats", "select_best", "rle_")):
                role = "decoder / output side"
            elif any(k in n for k in ("weight_col_norms", "weight_row_norms", "outlier_weight_ext", "outlier_side_weight", "mx4_pack_kernel")):
                role = "engine load (once per handle)"
            elif "outlier_" in n:
                role = "outlier-column side operands"
            out[i] = role
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=float, default=0, help="timed + warm-up steps in the trace (for the ms / step column)")
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = load_db(a.trace) if a.trace.endswith(".db") else load_csv(a.trace)
    lab = roles(rows, a.tiles * 4096)
    agg = defaultdict(list)
    for r, role in zip(rows, lab):
        agg[(role, short(r["name"]), r["grid"])].append((r["end"] - r["start"]) / 1e3)
    total = sum(sum(v) for v in agg.values())
    table = []
    for (role, name, grid), v in agg.items():
        table.append(dict(role=role, kernel=name[:70], grid=grid, calls=len(v), avg_us=sum(v) / len(v), min_us=min(v), max_us=max(v),
                          total_ms=sum(v) / 1e3, ms_per_step=(sum(v) / 1e3 / a.steps) if a.steps else 0.0, share=100 * sum(v) / total))
    table.sort(key=lambda t: -t["total_ms"])
    cols = ["role", "kernel", "grid", "calls", "avg_us", "min_us", "max_us", "total_ms", "ms_per_step", "share"]
    fmt = lambda t, c: (f"{t[c]:.1f}" if c in ("avg_us", "min_us", "max_us") else f"{t[c]:.2f}" if c in ("total_ms", "ms_per_step", "share") else str(t[c]))
    if a.out:
        with open(a.out + ".csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            for t in table:
                w.writerow([fmt(t, c) for c in cols])
        with open(a.out + ".md", "w") as f:
            f.write(f"rocprofv3 kernel trace `{a.trace}`: {len(rows)} launches, {total / 1e3:.1f} ms of kernel time"
                    + (f", {a.steps:g} steps" if a.steps else "") + "\n\n| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
            for t in table[:48]:
                f.write("| " + " | ".join(fmt(t, c) for c in cols) + " |\n")
    for t in table[:40]:
        print(f"{t['role'][:34]:34s} {t['kernel'][:44]:44s} grid {t['grid']:7d} x{t['calls']:5d} avg {t['avg_us']:8.1f} us  min {t['min_us']:8.1f}  "
              f"{t['ms_per_step']:6.2f} ms/step {t['share']:5.1f} %")


if __name__ == "__main__":
    sys.exit(main())
