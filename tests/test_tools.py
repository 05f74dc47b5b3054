"""CPU: the evidence tools under tools/ that DESIGN.md's tables are generated from (they parse profiler output; a silent mis-label
would put proj's time in lin2's row)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z",
                    "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Queue_Id", "Stream_Id"])
        t = 0
        for name, dur_us, wg, grid in rows:
            w.writerow([name, t, t + dur_us * 1000, wg[0], wg[1], wg[2], grid[0], grid[1], grid[2], 1, 1])
            t += dur_us * 1000 + 2000


def test_trace_summary_names_roles_from_neighbours_and_counts_all_grid_axes(tmp_path):
    x64 = "void (anonymous namespace)::gemm_et_x64_kernel<1, true, false, 5, 3, 0, false, false, false>(unsigned short const*)"
    rows = [
        ("void (anonymous namespace)::layernorm_kernel<1>(float const*)", 50, (256, 1, 1), (8192 * 256, 1, 1)),
        ("void (anonymous namespace)::gemm_et_x64p_kernel<1, false, 0, false, false, false, false>(unsigned short const*)", 280, (512, 1, 1), (256 * 512, 1, 1)),
        ("void (anonymous namespace)::window_attention_kernel<1, 80, 0, 0>(unsigned short const*)", 150, (512, 1, 1), (256 * 512, 1, 1)),
        (x64, 160, (512, 1, 1), (512 * 512, 1, 1)),                                                  # after attention -> proj
        ("void (anonymous namespace)::layernorm_kernel<1>(float const*)", 50, (256, 1, 1), (8192 * 256, 1, 1)),
        ("void (anonymous namespace)::gemm_et_w4x_kernel<1, false, 2, true>(unsigned short const*)", 400, (256, 1, 1), (256 * 256, 1, 1)),
        (x64, 390, (512, 1, 1), (512 * 512, 1, 1)),                                                  # after lin1 -> lin2
        ("void (anonymous namespace)::i2t_fused_kernel<1, true>(unsigned short const*)", 90, (256, 1, 1), (32 * 256, 32, 1)),   # 2-D grid
        ("void (anonymous namespace)::weight_col_norms_kernel(float const*, int, int, float*)", 80, (1024, 1, 1), (20 * 1024, 1, 1)),
        ("void (anonymous namespace)::layernorm_kernel<1>(float const*)", 5, (256, 1, 1), (56 * 256, 1, 1)),
    ]
    tr = tmp_path / "t_kernel_trace.csv"
    _trace(tr, rows)
    out = tmp_path / "sum"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_summary.py"), str(tr), "--steps", "1", "--out", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {(row["role"], row["kernel"].split("<")[0], int(row["grid"])): row for row in csv.DictReader(open(str(out) + ".csv"))}
    roles = {(k[0], k[1]) for k in got}
    assert ("proj + residual", "gemm_et_x64_kernel") in roles and ("lin2 + residual", "gemm_et_x64_kernel") in roles
    assert ("lin1 + GELU", "gemm_et_w4x_kernel") in roles and ("qkv (windowed block)", "gemm_et_x64p_kernel") in roles
    assert ("LayerNorm (encoder block)", "layernorm_kernel") in roles and ("LayerNorm (decoder / neck)", "layernorm_kernel") in roles
    assert ("engine load (once per handle)", "weight_col_norms_kernel") in roles
    assert ("decoder / output side", "i2t_fused_kernel", 32 * 32) in got                     # all three grid axes, in blocks
    proj = got[("proj + residual", "gemm_et_x64_kernel", 512)]
    lin2 = got[("lin2 + residual", "gemm_et_x64_kernel", 512)]
    assert float(proj["avg_us"]) == 160.0 and float(lin2["avg_us"]) == 390.0
