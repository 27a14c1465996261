#!/usr/bin/env python3
"""Per-SHAPE table of the projections of the replayed prefill passes (one number per kernel AND shape, never a mean over
two launch sizes).  Reads the raw rocprofv3 kernel trace of `tools/prof_phases.py prefill` -- per rep a 16 x 802-row pass and a 16 x
330-row pass, 36 layers each, every layer qkv <NONE>, o <RESIDUAL>, gate/up <SWIGLU>, down <RESIDUAL> in that order between the pass's
k_embed_rows and its k_logits_multi -- and writes kernel, projection, rows, grid (workgroups), dispatches, mean / min / max us,
TFLOP/s (2 x rows x N x K, unpadded) and the fraction of the 2.5 PFLOP/s dense bf16 peak.
usage: prefill_by_shape.py <rocprof_out_dir> <out.csv> [rows_a=12832 rows_b=5280]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
rows_ab = (int(sys.argv[3]) if len(sys.argv) > 3 else 12832, int(sys.argv[4]) if len(sys.argv) > 4 else 5280)
SHAPES = {"qkv": (2560, 2048), "o": (2048, 2048), "gate_up": (22016, 2048), "down": (2048, 11008)}
disp = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            wg = max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
            disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // wg))
disp.sort()
acc = defaultdict(list)
in_pass, n_pass, seq = False, 0, 0
for t0, t1, name, grid in disp:
    if name.startswith("k_embed_rows"):
        in_pass, seq = True, 0
        continue
    if "k_logits_multi" in name:
        if in_pass:
            n_pass += 1
        in_pass = False
        continue
    if not in_pass or "k_gemm" not in name:
        continue
    proj = ("qkv", "o", "gate_up", "down")[seq % 4]
    seq += 1
    acc[(name.split("(")[0].replace("void ", ""), proj, rows_ab[n_pass % 2], grid)].append((t1 - t0) / 1000.0)
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "projection", "rows", "N", "K", "grid_workgroups", "dispatches", "mean_us", "min_us", "max_us", "TFLOPs_at_mean", "frac_of_2500_TFLOPs"])
    for (k, proj, rows, grid), us in sorted(acc.items(), key=lambda kv: (-kv[0][2], kv[0][1])):
        n, kk = SHAPES[proj]
        m = sum(us) / len(us)
        tf = 2.0 * rows * n * kk / (m * 1e-6) / 1e12
        w.writerow([k, proj, rows, n, kk, grid, len(us), round(m, 1), round(min(us), 1), round(max(us), 1), round(tf, 1), round(tf / 2500.0, 4)])
        print(f"{proj:8s} rows {rows:6d} grid {grid:5d} x{len(us):4d}  {m:8.1f} us  {tf:7.1f} TFLOP/s = {tf / 2500.0:.3f}   {k}")
print(f"wrote {dst} ({n_pass} passes)")
