#!/usr/bin/env python3
"""Per-shape roofline table of conv_f16x2_kernel from a rocprofv3 kernel trace of bench.py (config 1: 64x1024, batch 8), plus the
cost of the gn_finalize hops.  Usage: scripts/per_shape_table.py gpurun_out/<job>/bench_kt_kernel_trace.csv [batch] > profiles/<tag>_conv_shapes.txt

The 54 launches of a reverse step come in the engine's fixed order (r2dm_amd/csrc/engine.hip: d_block1..4, u_block4..1;
reference efficient_unet.py:95-110,132-139,169-176); launch i of every step is averaged over all steps in the trace.
peak = dense 16-bit MFMA 2500 TF/s / 3 products per fp32 product = 833.3 TF/s of algorithmic fp32 FLOPs."""
import csv, sys, collections
path = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, W = 64, 1024
def res(c, h, w, n): return [(f"res conv{j + 1}", c, c, h, w) for _ in range(n) for j in range(2)]
plan = []
plan += [("d1 " + t, ci, co, h, w) for (t, ci, co, h, w) in res(64, H, W, 3)]
plan += [("d2 downsample conv", 64, 128, H, W)] + [("d2 " + t, ci, co, h, w) for (t, ci, co, h, w) in res(128, H // 2, W // 2, 3)]
plan += [("d3 downsample conv", 128, 256, H // 2, W // 2)] + [("d3 " + t, ci, co, h, w) for (t, ci, co, h, w) in res(256, H // 4, W // 4, 3)]
plan += [("d4 downsample conv", 256, 512, H // 4, W // 4)] + [("d4 " + t, ci, co, h, w) for (t, ci, co, h, w) in res(512, H // 8, W // 8, 3)]
def up(name, cin, c, h, w):
    return [(name + " res0 conv1", cin, c, h, w), (name + " res0 conv2", c, c, h, w)] + [(name + " " + t, ci, co, hh, ww) for (t, ci, co, hh, ww) in res(c, h, w, 2)]
plan += up("u4", 512, 256, H // 8, W // 8) + [("u4 upsample conv", 256, 256, H // 4, W // 4)]
plan += up("u3", 512, 128, H // 4, W // 4) + [("u3 upsample conv", 128, 128, H // 2, W // 2)]
plan += up("u2", 256, 64, H // 2, W // 2) + [("u2 upsample conv", 64, 64, H, W)]
plan += up("u1", 128, 64, H, W)
assert len(plan) == 54
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
f2 = [r for r in rows if "conv_f16x2_kernel" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
nsteps = len(f2) // 54
f2 = f2[len(f2) - nsteps * 54:]  # whole steps, counted from the end (the trace starts with the weight load)
dur = collections.defaultdict(list); grid = {}
for i, r in enumerate(f2):
    dur[i % 54].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    grid[i % 54] = int(r["Grid_Size_X"]) // 512 if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) // 512
PEAK = 2500.0 / 3
print(f"# conv_f16x2_kernel, per launch position of a reverse step (batch {B}, {nsteps} steps of the trace averaged; rocprofv3 kernel trace of bench.py)")
print("%-3s %-22s %-10s %-9s %6s %8s %8s %8s %6s" % ("#", "layer", "Cin->Cout", "HxW", "blocks", "GFLOP", "us", "TF/s", "frac"))
agg = collections.OrderedDict(); tot_us = tot_gf = 0.0
for i, (name, ci, co, h, w) in enumerate(plan):
    gf = 2.0 * B * co * ci * 9 * h * w / 1e9
    us = sum(dur[i]) / len(dur[i])
    tf = gf / us / 1e3 * 1e3  # GFLOP / us = PFLOP/s*1e-3 -> TF/s: gf*1e9 / (us*1e-6) / 1e12
    tf = gf * 1e9 / (us * 1e-6) / 1e12
    print("%-3d %-22s %-10s %-9s %6d %8.2f %8.1f %8.1f %6.3f" % (i, name, f"{ci}->{co}", f"{h}x{w}", grid.get(i, 0), gf, us, tf, tf / PEAK))
    k = (f"{ci}->{co}", f"{h}x{w}", "plain input" if "sample conv" in name else "GroupNorm+SiLU input")
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += gf
    tot_us += us; tot_gf += gf
print("\n# by shape")
print("%-10s %-9s %-22s %8s %9s %9s %8s %6s" % ("Cin->Cout", "HxW", "input", "launches", "avg us", "total us", "TF/s", "frac"))
for (cc, hw, kind), (n, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = gf * 1e9 / (us * 1e-6) / 1e12
    print("%-10s %-9s %-22s %8d %9.1f %9.1f %8.1f %6.3f" % (cc, hw, kind, n, us / n, us, tf, tf / PEAK))
tf = tot_gf * 1e9 / (tot_us * 1e-6) / 1e12
print(f"\nall 54 launches: {tot_us / 1e3:.3f} ms per step, {tot_gf:.1f} GFLOP, {tf:.1f} TF/s = {tf / PEAK:.3f} of {PEAK:.1f} TF/s")
# ---- what a gn_finalize hop costs: producer end -> consumer start, with a finalize launch in between, against a direct hop
names = [r["Kernel_Name"] for r in rows]
hop_fin, hop_direct, fin_us = [], [], []
for i in range(1, len(rows) - 1):
    if "gn_finalize" in names[i] and "conv" in names[i - 1] and "conv" in names[i + 1]:
        hop_fin.append((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
        fin_us.append((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3)
    if "conv" in names[i] and "conv" in names[i + 1]:
        hop_direct.append((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print(f"\n# gn_finalize hops (conv -> gn_finalize -> conv): median {med(hop_fin):.1f} us from the producer's end to the consumer's start, of which the kernel {med(fin_us):.1f} us;"
      f" a direct conv -> conv hop: median {med(hop_direct):.1f} us.  {len(hop_fin) / max(nsteps, 1):.0f} such hops per step -> folding every finalize into a neighbour"
      f" would return at most {(med(hop_fin) - med(hop_direct)) * len(hop_fin) / max(nsteps, 1) / 1e3:.3f} ms per step")
