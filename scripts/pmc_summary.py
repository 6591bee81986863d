#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters from one or more rocpd sqlite results (or counter_collection CSVs).

    python scripts/pmc_summary.py gpurun_out/j1/pmc1/pmc1_results.db [more ...]  [--filter conv_]

Prints, per kernel name (template arguments kept), the number of dispatches, the mean duration and the mean of every
counter; plus derived matrix-pipe occupancy when SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE are present:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
(on gfx950 rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: value / 8 / duration = 1.8-2.3 GHz; SQ_VALU_MFMA_BUSY_CYCLES
is exactly 32 x SQ_INSTS_MFMA for v_mfma_f32_32x32x16_bf16, i.e. matrix-pipe cycles summed over all 1024 SIMDs).
"""
import csv, re, sqlite3, sys
from collections import defaultdict

def rows(path):
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        for k, c, v, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            yield k, c, float(v), float(d)
    else:
        for r in csv.DictReader(open(path)):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

flt = None
paths = []
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--filter": flt = a.pop(0)
    else: paths.append(x)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for p in paths:
    seen = set()
    for k, c, v, d in rows(p):
        k = re.sub(r"r2dm::", "", k); k = re.sub(r"\(.*", "", k).replace("void ", "")
        if flt and flt not in k: continue
        acc[k][c][0] += v; acc[k][c][1] += 1
        dur[k][0] += d; dur[k][1] += 1
names = sorted(acc, key=lambda k: -dur[k][0])
for k in names:
    cs = acc[k]
    n = max(v[1] for v in cs.values())
    print(f"{k}\n    dispatches {n}  mean duration {dur[k][0] / dur[k][1] / 1e3:.1f} us")
    m = {c: v[0] / v[1] for c, v in cs.items()}
    for c in sorted(m): print(f"    {c:32s} {m[c]:16.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        busy = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024)
        clk = m['GRBM_GUI_ACTIVE'] / 8 / (dur[k][0] / dur[k][1])
        print(f"    -> matrix pipe busy = MFMA_BUSY / (GUI_ACTIVE/8 x 1024 SIMD) = {busy:.3f}")
        print(f"    -> effective clock  = GUI_ACTIVE/8 / duration = {clk:.3f} GHz   (busy x clock / 2.4 GHz = {busy * clk / 2.4:.3f} of the dense bf16 peak)")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in m: print(f"    -> {c} / SQ_WAVE_CYCLES = {m[c] / w:.3f}")
