#!/usr/bin/env python3
"""Per-conv-launch durations in launch order from a rocprofv3 kernel trace of scripts/time_forward.py (last forwards)."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = min(8, len(rows) // N)
per = [[] for _ in range(N)]
for i, r in enumerate(rows[-N * reps:]):
    per[i % N].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for i, r in enumerate(rows[-N:]):
    nm = r['Kernel_Name']
    m = re.search(r'(conv_\w+)<(.*?)>', nm)
    avg = sum(per[i]) / len(per[i]); tot += avg
    print(i, m.group(1).replace('conv_', '').replace('_kernel', ''), m.group(2).replace(' ', ''), int(r['Grid_Size_X']) // 256, 'vgpr', r['VGPR_Count'], '+', r['Accum_VGPR_Count'], '%.1f us' % avg)
print('conv total %.2f ms' % (tot / 1e3))
