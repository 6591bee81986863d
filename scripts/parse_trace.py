#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of scripts/time_forward.py: per conv variant x grid, and other kernels."""
import csv, re, collections, sys
path, nfwd = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = list(csv.DictReader(open(path)))
agg, other = collections.defaultdict(list), collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'conv_mfma' not in n:
        other[re.sub(r'\(.*', '', n)[:50]].append(d); continue
    m = re.search(r'<(.*?)>', n).group(1).replace(' ', '')
    agg[(m, int(r['Grid_Size_X']) // 256, r['VGPR_Count'], r['Accum_VGPR_Count'], r['Scratch_Size'])].append(d)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, 'n/fwd=%.0f' % (len(v) / nfwd), 'avg %.1f us' % (sum(v) / len(v)), 'min %.1f' % min(v), 'total/fwd %.2f ms' % (sum(v) / 1e3 / nfwd)); tot += sum(v)
print('conv total per fwd ms %.2f' % (tot / 1e3 / nfwd))
o = 0
for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print('%-50s n/fwd=%.1f avg %.1f us total/fwd %.3f ms' % (k, len(v) / nfwd, sum(v) / len(v), sum(v) / 1e3 / nfwd)); o += sum(v)
print('other (top10) per fwd ms %.2f' % (o / 1e3 / nfwd))
