"""Timeline of one steady-state bench step from a rocprofv3 kernel trace: kernels in launch order with start offset,
duration and the idle gap before each.  usage: step_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# a step starts at each camera_rays launch
starts = [i for i, r in enumerate(rows) if 'camera_rays' in r['Kernel_Name']]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
busy = 0
agg = {}
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].split('<')[0][-48:]
    gap = s - prev_end
    busy += e - s
    a = agg.setdefault(name, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    prev_end = max(prev_end, e)
span = int(rows[i1]['Start_Timestamp']) - t0
print('step span %.3f ms, kernels busy %.3f ms, idle %.3f ms, %d launches' % (span / 1e6, busy / 1e6, (span - busy) / 1e6, i1 - i0))
print('%-50s %5s %10s %12s' % ('kernel', 'n', 'busy us', 'gap-before us'))
for k, (n, b, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print('%-50s %5d %10.1f %12.1f' % (k, n, b / 1e3, g / 1e3))
