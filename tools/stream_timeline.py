"""Per-queue occupancy of the steady-state bench steps from a rocprofv3 kernel trace (--kernel-trace, CSV): for each HIP
queue the busy time per step and its top kernels, the union (GPU not idle) and the time with >= 2 kernels in flight.
usage: stream_timeline.py <kernel_trace.csv> [steps_to_skip_at_both_ends]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
starts = [i for i, r in enumerate(rows) if 'camera_rays' in r['Kernel_Name']]
i0, i1 = starts[skip], starts[-skip]
steps = len(starts) - 2 * skip
sel = rows[i0:i1]
t0, t1 = int(sel[0]['Start_Timestamp']), int(rows[i1]['Start_Timestamp'])
span = (t1 - t0) / 1e3
qkey = 'Queue_Id' if 'Queue_Id' in sel[0] else 'Stream_Id'
busy = defaultdict(float); names = defaultdict(lambda: defaultdict(float))
ev = []
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r[qkey]
    busy[q] += (e - s) / 1e3
    names[q][r['Kernel_Name'].split('(')[0].split('<')[0][-40:]] += (e - s) / 1e3
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = t0; union = 0.0; multi = 0.0
for t, d in ev:
    if depth >= 1: union += (t - last) / 1e3
    if depth >= 2: multi += (t - last) / 1e3
    depth += d; last = t
print('%d steps, span %.1f us per step; GPU busy (union) %.1f us per step, >= 2 kernels in flight %.1f us per step' % (
    steps, span / steps, union / steps, multi / steps))
for q in sorted(busy, key=lambda k: -busy[k]):
    top = sorted(names[q].items(), key=lambda kv: -kv[1])[:6]
    print('queue %s: busy %.1f us per step | %s' % (q, busy[q] / steps, ', '.join('%s %.0f' % (k, v / steps) for k, v in top)))
