"""Kernels of ONE steady-state step on the queue that launches camera_rays (the caller's stream), in order: the serial chain a
small batch is bound by.  usage: queue_listing.py <kernel_trace.csv> [step index from the end = 5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
qk = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
# the tail's queue: the one with mlp_backward16s / sg_render_bwd
from collections import Counter
q = Counter(r[qk] for r in rows if 'sg_render_bwd' in r['Kernel_Name'] or 'mc_shade_bwd' in r['Kernel_Name']).most_common(1)[0][0]
mine = [r for r in rows if r[qk] == q]
starts = [i for i, r in enumerate(mine) if 'idr_loss_kernel' in r['Kernel_Name']]
a, b = starts[-back - 1], starts[-back]
t0 = int(mine[a]['Start_Timestamp'])
tot = 0
for r in mine[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print('+%8.1f us %7.1f us  grid %-8s wg %-5s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'),
                                                       r['Kernel_Name'][:150]))
print('%d kernels, %.1f us busy, span %.1f us' % (b - a, tot / 1e3, (int(mine[b]['Start_Timestamp']) - t0) / 1e3))
