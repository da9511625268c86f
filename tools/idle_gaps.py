"""GPU-idle gaps of the steady-state steps in a rocprofv3 kernel trace: intervals in which NO kernel of any queue runs,
with the kernel that ended before and the one that started after each.  usage: idle_gaps.py <kernel_trace.csv> [min_gap_us=30]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
starts = [i for i, r in enumerate(rows) if 'camera_rays' in r['Kernel_Name']]
i0, i1 = starts[3], starts[-3]
steps = len(starts) - 6
short = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0][-44:]
end, last = int(rows[i0]['Start_Timestamp']), rows[i0]
gaps = []
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end:
        gaps.append(((s - end) / 1e3, short(last), short(r)))
    if e > end:
        end, last = e, r
span = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3 / steps
tot = sum(g[0] for g in gaps) / steps
big = [g for g in gaps if g[0] >= min_gap]
print('%d steps, %.1f us per step; idle %.1f us per step in %d gaps per step; gaps >= %.0f us: %.1f us per step' % (
    steps, span, tot, len(gaps) // steps, min_gap, sum(g[0] for g in big) / steps))
agg = {}
for g, a, b in big:
    k = (a, b)
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += g
for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%8.1f us per step  %5.1f x per step   after %-44s before %s' % (t / steps, n / steps, a, b))
if len(sys.argv) > 3:       # the biggest gaps of ONE step, with their neighbourhood
    k = int(sys.argv[3])
    a, b = starts[k], starts[k + 1]
    t0 = int(rows[a]['Start_Timestamp'])
    end, out = t0, []
    for i in range(a, b):
        s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
        if s - end > min_gap * 1e3:
            out.append((s - end, i))
        end = max(end, e)
    for g, i in sorted(out, reverse=True)[:14]:
        print('gap %8.1f us at +%9.1f us:' % (g / 1e3, (int(rows[i]['Start_Timestamp']) - t0) / 1e3))
        for j in range(max(a, i - 3), min(b, i + 3)):
            r = rows[j]
            print('     %s +%9.1f us %8.1f us  q%s grid %s  %s' % ('>>' if j == i else '  ', (int(r['Start_Timestamp']) - t0) / 1e3,
                  (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r.get('Grid_Size', '?'), short(r)))
