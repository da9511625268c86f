"""The longest whole-GPU idle intervals of a rocprofv3 kernel trace with the kernels around them.  usage: big_gaps.py trace.csv [n=12]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
short = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0][-40:]
t00 = int(rows[0]['Start_Timestamp'])
end, gaps = int(rows[0]['End_Timestamp']), []
for i, r in enumerate(rows[1:], 1):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end:
        gaps.append((s - end, i))
    end = max(end, e)
print('trace: %d kernels over %.1f ms; idle in total %.1f ms' % (len(rows), (end - t00) / 1e6, sum(g for g, _ in gaps) / 1e6))
for g, i in sorted(gaps, reverse=True)[:n]:
    print('--- %.2f ms idle at t = %.1f ms' % (g / 1e6, (int(rows[i]['Start_Timestamp']) - t00) / 1e6))
    for k in range(max(0, i - 4), min(len(rows), i + 4)):
        r = rows[k]
        print('   %s %10.1f ms  %8.1f us  q%-3s %s' % ('>' if k == i else ' ', (int(r['Start_Timestamp']) - t00) / 1e6,
              (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), short(r)))
