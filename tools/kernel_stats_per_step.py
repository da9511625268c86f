"""Per-STEP kernel times from a rocprofv3 --kernel-trace --stats CSV of bench.py: every kernel's total time divided by the number of
training steps the run held - counted, not assumed: the calls of idr_loss_kernel (one per training step; VERDICT r5 next #7: round 5's
documents divided a 35-step run by 41).  Prints the divisor, the evaluators' and the non-evaluator kernels' ms per step.

    python tools/kernel_stats_per_step.py <kernel_stats.csv> [out.txt]"""
import csv
import sys


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    name_k = 'Name' if 'Name' in rows[0] else 'KernelName'
    tot = lambda r: float(r.get('TotalDurationNs') or r.get('Total Duration (ns)') or 0.0) * 1e-6
    calls = lambda r: int(float(r.get('Calls') or 0))
    steps = sum(calls(r) for r in rows if 'idr_loss_kernel' in r[name_k])
    if steps == 0:
        raise SystemExit('no idr_loss_kernel call in %s: not a training run' % path)
    evalk = ('eval_kernel16', 'eval_kernel<', 'eval_kernel(')
    lines = ['%s: %d training steps (= calls of idr_loss_kernel: the divisor of every per-step figure below)' % (path, steps)]
    ev = sum(tot(r) for r in rows if any(k in r[name_k] for k in evalk))
    rest = sum(tot(r) for r in rows) - ev
    lines.append('evaluators (eval_kernel*) %.2f ms per step, every other kernel %.2f ms per step, all kernels %.2f ms per step'
                 % (ev / steps, rest / steps, (ev + rest) / steps))
    for r in sorted(rows, key=tot, reverse=True)[:24]:
        lines.append('  %9.3f ms/step  %7.1f calls/step  %s' % (tot(r) / steps, calls(r) / steps, r[name_k][:110]))
    out = '\n'.join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + '\n')


if __name__ == '__main__':
    main()
