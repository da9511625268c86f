"""Why is config 1 slower nested in the flag-less bench run (1.20 ms) than alone (0.92)?  A/B within one process, with the
garbage collector's own clock (gc.callbacks): collections per generation and their total time per cfg1 run."""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nefii_amd import _lib

lib = _lib.lib()
dev = torch.device('cuda', 0)
args = argparse.Namespace(repeats=3, scaling='weak')
stat = {'n': [0, 0, 0], 't': [0.0, 0.0, 0.0], 't0': 0.0}

def cb(phase, info):
    if phase == 'start':
        stat['t0'] = time.perf_counter()
    else:
        g = info['generation']; stat['n'][g] += 1; stat['t'][g] += time.perf_counter() - stat['t0']
gc.callbacks.append(cb)

def cfg1(tag):
    stat['n'] = [0, 0, 0]; stat['t'] = [0.0, 0.0, 0.0]
    r = bench.run_workload('cfg1', args, 30, 12, 0, 1, dev, 'nccl', lib, side=False)
    print('%-34s %.3f ms  %s | gc runs %s  ms %s (whole call: ~150 steps + set-up)' % (
        tag, r['ms_per_step'], ['%.3f' % x for x in r['ms_per_step_repeats']], stat['n'], ['%.1f' % (x * 1e3) for x in stat['t']]), flush=True)

cfg1('alone, first')
cfg1('alone, again')
cfg1('alone, third')
os.environ['NEFII_TRACE_STREAMS'] = '5'      # one more than the pool holds: a fresh stream joins
cfg1('with a fifth trace stream')
