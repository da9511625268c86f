"""Does Python's garbage collector stall config 3's steps?  (tools/idle_gaps.py shows the whole GPU idle for ~49 ms about
every fifth step, inside a backward pass.)  gc.callbacks clock every collection during run_workload; A/B with gc.freeze()
+ a raised threshold."""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nefii_amd import _lib

lib = _lib.lib()
dev = torch.device('cuda', 0)
args = argparse.Namespace(repeats=3, scaling='weak')
stat = {'n': [0, 0, 0], 't': [0.0, 0.0, 0.0], 't0': 0.0, 'big': []}

def cb(phase, info):
    if phase == 'start':
        stat['t0'] = time.perf_counter()
    else:
        g = info['generation']; dt = time.perf_counter() - stat['t0']
        stat['n'][g] += 1; stat['t'][g] += dt
        if dt > 5e-3: stat['big'].append((g, round(dt * 1e3, 1), info['collected']))
gc.callbacks.append(cb)

def run(tag, w='cfg3', steps=10, warmup=3):
    stat['n'] = [0, 0, 0]; stat['t'] = [0.0, 0.0, 0.0]; stat['big'] = []
    r = bench.run_workload(w, args, steps, warmup, 0, 1, dev, 'nccl', lib, side=False)
    print('%-28s %.2f ms  %s | gc runs %s  ms %s  long ones (gen, ms, collected) %s' % (
        tag, r['ms_per_step'], ['%.2f' % x for x in r['ms_per_step_repeats']], stat['n'],
        ['%.1f' % (x * 1e3) for x in stat['t']], stat['big'][:8]), flush=True)

run('default gc')
gc.collect(); gc.freeze()
run('after gc.freeze')
gc.disable()
run('gc disabled')
gc.enable()
run('default gc again')
