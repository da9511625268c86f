"""Probe which torch thread count gives the best CPU-oracle step time on this host (bench.py cpu_baseline)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for t in (int(a) for a in sys.argv[1:]):
    # bench.cpu_baseline pins its own thread count; probe by overriding os.cpu_count
    os.cpu_count = lambda t=t: t
    r, _ = bench.cpu_baseline('cfg2', 512, steps=2, warmup=1)
    print(t, r['value'], r['cores'], flush=True)
