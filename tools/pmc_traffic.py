"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of bench.py into the per-launch HBM
traffic of the dominant kernel, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half the
bytes of wide coalesced reads; both counters are in KB).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> out.json
"""
import csv
import json
import sys


def per_launch(path, counter, kern):
    """bytes summed over every dispatch of the kernel; launches = tracer rounds.  The pipelined kernel is dispatched twice
    per round (64-query and 32-query instance, one of them exits at once): the `<8, 1>` (16p) / `<2, FT>` (16q) dispatches add their bytes but
    are not counted as launches, like the HIP events of bench.py bracket both dispatches of a round."""
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value'])
            name = r['Kernel_Name']
            # one launch per tracer round: the round's big-tile split-precision dispatch stands for it (the 32-query
            # instance `<2, FT>` and the coarse evaluator eval_kernel16s are further dispatches of the same round)
            if ', 1>(' not in name and '<2>(' not in name and 'eval_kernel16q<2, ' not in name and 'eval_kernel16s' not in name:
                disp.add(r['Dispatch_Id'])
    return tot, len(disp)


fetch_csv, write_csv, kern, out = sys.argv[1:5]
f, nf = per_launch(fetch_csv, 'FETCH_SIZE', kern)
w, nw = per_launch(write_csv, 'WRITE_SIZE', kern)
res = {'kernel': kern, 'launches_fetch_pass': nf, 'launches_write_pass': nw,
       'fetch_kb_raw_per_launch': f / max(nf, 1), 'write_kb_per_launch': w / max(nw, 1),
       'hbm_bytes_per_launch': (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
       'note': 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as reported; averages over ALL '
               'launches of the kernel in the run, empty rounds included, like roofline.achieved'}
json.dump(res, open(out, 'w'), indent=1)
print(res)
