#!/usr/bin/env python3
"""Time line of the device during one file-runner run, from a rocprofv3 kernel trace (kernel_trace.csv): per bin of the
last run in the trace, the share of the time in which the decoder / any batch kernel / nothing was running, and the batches
finished (k_batch launches).  Diagnostic tool.

  python tools/exp/run_timeline.py kernel_trace.csv [bin_ms]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
bin_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)) for r in rows)
# a run's first decoder launch is the short first stretch (4 096 blocks at most, a wave each; the later ones hold 16 384 or
# more); the run begins with the block-table scan before it.  The last run in the trace is taken.
first = [i for i, e in enumerate(ev) if "inflate" in e[2] and e[3] <= 4096 * 64]
cut = 0
if first:
    cut = first[-1]
    while cut > 0 and ("bgzf" in ev[cut - 1][2] or "inflate" in ev[cut - 1][2] or ev[cut][0] - ev[cut - 1][1] < 2e6):
        cut -= 1
        if "k_bgzf_count" in ev[cut][2]:
            break
ev = [(s0, e0, n) for s0, e0, n, _ in ev[cut:]]
t0 = ev[0][0]
t1 = max(e[1] for e in ev)


def union(iv, a, b):
    tot, cur = 0, a
    for s, e in sorted(iv):
        s, e = max(s, cur), min(e, b)
        if e > s:
            tot += e - s
            cur = e
    return tot


print("run of %.1f ms, %d launches" % ((t1 - t0) / 1e6, len(ev)))
print("  from ms   decoder   batch kernels   any    batches done   decoder launches running")
a = t0
while a < t1:
    b = a + bin_ns
    inf = [(s, e) for s, e, n in ev if "inflate" in n and e > a and s < b]
    oth = [(s, e) for s, e, n in ev if "inflate" not in n and "bgzf" not in n and e > a and s < b]
    al = [(s, e) for s, e, n in ev if e > a and s < b]
    nb = sum(1 for s, e, n in ev if "k_batch" in n and a <= e < b)
    print("  %7.1f   %5.0f %%   %5.0f %%        %5.0f %%   %4d          %d" % ((a - t0) / 1e6, 100 * union(inf, a, b) / bin_ns, 100 * union(oth, a, b) / bin_ns,
                                                                       100 * union(al, a, b) / bin_ns, nb, len(inf)))
    a = b
