#!/usr/bin/env python3
"""GPU box: per-queue busy time and overlap of the multi-stream crop pass from a rocprofv3 kernel trace (analysis tool).
    rocprofv3 --kernel-trace -d DIR --output-format csv -- python3 bench.py --steps 6 --warmup 3 --precision f16x3 --no-prof ...
    python tools/timeline.py DIR"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# window = the last 4 whole steps (one argmax_hw launch per step marks a step)
marks = [s for s, e, q, n in rows if "argmax_hw" in n]
t_lo, t_hi = marks[-5], marks[-1]
rows = [r for r in rows if t_lo <= r[0] < t_hi]
span = t_hi - t_lo
print(f"4 steps: {span / 4e6:.2f} ms per step")
perq = collections.defaultdict(int)
for s, e, q, n in rows:
    perq[q] += e - s
# union of busy intervals
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, q, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {span / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), idle {100 - 100 * busy / span:.1f} %")
for q, t in sorted(perq.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: kernels busy {t / 1e6:.2f} ms ({100 * t / span:.1f} % of the window)")
# concurrency histogram
ev = sorted([(s, 1) for s, e, q, n in rows] + [(e, -1) for s, e, q, n in rows])
lvl, last, hist = 0, ev[0][0], collections.defaultdict(int)
for t, dlt in ev:
    hist[lvl] += t - last
    last = t
    lvl += dlt
print("time with k kernels in flight:", {k: f"{100 * v / span:.1f}%" for k, v in sorted(hist.items())})
