#!/usr/bin/env python3
"""Per-dispatch view of ONE serialised step from a rocprofv3 --kernel-trace CSV (analysis tool, any machine).
    python tools/trace_steps.py kernel_trace.csv [--list] [--cmp other.csv]
The last complete step is the dispatches between the last two `argmax_hw_kernel` launches (the hourglass branch ends the
serialised pass).  Prints per-kernel-family totals (us, launches), the sum of kernel time and of the gaps between
dispatches; --list prints every dispatch (duration, gap to the previous one, grid); --cmp puts a second trace beside it."""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    m = re.search(r"fusg::(\w+)(<[^>]*>)?", name)
    if not m:
        return name[:40]
    return m.group(1) + (m.group(2) or "")


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "argmax_hw_kernel" in r[2]]
    if len(marks) < 2:
        raise SystemExit("need at least two steps in the trace")
    return rows[marks[-2] + 1: marks[-1] + 1]


def summarise(step):
    fam = OrderedDict()
    gaps = 0
    for i, (s, e, n, g) in enumerate(step):
        k = short(n)
        a = fam.setdefault(k, [0, 0])
        a[0] += e - s
        a[1] += 1
        if i:
            gaps += max(0, s - step[i - 1][1])
    return fam, gaps


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    step = load(args[0])
    fam, gaps = summarise(step)
    other = None
    if "--cmp" in sys.argv:
        other = summarise(load(sys.argv[sys.argv.index("--cmp") + 1]))[0]
    tot = sum(v[0] for v in fam.values())
    conv = sum(v[0] for k, v in fam.items() if k.startswith(("conv_", "hg_bneck")))
    print(f"step: {len(step)} dispatches, kernel time {tot / 1e3:.1f} us (conv families {conv / 1e3:.1f} us), gaps {gaps / 1e3:.1f} us, "
          f"span {(step[-1][1] - step[0][0]) / 1e3:.1f} us")
    for k, (ns, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        line = f"{k:52s} {n:4d} {ns / 1e3:10.1f} us {ns / n / 1e3:8.1f} avg"
        if other is not None and k in other:
            line += f"   | {other[k][1]:4d} {other[k][0] / 1e3:10.1f} us  ({(ns - other[k][0]) / 1e3:+.1f})"
        print(line)
    if "--list" in sys.argv:
        for i, (s, e, n, g) in enumerate(step):
            gap = s - step[i - 1][1] if i else 0
            print(f"{i:4d} {short(n):52s} wgs {g:7d} {(e - s) / 1e3:9.1f} us  gap {gap / 1e3:7.1f}")


if __name__ == "__main__":
    main()
