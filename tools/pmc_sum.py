#!/usr/bin/env python3
"""Sum rocprofv3 PMC counters per kernel family over the counter_collection CSVs found under the given
directories (analysis tool):  python tools/pmc_sum.py DIR [DIR...] [--match conv_halo]"""
import collections
import csv
import glob
import os
import sys


def main():
    args = sys.argv[1:]
    match = "conv_halo"
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    tot = collections.defaultdict(float)
    nd = collections.defaultdict(set)
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if match not in r["Kernel_Name"]:
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                nd[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(tot):
        print(f"{k:32s} {tot[k]:18.0f}  over {len(nd[k])} dispatches  ({tot[k] / max(1, len(nd[k])):.4g} per dispatch)")


if __name__ == "__main__":
    main()
