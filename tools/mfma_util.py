#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel family from one rocprofv3 PMC pass of bench.py (analysis tool):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d DIR --output-format csv -- python3 bench.py ...
    python tools/mfma_util.py DIR
pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), at whatever clock the card ran, summed over all
dispatches of a family."""
import collections
import csv
import glob
import os
import sys


def family(name):
    for k in ("conv_halo_h3", "conv_tapunit_h3", "conv_igemm_h3", "conv_igemm_f32", "conv_splitk_reduce", "hg_bneck_h3"):
        if k in name:
            return k
    return name.split("(")[0].split("::")[-1][:28]


def main():
    d = sys.argv[1]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            tot[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[fam].add(r["Dispatch_Id"])
    print("MFMA pipe utilisation of the crop pass from PMC counters (MI355X, branches serialised; leg = the bench.py command of the pass)")
    print(f"{'kernel family':28s} {'dispatches':>10s} {'CU-busy cycles':>16s} {'MFMA pipe busy':>15s} {'MFMA MOPS F16+BF16':>18s}")
    conv_busy = conv_cu = 0.0
    rows = []
    for fam, c in tot.items():
        cu, busy = c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if (fam.startswith("conv_") or fam == "hg_bneck_h3") and fam != "conv_splitk_reduce":
            conv_busy += busy
            conv_cu += cu
        rows.append((cu, fam, len(nd[fam]), busy / (4 * cu) if cu else 0.0,
                     c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)))
    print(f"{'all conv kernels':28s} {'':>10s} {conv_cu:16.4g} {conv_busy / (4 * conv_cu) if conv_cu else 0:15.3f}")
    for cu, fam, n, util, mops in sorted(rows, reverse=True)[:14]:
        print(f"{fam:28s} {n:10d} {cu:16.4g} {util:15.3f} {mops:15.4g}")


if __name__ == "__main__":
    main()
