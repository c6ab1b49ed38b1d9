#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (FETCH_SIZE, then WRITE_SIZE - they do not fit one pass on
gfx950) into the per-launch HBM traffic of the conv kernels, with the guide's gfx950 correction
(MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of wide streaming reads -> x2; both
counters are in KiB).  Writes profiles/hbm_traffic_latest.json, which bench.py reports as
roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/f --output-format csv -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/w --output-format csv -- python bench.py ...
    python tools/hbm_traffic.py out/f out/w [key]

`key` names the leg the passes were taken on (default "f16x3_256"; e.g. "bf16_256", "bf16_512"): the result is stored
under that key of profiles/hbm_traffic_latest.json (the key-less top level keeps mirroring the f16x3_256 leg for older
readers).
"""
import collections
import csv
import glob
import json
import os
import sys


CONV_KERNELS = ("conv_igemm", "conv_halo", "conv_tapunit", "conv_splitk_reduce", "hg_bneck", "conv_small", "conv_pointwise")     # every dispatch a fusg_conv2d call makes
# one fusg_conv2d call = one launch, except split-K launches, which add their reduce dispatch: calls = main dispatches
# (round 4: + the small-image kernel and the streaming pointwise kernel, which the round-3 list missed)
MAIN_KERNELS = ("conv_igemm", "conv_halo", "conv_tapunit", "hg_bneck", "conv_small", "conv_pointwise")


def load(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        k = "conv" if any(t in name for t in CONV_KERNELS) else "other"
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
        if any(t in name for t in MAIN_KERNELS):
            n["conv_calls"].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    ft, fn = load(fd, "FETCH_SIZE")
    wt, wn = load(wd, "WRITE_SIZE")
    fetch_b = ft["conv"] * 1024 * 2          # KiB -> B, x2 gfx950 wide-read correction
    write_b = wt["conv"] * 1024
    launches = fn["conv_calls"]
    import datetime
    out = {"conv_launches": launches, "conv_dispatches": fn["conv"], "conv_fetch_bytes_total": fetch_b, "conv_write_bytes_total": write_b,
           "conv_bytes_per_launch": (fetch_b + write_b) / max(1, launches),
           "definition": "HBM bytes of every dispatch a conv call (fusg_conv2d / fusg_hg_bottleneck) makes (generic / halo / tap-unit / "
                         "fused-bottleneck kernels + split-K reduce), per conv call - the same unit as bench.py's roofline.achieved",
           "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 wide streaming reads), WRITE_SIZE KiB x1024",
           "collected": datetime.date.today().isoformat() + ", rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two separate passes of "
                        "bench.py --precision f16x3 (serialised branches)",
           "source": [fd, wd]}
    key = sys.argv[3] if len(sys.argv) > 3 else "f16x3_256"
    out["leg"] = key
    out["collected"] = out["collected"].replace("--precision f16x3", "on the leg '%s'" % key)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(repo, "profiles", "hbm_traffic_latest.json")
    try:
        allk = json.load(open(path))
    except (OSError, ValueError):
        allk = {}
    legs = allk.get("legs", {})
    legs[key] = out
    top = dict(legs.get("f16x3_256", out))
    top["legs"] = legs
    with open(path, "w") as f:
        json.dump(top, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
