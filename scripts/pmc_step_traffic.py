#!/usr/bin/env python
"""HBM traffic of ONE training step from the FETCH_SIZE / WRITE_SIZE passes of scripts/prof_step_pmc.sh: the
dispatches between the last two loader gathers (`pool_gather_kernel`) of each pass, counters in KiB.

    python scripts/pmc_step_traffic.py gpurun_out/pmc_step
"""
import csv
import glob
import os
import sys


def step_sum(path, counter):
    rows = [r for r in csv.DictReader(open(path, newline="")) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gath = [i for i, r in enumerate(rows) if "pool_gather_kernel" in r["Kernel_Name"]]
    win = rows[gath[-2]: gath[-1]]
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
    return sum(float(r["Counter_Value"]) for r in win) * 1024.0, len(win), dur


def main():
    root = sys.argv[1]
    total = 0.0
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = glob.glob(os.path.join(root, name, "*", "*counter_collection.csv"))[0]
        b, n, dur = step_sum(path, counter)
        total += b
        print("%-11s %7.2f GB over %d dispatches (kernels one at a time: %.2f ms)" % (counter, b / 1e9, n, dur / 1e6))
    print("per step    %7.2f GB (10^9 bytes, as in profiles/r2_pmc_training_step_final.md) = %.2f GiB" % (total / 1e9, total / 2 ** 30))


if __name__ == "__main__":
    main()
