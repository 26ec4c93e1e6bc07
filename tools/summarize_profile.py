"""Condense the rocprofv3 outputs of tools/profile.sh: kernel statistics and, per kernel,
the sum of every PMC counter over the launches of the profiled command."""
import csv
import glob
import os
import sys


def main(out):
    for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("# kernel statistics:", os.path.relpath(path, out))
        for row in list(csv.reader(open(path)))[:8]:
            print(",".join(row))
    print("# PMC passes (bench.py --steps 1 --warmup 0: one launch of each pass kernel); counter totals per kernel")
    print("kernel,counter,value")
    for path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        totals = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            short = "traceKernel" if "traceKernel" in name else "emissionTileKernel" if "emissionTile" in name else None
            if short is None:
                continue
            key = (short, row["Counter_Name"])
            totals[key] = totals.get(key, 0.0) + float(row["Counter_Value"])
        for (kernel, counter), value in sorted(totals.items()):
            print(f"{kernel},{counter},{value}")


if __name__ == "__main__":
    main(sys.argv[1])
