"""Condense the rocprofv3 outputs of tools/profile.sh: kernel statistics and, per kernel of interest, the
sum of every PMC counter over that kernel's launches in ONE step of the bench (one photon pass: a pass
traced bounce level by bounce level launches the trace kernel once per level) together with the number of
launches."""
import csv
import glob
import os
import sys

KERNELS = {"traceKernel": "traceKernel", "gatherGroupKernel": "gatherKernel", "gatherKernel": "gatherKernel",
           "emissionKeysKernel": "emissionKeysKernel", "digitCountKernel": "digitCountKernel",
           "digitScatterKernel": "digitScatterKernel", "refineOrderKernel": "refineOrderKernel"}


def short_name(name):
    for key, short in KERNELS.items():
        if key in name:
            return short
    return "radixSort" if "rocprim" in name and "radix" in name.lower() else None


def main(out):
    for path in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("# kernel statistics:", os.path.relpath(path, out))
        for row in list(csv.reader(open(path)))[:10]:
            print(",".join(row))
    print("# PMC passes (bench.py --steps 1 --warmup 0: one step); counter totals per kernel over the step's launches")
    print("kernel,counter,value,launches")
    for path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        totals, launches = {}, {}
        for row in csv.DictReader(open(path)):
            short = short_name(row["Kernel_Name"])
            if short is None:
                continue
            key = (short, row["Counter_Name"])
            totals[key] = totals.get(key, 0.0) + float(row["Counter_Value"])
            launches.setdefault(key, set()).add(row["Dispatch_Id"])
        for (kernel, counter), value in sorted(totals.items()):
            print(f"{kernel},{counter},{value},{len(launches[(kernel, counter)])}")


if __name__ == "__main__":
    main(sys.argv[1])
