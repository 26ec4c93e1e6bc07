"""Copy the results of one tools/profile.sh run (gpurun_out/<tag>/) into profiles/:
    python tools/publish_profile.py r01f r01
writes profiles/<round>_kernel_stats.csv, _kernel_stats_bench.json, _bench_default.json,
_pmc_summary.csv and refreshes profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes."""
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd):
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(dst, f"{rnd}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "stats_bench.json"), os.path.join(dst, f"{rnd}_kernel_stats_bench.json"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{rnd}_bench_default.json"))
    lines = open(os.path.join(src, "summary.txt")).read().splitlines()
    start = next(i for i, line in enumerate(lines) if line.startswith("kernel,counter,value"))
    bench = json.load(open(os.path.join(src, "bench.json")))
    photons = bench["config"]["photons_per_light"] * bench["config"]["lights"]
    with open(os.path.join(dst, f"{rnd}_pmc_summary.csv"), "w") as f:
        f.write("# rocprofv3 --pmc passes (one counter group per run, tools/profile.sh) of: python3 bench.py --steps 1 "
                "--warmup 0 --no-cpu-baseline\n")
        f.write(f"# one launch of each kernel = {photons // 1000000}M photons, {bench['config']['sink']} sink; "
                "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them\n")
        f.write("\n".join(lines[start:]) + "\n")
    counters = {}
    for line in lines[start + 1:]:
        kernel, counter, value = line.split(",")
        if kernel == "traceKernel":
            counters[counter] = float(value)
    path = os.path.join(dst, "traffic.json")
    traffic = json.load(open(path))
    fetch, write = counters["FETCH_SIZE"] * 1024.0, counters["WRITE_SIZE"] * 1024.0
    entry = {"workload": bench["config"]["workload"], "sink": bench["config"]["sink"], "photons_per_launch": photons,
             "hbm_bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write}
    traffic["launches"] = [e for e in traffic["launches"]
                           if (e["workload"], e["sink"], e["photons_per_launch"]) !=
                           (entry["workload"], entry["sink"], entry["photons_per_launch"])] + [entry]
    traffic["note"] = traffic["note"].rsplit("Source:", 1)[0] + f"Source: profiles/{rnd}_pmc_summary.csv."
    json.dump(traffic, open(path, "w"), indent=1)
    print("published", tag, "->", dst, entry)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r01")
