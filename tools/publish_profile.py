"""Copy the results of one tools/profile.sh run (gpurun_out/<tag>/) into profiles/:
    python tools/publish_profile.py <tag> <round> [suffix]
(suffix: appended to the file names, for a run at another photon count than the scene's own)
writes profiles/<round>_<workload>_kernel_stats.csv, _bench.json, _pmc.csv and refreshes the workload's entry
of profiles/counters.json, the per-pass counter record bench.py's roofline reads."""
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd, suffix=""):
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    bench = json.load(open(os.path.join(src, "bench.json")))
    workload, sink = bench["config"]["workload"], bench["config"]["sink"]
    photons = bench["config"]["photons_per_light"] * bench["config"]["lights"]
    # a scene profiled with the sink that is not its own keeps both sets of files
    own = bench["config"].get("scene_sink", sink)
    name = (workload if sink == own else f"{workload}_{sink}") + suffix
    stem = os.path.join(dst, f"{rnd}_{name}")
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, stem + "_kernel_stats.csv")
    shutil.copy(os.path.join(src, "stats_bench.json"), stem + "_kernel_stats_bench.json")
    shutil.copy(os.path.join(src, "bench.json"), stem + "_bench.json")
    lines = open(os.path.join(src, "summary.txt")).read().splitlines()
    start = next(i for i, line in enumerate(lines) if line.startswith("kernel,counter,value"))
    with open(stem + "_pmc.csv", "w") as f:
        f.write("# rocprofv3 --pmc passes (one counter group per run, tools/profile.sh) of: python3 bench.py --scene "
                f"scenes/{workload}.json --sink {sink} --photons {bench['config']['photons_per_light']} --steps 1 --warmup 0 "
                "--no-cpu-baseline\n")
        f.write(f"# one step = one photon pass of {photons} photons ({sink} sink) and its gather; counters are summed over "
                "a kernel's launches in the step; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them, uncorrected\n")
        f.write("\n".join(lines[start:]) + "\n")
    counters, launches = {}, 1
    for line in lines[start + 1:]:
        kernel, counter, value, count = line.split(",")
        if kernel == "traceKernel":
            counters[counter] = float(value)
            launches = int(count)
    fetch, write = counters["FETCH_SIZE"] * 1024.0, counters["WRITE_SIZE"] * 1024.0
    hits, misses = counters.get("TCC_HIT_sum", 0.0), counters.get("TCC_MISS_sum", 0.0)
    entry = {
        "workload": workload, "sink": sink, "photons_per_launch": photons, "trace_launches_per_pass": launches,
        "source": f"profiles/{rnd}_{name}_pmc.csv",
        "SQ_ACTIVE_INST_VALU": counters["SQ_ACTIVE_INST_VALU"], "SQ_INSTS_VALU": counters["SQ_INSTS_VALU"],
        "SQ_INSTS_SALU": counters.get("SQ_INSTS_SALU"), "SQ_WAVE_CYCLES": counters.get("SQ_WAVE_CYCLES"),
        "SQ_BUSY_CYCLES": counters.get("SQ_BUSY_CYCLES"),
        "hbm_bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
        "l2_hit_rate": hits / (hits + misses) if hits + misses > 0 else None,
        # a TCC request moves one 128-byte line at most
        "l2_request_bytes": 128.0 * counters["TCC_REQ_sum"] if "TCC_REQ_sum" in counters else None,
        "l1_accesses": counters.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "lanes_active": counters["SQ_THREAD_CYCLES_VALU"] / (64.0 * counters["SQ_ACTIVE_INST_VALU"])
        if counters.get("SQ_THREAD_CYCLES_VALU") else None,
        "wait_memory_frac": counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"]
        if counters.get("SQ_WAIT_ANY") and counters.get("SQ_WAVE_CYCLES") else None,
        "wait_issue_frac": counters["SQ_WAIT_INST_ANY"] / counters["SQ_WAVE_CYCLES"]
        if counters.get("SQ_WAIT_INST_ANY") and counters.get("SQ_WAVE_CYCLES") else None,
    }
    path = os.path.join(dst, "counters.json")
    record = {"note": "", "launches": []}
    if os.path.exists(path):
        record = json.load(open(path))
    record["note"] = ("Counter totals of the traceKernel launches of ONE photon pass, from rocprofv3 PMC passes (one counter "
                      "group per run, tools/profile.sh); SQ_ACTIVE_INST_VALU and the other SQ cycle counters tick once per "
                      "four cycles; FETCH_SIZE / WRITE_SIZE converted from KiB to bytes, uncorrected (the factor 2 the "
                      "hardware notes give for FETCH_SIZE holds for 16-byte-per-lane streaming reads, which this kernel "
                      "does not issue).")
    key = (entry["workload"], entry["sink"], entry["photons_per_launch"])
    record["launches"] = [e for e in record["launches"]
                          if (e["workload"], e["sink"], e["photons_per_launch"]) != key] + [entry]
    json.dump(record, open(path, "w"), indent=1)
    print("published", tag, "->", dst, json.dumps(entry))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02", sys.argv[3] if len(sys.argv) > 3 else "")
