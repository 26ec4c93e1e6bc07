# Atomic requests of the splat sink with and without the lanes of a wave adding together
# (rocprofv3 PMC pass of one bench step per setting; run on the GPU box from the repository root).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/splat_atomics
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in MultiCameraTest MirrorKnot4k; do
  for T in 1 0; do
    export RT_AMD_SPLAT_TOGETHER=$T
    timeout -s KILL 150 rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum --output-format csv -d $OUT/${S}_$T -- python3 $R/bench.py --scene $R/scenes/$S.json --sink splat --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${S}_$T.json 2> $OUT/${S}_$T.err
    timeout -k 10 150 python3 $R/bench.py --scene $R/scenes/$S.json --sink splat --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${S}_${T}_timed.json 2>> $OUT/${S}_$T.err
    python3 - $OUT/${S}_$T $S $T $OUT/${S}_${T}_timed.json <<'PY'
import csv, glob, json, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0])))
totals = {}
for r in rows:
    if "traceKernel" in r["Kernel_Name"]:
        totals[r["Counter_Name"]] = totals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
timed = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(sys.argv[2], "splat, lanes add together =", sys.argv[3], {k: int(v) for k, v in totals.items()},
      "splats", timed["counters"]["splat_total"], "trace kernel ms %.3f" % timed["roofline"]["kernel_ms"])
PY
  done
done
