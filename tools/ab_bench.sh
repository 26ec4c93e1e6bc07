# A/B of library builds on a few scenes (GPU box, repository root):  bash tools/ab_bench.sh "<lib ...>" "<scene ...>"
for lib in $1; do for s in $2; do
  RT_AMD_LIBRARY=$PWD/ray-tracer_amd/csrc/$lib timeout -k 10 200 python bench.py --scene scenes/$s.json --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib $s failed"; tail -3 gpurun_out/ab.err; }
  python - "$lib" "$s" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json")); st = d["stages_ms"]
print(f"{sys.argv[1]:22s} {sys.argv[2]:30s} step {d['ms_per_step']:7.2f} sort {st['sort']:5.2f} trace {st['trace']:6.2f} gather {st['gather']:6.2f}")
PY
done; done
