# Bench lines of a few configurations with the library as it is built (run on the GPU box from the repository root):
#   bash tools/quick_bench.sh <tag> [scene ...]
TAG=${1:-quick}; shift
SCENES=${@:-MirrorBombardment-200M MirrorKnot4k CornellBoxGlassMirrorDiffuse CornellBox MultiCameraTest CBGMD_stochastic128}
mkdir -p gpurun_out/$TAG
for s in $SCENES; do
  timeout -k 10 200 python bench.py --scene scenes/$s.json --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/$s.json 2> gpurun_out/$TAG/$s.err || { echo "$s failed"; tail -3 gpurun_out/$TAG/$s.err; exit 1; }
  python - "$s" "gpurun_out/$TAG/$s.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
st = d["stages_ms"]
print(f"{sys.argv[1]:32s} step {d['ms_per_step']:7.2f} ms  sort {st['sort']:5.2f} trace {st['trace']:6.2f} gather {st['gather']:6.2f}  "
      f"{d['value']:.3e} b/s  pass {d['photon_pass_bounces_per_s']:.3e}  frac {d['roofline']['frac']:.3f}")
PY
done
