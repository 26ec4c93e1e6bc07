# Full-size frames of the configuration scenes through the CLI (run on the GPU box from the repository root).
mkdir -p gpurun_out/renders
for S in ${@:-CornellBox CornellBoxGlassMirrorDiffuse MirrorKnot4k MultiCameraTest CBGMD_stochastic128 MirrorBombardment-200M}; do
  python3 - "$S" <<'PY'
import json, sys
name = sys.argv[1]
scene = json.load(open(f"scenes/{name}.json"))
scene["$renderConfiguration"]["$renderPath"] = "/tmp/rt_renders"  # 4096 x 4096 x 16-bit frames: not for gpurun_out
json.dump(scene, open(f"gpurun_out/renders/{name}.json", "w"))
PY
  START=$(date +%s.%N)
  timeout -k 10 280 ray-tracer_amd/host/ray-tracer gpurun_out/renders/$S.json > gpurun_out/renders/$S.log 2>&1 || { tail -5 gpurun_out/renders/$S.log; exit 1; }
  echo "$S wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),2))" $START) s"
  grep -E "Photon pass|Probe gather: kept|Render time|total:" gpurun_out/renders/$S.log | head -6
  tail -1 gpurun_out/renders/$S.log
done
ls -la /tmp/rt_renders/*.png
