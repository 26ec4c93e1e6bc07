# Trace-kernel time of the benchmark scenes (run on the GPU box from the repository root).
for S in scenes/MirrorBombardment-200M.json scenes/MirrorKnot4k.json scenes/CornellBoxGlassMirrorDiffuse.json scenes/CornellBox.json; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --scene $S "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], 'ms/step', round(d['ms_per_step'],3), 'trace kernel ms', round(d['roofline']['kernel_ms'],3), 'bounces/s %.3e' % d['value'])" || exit 1
done
