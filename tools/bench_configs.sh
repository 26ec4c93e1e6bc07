# One bench line per configuration scene (run on the GPU box from the repository root);
# the lines are appended to gpurun_out/bench_configs.jsonl.
mkdir -p gpurun_out && : > gpurun_out/bench_configs.jsonl
for S in MirrorBombardment-200M MirrorKnot4k CornellBoxGlassMirrorDiffuse CornellBox MultiCameraTest CBGMD_stochastic128; do
  timeout -k 10 240 python bench.py --no-cpu-baseline --steps 3 --scene scenes/$S.json "$@" >> gpurun_out/bench_configs.jsonl || exit 1
  tail -1 gpurun_out/bench_configs.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['sink'], 'ms/step', round(d['ms_per_step'],3), 'trace kernel ms', round(d['roofline']['kernel_ms'],3), 'bounces/s %.3e' % d['value'])"
done
