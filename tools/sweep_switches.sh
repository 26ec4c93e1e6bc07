# Sweeps of the library's environment switches on the bench line of a scene (GPU box, repository root):
#   bash tools/sweep_switches.sh <scene> "<VAR=value ...>" ...     one bench run per quoted setting
S=$1; shift
for SETTING in "" "$@"; do
  env $SETTING timeout -k 10 200 python bench.py --scene scenes/$S.json --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$S [$SETTING] failed"; tail -2 gpurun_out/sweep.err; continue; }
  python - "$S" "$SETTING" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep.json")); st = d["stages_ms"]
print(f"{sys.argv[1]:28s} [{sys.argv[2]:34s}] step {d['ms_per_step']:7.3f} sort {st['sort']:5.2f} trace {st['trace']:7.3f} gather {st['gather']:6.3f}")
PY
done
