# rocprofv3 kernel statistics of one full-size CLI frame (run on the GPU box from the repository root):
#   bash tools/profile_cli.sh MirrorKnot4k
set -e
S=${1:-MirrorKnot4k}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/cli_$S /tmp/rt_renders
python3 - "$S" "$R" <<'PY'
import json, sys
name, root = sys.argv[1], sys.argv[2]
scene = json.load(open(f"{root}/scenes/{name}.json"))
scene["$renderConfiguration"]["$renderPath"] = "/tmp/rt_renders"
json.dump(scene, open(f"/tmp/rt_renders/{name}.json", "w"))
PY
cp -r $R/scenes/meshes /tmp/rt_renders/ 2>/dev/null || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cli_$S -- $R/ray-tracer_amd/host/ray-tracer /tmp/rt_renders/$S.json > $R/gpurun_out/cli_$S/run.log 2>&1
python3 - $R/gpurun_out/cli_$S <<'PY'
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for row in list(csv.reader(open(path)))[:14]:
    print(row[0][:90], row[1], row[2], row[4])
PY
