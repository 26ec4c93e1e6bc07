set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for S in ${SCENES:-MirrorBombardment-200M CornellBox MirrorKnot4k CornellBoxGlassMirrorDiffuse}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g_$S -- python3 $R/tools/stage_times.py --repeat 2 $R/scenes/$S.json > $R/gpurun_out/prof_g_$S.log 2>&1
  F=$(ls $R/gpurun_out/prof_g_$S/*/*kernel_stats.csv | head -1)
  echo "== $S"; tail -2 $R/gpurun_out/prof_g_$S.log; python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("TOP", "14"))]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
done
