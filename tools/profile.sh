# Measurement recipe behind profiles/ (run on the GPU box from the repository root):
#   bash tools/profile.sh <tag> [scene name] [sink] [photons per light] [ranks whose store share the run keeps]
# the bench line of the scene (default MirrorBombardment-200M), the rocprofv3 kernel statistics of the
# same command, and the PMC passes (one counter group per run: the hardware holds 8 SQ, 4 TCC and
# 4 TCP counters at a time; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  Results land in
# gpurun_out/<tag>/; tools/publish_profile.py copies what is kept into profiles/.
set -e
TAG=${1:-profile}
SCENE=${2:-MirrorBombardment-200M}
SINK=${3:-scene}
PHOTONS=${4:-0}   # 0: the scene's own budget; otherwise e.g. one rank's share of a sharded frame
SHARE=${5:-0}     # with a share of the photons: the number of ranks, so that the store is held to one rank's part
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
BASELINE=--no-cpu-baseline
if [ "$SCENE" = "MirrorBombardment-200M" ] && [ "$PHOTONS" = "0" ]; then BASELINE=; fi
timeout -k 10 500 python3 $R/bench.py --scene $R/scenes/$SCENE.json --sink $SINK --photons $PHOTONS --store-share $SHARE $BASELINE > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --scene $R/scenes/$SCENE.json --sink $SINK --photons $PHOTONS --store-share $SHARE --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
for GROUP in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $GROUP | tr ' ' '+')
  timeout -s KILL 150 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pmc_$NAME -- python3 $R/bench.py --scene $R/scenes/$SCENE.json --sink $SINK --photons $PHOTONS --store-share $SHARE --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$NAME.json 2> $OUT/pmc_$NAME.err
done
cd $R && python3 tools/summarize_profile.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
