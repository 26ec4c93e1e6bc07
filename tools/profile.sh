# Measurement recipe of the default bench (run on the GPU box from the repository root):
# the bench line, the rocprofv3 kernel statistics of the same command, and the PMC passes
# (one counter group per run) behind roofline.traffic.  Results land in gpurun_out/<tag>/.
set -e
TAG=${1:-profile}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
for GROUP in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  NAME=$(echo $GROUP | tr ' ' '+')
  timeout -s KILL 150 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pmc_$NAME -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$NAME.err
done
cd $R && python3 tools/summarize_profile.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
