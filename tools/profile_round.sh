#!/bin/bash
# Run on the GPU box (through gpurun): every rocprofv3 pass whose summary goes to profiles/<tag>_*.
#   stats   : --kernel-trace --stats of bench.py per solver (default streams)
#   counters: one --pmc pass per group per solver, serial launches (--streams 1), no trace domains
# usage: bash tools/profile_round.sh <tag>      (then, in the build container: python tools/prof_summarize.py <tag>)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in PSGCFS CFS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$MODE -- python3 $ROOT/bench.py --mode $MODE --steps 20 --warmup 5 --blocks 2 --no-cpu-baseline --no-other-mode > $OUT/prof_stats_$MODE.log 2>&1
  echo "stats $MODE done"
  i=0
  for GROUP in "FETCH_SIZE" "WRITE_SIZE" \
               "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
               "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
               "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $GROUP --output-format csv -d $OUT/prof_pmc${i}_$MODE -- python3 $ROOT/bench.py --mode $MODE --steps 5 --warmup 2 --blocks 1 --streams 1 --no-cpu-baseline --no-other-mode > $OUT/prof_pmc${i}_$MODE.log 2>&1
    echo "pmc group $i $MODE done"
  done
done
# RRT kernel and the config-5 reference map: kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_rrt -- python3 $ROOT/tests/tools/rrt_bench.py --steps 5 --rounds 1 > $OUT/prof_stats_rrt.log 2>&1
echo "stats rrt done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_mesh -- python3 $ROOT/bench.py --config 5 --map reference --steps 5 --warmup 1 --blocks 1 --streams 1 --no-cpu-baseline > $OUT/prof_stats_mesh.log 2>&1
echo "stats mesh done"
