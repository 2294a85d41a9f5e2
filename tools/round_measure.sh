#!/bin/bash
# Run on the GPU box (through gpurun): the bench lines and probes whose output is kept under profiles/<tag>_* .
# usage: bash tools/round_measure.sh <tag> [part ...]     parts: bench configs ranks fuzz (default: all)
set -u
TAG=${1:-r03}; shift || true
PARTS=${*:-bench configs ranks fuzz}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}m
mkdir -p $OUT
cd $ROOT
run() { n=$1; shift; timeout -k 10 600 "$@" > $OUT/$n.json 2> $OUT/$n.err; echo "$n rc=$?"; }
for P in $PARTS; do
case $P in
bench)
  run bench python3 bench.py
  run bench_driver_flags python3 bench.py --steps 20 --warmup 5
  run bench_batch16384_single_stream python3 bench.py --no-other-mode --no-cpu-baseline --batch 16384 --steps 12 --warmup 2 --blocks 3 --streams 1
  ;;
configs)
  run config4_jittered_routes python3 bench.py --config 4 --steps 20 --warmup 4 --blocks 3
  run config4_device_resampling python3 tests/tools/config4_bench.py --check 512
  run config4_grown_routes python3 tests/tools/rrt_bench.py --check 256
  run rrt_star_trees python3 tests/tools/rrt_bench.py --solver "RRT*" --rounds 1 --steps 2
  run mesh_config5_psgcfs python3 bench.py --config 5 --steps 20 --warmup 4 --blocks 3
  run mesh_config5_cfs python3 bench.py --config 5 --mode CFS --steps 20 --warmup 4 --blocks 3
  run mesh_config5_reference_map_psgcfs python3 bench.py --config 5 --map reference --steps 20 --warmup 4 --blocks 3
  run mesh_config5_reference_map_cfs python3 bench.py --config 5 --map reference --mode CFS --steps 20 --warmup 4 --blocks 3
  ;;
ranks)
  # two ranks sharing the one GPU, gloo collective on host copies: the multi-GPU code path end to end (the driver's 8-GPU run uses RCCL)
  export CFS_BENCH_BACKEND=gloo CFS_BENCH_DEVICE=0
  run ranks2_config3_weak python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline
  run ranks2_config4_strong python3 bench.py --gpus 2 --config 4 --scaling strong --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  run ranks2_config5_strong python3 bench.py --gpus 2 --config 5 --scaling strong --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  unset CFS_BENCH_BACKEND CFS_BENCH_DEVICE
  run ranks1_config4_strong python3 bench.py --config 4 --scaling strong --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  run ranks1_config5_strong python3 bench.py --config 5 --scaling strong --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  # what a 1/8 shard does on one GPU (strong scaling at 8 GPUs, per GPU): 512 config-4 routes, 32 config-5 seeds
  run shard8_config4 python3 bench.py --config 4 --batch 512 --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  run shard8_config5 python3 bench.py --config 5 --batch 32 --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline
  ;;
fuzz)
  for sd in 31 32 33; do timeout -k 10 500 python3 tests/tools/fuzz_shapes.py $sd 40 > $OUT/fuzz_$sd.txt 2>&1; echo "fuzz $sd rc=$?"; done
  timeout -k 10 500 python3 tests/tools/fuzz_shapes.py 34 24 mesh > $OUT/fuzz_34_mesh.txt 2>&1; echo "fuzz mesh rc=$?"
  ;;
esac
done
