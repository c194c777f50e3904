#!/bin/bash
# Run on the GPU box: bench.py (2048^2 and a 1024x512 tile) with several builds of the library.
# usage: scripts/ab_libs.sh <tag> <lib> [<lib> ...]   (lib = default | suffix of climaseaice.jl_amd/libcsi_hip_<suffix>.so)
TAG=$1; shift
R=$GRAFT_REPO_ROOT/climaseaice.jl_amd
for lib in "$@"; do
  if [ $lib = default ]; then unset CSI_HIP_LIBRARY; else export CSI_HIP_LIBRARY=$R/libcsi_hip_$lib.so; fi
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > gpurun_out/${TAG}_bench_$lib.log 2>&1
  timeout 120 python bench.py --steps 10 --warmup 3 --tile 1024x512 --no-cpu-baseline --no-full-step > gpurun_out/${TAG}_tile_$lib.log 2>&1
  CSI_PAIR_TILES=1024 timeout 120 python bench.py --steps 10 --warmup 3 --tile 1024x512 --no-cpu-baseline --no-full-step > gpurun_out/${TAG}_tile_t1024_$lib.log 2>&1
done
