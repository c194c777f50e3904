#!/bin/bash
# Round 4: first contact of the MARCH mode (all pairs of a sub-cycle in one launch on the peer transport)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "peer_halo_transport" 2>&1 | tail -15 > gpurun_out/r04_march_tests.txt
timeout 900 python -m pytest tests/test_gpu_local_tiles.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/r04_march_tests.txt
cat gpurun_out/r04_march_tests.txt
out=gpurun_out/r04_march_bench.txt
: > $out
run() {
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print(round(d['value'] / 1e9, 2), 'G', 'launch_us', round(r['avg_launch_ms'] * 1e3, 2), 'ms/step', round(d['ms_per_step'], 3), d['path'])"
}
for rep in 1 2; do
echo "tile untiled           $(run --tile 1024x512)" >> $out
echo "tile peer march=0      $(CSI_MARCH=0 run --tile 1024x512 --force-connected --no-compare)" >> $out
echo "tile peer march=1      $(CSI_MARCH=1 run --tile 1024x512 --force-connected --no-compare)" >> $out
echo "tile peer march pairs=2 $(CSI_MARCH=1 CSI_MARCH_PAIRS=2 run --tile 1024x512 --force-connected --no-compare)" >> $out
echo "tile peer march pairs=10 $(CSI_MARCH=1 CSI_MARCH_PAIRS=10 run --tile 1024x512 --force-connected --no-compare)" >> $out
done
echo "t1024 peer march=0     $(CSI_MARCH=0 run --tile 1024x1024 --force-connected --no-compare)" >> $out
echo "t1024 peer march=1     $(CSI_MARCH=1 run --tile 1024x1024 --force-connected --no-compare)" >> $out
echo "t2048x1024 peer march=0 $(CSI_MARCH=0 run --tile 2048x1024 --force-connected --no-compare)" >> $out
echo "t2048x1024 peer march=1 $(CSI_MARCH=1 run --tile 2048x1024 --force-connected --no-compare)" >> $out
cat $out
