#!/bin/bash
# Round 4: launch time of the pair kernel on the metric's N = 8 tile (1024 x 512, untiled) as a function of rows per tile
# (iterations = rows + 6; tiles = 19 x ceil(512 / rows)): fits T = T0 + iterations x t_iter at each occupancy.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r04_rows_model.txt
: > $out
run() {   # label, env..., -- args
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print(round(d['value'] / 1e9, 2), 'G', 'launch_us', round(r['avg_launch_ms'] * 1e3, 2), 'ms/step', round(d['ms_per_step'], 3))"
}
echo "full2048 $(run)" >> $out
for rows in 3 4 5 6 7 8 10 12 14 16 20 26 32 48 64 128; do
  echo "rows=$rows untiled $(CSI_PAIR_ROWS=$rows run --tile 1024x512)" >> $out
done
for rows in 7 10 14 20; do
  echo "rows=$rows peer $(CSI_PAIR_ROWS=$rows run --tile 1024x512 --force-connected --no-compare)" >> $out
done
cat $out
