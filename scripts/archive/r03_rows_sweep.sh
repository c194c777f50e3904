#!/bin/bash
# rows per tile of the pair kernel on the metric's N = 8 tile (1024 x 512), untiled and peer-connected to itself
cd $GRAFT_REPO_ROOT
for rows in default 7 8 9 10 11 12 13 14 16 18; do
  R=$rows; [ $rows = default ] && R=""
  a=$(CSI_PAIR_ROWS=$R python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile 1024x512 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(round(d['value'] / 1e9, 2))")
  b=$(CSI_PAIR_ROWS=$R python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --tile 1024x512 --force-connected --no-compare 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(round(d['value'] / 1e9, 2))")
  echo "rows=$rows untiled $a peer $b"
done
