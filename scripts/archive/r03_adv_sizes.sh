#!/bin/bash
# one launch per RK stage against separate launches (fusion 0), by grid size
cd $GRAFT_REPO_ROOT
for n in 256 512 768 1024 1536 2048; do
  for f in 0 2; do echo -n "N=$n fusion=$f: "; python3 examples/advection_only.py $n 200 $f 2>/dev/null | head -1; done
done
