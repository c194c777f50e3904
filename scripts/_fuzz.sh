for g in fuzz_fused fuzz_tiles fuzz_fold; do echo "== $g 80000 .. 80800"; timeout 1500 python scripts/$g.py 80000 80800 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -2; done
echo "== fuzz_steps 80000 .. 80200"; timeout 1200 python scripts/fuzz_steps.py 80000 80200 2>&1 | tail -1
echo "== fuzz_local_tiles 80000 .. 80400"; timeout 1500 python scripts/fuzz_local_tiles.py 80000 80400 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -1
