for g in fuzz_fused fuzz_tiles fuzz_fold; do echo "== $g 90000 .. 91600"; timeout 2400 python scripts/$g.py 90000 91600 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -1; done
echo "== fuzz_steps 90000 .. 90400"; timeout 2000 python scripts/fuzz_steps.py 90000 90400 2>&1 | tail -1
echo "== fuzz_local_tiles 90000 .. 90800"; timeout 2400 python scripts/fuzz_local_tiles.py 90000 90800 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -1
