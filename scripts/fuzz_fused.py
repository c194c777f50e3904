"""Extended fuzz of the fused EVP paths against the three-kernel path (run on the GPU box; 24 seeds of the same test are in the suite)."""
import sys, numpy as np, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import test_gpu_evp as T
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 200)
for seed in range(lo, hi):
    try:
        T.test_fused_paths_fuzz_bitwise(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:300])
    except Exception as e:
        bad += 1
        print("ERR", seed, type(e).__name__, str(e)[:200])
print("done, failures:", bad)
