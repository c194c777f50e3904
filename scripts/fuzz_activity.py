"""Fuzz campaign of the round-6 structure cuts (tile activity, row-constant rows): random configurations, both cuts on vs off, every
parent cell of u, v, sigma, alpha, zeta, Delta bit for bit.  python scripts/fuzz_activity.py <seed lo> <seed hi>  (GPU box)
python scripts/fuzz_activity.py band <seed lo> <seed hi>: random north-fold configurations, the band's copy-free launches against the
three kernels + copies (CSI_BAND_FUSED=0)."""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import test_gpu_activity as T
if len(sys.argv) > 1 and sys.argv[1] == "band":
    import pytest
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    bad = used = 0
    for seed in range(lo, hi):
        mp = pytest.MonkeyPatch()
        try:
            T.test_band_fuzz_bitwise(seed, mp)
            used += bool(T.LAST_FUZZ.get("band"))
        except AssertionError as e:
            bad += 1; print("FAIL", seed, str(e)[:400], flush=True)
        except Exception as e:
            bad += 1; print("ERR", seed, type(e).__name__, str(e)[:300], flush=True)
        finally:
            mp.undo()
    print(f"band seeds {lo} .. {hi - 1}: done, failures: {bad}; the fold band ran (fewer launches than with the three kernels) in {used} of {hi - lo} cases")
    sys.exit(1 if bad else 0)
bad = skipped = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 200)
live = []
for seed in range(lo, hi):
    try:
        T.test_cuts_fuzz_bitwise(seed)
        t, l, u = T.LAST_FUZZ["activity"]
        live.append((u, t, l, T.LAST_FUZZ["row_constant_rows"]))
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:400], flush=True)
    except Exception as e:
        bad += 1
        print("ERR", seed, type(e).__name__, str(e)[:300], flush=True)
used = [x for x in live if x[0]]
print(f"seeds {lo} .. {hi - 1}: done, failures: {bad}; live launches used in {len(used)} of {len(live)} cases, tiles skipped in "
      f"{sum(1 for x in used if x[2] < x[1])} (mean live fraction there {sum(x[2] / x[1] for x in used if x[2] < x[1]) / max(1, sum(1 for x in used if x[2] < x[1])):.2f}); "
      f"row-constant rows marked in {sum(1 for x in live if x[3])} cases")
