# A/B of two library builds on the other configurations: scripts/ab_cases.sh "<config substring>" ...
cd $GRAFT_REPO_ROOT
for c in "$@"; do
  for rep in 1 2; do
    for lib in old new; do
      if [ $lib = old ]; then export CSI_HIP_LIBRARY=$GRAFT_REPO_ROOT/climaseaice.jl_amd/libcsi_hip_old.so; else unset CSI_HIP_LIBRARY; fi
      python scripts/bench_cases.py 2048 "$c" 2>/dev/null | grep -v "^{" | grep level2 | sed "s/{.level0.*level2/ $lib level2/"
    done
  done
done
