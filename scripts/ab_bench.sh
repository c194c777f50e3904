cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export CSI_HIP_LIBRARY=$GRAFT_REPO_ROOT/climaseaice.jl_amd/libcsi_hip_old.so; else unset CSI_HIP_LIBRARY; fi
    python bench.py --no-cpu-baseline --no-full-step --steps 10 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', round(j['value']/1e9,2), j['roofline']['avg_launch_ms'])"
  done
done
