#!/bin/bash
# peer transport for odd sub-step counts: the peer tests, then the tile fuzz (half of it on the peer transport, any count)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_evp.py -m gpu -q -k "peer" > gpurun_out/odd_peer_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/odd_peer_tests.log
python scripts/fuzz_tiles.py 0 600 > gpurun_out/odd_peer_fuzz.log 2>&1
tail -4 gpurun_out/odd_peer_fuzz.log
