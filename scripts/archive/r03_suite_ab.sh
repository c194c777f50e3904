#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/gpu_tests.sh
bash scripts/r03_ab.sh ab_fold_peer "peer:--tile 1024x512 --force-connected --no-compare" "head:"
