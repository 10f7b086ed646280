#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/j_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/j_tests.log
bash tools/run_r02_h.sh
