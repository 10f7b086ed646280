#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sparse or fuzz or table or lowentropy or config4_5 or recording" > gpurun_out/r3_sparse_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3_sparse_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_r3_ab.sh "lowentropy" "norec"
