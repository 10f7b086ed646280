#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "recording or medium or split or automat or sparse or fuzz or table or text or synchronise or damage" > gpurun_out/r3_dfa_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3_dfa_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_r3_ab.sh "text8g text text_32m" "nox6"
