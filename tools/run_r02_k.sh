#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "graph or split or host_batch or medium_class" > gpurun_out/k_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/k_tests.log
