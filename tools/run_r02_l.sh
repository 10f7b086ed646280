#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_cpp_host.py -m gpu -x -q > gpurun_out/l_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/l_tests.log
