#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/e_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/e_tests.log
timeout -k 10 300 python bench.py --workload text --bytes-per-gpu 8589934592 --cpu-sample-mib 0 --steps 3 > gpurun_out/e_text8g.json 2> gpurun_out/e_text8g.err || echo "text8g failed"
timeout -k 10 300 python bench.py --workload text --cpu-sample-mib 0 --steps 3 > gpurun_out/e_text1g.json 2> gpurun_out/e_text1g.err || echo "text1g failed"
python tools/summ.py gpurun_out/e_text8g.json gpurun_out/e_text1g.json
