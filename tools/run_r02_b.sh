#!/bin/bash
# round 2, GPU call B: full GPU test suite + default bench + config 3 on the fixed-length-class build
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/b_tests.log
timeout -k 10 300 python bench.py --cpu-sample-mib 0 > gpurun_out/b_random8g.json 2> gpurun_out/b_random8g.err || echo "bench failed"
timeout -k 10 300 python bench.py --workload random256m --cpu-sample-mib 0 > gpurun_out/b_random256m.json 2> gpurun_out/b_random256m.err || echo "bench 256m failed"
python tools/summ.py gpurun_out/b_random8g.json gpurun_out/b_random256m.json
