#!/bin/bash
# full GPU test suite, then every bench workload (one line each): the check before a commit that touches kernels
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/o_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/o_tests.log
for w in random8g random256m text text8g lowentropy text_32m; do timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 > gpurun_out/o_$w.json 2>/dev/null || echo "bench $w failed"; done
python tools/summ.py gpurun_out/o_random8g.json gpurun_out/o_random256m.json gpurun_out/o_text.json gpurun_out/o_text8g.json gpurun_out/o_lowentropy.json gpurun_out/o_text_32m.json
