#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/p_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/p_tests.log
for w in text8g text text_32m; do timeout -k 10 300 python bench.py --workload $w --cpu-sample-mib 0 --steps 3 > gpurun_out/p_$w.json 2>/dev/null || echo "bench $w failed"; done
python tools/summ.py gpurun_out/p_text8g.json gpurun_out/p_text.json gpurun_out/p_text_32m.json
