#!/bin/bash
# quick check of a K4 change: the decode parity tests, then the text-like bench workloads
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decode or split or medium or sparse or classes or config4 or roundtrip or trunc or golden or fuzz or padding or periodic" > gpurun_out/q_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/q_tests.log
[ $rc -eq 0 ] || exit 1
for w in text text8g lowentropy text_32m; do timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 3 > gpurun_out/q_$w.json 2>/dev/null || echo "bench $w failed"; done
python tools/summ.py gpurun_out/q_text.json gpurun_out/q_text8g.json gpurun_out/q_lowentropy.json gpurun_out/q_text_32m.json
