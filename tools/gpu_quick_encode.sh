#!/bin/bash
# quick check of an encoder-side change: histogram / code build / encode parity tests, then all bench workloads
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram or code_build or parity_sizes or ragged or long_codes or pins or config or fuzz_parity or classes" > gpurun_out/r_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r_tests.log
[ $rc -eq 0 ] || exit 1
for w in text text_32m text8g lowentropy; do timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 3 > gpurun_out/r_$w.json 2>/dev/null || echo "bench $w failed"; done
python tools/summ.py gpurun_out/r_text.json gpurun_out/r_text_32m.json gpurun_out/r_text8g.json gpurun_out/r_lowentropy.json
