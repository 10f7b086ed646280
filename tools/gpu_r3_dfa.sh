#!/bin/bash
# k4_dfa recording walk: decoder parity tests first, then A/B against the three-walk build on the text workloads
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not 8gib and not config" > gpurun_out/r3_dfa_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3_dfa_tests.log
[ $rc -eq 0 ] || exit $rc
for w in text8g text text_32m; do
  timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 5 --warmup 2 > gpurun_out/dfa_rec_$w.json 2> gpurun_out/dfa_rec_$w.err || echo "rec $w failed"
  DCZ_LIB=$R/variants/lib_norec.so timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 5 --warmup 2 > gpurun_out/dfa_norec_$w.json 2> gpurun_out/dfa_norec_$w.err || echo "norec $w failed"
  python tools/summ.py gpurun_out/dfa_rec_$w.json gpurun_out/dfa_norec_$w.json
done
