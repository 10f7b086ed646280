#!/bin/bash
# full GPU suite + the driver's command (default bench run with secondary workloads).  If the suite dies on a GPU exception,
# rocgdb reads the GPU core dump (kernel, pc, instruction of the wave that took it) into gpurun_out/diag_rocgdb.txt.
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpucore.*
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "full tests rc=$rc"; tail -3 gpurun_out/r3_tests.log
if [ $rc -ne 0 ]; then
  c=$(ls gpucore.* 2>/dev/null | head -1)
  if [ -n "$c" ]; then
    timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info dispatches" -ex "info threads" \
        -ex "bt" -ex "info registers pc exec" -ex "x/12i \$pc-24" -ex "info registers" $(which python3) -c $c > gpurun_out/diag_rocgdb.txt 2>&1
    grep -n "fault\|xception\|signal\|SIG\|k4_\|dcz" gpurun_out/diag_rocgdb.txt | head -40
  fi
  exit $rc
fi
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench_default.json'))
print("headline %.1f GB/s %.3f ms roofline %s %.3f" % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']), d['launch_shapes'])
for n,s in d.get('secondary',{}).items():
    print("%-12s %8.1f GB/s %8.3f ms  roofline %s %.3f  k4 %.3f" % (n, s['value'], s['ms_per_step'], s['roofline']['kernel'], s['roofline']['frac'], s['kernels']['k4_decode']['avg_ms']))
PY
