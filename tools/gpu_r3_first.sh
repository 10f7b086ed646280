#!/bin/bash
# round 3, first GPU pass: the new queued-call tests, then the DRIVER'S EXACT COMMAND at 20 and 100 steps
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "queued or wrong_launch or launch_shape or in_place or hip_graph" > gpurun_out/r3_hint_tests.log 2>&1; rc=$?; echo "hint tests rc=$rc"; tail -5 gpurun_out/r3_hint_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_random8g_s20.json 2> gpurun_out/r3_bench_s20.err && python tools/summ.py gpurun_out/r3_bench_random8g_s20.json &&
timeout -k 10 300 python3 bench.py --gpus 1 --steps 100 --warmup 5 --cpu-sample-mib 0 > gpurun_out/r3_bench_random8g_s100.json 2> gpurun_out/r3_bench_s100.err && python tools/summ.py gpurun_out/r3_bench_random8g_s100.json
