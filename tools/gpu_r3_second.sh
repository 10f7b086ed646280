#!/bin/bash
# round 3, second GPU pass: full GPU suite, the cost of a wrong hint (plain + under rocprofv3 --stats), rocprof stats of the driver's command
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/mispredict.py 6 > gpurun_out/r3_mispredict.txt 2> gpurun_out/r3_mispredict.err; cat gpurun_out/r3_mispredict.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_mis -- python3 $R/tools/mispredict.py 4 > $R/gpurun_out/r3_mispredict_prof.txt 2> $R/gpurun_out/r3_mispredict_prof.err
cp $(ls $R/gpurun_out/st_mis/*/*_kernel_stats.csv | head -1) $R/gpurun_out/r3_mispredict_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --cpu-sample-mib 0 > $R/gpurun_out/r3_driver_cmd_prof.json 2> $R/gpurun_out/r3_driver_cmd_prof.err
cp $(ls $R/gpurun_out/st_drv/*/*_kernel_stats.csv | head -1) $R/gpurun_out/r3_driver_cmd_kernel_stats.csv
rm -rf $R/gpurun_out/st_mis $R/gpurun_out/st_drv
head -12 $R/gpurun_out/r3_mispredict_kernel_stats.csv; head -8 $R/gpurun_out/r3_driver_cmd_kernel_stats.csv
