#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_r3_ab.sh "text8g" "x8 x24 x32"
bash tools/pmc.sh r3text --workload text8g > gpurun_out/r3_text8g_sq.txt 2>&1; grep -A3 "k4_dfa" gpurun_out/r3_text8g_sq.txt | head -20
rm -rf gpurun_out/pmc_r3text_*
