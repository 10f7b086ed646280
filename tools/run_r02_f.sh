#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_workload.sh r02rw1_text8g --workload text --bytes-per-gpu 8589934592
grep -A2 "regwin" gpurun_out/prof_r02rw1_text8g_sq.txt
