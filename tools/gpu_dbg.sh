#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for n in 26 32 48 64 100; do timeout -k 10 120 python tools/dbg_split.py $n 2>&1 | grep -v amdgpu | head -3 | cut -c1-160 | tr '\n' ' '; echo; done
