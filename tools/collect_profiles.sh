#!/bin/bash
# usage: collect_profiles.sh <tag> <round prefix, e.g. r02>   -- copies the summaries of refresh_profiles.sh into profiles/
V=$1; P=$2
for w in random8g text8g lowentropy text32m text256 random256m; do for k in bench.json kernel_stats.csv sq.txt traffic.txt; do
  [ -f gpurun_out/prof_${V}_${w}_$k ] && cp gpurun_out/prof_${V}_${w}_$k profiles/${P}_${V}_${w}_$k; done; done
for w in random8g random256m text text8g lowentropy text_32m 2rank; do [ -f gpurun_out/bench_${w}_$V.json ] && cp gpurun_out/bench_${w}_$V.json profiles/${P}_bench_${w}_$V.json; done
ls profiles | grep "${P}_" | wc -l
# HBM traffic of the default workload: mk_traffic.py printed it on the GPU box (gpurun_out/traffic_json_<tag>.txt); the file
# bench.py reads is keyed by workload
[ -f gpurun_out/traffic_json_$V.txt ] && python3 -c "import json,sys; r=json.load(open('gpurun_out/traffic_json_$V.txt')); json.dump({'random8g': r}, open('profiles/pmc_traffic.json','w'), indent=1); json.dump({'random8g': r}, open('profiles/${P}_${V}_pmc_traffic_random8g.json','w'), indent=1)"
