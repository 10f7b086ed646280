cd $GRAFT_REPO_ROOT
for w in random256m text random8g; do
  for f in "" "--service-stream"; do
    timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 20 --warmup 5 $f > gpurun_out/abs_${w}_${f#--}.json 2> gpurun_out/abs.err || echo "failed $w $f: $(tail -2 gpurun_out/abs.err)"
    echo "$w [$f]"; python tools/summ.py gpurun_out/abs_${w}_${f#--}.json
  done
done
