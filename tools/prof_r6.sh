# round 6 evidence: bench line, serial kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE in their own runs) for the
# inference bench AND the two training programs, kernel stats of the HC training step (single stream)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r6
rm -rf $O; mkdir -p $O
cd $R && timeout 900 python bench.py --live-traffic --profile-json $O/profile.json > $O/bench_n1.json 2> $O/bench_n1.err; cd /tmp
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train > $O/serial_bench.json 2> $O/serial.err
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> $O/fetch.err
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> $O/write.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hc_single -- python $R/tools/train_hc_bench.py --steps 3 --warmup 2 > $O/hc_single.json 2> $O/hc_single.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/hc_fetch -- python $R/tools/train_hc_bench.py --steps 1 --warmup 1 > /dev/null 2> $O/hc_fetch.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/hc_write -- python $R/tools/train_hc_bench.py --steps 1 --warmup 1 > /dev/null 2> $O/hc_write.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l_single -- python $R/tools/train_bench.py --steps 50 --warmup 5 > $O/l_single.json 2> $O/l_single.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/l_fetch -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2> $O/l_fetch.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/l_write -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2> $O/l_write.err
cd $R
python tools/pmc_traffic.py $O/fetch $O/write $O/hc_fetch $O/hc_write $O/l_fetch $O/l_write > $O/r6_pmc_traffic.json 2> $O/pmc_traffic.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
find $O/serial -name "*kernel_stats.csv" -exec cp {} $O/r6_rocprofv3_kernel_stats_serial.csv \;
find $O/hc_single -name "*kernel_stats.csv" -exec cp {} $O/r6_train_hc_kernel_stats_single_stream.csv \;
find $O/l_single -name "*kernel_stats.csv" -exec cp {} $O/r6_train_lifter_kernel_stats_single_stream.csv \;
du -sh $O; head -c 300 $O/bench_n1.json; echo; tail -c 200 $O/hc_single.json; python - <<'PY'
import json
t=json.load(open('gpurun_out/prof_r6/r6_pmc_traffic.json'))
for k,v in sorted(t['kernels'].items(), key=lambda kv:-kv[1]['hbm_bytes_per_launch'])[:12]: print('%-70s %8.1f MB x %d'%(k[:70],v['hbm_bytes_per_launch']/1e6,v['dispatches']))
PY
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_r6/profile.json'))
tot=sum(c['ms'] for c in d['classes'])
with open('gpurun_out/prof_r6/r6_bench_profile_classes.txt','w') as f:
    f.write('# bench.py --profile-json: per-class time of the 64-crop inference program, serial per-launch hipEvents (mean of 3 passes)\n')
    f.write('# total %.3f ms over %d classes\n' % (tot, len(d['classes'])))
    for c in d['classes']:
        f.write('%-36s %3d launches %8.1f us avg %7.3f ms %5.1f%%\n' % (c['name'], c['launches'], c['avg_us'], c['ms'], 100*c['ms']/tot))
b=json.loads(open('gpurun_out/prof_r6/bench_n1.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline'].get('traffic'), b['roofline'].get('traffic_source'), b['roofline'].get('traffic_live_error'))
print(b.get('train_hc',{}).get('ms_per_step'), b.get('train_lifter',{}).get('ms_per_step'))
PY
