# kernel-time split of the HC training step (single stream: durations do not overlap)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_hc
rm -rf $O; mkdir -p $O
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hc_single -- python $R/tools/train_hc_bench.py --steps 3 --warmup 2 > $O/hc_single.json 2> $O/hc_single.err
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*kernel_stats.csv" -exec cp {} $O/hc_kernel_stats.csv \;
cd $R; timeout 200 python tools/train_hc_bench.py --steps 5 --warmup 2 > $O/hc_noprof.json 2>/dev/null
tail -c 300 $O/hc_noprof.json
