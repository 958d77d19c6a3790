R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_wgrad
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w -- python $R/tools/wgrad_probe.py --no-check --iters 20 > $O/out.txt 2>$O/err.txt
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
cat $O/out.txt; cut -d, -f1-4 $O/kernel_stats.csv | head -8
