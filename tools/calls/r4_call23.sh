# kernel-level breakdown of cfg 83 (memset + conv_wino4c_kernel<0, 2> + conv_wino4_finish_kernel) and cfg 82 / 61
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c23; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 16 64; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p$n -o p -- python $R/tools/wino_probe.py --shape $n,8,8,384,384 --direct 0 --wino 61,82,83 --iters 50 > $O/probe_$n.log 2>&1
f=$(ls $O/p$n/*kernel_stats.csv 2>/dev/null | head -1)
echo "== n=$n"; head -12 $f | cut -c1-160
done
