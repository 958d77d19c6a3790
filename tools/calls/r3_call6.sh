# round 3, GPU call 6: stem kernel + new kernel tests, lifter step with GEMMs + in-kernel dropout, lifter kernel stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c6
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_autograd.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 200 python tools/conv_probe.py --shape 64,256,256,3,64,3,2,1 --shape 32,256,256,3,64,3,2,1 --cfg 64,8,18,28 --iters 20 --rounds 3 --res 0 > $O/stem_probe.txt 2>&1
cat $O/stem_probe.txt
timeout 300 python tools/train_bench.py --steps 50 > $O/train_lifter.txt 2>&1; tail -2 $O/train_lifter.txt
EGONET_AMD_RNG_DROPOUT=0 timeout 300 python tools/train_bench.py --steps 50 > $O/train_lifter_maskdropout.txt 2>&1; tail -1 $O/train_lifter_maskdropout.txt
cd /tmp && export TMPDIR=/tmp
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lifter_prof -- python $R/tools/train_bench.py --steps 45 --warmup 5 > $O/lifter_prof.json 2> $O/lifter_prof.err
find $O/lifter_prof -name "*kernel_trace.csv" -delete; find $O/lifter_prof -name "*kernel_stats.csv" | head -2
