# round 3, GPU call 1: instruction-tax microbenchmark, the new parity / autograd tests, the bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c1
mkdir -p $O
cd $R
timeout 120 ./tools/micro/mfma_tax > $O/mfma_tax.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_autograd.py -q -s -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "pytest rc $?" >> $O/pytest_new.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
EGONET_AMD_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-train --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
tail -5 $O/pytest_new.log; head -c 600 $O/bench_n1.json; echo; head -c 300 $O/bench_gloo2.json; echo; tail -3 $O/bench_gloo2.err
