R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c8
mkdir -p $O
cd $R
timeout 300 python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1
cat $O/wgrad_probe.txt
timeout 900 python -m pytest tests/test_gpu_train_hrnet.py tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad or gradients or two_steps or pedestrian or winograd" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 300 python tools/train_hc_bench.py --steps 20 --warmup 5 > $O/hc.json 2> $O/hc.err; tail -c 400 $O/hc.json
