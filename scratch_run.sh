cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
EGONET_AMD_RETUNE=1 EGONET_AMD_TUNE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/tuned_inf.json timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | cut -c1-120
EGONET_AMD_RETUNE=1 EGONET_AMD_TUNE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/tuned_train.json timeout 900 python tools/train_hc_bench.py --batch 32 --steps 1 --warmup 1 2>/dev/null | cut -c1-120
ls -la gpurun_out/tuned_*.json
