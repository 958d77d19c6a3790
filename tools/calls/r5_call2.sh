# round 5 call 2: small batches replayed as a hipGraph (engine._forward_graphed), packer test, data-gradient error list of the
# F(4x4,3x3) tape at 32 crops, small-batch bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu 2>&1 | tail -8 | tee $O/pytest_models.txt
timeout 300 python -m pytest tests/test_gpu_train_ops.py -q -m gpu -k "f43" 2>&1 | tail -4 | tee $O/pytest_pack.txt
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k training -s 2>&1 | grep -E "data gradient|worst|launches|passed|failed" | tee $O/train32_dgrad.txt
bench() { EGONET_AMD_AUTOTUNE=0 EGONET_AMD_GRAPH_MAX_N=$3 timeout 600 python bench.py --batch $1 --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 batch $1: %.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; }
for b in 1 4 16; do bench $b eager 0; bench $b graph 16; done | tee $O/small_batch_bench.txt
