# stage 0 of the next item loaded during the last stage of an item (upper halo buffer): parity of all conv_wino4 kernels,
# stream stress of the F(4x4,3x3) configurations and the ticket hand-off, 64-crop bench
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4 or winograd_at_the_bench" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_stress_streams.py -q -m gpu -k "70- or 80- or 82- or 83- or ticket" 2>&1 | tail -4
for i in 1 2 3; do timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; done
