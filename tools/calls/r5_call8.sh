# round 5 call 8: the whole GPU suite on the final code, smoke(), SQ counters of the F(4x4,3x3) kernels with this build,
# and the 64-crop step replayed as a hipGraph (experiment)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c8; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.txt | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
for g in 0 64; do echo "GRAPH_MAX_N=$g"; EGONET_AMD_GRAPH_MAX_N=$g timeout 400 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; done | tee $O/graph64.txt
bash tools/pmc_wino4.sh r5 > $O/pmc_wino4.log 2>&1; tail -5 $O/pmc_wino4.log; cp gpurun_out/pmc_wino4/r5_pmc_sq_wino4.txt $O/ 2>/dev/null
