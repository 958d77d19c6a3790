cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_models.py -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
run chain
EGONET_AMD_CHAIN=0 run nochain
run chain
EGONET_AMD_CHAIN=0 run nochain
