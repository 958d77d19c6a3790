# sanity of the last host-side change (programs built under their device's context): model tests
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
