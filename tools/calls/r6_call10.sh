# round 6 call 10: the whole GPU suite in ONE process with graph replay on for n <= 16 + the in-process graph case at the end
export PYTHONFAULTHANDLER=1
export EGONET_AMD_GRAPH_MAX_N=16
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider tests/graph_inproc_case.py 2>&1 | tail -40
