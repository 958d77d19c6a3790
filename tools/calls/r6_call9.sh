# round 6 call 9: the hipGraphLaunch crash of round 5 (profiles/r5_graph_replay_crash.txt) under rocgdb: backtrace of the fault
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_autograd.py tests/graph_inproc_case.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -40
echo "=== under rocgdb"
timeout 900 rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "info threads" -ex "thread apply all bt 12" --args python -m pytest tests/test_gpu_autograd.py tests/graph_inproc_case.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -150
