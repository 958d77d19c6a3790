#!/bin/bash
# round 5, call 28: the driver's own command line (no flags) at HEAD, on whichever box the pool hands out
cd /root/repo
mkdir -p gpurun_out/head
timeout 400 python bench.py > gpurun_out/head/bench_n1.json 2> gpurun_out/head/bench_n1.err
echo "bench exit $?"; tail -c 400 gpurun_out/head/bench_n1.json
