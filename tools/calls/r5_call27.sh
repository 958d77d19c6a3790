#!/bin/bash
# round 5, call 27: the whole -m gpu suite at HEAD after the fixture rename, then smoke()
cd /root/repo
mkdir -p gpurun_out/final
timeout 700 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/final/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> gpurun_out/final/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
tail -3 gpurun_out/final/pytest_gpu.txt; tail -1 gpurun_out/final/smoke.txt
