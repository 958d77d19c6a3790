# round 6 call 12: PCIe-inclusive rates refreshed (round 3: resident 4 284 / host fp32 4 021 / uint8 frames 3 953 crops/s), with the pipelined forms
timeout 300 python -m pytest tests/test_crop.py -q -m gpu 2>&1 | tail -3
timeout 600 python tools/frontend_bench.py --steps 30 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r6_frontend_bench.json
