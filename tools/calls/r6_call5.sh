python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4h" 2>&1 | grep -E "assert|Error|passed|failed|^E" | head -40
