# round 6 call 20: apply_dropout routing (lifter), trainer tests
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_trainer.py tests/test_gpu_train.py -q -m gpu -k "apply_dropout or trainer or lifter" 2>&1 | grep -E "passed|failed|^E" | tail -8
