# the device-side F(4x4,3x3) filter transform (egn_wino4_pack_weight_f32) against the host pack
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "filter_transform_on_the_device" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
