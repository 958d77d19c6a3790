R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
echo "== new (scalar transform instructions), product library"
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --wino 70,80 --rounds 4 2>&1 | grep "wino.*us "
echo "== old (compiler-packed v_pk_*), probe library built before the change"
EGONET_AMD_LIB=$R/tools/_build/libegonet_hip_probes.so timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --wino 70,80 --rounds 4 2>&1 | grep "wino.*us "
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k wino4 2>&1 | tail -2
