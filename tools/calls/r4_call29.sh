# A/B of the next-item prefetch: product build vs the build before it (tools/_build/libegonet_hip_cox99.so), F(4x4,3x3) probes
cd $GRAFT_REPO_ROOT
for lib in "" tools/_build/libegonet_hip_cox99.so; do
echo "== lib '$lib'"
EGONET_AMD_LIB=$lib timeout 300 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 128,64,64,48,48 --shape 128,32,32,96,96 --direct 0 --wino 70,80 --iters 50 2>&1 | grep "wino.*us"
EGONET_AMD_LIB=$lib timeout 300 python tools/wino_probe.py --shape 128,16,16,192,192 --shape 128,8,8,384,384 --direct 0 --wino 80,82 --iters 50 2>&1 | grep "wino.*us"
done
