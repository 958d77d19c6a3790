R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c3
mkdir -p $O
cd $R
export EGONET_AMD_LIB=$R/tools/_build/libegonet_hip_probes.so
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,64,64,96,48 --wino 70,82,83,84 > $O/sched_geo0.txt 2>&1
grep -v "rc -2" $O/sched_geo0.txt | grep "us \|wino8[2-4].*max"
timeout 600 python tools/wino_probe.py --shape 64,16,16,192,192 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 16,64,64,48,48 --wino 80,85,86,87 > $O/sched_geo1.txt 2>&1
grep -v "rc -2" $O/sched_geo1.txt | grep "us \|wino8[5-7].*max"
