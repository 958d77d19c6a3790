R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 64,64,64,256,48 --wino 59,70,80 > $O/probe.txt 2>&1
grep -v "rc -2" $O/probe.txt | grep "us "
bash tools/pmc_wino4.sh r4 > $O/pmc.txt 2>&1; grep -A9 "wino4" $O/pmc.txt | grep "^void\|BANK_CONFLICT\|MFMA_BUSY\|BUSY_CU\|ACTIVE_INST_LDS\|WAIT_ANY\|INSTS_VALU\|^#" | head -60
