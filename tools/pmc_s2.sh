# SQ counters of the stride-2 3x3 classes on the direct kernels (cfg from the shipped table) next to a stride-1 direct launch
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_s2
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
run() {
  timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A$1 -- python $R/tools/conv_probe.py --shape $2 --cfg $3 --iters 3 --res 0 > $O/A$1.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B$1 -- python $R/tools/conv_probe.py --shape $2 --cfg $3 --iters 3 --res 0 > $O/B$1.txt 2>&1
  echo "# shape $2 (N,H,W,Cin,Cout,k,stride,pad)  cfg $3"; grep "us " $O/B$1.txt
  python $R/tools/pmc_summary.py $O/A$1 | grep -A9 "conv_"; python $R/tools/pmc_summary.py $O/B$1 | grep -A9 "conv_"
}
( run 1 64,64,64,48,96,3,2,1 7
  run 2 64,64,64,48,48,3,2,1 8
  run 3 64,32,32,96,192,3,2,1 17
  run 4 64,64,64,48,48,3,1,1 8,18 ) > $O/r4_pmc_sq_s2.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep "^#\|^void\|BANK\|MFMA_BUSY\|BUSY_CU\|ACTIVE_INST_LDS\|WAIT_ANY\|WAIT_INST_LDS\|INSTS_LDS\|INSTS_VALU" $O/r4_pmc_sq_s2.txt
