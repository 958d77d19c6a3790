# SQ counters of conv_s2r_kernel (cfg 85, csrc/conv_s2r.hip) on its two largest classes at 64 crops (two passes of 8 counters,
# counters in their own runs, kernel trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_s2r
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
run() {
  timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A$1 -- python $R/tools/conv_probe.py --shape $2 --cfg $3 --iters 3 --res 0 > $O/A$1.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B$1 -- python $R/tools/conv_probe.py --shape $2 --cfg $3 --iters 3 --res 0 > $O/B$1.txt 2>&1
  echo "# shape $2 (N,H,W,Cin,Cout,k,stride,pad)  cfg $3"; grep "us " $O/B$1.txt
  python $R/tools/pmc_summary.py $O/A$1 | grep -A9 "conv_"; python $R/tools/pmc_summary.py $O/B$1 | grep -A9 "conv_"
}
( run 1 64,64,64,48,48,3,2,1 85
  run 2 64,64,64,48,96,3,2,1 85 ) > $O/r5_pmc_sq_s2r.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep "^#\|^conv\|^void\|BANK\|MFMA_BUSY\|BUSY_CU\|ACTIVE_INST_LDS\|WAIT_ANY\|INSTS_VALU\|INSTS_VMEM" $O/r5_pmc_sq_s2r.txt
