# SQ counters of conv_wino4_kernel (cfg 70) / conv_wino4b_kernel (cfg 80) beside conv_wino9_kernel (cfg 59) on the shape
# classes they serve: usage  bash tools/pmc_wino4.sh [tag]   (tag names the summary file, default r4)
# (two passes of 8 counters; counters in their own runs, kernel trace only)
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_wino4
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
CMD="python $R/tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 59,70 --iters 3 --rounds 1"
CMDB="python $R/tools/wino_probe.py --shape 64,16,16,192,192 --wino 59,80 --iters 3 --rounds 1"
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A -- $CMD > $O/A.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B -- $CMD > $O/B.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A2 -- $CMDB > $O/A2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B2 -- $CMDB > $O/B2.txt 2>&1
( echo "# rocprofv3 --kernel-trace --pmc <pass A | pass B> -- $CMD   (tools/pmc_wino4.sh)"
  echo "# counter sums over the device per dispatch, mean over the dispatches of BOTH shapes; condensed by tools/pmc_summary.py"
  echo "# pass A"; python $R/tools/pmc_summary.py $O/A | grep -A8 "wino4\|wino9"
  echo "# pass B"; python $R/tools/pmc_summary.py $O/B | grep -A8 "wino4\|wino9"
  echo "# rocprofv3 ... -- $CMDB"
  echo "# pass A"; python $R/tools/pmc_summary.py $O/A2 | grep -A8 "wino4\|wino9"
  echo "# pass B"; python $R/tools/pmc_summary.py $O/B2 | grep -A8 "wino4\|wino9" ) > $O/${TAG}_pmc_sq_wino4.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/${TAG}_pmc_sq_wino4.txt
