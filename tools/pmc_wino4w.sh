# SQ counters of conv_wino4w_kernel (cfg 86, round 6) beside conv_wino4_kernel (cfg 70) on the 96-channel class at 64 crops,
# and of conv_wino4_kernel on the 48-channel class: usage  bash tools/pmc_wino4w.sh [tag]
# (two passes of 8 counters per command; counters in their own runs, kernel trace only)
R=$GRAFT_REPO_ROOT
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_wino4w
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
CMD="python $R/tools/wino_probe.py --shape 64,32,32,96,96 --wino 70,86 --iters 3 --rounds 1"
CMDB="python $R/tools/wino_probe.py --shape 64,64,64,48,48 --wino 59,70 --iters 3 --rounds 1"
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A -- $CMD > $O/A.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B -- $CMD > $O/B.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A2 -- $CMDB > $O/A2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B2 -- $CMDB > $O/B2.txt 2>&1
( echo "# rocprofv3 --kernel-trace --pmc <pass A | pass B> -- $CMD   (tools/pmc_wino4w.sh)"
  echo "# counter sums over the device per dispatch, mean over the dispatches; condensed by tools/pmc_summary.py"
  echo "# MFMA pipe busy of the CU-busy time = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CU_CYCLES / 256 CUs)"
  echo "# pass A (96 -> 96 @ 32 x 32, 64 crops)"; python $R/tools/pmc_summary.py $O/A | grep -A8 "wino4"
  echo "# pass B"; python $R/tools/pmc_summary.py $O/B | grep -A8 "wino4"
  echo "# rocprofv3 ... -- $CMDB"
  echo "# pass A (48 -> 48 @ 64 x 64, 64 crops)"; python $R/tools/pmc_summary.py $O/A2 | grep -A8 "wino4\|wino9"
  echo "# pass B"; python $R/tools/pmc_summary.py $O/B2 | grep -A8 "wino4\|wino9" ) > $O/${TAG}_pmc_sq_wino4.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/${TAG}_pmc_sq_wino4.txt
