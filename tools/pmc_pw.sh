# SQ counters of conv_pw_kernel (layer1's 1x1 pair, csrc/conv_pw.hip) at 64 crops: two passes of 8 counters, counters in
# their own runs (kernel trace only); usage  bash tools/pmc_pw.sh [tag]
R=$GRAFT_REPO_ROOT
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_pw
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
CMD="python $R/tools/pw_probe.py --crops 64 --iters 3"
python $R/tools/pw_probe.py --crops 64 > $O/probe.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A -- $CMD > $O/A.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B -- $CMD > $O/B.txt 2>&1
( echo "# python tools/pw_probe.py --crops 64   (hipEvents, min of 20)"; grep crops $O/probe.txt
  echo "# rocprofv3 --kernel-trace --pmc <pass A | pass B> -- $CMD   (tools/pmc_pw.sh)"
  echo "# counter sums over the device per dispatch, mean over the dispatches (conv_pw_kernel<true>: the pair; <false>: both one-product forms)"
  echo "# pass A"; python $R/tools/pmc_summary.py $O/A | grep -A8 "conv_pw"
  echo "# pass B"; python $R/tools/pmc_summary.py $O/B | grep -A8 "conv_pw" ) > $O/${TAG}_pmc_sq_pw.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/${TAG}_pmc_sq_pw.txt
