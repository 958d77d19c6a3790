# SQ counters of the Winograd weight-gradient kernel on two shape classes (two passes of 8 counters)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_wgrad
rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
CMD="python $R/tools/wgrad_probe.py --no-check --iters 3 --shape 48,48,64,64 --shape 192,192,16,16"
timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/A -- $CMD > $O/A.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/B -- $CMD > $O/B.txt 2>&1
python $R/tools/pmc_summary.py $O/A > $O/summary_A.txt
python $R/tools/pmc_summary.py $O/B > $O/summary_B.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +5M -delete
cat $O/summary_A.txt $O/summary_B.txt | grep -A9 "wgrad_wino"
