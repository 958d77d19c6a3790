# round 5 call 13: the whole GPU suite on the final code (graph replay opt-in, its case in a subprocess); SQ counters of conv_pw
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c13; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -X faulthandler -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Fatal" $O/pytest_gpu.txt | tail -8
bash tools/pmc_pw.sh r5 > $O/pmc_pw.log 2>&1; cp gpurun_out/pmc_pw/r5_pmc_sq_pw.txt $O/ 2>/dev/null; head -8 $O/r5_pmc_sq_pw.txt
