R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4c31
bash tools/prof_r4.sh > gpurun_out/r4c31/prof.log 2>&1; tail -22 gpurun_out/r4c31/prof.log | cut -c1-300
