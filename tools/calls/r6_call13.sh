# round 6 call 13: the round's evidence set on the current code (bench line with live traffic, serial kernel stats, PMC traffic,
# training kernel stats) + SQ counters of conv_wino4w_kernel
bash tools/prof_r6.sh > gpurun_out/prof_r6.log 2>&1
tail -30 gpurun_out/prof_r6.log
bash tools/pmc_wino4w.sh r6 > gpurun_out/pmc_wino4w.log 2>&1
tail -5 gpurun_out/pmc_wino4w.log
