R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r2
mkdir -p $O
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train > $O/serial_bench.json 2> $O/serial.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train > $O/lanes_bench.json 2> $O/lanes.err
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> $O/fetch.err
EGONET_AMD_LANES=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> $O/write.err
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hc_single -- python $R/tools/train_hc_bench.py --steps 3 --warmup 2 > $O/hc_single.json 2> $O/hc_single.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lifter -- python $R/tools/train_bench.py --steps 45 --warmup 5 > $O/lifter.json 2> $O/lifter.err
# keep only the stats / counter CSVs (traces are large)
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O; find $O -name "*.csv" | head -30
