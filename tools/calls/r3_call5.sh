# round 3, GPU call 5: top up the tile table with conv_wino9_kernel (cfg 59-62, the old Winograd ids re-timed in
# the same session), then the whole GPU suite and the bench with the new table
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c5
mkdir -p $O
cd $R
timeout 900 python tools/retune.py --out $O/gfx950.json --match k3x3_s1 --retime 45,51,52,56,57 > $O/retune.log 2>&1
tail -40 $O/retune.log
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
head -c 400 $O/bench_n1.json; echo
