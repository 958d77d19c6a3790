R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 python bench.py --no-train --no-cpu-baseline 2>/dev/null | head -c 220; echo
