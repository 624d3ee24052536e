cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
