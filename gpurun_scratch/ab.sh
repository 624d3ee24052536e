cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 1 0; do
USC3D_GATHER_INTO_GRAPH_INPUTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('into_inputs=$v', round(r['ms_per_step'],2))"
done; done
