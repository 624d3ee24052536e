cd $GRAFT_REPO_ROOT
timeout 1700 python gpurun_scratch/stress_mr4.py 2>&1 | tail -8
