cd $GRAFT_REPO_ROOT
timeout 1500 python gpurun_scratch/stress_mr.py 2>&1 | tail -60
