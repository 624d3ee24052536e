cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > gpurun_out/r06_t6.log 2>&1; tail -3 gpurun_out/r06_t6.log
bash tools/ab.sh -r 2 -x "--force-dist" fd_laneordered: fd_r05:USC3D_LANE_ORDERED_COLLECTIVES=0 fd_norecheck:USC3D_STREAM_RECHECK=0
bash tools/ab.sh -r 2 plain: cu7of8:USC3D_LANE_CU_PATTERN=7f7f7f7f cu3of4:USC3D_LANE_CU_PATTERN=77777777 cu1of2:USC3D_LANE_CU_PATTERN=55555555 grid192:USC3D_WGRAD_GRID_LIMIT=192
