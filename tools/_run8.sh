cd $GRAFT_REPO_ROOT
bash tools/ab.sh -r 2 plain: rel3000:USC3D_LANE_RELEASE_ROWS=3000 rel600:USC3D_LANE_RELEASE_ROWS=600
bash tools/ab.sh -r 2 -x "--force-dist" fd: fd_nooverlap:USC3D_OVERLAP_ALLREDUCE=0
