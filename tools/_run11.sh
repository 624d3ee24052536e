cd $GRAFT_REPO_ROOT
./tools/probes/mfma_rate
bash tools/ab.sh -r 3 plain: expseg:PYTORCH_HIP_ALLOC_CONF=expandable_segments:True,PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True
