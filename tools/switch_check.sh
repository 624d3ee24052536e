# every A/B switch of the model path off, one at a time: the step must run and print the loss of the reference configuration
# (same seed, same scene, 6 steps) — bit-identical where the docs say so.  Usage (GPU box): bash tools/switch_check.sh
cd $GRAFT_REPO_ROOT
run() { env "$@" python - <<'P'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse(["--no-cpu-baseline", "--voxels", "60000"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
step = bench.make_mask3d_step(args, dev, 0, 1)
ls = [float(step(1)[0]) for _ in range(6)]
step.close()
print(" ".join(f"{v:.7f}" for v in ls))
P
}
echo -n "reference configuration      : "; run USC3D_X=1 2>/dev/null | tail -1
for s in USC3D_GRAD_SINKS USC3D_LN_PASSTHROUGH USC3D_GATHER_INTO_GRAPH_INPUTS USC3D_FUSED_KEY_SAMPLING USC3D_RESIDUAL_IN_PROJECTION USC3D_PADDED_MASK_EMBED USC3D_FUSED_QKV USC3D_GROUP_WGRAD USC3D_BACKBONE_PROGRAM USC3D_NATIVE_UNITS USC3D_GRAD_IN_PLACE USC3D_PREFETCH_THREAD USC3D_FUSED_ATTN_MASK USC3D_FUSED_CRITERION; do
  printf "%-29s: " "$s=0"; run $s=0 2>/dev/null | tail -1
done
