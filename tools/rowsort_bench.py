"""Developer aid: device time of usc_rowsort_build on the bench scene's maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import MinkowskiEngine as ME, ops
from unscene3d_amd.synthetic import make_scene
dev = torch.device("cuda:0")
sc = make_scene(2000, target_voxels=150000)
c3, umap, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
cmap, _, _ = ops.coordmap_build(coords)
for lvl in range(3):
    nbr = ops.kernel_map_cube(cmap, 3)
    def run():
        nbr._usc_rowsort = None
        ops.rowsort(nbr)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"rows {cmap.n:7d}: rowsort {e0.elapsed_time(e1)/20*1e3:7.1f} us")
    cmap, _, _ = ops.coordmap_build(cmap.coords, quant=2 * cmap.tensor_stride, tensor_stride=2 * cmap.tensor_stride)
