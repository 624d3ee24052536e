"""Developer/test aid: generalized Fiedler vector of a synthetic segment scene -> .npz (used by the test that compares the
one-launch and the stepwise tridiagonalisation: run it twice, once with USC3D_TRI_STEPWISE=1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from unscene3d_amd.pseudo_masks import ncut
from unscene3d_amd.synthetic import make_segment_scene

side, out = int(sys.argv[1]), sys.argv[2]
feats, conn, label = make_segment_scene(3, side=side, n_objects=min(16, max(2, side // 2)))
dev = torch.device("cuda:0")
A, D = ncut.get_affinity_matrix(tuple(torch.from_numpy(f).to(dev) for f in feats), tau=0.6)
_, vec = ncut.second_smallest_eigenvector(A, D)
np.savez(out, vec=np.asarray(vec), A=A.cpu().numpy(), D=D.cpu().numpy())
