import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import ncut_ref as NR
from unscene3d_amd.pseudo_masks import ncut
z = np.load("tests/golden/ncut.npz"); name="single"
f = torch.from_numpy(z[f"{name}/feat0"]); S=f.shape[0]
dev=torch.device("cuda:0")
A,D = ncut.get_affinity_matrix(f.to(dev), tau=0.6)
Ah = A.cpu().numpy().astype(bool); Dh = D.cpu().numpy()
A0 = np.unpackbits(z[f"{name}/A0"],axis=1)[:,:S].astype(bool)
print("A mismatches", (Ah!=A0).sum(), "asym", (Ah!=Ah.T).sum())
Af = np.where(Ah, 1.0, 1e-5)
w,v = NR.fiedler(Af, Dh)
_, vec = ncut.second_smallest_eigenvector(A, D)
print("scipy evals", w, "corr(dev, scipy)", float(vec@(Dh*v)), "norm", float(vec@(Dh*vec)))
print("corr with golden", float(vec@(z[f"{name}/deg0"]*z[f"{name}/vec0"])))
# eigenvalue check: residual
L = np.diag(Dh)-np.tril(Af)-np.tril(Af,-1).T
r = L@vec - w[0]*Dh*vec
print("residual", np.abs(r).max())
