"""Developer aid (not a test): run the device NCut loop on the bench's 625-segment scene and, at every
iteration, compare the device eigenvector with scipy's on the SAME device-built (A, D)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.linalg import eigh
from oracle import ncut_ref
from unscene3d_amd.pseudo_masks import ncut
from unscene3d_amd.synthetic import make_segment_scene

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 75
feats, conn, label = make_segment_scene(seed)
S = feats[0].shape[0]
dev = torch.device("cuda:0")
tr = []
ref = ncut_ref.unscene3d_ref(tuple(torch.from_numpy(f) for f in feats), np.arange(S), conn, tau=0.6, trace=tr)
print("oracle masks", ref.shape[0])
for t in tr:
    print("  oracle it", t["it"], "evals", t["evals"], "n_fg", t["n_fg"], "part", len(t["part"]))

orig = ncut.second_smallest_eigenvector
log = []
def wrapped(A, D, eps=1e-5):
    _, vec = orig(A, D, eps)
    Ah = np.where(A.cpu().numpy() > 0, 1.0, eps); Dh = D.cpu().numpy()
    w, v = eigh(np.diag(Dh) - Ah, np.diag(Dh), subset_by_index=[1, 2])
    sv = v[:, 0]
    corr = float(vec @ (Dh * sv))
    bip = vec > vec.mean()
    log.append((w, corr, bip.mean(), int(np.argmax(np.abs(vec))), float(np.abs(vec).max()), float(np.sort(np.abs(vec))[-2])))
    return np.copy(vec), vec
ncut.second_smallest_eigenvector = wrapped
m = ncut.unscene3d(tuple(torch.from_numpy(f).to(dev) for f in feats), torch.arange(S), torch.from_numpy(conn), affinity_tau=0.6)
print("device masks", m.shape[0])
for i, l in enumerate(log):
    print("  dev it", i, "scipy evals on dev A", l[0], "corr(dev,scipy)", round(l[1], 6), "fg ratio", round(l[2], 3), "argmax|v|", l[3], l[4], l[5])
# with scipy's sign imposed
log2 = list(log); log.clear()
def hook_sign():
    def wrapped2(A, D, eps=1e-5):
        _, vec = orig(A, D, eps)
        Ah = np.where(A.cpu().numpy() > 0, 1.0, eps); Dh = D.cpu().numpy()
        w, v = eigh(np.diag(Dh) - Ah, np.diag(Dh), subset_by_index=[1, 2])
        if float(vec @ (Dh * v[:, 0])) < 0:
            vec = -vec
        return np.copy(vec), vec
    return wrapped2
ncut.second_smallest_eigenvector = hook_sign()
m2 = ncut.unscene3d(tuple(torch.from_numpy(f).to(dev) for f in feats), torch.arange(S), torch.from_numpy(conn), affinity_tau=0.6)
print("device masks with scipy sign", m2.shape[0])
if m2.shape == ref.shape:
    print("iou", [(float((a & b).sum()) / max(1, (a | b).sum())) for a, b in zip(m2, ref)])
