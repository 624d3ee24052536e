"""CPU: oracle/ncut_ref.py (affinity + scipy eigenvector) against the golden vectors captured from the
reference's own get_affinity_matrix / second_smallest_eigenvector (tests/golden/ncut.npz)."""
import os

import numpy as np
import torch

from oracle import ncut_ref as NR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ncut.npz")


def _case(z, name, it=0):
    feats = [torch.from_numpy(z[f"{name}/feat{j}"]) for j in range(2) if f"{name}/feat{j}" in z.files]
    S = feats[0].shape[0]
    A = np.unpackbits(z[f"{name}/it{it}/A"], axis=1)[:, :S].astype(bool)
    return feats, S, A, z[f"{name}/it{it}/deg"], z[f"{name}/it{it}/vec"], float(z[f"{name}/tau"])


def test_affinity_and_eigenvector_match_reference():
    z = np.load(GOLD)
    for name in ("single", "dual", "irregular"):
        feats, S, A0, deg0, vec0, tau = _case(z, name)
        A, d = NR.affinity(feats[0] if len(feats) == 1 else (feats[0], feats[1]), tau)
        assert np.array_equal(A > 0.5, A0)
        np.testing.assert_allclose(d, deg0, rtol=1e-12)
        w, v = NR.fiedler(A, d)
        # LAPACK's eigenvector sign is not portable across BLAS builds / core counts (observed: this
        # container and the GPU box disagree), so the oracle is compared up to sign here; the device
        # kernel restates netlib's conventions and is compared WITH sign against the golden vector.
        assert abs(float(v @ (d * vec0))) > 0.999999
        np.testing.assert_allclose(w, z[f"{name}/it0/evals"], rtol=1e-6)


def test_oracle_loop_reproduces_reference_masks_up_to_lapack_sign():
    """The restated loop gives the reference's masks when LAPACK returns the same signs as in the build
    container; elsewhere it must still produce valid masks (disjoint, non-empty)."""
    z = np.load(GOLD)
    name = "single"
    f = torch.from_numpy(z[f"{name}/feat0"])
    S = f.shape[0]
    masks = NR.unscene3d_ref(f.clone(), np.arange(S), z[f"{name}/conn"], tau=float(z[f"{name}/tau"]))
    assert masks.shape[1] == S and masks.shape[0] >= 1
    assert (masks.sum(0) <= 1).all() and (masks.sum(1) > 0).all()


def test_oracle_loop_reproduces_reference_masks_on_irregular_scene():
    """Per-segment noise levels give irregular graphs without near-breakdowns in the tridiagonalisation:
    LAPACK's sign is then well defined and the restated loop must give exactly the reference's masks."""
    z = np.load(GOLD)
    name = "irregular"
    f = tuple(torch.from_numpy(z[f"{name}/feat{j}"]) for j in range(2))
    S = f[0].shape[0]
    masks = NR.unscene3d_ref(f, np.arange(S), z[f"{name}/conn"], tau=float(z[f"{name}/tau"]))
    ref = np.unpackbits(z[f"{name}/masks"], axis=1)[:int(z[f"{name}/n_masks"]), :S].astype(bool)
    assert masks.shape == ref.shape
    assert np.array_equal(masks, ref)


def test_oracle_aggregate_features_matches_reference():
    """N1 restatement vs the reference's own aggregate_features (tests/golden/aggregate.npz): gapped segment ids,
    invalid rows, all-zero segments filled from `zero_segments[0]`'s neighbours (:387) or from the global mean."""
    z = np.load(os.path.join(os.path.dirname(GOLD), "aggregate.npz"))
    for case in ("neigh", "global"):
        for mode in ("mean", "max"):
            agg, uniq = NR.aggregate_features(z[f"{case}/feats"], z[f"{case}/seg"], z[f"{case}/conn"], mode)
            assert np.array_equal(uniq, z[f"{case}/uniq"])
            np.testing.assert_allclose(agg, z[f"{case}/{mode}"], rtol=2e-6, atol=1e-7)


def test_host_blob_logic_equals_the_reference_scan():
    """unscene3d_amd.pseudo_masks.ncut.separate_segments (cached neighbour sets, owner lookup, the reference's
    positional scan only for an id that touches several blobs) against the oracle's restatement of the reference loop
    (pseudo_masks/unscene3d_pseudo_main.py:181-250) on random directed and symmetric segment graphs: the same blob
    for the arg-max seed — including the scan's skip-after-merge quirk."""
    from unscene3d_amd.pseudo_masks import ncut

    rng = np.random.default_rng(3)
    for t in range(300):
        S = int(rng.integers(5, 120))
        E = int(rng.integers(0, 6 * S))
        uniq = np.sort(rng.choice(400, S, replace=False))
        conn = np.stack([rng.choice(uniq, E), rng.choice(uniq, E)], 1) if E else np.zeros((0, 2), np.int64)
        if t % 2 == 0 and E:
            conn = np.concatenate([conn, conn[:, ::-1]])
        vec = rng.standard_normal(S)
        bip = vec > vec.mean()
        exp = NR._separate(bip, vec, uniq, conn)
        nbs = ncut.neighbour_sets(uniq, conn)
        assert nbs == {int(s): set(conn[conn[:, 0] == s, 1].tolist()) for s in uniq}
        assert ncut.separate_segments(bip, vec, uniq, conn, mode="max") == exp
        assert ncut.separate_segments(bip, vec, uniq, conn, mode="max", neighbours=nbs) == exp
