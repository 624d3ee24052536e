"""Host-only checks of the library's launch planning and workspace sizing (no GPU needed: these entry points do not
touch the device).  They pin the dispatch rules DESIGN.md describes."""
import pytest

from unscene3d_amd._lib import lib


def _plan(kind, n, cin, cout, K):
    code = lib.usc_spconv_plan(kind, n, cin, cout, K)
    return {"NB": code & 0xFF, "aligned": (code >> 8) & 1, "compact": (code >> 12) & 1, "full": (code >> 13) & 1,
            "hi": code >> 16}


def test_conv_plan_dispatch_rules():
    # large maps, >= 64 input channels, table form -> tile-compacted kernel, NB from the column blocks
    assert _plan(0, 148564, 96, 96, 27)["compact"] == 1 and _plan(0, 148564, 96, 96, 27)["NB"] == 3
    assert _plan(0, 40421, 96, 128, 27)["compact"] == 1 and _plan(0, 40421, 96, 128, 27)["NB"] == 2
    # small maps, 32 input channels, 1x1 convs, non-aligned widths -> not compact
    assert _plan(0, 9402, 128, 128, 27)["compact"] == 0
    assert _plan(0, 148564, 32, 32, 27)["compact"] == 0
    assert _plan(0, 148564, 96, 128, 1)["compact"] == 0
    assert _plan(0, 148564, 3, 32, 27)["aligned"] == 0
    # small maps split the offsets over groups (hi = G > 1), large ones do not
    assert _plan(0, 507, 256, 256, 27)["hi"] > 1
    assert _plan(0, 148564, 96, 128, 1)["hi"] == 1


def test_empty_maps_plan_without_faulting():
    for kind in (0, 1, 2):
        lib.usc_spconv_plan(kind, 0, 32, 32, 27)
    assert lib.usc_spconv_gather_gemm_ws_bytes(0, 32, 32, 27) == 0
    assert lib.usc_spconv_sorted_ws_bytes(0, 32, 32, 27) == 0


def test_wgrad_plan_matches_design():
    p = _plan(2, 4011228, 96, 96, 27)
    assert p["full"] == 1 and p["NB"] == 3 and p["hi"] == 3            # 3 x 3 accumulator tiles per wave
    p = _plan(2, 253854, 128, 128, 27)
    assert p["full"] == 1 and p["NB"] == 4 and p["hi"] == 2
    p = _plan(2, 4011228, 128, 96, 27)                                  # round 5: 4 x 3 tiles, accumulators in the AGPR half
    assert p["full"] == 1 and p["NB"] == 3 and p["hi"] == 4
    assert _plan(2, 148564, 96, 128, 1)["hi"] == 1                      # (the mirror form <3,4> measured slower: not used)
    assert _plan(2, 4011228, 3, 32, 27)["full"] == 0 and _plan(2, 4011228, 3, 32, 27)["aligned"] == 0


@pytest.mark.parametrize("n,cin,cout,K", [(148564, 96, 96, 27), (507, 256, 256, 27), (9402, 128, 192, 27), (40421, 32, 32, 8)])
def test_workspace_sizes_are_consistent(n, cin, cout, K):
    ws = lib.usc_spconv_gather_gemm_ws_bytes(n, cin, cout, K)
    p = _plan(0, n, cin, cout, K)
    if p["compact"]:
        assert ws == K * cin * cout * 4                                 # packed weight slices
    elif p["hi"] > 1:
        assert ws == p["hi"] * n * cout * 4                             # partial sums of the offset groups
    else:
        assert ws == 0
    ws_sorted = lib.usc_spconv_sorted_ws_bytes(n, cin, cout, K)
    assert ws_sorted % (n * cout * 4) == 0                              # G partial buffers or none
    assert lib.usc_rowsort_ws_bytes(K, n) >= n * 4 + 2 * 4096 * 4       # row masks + bucket counters and cursors
    assert lib.usc_spconv_wgrad_ws_bytes(K, cin, cout) >= K * cin * cout * 4


def test_attention_and_layernorm_workspaces():
    for S in (200, 800, 3200, 12800):
        ws = lib.usc_attn_ws_bytes(100, S, 1, 8)
        assert ws >= 1 * S * 16 + 8 * 128 * 16 * 4                      # mask bits + at least one partial (o) slab
    assert lib.usc_attn_ws_bytes(100, 12800, 1, 8) >= lib.usc_attn_ws_bytes(100, 200, 1, 8)
    assert lib.usc_layernorm_bwd_ws_bytes(100, 128) == 0                # one workgroup: no partials
    assert lib.usc_layernorm_bwd_ws_bytes(5000, 128) == 5 * 2 * 128 * 4


def test_steady_state_preparation_on_a_host_device_only_freezes_the_heap():
    """trainer.prepare_steady_state(cpu): no stream pools to size; the interpreter heap is frozen (and thawed here)."""
    import gc

    import torch

    from unscene3d_amd.trainer.trainer import prepare_steady_state
    before = gc.get_freeze_count()
    try:
        out = prepare_steady_state(torch.device("cpu"))
        assert "main_bytes" not in out and out["frozen_objects"] > before
        assert prepare_steady_state(torch.device("cpu"), freeze_heap=False) == {}
    finally:
        gc.unfreeze()
