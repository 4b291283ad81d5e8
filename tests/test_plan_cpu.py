"""CPU: host-side geometry of the tiled encoder (pure integer logic) and level resolutions vs goldens."""
import numpy as np

from collision_handling_in_instantngp_amd import ops, models


def test_product_level_resolutions_match_reference(golden):
    g = golden("G1_level_resolutions")
    i = 0
    while f"case{i}" in g:
        a, b, L = (int(v) for v in g[f"case{i}"])
        assert np.array_equal(models.level_resolutions(a, b, L), g[f"n_ls{i}"])
        i += 1


def test_plan_cfg2_stages_every_level():
    n = [int(v) for v in models.level_resolutions(16, 512, 16)]
    pl = ops.EncodePlan(2 ** 20, n, 2)
    assert pl.Ls == 16 and pl.tile_shift == 5 and pl.lds_bytes <= ops.TILED_LDS_LIMIT
    assert pl.max_items >= 2 ** 20 // pl.chunk + pl.ntiles
    assert pl.vtot == sum((v + 2) ** 2 for v in n)
    # every tile's sub-grid fits the LDS bound used for the launch
    TS = 1 << pl.tile_shift
    for tx in (0, 1, TS // 2, TS - 1):
        tot = 0
        for v in n:
            cx = (tx * v) >> pl.tile_shift
            hx = min((((tx + 1) * v) >> pl.tile_shift) + 1, v + 1)
            tot += (hx - cx + 1) ** 2
        assert tot * 2 * 4 <= pl.lds_bytes


def test_plan_sparse_fine_levels_go_direct():
    n = [int(v) for v in models.level_resolutions(16, 4096, 16)]
    pl = ops.EncodePlan(2 ** 20, n, 2)
    c = ops.TILED_CELLS_PER_PIXEL
    assert 0 < pl.Ls < 16 and all(v * v <= c * 2 ** 20 for v in n[:pl.Ls]) and n[pl.Ls] ** 2 > c * 2 ** 20


def test_plan_small_batches_use_direct_form():
    pl = ops.EncodePlan(1000, [8, 12, 20, 32], 2)
    assert pl.Ls == 0
    assert ops.EncodePlan(2000, [8, 12, 20, 32], 2, "tiled").Ls == 4
    assert ops.EncodePlan(100, [8, 12, 20, 32], 2, "tiled").Ls < 4           # too sparse at the finest level
    assert ops.EncodePlan(2 ** 20, [8, 12, 20, 32], 2, "direct").Ls == 0


def test_work_item_size_follows_mean_tile_population():
    """ops.EncodePlan: chunk = twice the mean pixels per tile as a power of two in [1024, 4096] (an item that is a sliver of
    a split tile still stages every sub-grid), overridable through ops.TILED_CHUNK."""
    n = [int(v) for v in models.level_resolutions(16, 512, 16)]
    assert ops.EncodePlan(2 ** 20, n, 2, "tiled").chunk == 2048         # 1024 tiles of ~1024 px
    assert ops.EncodePlan(2 ** 16, n, 2, "tiled").chunk == 1024         # floor
    assert ops.EncodePlan(2 ** 23, n, 2, "tiled").chunk == 4096         # cap
    p = ops.EncodePlan(2 ** 20, n, 2, "tiled")
    assert p.max_items >= -(-p.P // p.chunk) + p.ntiles
    old = ops.TILED_CHUNK
    try:
        ops.TILED_CHUNK = 512
        assert ops.EncodePlan(2 ** 20, n, 2, "tiled").chunk == 512
    finally:
        ops.TILED_CHUNK = old


def test_fixed_point_grid_is_chosen_by_the_launchers_own_decision():
    """the fixed-point vertex grid dG64 is filled by the level-interleaved backward only, and whether that
    kernel runs is the LAUNCHER's decision (gngf_tiled_interleaved_applies, host logic).  cfg2 stages every level in a 92 KB
    image: interleaved.  The 4096^2 shape (finest staged level N = 1955, tile_shift 6) would need 298 KB: the generic kernels
    run, and the Python side must then hand over a ZEROED fp32 grid, not an uninitialised one next to a dG64."""
    from collision_handling_in_instantngp_amd import _lib
    n2 = [int(v) for v in models.level_resolutions(16, 512, 16)]
    n4 = [int(v) for v in models.level_resolutions(16, 4096, 16)]
    p2, p4 = ops.EncodePlan(2 ** 20, n2, 2), ops.EncodePlan(2 ** 20, n4, 2)
    assert p2.interleaved(backward=True) and p2.interleaved(backward=False)
    assert 0 < p4.Ls < 16 and not p4.interleaved(backward=True)
    assert not ops.EncodePlan(2 ** 20, n2, 4).interleaved(backward=True)          # F = 4: never
    prev = _lib.query("gngf_set_tiled_interleaved", 0)
    try:
        assert not p2.interleaved(backward=True)                                  # the process switch is part of the decision
    finally:
        _lib.query("gngf_set_tiled_interleaved", prev)
    assert p2.interleaved(backward=True)


def test_which_direct_levels_take_the_bucketed_backward():
    """host policy + the library's own sizing (gngf_encode_bwd_bucketed_plan; no GPU needed): the bucketed form above half a
    contribution per table row — a fifth when it also replaces the clear of the levels it writes — hash source, F in {1, 2, 4}"""
    n4 = [int(v) for v in models.level_resolutions(16, 4096, 16)]
    p4 = ops.EncodePlan(2 ** 20, n4, 2)
    assert p4.Ls == 14
    plan = ops.bucketed_plan(2 ** 20, 2, 2 ** 22, 16 - p4.Ls)                    # the 4096^2 shape: one contribution per row
    assert plan is not None and plan[:3] == (12, 1024, 256) and plan[5] == 2 ** 20 * 4 * 2 * 16
    assert plan[3] == 2 * 256 * 1024 and plan[4] == (2 * 1024 + 1) + 2 * 1024 + 32
    assert ops.bucketed_plan(2 ** 20, 4, 2 ** 24, 4) is None                      # the 8192^2 shape: a quarter per row ...
    assert ops.bucketed_plan(2 ** 20, 4, 2 ** 24, 4, fresh=True) is not None      # ... unless the levels are written (no clear)
    assert ops.bucketed_plan(2 ** 20, 8, 2 ** 19, 2) is None                      # F = 8: not served by the library
    assert ops.bucketed_plan(2 ** 14, 2, 2 ** 12, 2) is None                      # too few pixels for five launches
    assert ops.bucketed_plan(2 ** 20, 2, 2 ** 22, 0) is None
    prev = ops.BUCKETED_DIRECT_BWD
    try:
        ops.BUCKETED_DIRECT_BWD = False
        assert ops.bucketed_plan(2 ** 20, 2, 2 ** 22, 2) is None
    finally:
        ops.BUCKETED_DIRECT_BWD = prev
