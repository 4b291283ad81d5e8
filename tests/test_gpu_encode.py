"""GPU parity: direct-form HIP kernels (through the C-ABI) vs the CPU oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import gngf_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def t(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def close(a, b, rtol, atol, msg=""):
    """assert_allclose + a row in the achieved-error report (tests/conftest.py: ParityRecorder)"""
    import inspect
    from conftest import parity_close
    if not msg:
        ctx = inspect.stack()[1].code_context
        msg = (ctx[0].strip() if ctx else "")[:100]
    parity_close(a, b, rtol, atol, msg)


@pytest.fixture(scope="module")
def ops():
    from collision_handling_in_instantngp_amd import ops as o
    return o


@pytest.mark.parametrize("tag,cfg", [("cfg1", (8, 32, 4)), ("cfg2", (16, 512, 16)), ("cfg4", (16, 4096, 16))])
def test_hash_indices_bit_exact_vs_reference_golden(ops, golden, tag, cfg):
    g = golden("G2G3_corners_hash")
    n_ls = t(orc.level_resolutions(*cfg), torch.int32)
    for T in (2 ** 8, 2 ** 19, 1000):
        idx = ops.hash_indices(t(g["x"]), n_ls, T)
        assert idx.dtype == torch.int64
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_hash_T{T}"]), T


@pytest.mark.parametrize("tag", ["small", "mid", "f4k3"])
def test_mrhe_boundary_vs_reference_golden(ops, golden, tag):
    g = golden("G4_encoding")
    tables = t(g[f"{tag}_tables"]).requires_grad_()
    out = ops.MrheFunction.apply(tables, t(g[f"{tag}_hash_idx"]), None, 0)
    assert np.array_equal(out.detach().cpu().numpy(), g[f"{tag}_hash_out"])          # pure gather: bit-exact
    out.backward(t(g[f"{tag}_hash_gout"]))
    close(tables.grad, g[f"{tag}_hash_dtables"], 1e-5, 1e-6)
    for vname, flag in (("softmax", True), ("raw", None), ("norm", False)):
        tables.grad = None
        probs = t(g[f"{tag}_gngf_probs"]).requires_grad_()
        out = ops.MrheFunction.apply(tables, t(g[f"{tag}_gngf_idx"]), probs, ops.BLEND_CODES[flag])
        close(out, g[f"{tag}_gngf_{vname}_out"], 1e-5, 1e-9)
        out.backward(t(g[f"{tag}_gngf_gout"]))
        close(tables.grad, g[f"{tag}_gngf_{vname}_dtables"], 1e-4, 1e-6)
        close(probs.grad, g[f"{tag}_gngf_{vname}_dprobs"], 1e-4, 2e-7 if flag is not False else 2e-5)


@pytest.mark.parametrize("tag", ["cfg1", "cfg2", "f4"])
def test_bilinear_vs_reference_golden(ops, golden, tag):
    g = golden("G5_bilinear")
    a, b, L, F = (int(v) for v in g[f"{tag}_cfg"])
    n_ls = t(orc.level_resolutions(a, b, L), torch.int32)
    feats = t(g[f"{tag}_feats"]).requires_grad_()
    out = ops.BilinearFunction.apply(t(g[f"{tag}_x"]), n_ls, feats)
    close(out, g[f"{tag}_out"], 1e-6, 1e-6)
    out.backward(t(g[f"{tag}_gout"]))
    close(feats.grad, g[f"{tag}_dfeats"], 1e-6, 1e-7)


def _coords(P, rng):
    x = rng.random((P, 2), dtype=np.float32)
    edge = np.array([[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [1 / 32, 31 / 32], [0.99999994, 1e-8]], np.float32)
    x[: len(edge)] = edge
    return x


@pytest.mark.parametrize("cfg", [(8, 32, 4, 2, 256), (16, 512, 16, 2, 2 ** 14), (16, 128, 5, 4, 1000), (4, 64, 3, 1, 64),
                                 (16, 256, 8, 8, 4096)])
def test_fused_encode_hash_vs_oracle(ops, cfg):
    n_min, n_max, L, F, T = cfg
    rng = np.random.default_rng(1)
    P = 3001
    x = _coords(P, rng)
    tables = (rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4
    n_ls = orc.level_resolutions(n_min, n_max, L)
    _, grid = orc.scale_to_grid(x, n_ls)
    idx = orc.spatial_hash(grid.astype(np.int32), T)
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx))
    tt = t(tables).requires_grad_()
    enc = ops.EncodeDirectFunction.apply(t(x), t(n_ls, torch.int32), tt, None, None, 0)
    close(enc, want, 1e-6, 1e-9)
    g = rng.standard_normal(want.shape).astype(np.float32)
    enc.backward(t(g))
    dfe = orc.bilinear_backward(x, n_ls, g, F)
    dt, _ = orc.encoding_backward(tables, idx, None, True, dfe)
    close(tt.grad, dt, 1e-4, 1e-6)


@pytest.mark.parametrize("cfg", [(8, 32, 4, 2, 256, 4), (16, 128, 8, 2, 4096, 4), (8, 64, 5, 4, 512, 1), (8, 40, 3, 2, 300, 7)])
def test_fused_encode_vertex_table_vs_oracle(ops, cfg):
    n_min, n_max, L, F, T, K = cfg
    rng = np.random.default_rng(2)
    P = 2049
    x = _coords(P, rng)
    tables = (rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4
    n_ls = orc.level_resolutions(n_min, n_max, L)
    vstride = n_max + 2
    NV = vstride * vstride
    vidx = rng.integers(0, T, (NV, K)).astype(np.int32)
    vw = rng.random((NV, K), dtype=np.float32)
    _, grid = orc.scale_to_grid(x, n_ls)
    gi = grid.astype(np.int64)                                   # (P,2,L,4)
    vid = gi[:, 1] * vstride + gi[:, 0]                          # (P,L,4)
    idx_inst = vidx[vid].astype(np.int64)                        # (P,L,4,K)
    w_inst = vw[vid]
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx_inst, w_inst, None))
    tt, tw = t(tables).requires_grad_(), t(vw).requires_grad_()
    enc = ops.EncodeDirectFunction.apply(t(x), t(n_ls, torch.int32), tt, t(vidx), tw, vstride)
    close(enc, want, 2e-6, 1e-9)
    g = rng.standard_normal(want.shape).astype(np.float32)
    enc.backward(t(g))
    dfe = orc.bilinear_backward(x, n_ls, g, F)
    dt, dw_inst = orc.encoding_backward(tables, idx_inst, w_inst, None, dfe)
    close(tt.grad, dt, 1e-4, 1e-6)
    dvw = np.zeros((NV, K), np.float64)
    np.add.at(dvw, vid.reshape(-1), dw_inst.reshape(-1, K).astype(np.float64))
    close(tw.grad, dvw, 1e-4, 1e-7)


def test_empty_batch_and_bad_args(ops):
    n_ls = t(orc.level_resolutions(8, 32, 4), torch.int32)
    tables = torch.zeros((4, 256, 2), device=DEV)
    enc = ops.EncodeDirectFunction.apply(torch.zeros((0, 2), device=DEV), n_ls, tables, None, None, 0)
    assert enc.shape == (0, 8)
    with pytest.raises(RuntimeError):      # F = 3 is not a supported feature width
        ops.EncodeDirectFunction.apply(torch.zeros((4, 2), device=DEV), n_ls, torch.zeros((4, 256, 3), device=DEV), None, None, 0)
    with pytest.raises(Exception):         # CPU tensors never fall back
        ops.hash_indices(torch.zeros((4, 2)), n_ls.cpu(), 256)


# ------------------------------------------------------------------------------------------------ tiled form
def _oracle_hash(x, n_ls, tables, g):
    L, T, F = tables.shape
    _, grid = orc.scale_to_grid(x, n_ls)
    idx = orc.spatial_hash(grid.astype(np.int32), T)
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx))
    dt, _ = orc.encoding_backward(tables, idx, None, True, orc.bilinear_backward(x, n_ls, g, F))
    return want, dt


@pytest.mark.parametrize("cfg", [(8, 32, 4, 2, 256, 3001), (16, 512, 16, 2, 2 ** 14, 40000), (16, 128, 5, 4, 1000, 20000),
                                 (4, 64, 3, 1, 64, 5000), (16, 512, 16, 2, 2 ** 12, 3001), (16, 256, 8, 8, 4096, 70001)])
def test_tiled_encode_hash_vs_oracle(ops, cfg):
    """tiled form (all staged, or staged + direct levels mixed when the batch is sparse at the fine levels)"""
    n_min, n_max, L, F, T, P = cfg
    rng = np.random.default_rng(P)
    x = _coords(P, rng)
    x[100:200] = x[100]                                  # many pixels in one cell: LDS atomics on one address
    tables = (rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4
    n_ls = orc.level_resolutions(n_min, n_max, L)
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    want, dt = _oracle_hash(x, n_ls, tables, g)
    tt = t(tables).requires_grad_()
    enc = ops.encode_apply(t(x), t(n_ls, torch.int32), [int(n) for n in n_ls], tt, None, None, 0, path="tiled")
    close(enc, want, 1e-6, 1e-9)
    enc.backward(t(g))
    close(tt.grad, dt, 2e-4, 2e-6 * max(1.0, float(np.abs(dt).max())))   # fp32 sums of ~P/cells terms, any order
    # and the two forms agree bit-for-bit in the forward direction
    enc_d = ops.encode_apply(t(x), t(n_ls, torch.int32), [int(n) for n in n_ls], t(tables), None, None, 0, path="direct")
    assert torch.equal(enc_d, enc.detach())


@pytest.mark.parametrize("cfg", [(8, 32, 4, 2, 256, 4, 5000), (16, 128, 8, 2, 4096, 4, 30000), (8, 64, 5, 4, 512, 1, 9000),
                                 (16, 512, 16, 2, 2 ** 13, 4, 2500)])
def test_tiled_encode_vertex_table_vs_oracle(ops, cfg):
    n_min, n_max, L, F, T, K, P = cfg
    rng = np.random.default_rng(P + 1)
    x = _coords(P, rng)
    tables = (rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4
    n_ls = orc.level_resolutions(n_min, n_max, L)
    vstride = n_max + 2
    NV = vstride * vstride
    vidx = rng.integers(0, T, (NV, K)).astype(np.int32)
    vw = rng.random((NV, K), dtype=np.float32)
    _, grid = orc.scale_to_grid(x, n_ls)
    gi = grid.astype(np.int64)
    vid = gi[:, 1] * vstride + gi[:, 0]
    idx_inst, w_inst = vidx[vid].astype(np.int64), vw[vid]
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx_inst, w_inst, None))
    tt, tw = t(tables).requires_grad_(), t(vw).requires_grad_()
    enc = ops.encode_apply(t(x), t(n_ls, torch.int32), [int(n) for n in n_ls], tt, t(vidx), tw, vstride, path="tiled")
    close(enc, want, 2e-6, 1e-9)
    g = rng.standard_normal(want.shape).astype(np.float32)
    enc.backward(t(g))
    dt, dw_inst = orc.encoding_backward(tables, idx_inst, w_inst, None, orc.bilinear_backward(x, n_ls, g, F))
    close(tt.grad, dt, 2e-4, 2e-6 * max(1.0, float(np.abs(dt).max())))
    dvw = np.zeros((NV, K), np.float64)
    np.add.at(dvw, vid.reshape(-1), dw_inst.reshape(-1, K).astype(np.float64))
    close(tw.grad, dvw, 2e-4, 2e-6 * max(1e-1, float(np.abs(dvw).max())))


def test_binning_is_a_permutation_grouped_by_tile(ops):
    rng = np.random.default_rng(9)
    P = 50000
    x = _coords(P, rng)
    plan = ops.EncodePlan(P, [16, 64, 512], 2, "tiled")
    ws = ops.TiledWorkspace(plan, t(x))
    srt = ws.sorted.cpu().numpy()
    ids = srt[:, 2].copy().view(np.int32)
    assert np.array_equal(np.sort(ids), np.arange(P))                      # every pixel exactly once
    assert np.array_equal(srt[:, :2], x[ids])                              # coordinates travel with their index
    TS = 1 << plan.tile_shift
    tile = np.minimum((srt[:, 1] * TS).astype(np.int64), TS - 1) * TS + np.minimum((srt[:, 0] * TS).astype(np.int64), TS - 1)
    assert np.all(np.diff(tile) >= 0)                                      # sorted by tile
    n_items = int(ws.n_items[0].item())                                     # ([1..3]: work counters of the persistent kernels)
    items = ws.items.cpu().numpy()[:n_items]
    assert items[:, 1].sum() == P and items[:, 1].max() <= plan.chunk and items[:, 1].min() > 0
    for s_, c_, t_, _ in items[:: max(1, n_items // 50)]:
        assert np.all(tile[s_:s_ + c_] == t_)
    off = ws.tile_off.cpu().numpy()
    assert off[-1] == P and np.array_equal(np.bincount(tile, minlength=TS * TS), np.diff(off))


@pytest.mark.parametrize("mode", ["hash", "vt"])
@pytest.mark.parametrize("path", ["tiled", "direct"])
def test_fp16_tables_vs_oracle_on_rounded_tables(ops, mode, path):
    """BASELINE config 5 flavour: fp16 table storage, F = 4.  No reference counterpart: the oracle is the fp32 restatement
    on the fp16-ROUNDED tables; interpolation and gradient accumulation are fp32, so forward agrees to fp32 round-off and
    the gradient (returned in fp16) to fp16 rounding (tolerance 1e-3, stated in SURVEY §8d)."""
    rng = np.random.default_rng(77)
    n_min, n_max, L, F, T, K, P = 16, 256, 8, 4, 2 ** 12, 3, 30000
    x = _coords(P, rng)
    tables16 = ((rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-2).astype(np.float16)
    tables = tables16.astype(np.float32)
    n_ls = orc.level_resolutions(n_min, n_max, L)
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    _, grid = orc.scale_to_grid(x, n_ls)
    if mode == "hash":
        idx, w, vi, vw, vs = orc.spatial_hash(grid.astype(np.int32), T), None, None, None, 0
    else:
        vs = n_max + 2
        vi_np = rng.integers(0, T, (vs * vs, K)).astype(np.int32)
        vw_np = rng.random((vs * vs, K), dtype=np.float32)
        gi = grid.astype(np.int64)
        vid = gi[:, 1] * vs + gi[:, 0]
        idx, w = vi_np[vid].astype(np.int64), vw_np[vid]
        vi, vw = t(vi_np), t(vw_np)
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx, w, None))
    dt, _ = orc.encoding_backward(tables, idx, w, None, orc.bilinear_backward(x, n_ls, g, F))
    tt = t(tables16).requires_grad_()
    enc = ops.encode_apply(t(x), t(n_ls, torch.int32), [int(n) for n in n_ls], tt, vi, vw, vs, path=path)
    assert enc.dtype == torch.float32
    close(enc, want, 2e-6, 1e-9)
    enc.backward(t(g))
    assert tt.grad.dtype == torch.float16
    close(tt.grad.float(), dt, 1e-3, 1e-3 * float(np.abs(dt).max()))


def _degenerate_tables(kind, NV, K, T, rng):
    if kind == "one_slot":                       # every (vertex, k) entry -> slot 3: one run spanning every wave and workgroup
        return np.full((NV, K), 3 % T, np.int32)
    if kind == "few_slots":                      # a fresh HPD: ~46 slots in a few hundred runs
        pool = rng.choice(T, size=min(46, T), replace=False)
        return pool[rng.integers(0, len(pool), (NV, K))].astype(np.int32)
    if kind == "k_equal":                        # all K entries of a vertex on the same slot, slots vary slowly with the vertex
        return np.repeat((np.arange(NV) // 97 % T)[:, None], K, 1).astype(np.int32)
    if kind == "two_slots_alternating":          # (5, 9 | 5 ...) patterns across the level-group boundaries of the visiting order
        return np.where((np.arange(NV)[:, None] + np.arange(K)[None]) % 2 == 0, 5 % T, 9 % T).astype(np.int32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["one_slot", "few_slots", "k_equal", "two_slots_alternating"])
@pytest.mark.parametrize("cfg", [(2, 24, 6, 2, 16, 1, 6000), (4, 40, 5, 2, 64, 4, 9000), (16, 512, 16, 2, 2 ** 12, 4, 30000),
                                 (3, 11, 4, 4, 8, 3, 2000)])
def test_sorted_vertex_backward_degenerate_slot_distributions_vs_oracle(ops, cfg, kind):
    """vertex_bwd_sorted_kernel (runs merged by a wave-level segmented scan and chained across the 8 waves of a workgroup):
    slot distributions with very long runs, runs that span several workgroups, tiny level groups with K = 1 and the same
    slot on both sides of a level-group boundary — table gradient AND d vert_w against the per-instance oracle."""
    n_min, n_max, L, F, T, K, P = cfg
    rng = np.random.default_rng(hash((cfg, kind)) % (2 ** 31))
    x = _coords(P, rng)
    tables = (rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-1
    n_ls = orc.level_resolutions(n_min, n_max, L)
    vstride = n_max + 2
    NV = vstride * vstride
    vidx = _degenerate_tables(kind, NV, K, T, rng)
    vw = rng.random((NV, K), dtype=np.float32)
    _, grid = orc.scale_to_grid(x, n_ls)
    gi = grid.astype(np.int64)
    vid = gi[:, 1] * vstride + gi[:, 0]
    idx_inst, w_inst = vidx[vid].astype(np.int64), vw[vid]
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    dt, dw_inst = orc.encoding_backward(tables, idx_inst, w_inst, None, orc.bilinear_backward(x, n_ls, g, F))
    dvw = np.zeros((NV, K), np.float64)
    np.add.at(dvw, vid.reshape(-1), dw_inst.reshape(-1, K).astype(np.float64))
    n_host = [int(n) for n in n_ls]
    for with_dw in (True, False):
        tt = t(tables).requires_grad_()
        tw = t(vw).requires_grad_(with_dw)
        enc = ops.encode_apply(t(x), t(n_ls, torch.int32), n_host, tt, t(vidx), tw, vstride, path="tiled")
        plan = ops.EncodePlan(P, n_host, F, "tiled")
        assert plan.Ls > 0
        enc.backward(t(g))
        # sums of up to P*4 terms per row in fp32 (any order): tolerance relative to the largest row
        close(tt.grad, dt, 2e-4, 2e-5 * float(np.abs(dt).max()), f"sorted vertex bwd [{kind}] table gradient, dw={with_dw}")
        if with_dw:
            close(tw.grad, dvw, 2e-4, 2e-5 * float(np.abs(dvw).max()), f"sorted vertex bwd [{kind}] d vert_w")


@pytest.mark.parametrize("K", [0, 4])
def test_mrhe_boundary_fp16_tables_cfg5_feature_width_vs_oracle(ops, K):
    """MultiResHashEncoding.forward at the module boundary on fp16 tables (F = 4, BASELINE config 5): fp32 arithmetic on the
    stored values — oracle = fp32 restatement on the fp16-rounded tables; gradient returned in fp16 (tolerance 1e-3)."""
    rng = np.random.default_rng(50 + K)
    L, T, F, P = 16, 2 ** 14, 4, 3000
    tables16 = ((rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-2).astype(np.float16)
    tables = tables16.astype(np.float32)
    if K == 0:
        idx, probs = rng.integers(0, T, (P, L, 4)), None
    else:
        idx, probs = rng.integers(0, T, (P, L, 4, K)), (rng.random((P, L, 4, K), dtype=np.float32) * 0.01 + 1e-3)
    want = orc.encoding_forward(tables, idx, probs, True)
    gout = rng.standard_normal(want.shape).astype(np.float32)
    dt, dp = orc.encoding_backward(tables, idx, probs, True, gout)
    tt = t(tables16).requires_grad_()
    tp = t(probs).requires_grad_() if K else None
    out = ops.MrheFunction.apply(tt, t(idx), tp, 0)
    assert out.dtype == torch.float32
    close(out, want, 2e-6, 1e-9)
    out.backward(t(gout))
    assert tt.grad.dtype == torch.float16
    close(tt.grad.float(), dt, 1e-3, 1e-3 * float(np.abs(dt).max()))
    if K:
        close(tp.grad, dp, 1e-4, 1e-6)


@pytest.mark.parametrize("P,n_max,L,coords", [(2 ** 17 + 77, 512, 16, "unit"), (40001, 256, 12, "strip"), (2 ** 15, 512, 16, "outside"),
                                             (5000, 64, 3, "unit")])
def test_level_interleaved_pixel_stage_equals_back_to_back_layout(ops, P, n_max, L, coords):
    """The level-interleaved LDS images (csrc/encode_tiled.hip: tiled_fwd_il_kernel / tiled_bwd_il_kernel; F = 2, <= 16 staged
    levels) against the back-to-back layout on the same binned pixels: the forward rows and the vertex-grid gradient are
    BIT-IDENTICAL (same fp32 expressions; the backward sums exact 64-bit fixed-point terms, and integer adds commute — the
    private accumulator copies of the coarse levels included).  Ragged sizes, a 12-level table, pixels confined to a strip (many
    empty tiles, many pixels per cell) and coordinates outside [0,1]^2 (the global fall-back path)."""
    from collision_handling_in_instantngp_amd import _lib
    rng = np.random.default_rng(P)
    xy = rng.random((P, 2), dtype=np.float32)
    if coords == "strip":
        xy[:, 1] = 0.31 + 0.02 * xy[:, 1]
    if coords == "outside":
        xy[::7] = xy[::7] * 1.5 - 0.25                                  # a seventh of the pixels partly outside the unit square
    xy_t = t(xy)
    n_ls = orc.level_resolutions(16, n_max, L)
    n_host = [int(n) for n in n_ls]
    n_t = t(n_ls, torch.int32)
    Fd = 2
    tables = t((rng.random((L, 4096, Fd), dtype=np.float32) - 0.5) * 2e-1)
    genc = t((rng.standard_normal((P, L * Fd)) * np.exp(rng.standard_normal((P, 1)) * 3)).astype(np.float32))      # wide dynamic range
    plan = ops.EncodePlan(P, n_host, Fd, "tiled")
    assert plan.Ls >= min(L, 12)              # (levels finer than 4 P cells are left to the direct form: not this test's subject)
    res = {}
    prev = _lib.query("gngf_set_tiled_interleaved", 1)
    try:
        # ONE binning for both variants: the order of a tile's pixels comes out of LDS atomics, and a tile with more than `chunk`
        # pixels is cut into work items along that order — two binnings give different (equally valid) partial sums
        G = torch.empty((plan.vtot, Fd), dtype=torch.float32, device=DEV)
        ws = ops.TiledWorkspace(plan, xy_t, vertex=(tables, None, None, n_t, 0, G))
        for variant in (0, 1):
            _lib.query("gngf_set_tiled_interleaved", variant)
            enc = torch.full((P, L * Fd), float("nan"), dtype=torch.float32, device=DEV)
            ops.call("gngf_encode_tiled_fwd", ops.ptr(ws.sorted), ops.ptr(ws.items), ops.ptr(ws.n_items), plan.max_items, ops.ptr(n_t),
                     plan.n_ls_c, ops.ptr(G), ops.ptr(enc), L, plan.Ls, Fd, plan.tile_shift, plan.lds_bytes, ops.stream_ptr())
            dG = torch.zeros((plan.vtot, Fd), dtype=torch.float32, device=DEV)
            ops._pixel_bwd(plan, ws, n_t, genc, dG, L, Fd, None, None)
            torch.cuda.synchronize()
            res[variant] = (enc[:, :plan.Ls * Fd].clone(), dG.clone())
    finally:
        _lib.query("gngf_set_tiled_interleaved", prev)
    assert bool(torch.isfinite(res[0][0]).all())
    assert torch.equal(res[0][0], res[1][0]), "forward rows differ between the LDS layouts"
    if coords == "outside":        # out-of-sub-grid pixels add to dG with global FLOAT atomics in both kernels: order-dependent rounding
        close(res[1][1], res[0][1].cpu().numpy(), 1e-5, 1e-6 * float(res[0][1].abs().max()), "interleaved vs back-to-back dG (with out-of-domain pixels)")
    else:
        assert torch.equal(res[0][1], res[1][1]), "vertex-grid gradients differ between the LDS layouts"
    # ... and against the oracle (forward rows of a sample of pixels)
    sel = rng.choice(P, size=min(P, 2000), replace=False)
    want = orc.gngf_forward(xy[sel], n_ls, tables.cpu().numpy(), [np.zeros((64, L * Fd), np.float32), np.zeros((64, 64), np.float32), np.zeros((3, 64), np.float32)],
                            [np.zeros(64, np.float32), np.zeros(64, np.float32), np.zeros(3, np.float32)], hash_mode=True)["enc"]
    if coords != "outside":
        close(res[1][0][t(sel)], want[:, :plan.Ls * Fd], 1e-5, 1e-7, "interleaved forward rows vs oracle")


@pytest.mark.parametrize("P,coords", [(2 ** 17 + 77, "unit"), (60001, "strip"), (2 ** 15, "outside"), (400, "unit")])
def test_fixed_point_vertex_grid_is_exact_and_order_free(ops, P, coords):
    """dG64 path of the pixel-stage backward (F = 2, <= 16 staged levels, a bound on |genc| from its producer): the work items
    add their exact 64-bit fixed-point sums into ONE fixed-point vertex grid with global integer atomics.  (1) Two DIFFERENT
    binnings of the same pixels (the order inside a tile, hence the cut into work items, comes out of LDS atomics) give the
    BIT-IDENTICAL vertex-grid gradient — integer sums do not depend on order or grouping; (2) it equals the partial-image +
    gather path to fp32 rounding; (3) the table gradient of the hash source formed straight from the fixed-point grid equals
    vertex_grid_bwd on the fp32 grid; (4) a NaN bound poisons everything."""
    rng = np.random.default_rng(P + 1)
    xy = rng.random((P, 2), dtype=np.float32)
    if coords == "strip":
        xy[:, 1] = 0.31 + 0.02 * xy[:, 1]
    if coords == "outside":
        xy[::7] = xy[::7] * 1.5 - 0.25
    xy_t = t(xy)
    L, Fd, T = 16, 2, 4096
    n_ls = orc.level_resolutions(16, 256, L)
    if P < 2 ** 14:            # a few hundred pixels on four coarse levels: fewer than 2^10 terms per vertex (the scale's floor)
        L = 4
        n_ls = orc.level_resolutions(8, 24, L)
    n_host = [int(n) for n in n_ls]
    n_t = t(n_ls, torch.int32)
    tables = t((rng.random((L, T, Fd), dtype=np.float32) - 0.5) * 2e-1)
    genc = t((rng.standard_normal((P, L * Fd)) * np.exp(rng.standard_normal((P, 1)) * 3)).astype(np.float32))
    if P < 2 ** 14:
        genc[::3] = genc.abs().max()           # many terms at the bound itself
    am = genc.abs().max().reshape(1)
    plan = ops.EncodePlan(P, n_host, Fd, "tiled")
    assert plan.Ls == L
    grids = []
    for rep in range(2):
        G = torch.empty((plan.vtot, Fd), dtype=torch.float32, device=DEV)
        ws = ops.TiledWorkspace(plan, xy_t, vertex=(tables, None, None, n_t, 0, G))      # a fresh binning each time
        dG64 = torch.zeros((plan.vtot * Fd + 2,), dtype=torch.int64, device=DEV)
        dG = torch.full((plan.vtot, Fd), float("nan"), dtype=torch.float32, device=DEV)
        ops._pixel_bwd(plan, ws, n_t, genc, dG, L, Fd, (am, 1, 0), None, None, dG64)
        torch.cuda.synchronize()
        grids.append((dG.clone(), dG64[:-2].clone()))
    assert bool(torch.isfinite(grids[0][0]).all())
    assert torch.equal(grids[0][1], grids[1][1]), "the fixed-point sums depend on the binning"
    assert torch.equal(grids[0][0], grids[1][0])
    # (2) the partial-image + gather path on the last workspace
    dG_ref = torch.zeros((plan.vtot, Fd), dtype=torch.float32, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, dG_ref, L, Fd, (am, 1, 0), None)
    close(grids[0][0], dG_ref.cpu().numpy(), 2e-6, 1e-6 * float(dG_ref.abs().max()), "fixed-point vertex grid vs partial images + gather")
    # (3) hash source: table gradient straight from the fixed-point grid
    dt_a = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    dG64 = torch.zeros((plan.vtot * Fd + 2,), dtype=torch.int64, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, torch.empty_like(dG_ref), L, Fd, (am, 1, 0), None, (dt_a, T), dG64)
    dt_b = torch.zeros_like(dt_a)
    ops._vertex_bwd(plan, tables, None, None, n_t, 0, grids[0][0], dt_b, None)
    close(dt_a, dt_b.cpu().numpy(), 1e-5, 1e-6 * float(dt_b.abs().max()), "hash table gradient from the fixed-point grid")
    # (3a) round 5, hash source: NO grid at all — the store pass of the interleaved backward adds each item's exact sums (rounded to
    # fp32 once) to the table-gradient rows hash(gx, gy) itself (tiled_bwd_il_kernel<., HDT>); incl. the pixels outside the staged
    # sub-grids ("outside"), which go to their rows one float atomic per term.  Against the table gradient formed from the exact
    # per-vertex sums: the two differ only in how many fp32 roundings a row's partial sums meet (one per item that reaches the
    # vertex instead of one per vertex) — a few ulp of the row's absolute mass
    dt_h = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    trace = []
    prev_trace, ops.PIXEL_BWD_TRACE = ops.PIXEL_BWD_TRACE, trace
    try:
        ops._pixel_bwd(plan, ws, n_t, genc, None, L, Fd, (am, 1, 0), None, (dt_h, T), None)
    finally:
        ops.PIXEL_BWD_TRACE = prev_trace
    assert trace and trace[0]["direct_hash"] and trace[0]["interleaved"], trace
    gabs = torch.zeros((plan.vtot, Fd), dtype=torch.float32, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc.abs(), gabs, L, Fd, (am, 1, 0), None)                 # per-vertex absolute mass (c >= 0)
    mass = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    ops._vertex_bwd(plan, tables, None, None, n_t, 0, gabs, mass, None)
    err = (dt_h.double() - dt_a.double()).abs()
    assert bool(((mass == 0) <= (dt_h == 0)).all()), "a row no pixel reaches received something"
    worst = float((err[mass > 0] / mass[mass > 0].double()).max())
    print(f"[direct hash scatter {P} {coords}] worst |err| / row mass vs the exact per-vertex sums: {worst:.2e}")
    assert worst <= 2e-6, worst
    close(dt_h, dt_a.cpu().numpy(), 1e-5, 2e-6 * float(dt_a.abs().max()), "hash table gradient added by the store pass (no vertex grid) vs from the fixed-point grid")
    # ... without a bound on |genc| (a decoder that hands none over): per-item scales — float adds need no common one
    dt_u = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, None, L, Fd, None, None, (dt_u, T), None)
    err_u = (dt_u.double() - dt_a.double()).abs()
    assert float((err_u[mass > 0] / mass[mass > 0].double()).max()) <= 2e-6
    # ... and the same through the GENERIC pixel-stage kernel (tiled_bwd_kernel<F, HDT>: what the 4096^2 / 8192^2 shapes run), with
    # and without a bound on |genc|
    from collision_handling_in_instantngp_amd import _lib
    prev_il = _lib.query("gngf_set_tiled_interleaved", 0)
    try:
        for bound in ((am, 1, 0), None):
            dt_g = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
            trace = []
            prev_trace, ops.PIXEL_BWD_TRACE = ops.PIXEL_BWD_TRACE, trace
            try:
                ops._pixel_bwd(plan, ws, n_t, genc, None, L, Fd, bound, None, (dt_g, T), None)
            finally:
                ops.PIXEL_BWD_TRACE = prev_trace
            assert trace and trace[0]["direct_hash"] and not trace[0]["interleaved"], trace
            errg = (dt_g.double() - dt_a.double()).abs()
            assert bool(((mass == 0) <= (dt_g == 0)).all())
            worst_g = float((errg[mass > 0] / mass[mass > 0].double()).max())
            assert worst_g <= 2e-6, (worst_g, bound is not None)
    finally:
        _lib.query("gngf_set_tiled_interleaved", prev_il)
    dt_n = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, None, L, Fd, (torch.full((1,), float("nan"), device=DEV), 1, 0), None, (dt_n, T), None)
    hit = dt_a != 0
    assert bool(hit.any())
    if coords == "outside":      # (rows reached ONLY by pixels outside every staged sub-grid — never for in-domain coordinates — are not poisoned)
        assert float(torch.isnan(dt_n[hit]).float().mean()) > 0.9
    else:
        assert bool(torch.isnan(dt_n[hit]).all()), "a NaN bound poisons every row the batch reaches"
    # (3b) vertex-table source in slot order: the vertex stage reading the fixed-point grid itself (no fp32 grid at all)
    vstride = max(n_host) + 2
    NV, K = vstride * vstride, 3
    vi = t(rng.integers(0, T, size=(NV, K)).astype(np.int32), torch.int32)
    vw = t(rng.random((NV, K), dtype=np.float32))
    order = ops.slot_order(vi, n_host, vstride)
    dG64 = torch.zeros((plan.vtot * Fd + 2,), dtype=torch.int64, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, None, L, Fd, (am, 1, 0), None, None, dG64)          # dG = None: no conversion launch
    assert torch.equal(dG64[:-2], grids[0][1])
    dt_c, dt_d = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV), torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    ops._vertex_bwd(plan, tables, vi, vw, n_t, vstride, None, dt_c, None, order, dG64)
    ops._vertex_bwd(plan, tables, vi, vw, n_t, vstride, grids[0][0], dt_d, None, order)
    close(dt_c, dt_d.cpu().numpy(), 2e-6, 1e-6 * float(dt_d.abs().max()), "slot-ordered vertex backward on the fixed-point grid")
    # (4) poison
    dG64 = torch.zeros((plan.vtot * Fd + 2,), dtype=torch.int64, device=DEV)
    dG = torch.zeros((plan.vtot, Fd), dtype=torch.float32, device=DEV)
    ops._pixel_bwd(plan, ws, n_t, genc, dG, L, Fd, (torch.full((1,), float("nan"), device=DEV), 1, 0), None, None, dG64)
    assert bool(torch.isnan(dG).all())
    dt_e = torch.zeros((L, T, Fd), dtype=torch.float32, device=DEV)
    ops._vertex_bwd(plan, tables, vi, vw, n_t, vstride, None, dt_e, None, order, dG64)
    touched = dt_d != 0
    assert bool(torch.isnan(dt_e[touched]).all()) and bool(touched.any())
