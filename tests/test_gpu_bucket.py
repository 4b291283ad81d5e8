"""GPU: the bucketed backward of the direct levels (csrc/encode_bucket.hip — counting sort of the contributions by table slice, one
workgroup per slice summing in a 64-bit fixed-point LDS image) against the C oracle's sum of the reference's fp32 terms in double
precision, against the atomics kernel it replaces, and its own properties: bitwise reproducible, both store modes (add to / write
the levels), ragged shapes (T no power of two, P no multiple of a pixel block, a level sub-range, images from 1 KiB to 128 KiB),
non-finite terms propagate to the rows they belong to and nowhere else."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import c_oracle, gngf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(P, n_min, n_max, L, T, F, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((P, 2), dtype=np.float32)
    x[:7] = np.array([[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [1 / 32, 31 / 32], [0.99999994, 1e-8]], np.float32)
    x[7:2007] = x[7]                                   # 2000 pixels on one cell: long chains on a few rows
    n_ls = orc.level_resolutions(n_min, n_max, L).astype(np.int32)
    g = (rng.standard_normal((P, L * F)) * np.exp(rng.uniform(-12, 2, (P, 1)))).astype(np.float32)      # 6 decades of magnitudes
    return x, n_ls, g


def _bucketed(ops, _lib, x, n_ls, g, L, T, F, l0, l1, image, accumulate, into=None, order=None):
    P = x.shape[0]
    plan = (ctypes.c_int64 * 6)()
    assert _lib.query("gngf_encode_bwd_bucketed_plan", P, F, T, l1 - l0, image, plan) == 1
    dt = torch.zeros((L, T, F), device=DEV) if into is None else into
    matrix = torch.empty((plan[3],), dtype=torch.int32, device=DEV)
    base = torch.empty((plan[4],), dtype=torch.int32, device=DEV)
    items = torch.empty((plan[5],), dtype=torch.uint8, device=DEV)
    _lib.call("gngf_encode_bwd_bucketed", _lib.ptr(x), _lib.ptr(n_ls), _lib.ptr(g), _lib.ptr(dt), P, L, F, T, l0, l1, image, accumulate,
              _lib.ptr(matrix), _lib.ptr(base), _lib.ptr(items), _lib.ptr(order), _lib.stream_ptr())
    torch.cuda.synchronize()
    return dt, tuple(plan)


@pytest.mark.parametrize("F,T,image", [(2, 2 ** 18, 65536), (4, 2 ** 20, 65536), (1, 1000, 1024)])
def test_bucketed_backward_walking_the_pixels_in_tile_order_is_bit_identical(F, T, image):
    """gngf_encode_bwd_bucketed(..., pixel_order): the batch walked in the tiled form's binned order (records {x, y, bits(original
    index), 0}: the `sorted` output of the binning for the same xy) instead of in the caller's order — the sums are integer sums of
    the same terms, so the table gradient is the same bit for bit; both store modes; a ragged batch (P no multiple of the block)."""
    from collision_handling_in_instantngp_amd import ops, _lib
    P, L, l0, l1 = 2 ** 17 + 1001, 6, 3, 6
    x, n_ls, g = _case(P, 16, 4096, L, T, F, seed=F)
    xt, nt, gt = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
    n_host = [int(n) for n in n_ls]
    plan = ops.EncodePlan(P, n_host, F, "tiled")
    assert plan.Ls > 0
    ws = ops.TiledWorkspace(plan, xt)                           # binning only: ws.sorted = the pixels in tile order
    torch.cuda.synchronize()
    idx = ws.sorted[:, 2].contiguous().view(torch.int32).long()
    assert torch.equal(torch.sort(idx).values, torch.arange(P, device=DEV)), "the binned records are a permutation of the batch"
    assert torch.equal(ws.sorted[:, :2], xt[idx])
    for accumulate in (0, 1):
        a, _ = _bucketed(ops, _lib, xt, nt, gt, L, T, F, l0, l1, image, accumulate)
        b, _ = _bucketed(ops, _lib, xt, nt, gt, L, T, F, l0, l1, image, accumulate, order=ws.sorted)
        assert float(a[l0:l1].abs().max()) > 0
        assert torch.equal(a, b), f"accumulate={accumulate}: the tile-ordered walk changed the table gradient"


@pytest.mark.parametrize("shape", [
    # P, n_min, n_max, L, T, F, l0, l1, image bytes
    (70001, 16, 512, 6, 1000, 1, 0, 6, 1024),               # T no power of two, 8 buckets of 128 slots, the last one partial
    (2 ** 17 + 13, 16, 2048, 5, 2 ** 16, 2, 1, 5, 2048),    # a level sub-range, 512 buckets per level
    (2 ** 16, 64, 4096, 3, 2 ** 20, 4, 0, 3, 65536),        # the cfg5 item layout (values and slots in two arrays)
    (100000, 16, 4096, 4, 2 ** 18, 2, 2, 4, 131072),        # the largest image
    (5000, 16, 256, 2, 300, 2, 0, 2, 65536),                # one bucket per level, one pixel block
])
def test_bucketed_direct_backward_vs_oracle_and_atomics(shape):
    from collision_handling_in_instantngp_amd import _lib, ops
    P, n_min, n_max, L, T, F, l0, l1, image = shape
    x, n_ls, g = _case(P, n_min, n_max, L, T, F, seed=P % 97)
    tx, tn, tg = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
    want = c_oracle.encode_bwd_f64(x, (L, T, F), n_ls, g)               # the reference's fp32 terms, summed in double
    want[:l0] = 0
    want[l1:] = 0
    mass = np.zeros((L, T, F))                                            # sum |term| per row: what an fp32 sum's error scales with
    absg = c_oracle.encode_bwd_f64(x, (L, T, F), n_ls, np.abs(g))
    mass[l0:l1] = absg[l0:l1]
    _, grid = orc.scale_to_grid(x, n_ls)
    idx = orc.spatial_hash(grid.astype(np.int32), T)                      # (P, L, 4)
    count = np.zeros((L, T, 1))
    for l in range(l0, l1):
        np.add.at(count[l, :, 0], idx[:, l].reshape(-1), 1)
    got, plan = _bucketed(ops, _lib, tx, tn, tg, L, T, F, l0, l1, image, 1)
    got = got.cpu().numpy().astype(np.float64)
    # exact integer sums of the fp32 terms, one rounding at the end: half an fp32 ulp of the result + the image's quantum per term
    # (2^-50 of the largest term of the row's bucket, bounded here by the largest term of the batch)
    err = np.abs(got - want)
    room = min(50, 61 - int(np.ceil(np.log2(max(1.0, count.max())))))       # (the kernel: per bucket, from its fullest row)
    quantum = count * (2.0 ** -room) * float(np.abs(g).max())
    bound = 6.0e-8 * np.abs(want) + quantum + 1e-45
    assert np.all(err <= bound), float((err / bound).max())
    assert np.count_nonzero(got[:l0]) == 0 and np.count_nonzero(got[l1:]) == 0
    from conftest import parity_close            # (a row in PARITY.md: the elementwise bound above is the assertion proper)
    parity_close(got[l0:l1], want[l0:l1], 1.2e-7, float(quantum.max()) + 1e-30,
                 f"bucketed direct backward vs C oracle (fp32 terms summed in double), P={P} L={l1 - l0} T={T} F={F} image={image}")
    # the kernel it replaces (fp32 atomics in arrival order): equal up to its own accumulation error
    tables = torch.zeros((L, T, F), device=DEV)
    ref = torch.zeros((L, T, F), device=DEV)
    _lib.call("gngf_encode_bwd", _lib.ptr(tx), *ops._tab(tables), _lib.ptr(None), _lib.ptr(None), _lib.ptr(tn), _lib.ptr(tg), _lib.ptr(ref),
              _lib.ptr(None), P, L, F, T, 0, ops.MODE_HASH, 0, 0, l0, l1, _lib.stream_ptr())
    torch.cuda.synchronize()
    dev_ = np.abs(ref.cpu().numpy() - got)
    lim = 6e-8 * (count + 1) * mass + quantum + 1e-30
    worst = np.unravel_index(np.argmax(dev_ / lim), dev_.shape)
    assert np.all(dev_ <= lim), (worst, dev_[worst], lim[worst], count[worst[0], worst[1], 0], mass[worst], got[worst], want[worst])
    # bitwise reproducible (the order of a bucket's items changes from run to run; integer sums do not care)
    again, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, l0, l1, image, 1)
    assert np.array_equal(again.cpu().numpy().astype(np.float64), got)
    # write mode: every row of [l0, l1) is written over whatever was there, the other levels are not touched
    dirty = torch.full((L, T, F), float("nan"), device=DEV)
    dirty[:l0] = 3.0
    dirty[l1:] = 5.0
    wrote, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, l0, l1, image, 0, into=dirty)
    w = wrote.cpu().numpy().astype(np.float64)
    assert np.array_equal(w[l0:l1], got[l0:l1]) and np.all(w[:l0] == 3.0) and np.all(w[l1:] == 5.0)
    # add mode on top of earlier gradients
    twice, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, l0, l1, image, 1, into=again.clone())
    assert np.allclose(twice.cpu().numpy(), 2 * got, rtol=1e-6, atol=0)


def test_bucketed_direct_backward_on_random_ragged_shapes():
    """twenty seeded random shapes (P from 1, any level sub-range, T with and without a power of two, every image size the plan
    accepts): exact sums again, and nothing outside the levels asked for is touched"""
    from collision_handling_in_instantngp_amd import _lib, ops
    rng = np.random.default_rng(20261004)
    done = 0
    while done < 20:
        P = int(rng.integers(1, 60000))
        L = int(rng.integers(1, 6))
        F = int(rng.choice([1, 2, 4]))
        T = int(rng.choice([int(rng.integers(3, 5000)), 2 ** int(rng.integers(4, 18))]))
        l0 = int(rng.integers(0, L))
        l1 = int(rng.integers(l0 + 1, L + 1))
        image = 2 ** int(rng.integers(10, 18))
        n_max = int(rng.choice([64, 700, 4096]))
        plan = (ctypes.c_int64 * 6)()
        if _lib.query("gngf_encode_bwd_bucketed_plan", P, F, T, l1 - l0, image, plan) != 1:
            continue
        done += 1
        x = rng.random((P, 2), dtype=np.float32)
        x[: min(P, 3)] = np.array([[0, 0], [1, 1], [0.99999994, 1e-8]], np.float32)[: min(P, 3)]
        n_ls = orc.level_resolutions(16, n_max, L).astype(np.int32) if L > 1 else np.array([n_max], np.int32)
        g = rng.standard_normal((P, L * F)).astype(np.float32)
        want = c_oracle.encode_bwd_f64(x, (L, T, F), n_ls, g)
        tx, tn, tg = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
        into = torch.full((L, T, F), 7.0, device=DEV)
        got, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, l0, l1, image, 0, into=into)
        got = got.cpu().numpy().astype(np.float64)
        tag = (P, L, F, T, l0, l1, image, n_max)
        assert np.all(got[:l0] == 7.0) and np.all(got[l1:] == 7.0), tag
        err = np.abs(got[l0:l1] - want[l0:l1])
        lim = 6.0e-8 * np.abs(want[l0:l1]) + 4 * P * 2.0 ** -44 * float(np.abs(g).max()) + 1e-45
        assert np.all(err <= lim), (tag, float((err / lim).max()))


def test_a_batch_that_lands_in_four_rows_is_still_exact():
    from collision_handling_in_instantngp_amd import _lib, ops
    P, L, T, F = 2 ** 16 + 5, 2, 2 ** 16, 2
    rng = np.random.default_rng(2)
    x = np.tile(np.array([[0.3711, 0.6123]], np.float32), (P, 1))
    x[-3000:] = rng.random((3000, 2), dtype=np.float32)
    n_ls = np.array([1024, 4096], np.int32)
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    tx, tn, tg = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
    want = c_oracle.encode_bwd_f64(x, (L, T, F), n_ls, g)
    got, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, 0, L, 8192, 1)
    got = got.cpu().numpy().astype(np.float64)
    # 65 k terms per row: room = 61 - 17 = 44 bits below the largest term
    assert np.all(np.abs(got - want) <= 6.0e-8 * np.abs(want) + P * 4 * 2.0 ** -44 * float(np.abs(g).max()))
    assert float(np.abs(want).max()) > 100


def test_a_tiny_row_next_to_a_large_one_in_the_same_bucket():
    """the bucket's fixed-point scale comes from its LARGEST |term| — the error bound is absolute per bucket, not
    relative per row.  Two pixels in different cells whose rows share one bucket (T = one bucket): one with gradient ~1, one with
    ~1e-20.  The large rows are within half an fp32 ulp of the double-precision sum; the tiny row's error is bounded by the quantum
    2^-50 (2^-45 here: 65 535 terms on the fullest row) of the bucket's largest term per term — which here means it is FLUSHED to zero, where float atomics would have kept
    1e-20 at fp32 relative precision.  Stated in csrc/encode_bucket.hip's header; this test pins it."""
    from collision_handling_in_instantngp_amd import _lib, ops
    P, L, T, F = 2 ** 16, 1, 2048, 2
    rng = np.random.default_rng(3)
    x = np.zeros((P, 2), np.float32)
    x[:] = [0.26, 0.26]                                    # every pixel on one cell ...
    x[0] = [0.74, 0.74]                                    # ... except the tiny one
    n_ls = np.array([64], np.int32)
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    g[0] = 1e-20
    tx, tn, tg = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
    got, plan = _bucketed(ops, _lib, tx, tn, tg, L, T, F, 0, L, 65536, 0)
    assert plan[1] == 1, plan                             # ONE bucket holds the whole level: both cells' rows share its scale
    got = got.double().cpu().numpy()
    want = c_oracle.encode_bwd_f64(x, (L, T, F), n_ls, g) if c_oracle.available() else None
    _, grid = orc.scale_to_grid(x[[0, 1]], n_ls)
    idx = orc.spatial_hash(grid.astype(np.int32), T)                     # (2, L, 4)
    tiny_rows, big_rows = idx[0, 0], idx[1, 0]
    assert not set(tiny_rows.tolist()) & set(big_rows.tolist())
    if want is not None:
        big = np.zeros((L, T, F), bool)
        big[0, big_rows] = True
        ulp = np.spacing(np.abs(want[big]).astype(np.float32)).astype(np.float64)
        # 65 535 terms on the fullest row: room = 61 - 16 = 45 bits, |term| < 2^(e + 1) <= 2 max |g|  ->  quantum q <= 2^-44 max |g|
        q = 2.0 ** -44 * float(np.abs(g).max())
        assert np.all(np.abs(got[big] - want[big]) <= 0.5 * ulp + 0.5 * q * P)
        # the absolute bound on the tiny rows: one quantum (of the bucket's largest term) per term, one term per row here
        assert np.all(np.abs(got[0, tiny_rows] - want[0, tiny_rows]) <= q)
        assert np.all(want[0, tiny_rows] > 0)
    assert np.all(got[0, tiny_rows] == 0.0), "documented behaviour: 1e-20 next to 1 in one bucket is below the bucket's quantum"


def test_non_finite_terms_reach_their_rows_only():
    from collision_handling_in_instantngp_amd import _lib, ops
    P, L, T, F = 2 ** 16, 2, 2 ** 14, 2
    x, n_ls, g = _case(P, 256, 2048, L, T, F, seed=5)
    g[12345, 1] = np.nan                                  # level 0, feature 1 of one pixel
    g[777, 2] = np.inf                                    # level 1, feature 0 of another
    tx, tn, tg = torch.from_numpy(x).to(DEV), torch.from_numpy(n_ls).to(DEV), torch.from_numpy(g).to(DEV)
    got, _ = _bucketed(ops, _lib, tx, tn, tg, L, T, F, 0, L, 4096, 1)
    got = got.cpu().numpy()
    _, grid = orc.scale_to_grid(x[[12345, 777]], n_ls)
    idx = orc.spatial_hash(grid.astype(np.int32), T)                     # (2, L, 4)
    bad = np.zeros((L, T, F), bool)
    bad[0, idx[0, 0], 1] = True
    bad[1, idx[1, 1], 0] = True
    assert np.all(~np.isfinite(got[bad])) and np.all(np.isfinite(got[~bad]))
    g2 = g.copy()
    g2[12345, 1] = 0
    g2[777, 2] = 0
    clean, _ = _bucketed(ops, _lib, tx, tn, torch.from_numpy(g2).to(DEV), L, T, F, 0, L, 4096, 1)
    clean = clean.cpu().numpy()
    # rows of buckets without a non-finite term: identical; rows that share a bucket with one were summed in fp32 instead
    same_bucket = np.zeros((L, T, F), bool)
    slots = 4096 // (8 * F)
    for lv, rows in ((0, idx[0, 0]), (1, idx[1, 1])):
        for r in rows:
            same_bucket[lv, (r // slots) * slots:(r // slots + 1) * slots] = True
    assert np.array_equal(got[~same_bucket], clean[~same_bucket])
    ok = same_bucket & ~bad
    assert np.allclose(got[ok], clean[ok], rtol=1e-4, atol=1e-6 * np.abs(clean).max())


def test_shapes_the_bucketed_form_does_not_serve_are_refused_by_its_plan():
    from collision_handling_in_instantngp_amd import _lib
    plan = (ctypes.c_int64 * 6)()
    q = lambda *a: _lib.query("gngf_encode_bwd_bucketed_plan", *a, plan)
    assert q(2 ** 20, 2, 2 ** 22, 2, 65536) == 1 and tuple(plan)[:3] == (12, 1024, 256)
    assert q(2 ** 20, 4, 2 ** 24, 4, 131072) == 1 and tuple(plan)[:3] == (12, 4096, 256) and plan[5] == 2 ** 20 * 16 * 20
    assert q(2 ** 20, 8, 2 ** 19, 2, 65536) == 0             # F = 8
    assert q(2 ** 20, 4, 2 ** 24, 4, 16384) == 0             # 2^15 buckets per level
    assert q(2 ** 28, 2, 2 ** 22, 2, 65536) == 0             # 2^31 contributions
    assert q(0, 2, 2 ** 22, 2, 65536) == 0


@pytest.mark.parametrize("F,half", [(2, False), (4, True)])
def test_encoder_backward_with_direct_levels_bucketed_equals_atomics(F, half):
    """the op as the model calls it: staged levels through the tiled chain, the fine levels in the direct form — bucketed or not"""
    from collision_handling_in_instantngp_amd import ops
    P, L, T = 2 ** 17, 8, 2 ** 16
    rng = np.random.default_rng(3)
    n_host = [int(v) for v in orc.level_resolutions(16, 4096, L)]
    n_ls = torch.tensor(n_host, dtype=torch.int32, device=DEV)
    xy = torch.from_numpy(rng.random((P, 2), dtype=np.float32)).to(DEV)
    g = torch.from_numpy(rng.standard_normal((P, L * F)).astype(np.float32)).to(DEV)
    plan = ops.EncodePlan(P, n_host, F)
    assert 0 < plan.Ls < L, plan.Ls
    res = {}
    prev = ops.BUCKETED_DIRECT_BWD
    try:
        for on in (True, False):
            ops.BUCKETED_DIRECT_BWD = on
            tab = torch.from_numpy(((rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-2)).to(DEV)
            tab = (tab.half() if half else tab).requires_grad_()
            enc = ops.encode_apply(xy, n_ls, n_host, tab, None, None, 0)
            enc.backward(g)
            torch.cuda.synchronize()
            res[on] = tab.grad.float().clone()
    finally:
        ops.BUCKETED_DIRECT_BWD = prev
    scale = float(res[False].abs().max())
    tol = (2e-3 if half else 2e-6) * scale                       # fp16 .grad: one rounding to half
    assert float((res[True] - res[False]).abs().max()) <= tol
    assert float(res[True][plan.Ls:].abs().max()) > 0


def test_step_to_step_gradient_buffer_equals_a_cleared_allocation_per_step():
    """ops.PERSISTENT_TABLE_GRAD (hash source, big shapes): the table gradient lives in one buffer per model, the rows the staged
    levels can touch are cleared (gngf_clear_hashed_rows), the direct levels are written.  Three steps on DIFFERENT batches with
    zero_grad between them equal the same steps on freshly cleared allocations; without zero_grad the second pass takes a buffer of
    its own and the gradients accumulate as torch's do; a kept reference to the first gradient is not written over then."""
    from collision_handling_in_instantngp_amd import models, ops, train
    P, L, T = 2 ** 17, 8, 2 ** 16
    models.should_use_hash_function = True
    prev, prev_min = ops.PERSISTENT_TABLE_GRAD, ops.PERSISTENT_MIN_BYTES
    ops.PERSISTENT_MIN_BYTES = 0          # (this small model's decoder is the training kernel, which would hide an 8 MB clear for free)
    try:
        g = torch.Generator(device=DEV).manual_seed(7)
        xs = [torch.rand((P, 2), device=DEV, generator=g) for _ in range(3)]
        ys = [torch.rand((P, 3), device=DEV, generator=g) for _ in range(3)]
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        empty = torch.tensor([], device=DEV)

        def build():
            torch.manual_seed(11)
            net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=4096,
                                                  MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                                  HPD_out_features=T, feature_dim=4, topk_k=4)
            net.return_indices = False
            with torch.no_grad():
                net.encoding.packed_tables().mul_(100.0)
            return net

        def step(net, k, zero=True):
            if zero:
                net.zero_grad()
            with net.fused_mse(ys[k], gloss=1.0):
                rgb, probs, _i, _c = net(xs[k], 1.0)
            mse, kls, coll = loss_fn(rgb, ys[k], None, probs, empty, empty)
            train.assemble_loss(mse, kls, coll, 1, 1, 1e-3).backward()
            torch.cuda.synchronize()
            return torch.stack([m.weight.grad for m in net.encoding._hash_tables])

        plan = ops.EncodePlan(P, [int(v) for v in orc.level_resolutions(16, 4096, L)], 4)
        assert 0 < plan.Ls < L and not plan.interleaved(backward=True)    # staged (generic kernels: four features) + direct levels
        res = {}
        for on in (False, True):
            ops.PERSISTENT_TABLE_GRAD = on
            net = build()
            net.dp.persist_ok = True          # opt-in per model: this loop lets go of the gradients before every step
            res[on] = [step(net, k).clone() for k in range(3)]
            assert (net.dp.persist_grad is not None) == on
            if on:
                # accumulation: a second backward pass without zero_grad adds to the first (which sits in the buffer)
                in_buffer = lambda: (net.encoding._hash_tables[0].weight.grad.untyped_storage().data_ptr()
                                     == net.dp.persist_grad.untyped_storage().data_ptr())
                first = step(net, 0)
                assert in_buffer()
                both = step(net, 1, zero=False)
                assert in_buffer()                         # (the second pass took an allocation of its own and was ADDED to the first)
                want = res[False][0] + res[False][1]
                assert float((both - want).abs().max()) <= 2e-6 * float(want.abs().max())
                # ... and after zero_grad the buffer is taken again
                again = step(net, 2)
                assert in_buffer()
                assert float((again - res[False][2]).abs().max()) <= 2e-6 * float(res[False][2].abs().max())
        for k in range(3):
            scale = float(res[False][k].abs().max())
            assert scale > 0 and float((res[True][k] - res[False][k]).abs().max()) <= 2e-6 * scale, k
        # the staged levels (float atomics of the gather pass) differ in the order of additions only; the direct levels are exact sums
        assert torch.equal(res[True][2][plan.Ls:], res[False][2][plan.Ls:])
    finally:
        ops.PERSISTENT_TABLE_GRAD, ops.PERSISTENT_MIN_BYTES = prev, prev_min
        models.should_use_hash_function = False


def test_a_gradient_the_caller_keeps_survives_zero_grad_and_the_next_step():
    """ADVICE r4 (medium).  torch and the reference loop leave a tensor the caller still holds alone: g = p.grad kept for logging,
    clipping or gradient differences must still hold step 1's values after zero_grad() and step 2.  The step-to-step buffer is
    therefore opt-in per model (net.dp.persist_ok, set by the code that owns the loop: train.GraphedStep, bench.py): a bare
    net(x); loss.backward() loop gets an allocation per step — checked here on the shape where the buffer WOULD be taken — and the
    same loop with the opt-in shows the aliasing the opt-in accepts."""
    from collision_handling_in_instantngp_amd import models, ops, train
    P, L, T = 2 ** 17, 8, 2 ** 16
    models.should_use_hash_function = True
    prev_min = ops.PERSISTENT_MIN_BYTES
    ops.PERSISTENT_MIN_BYTES = 0
    try:
        g = torch.Generator(device=DEV).manual_seed(21)
        xs = [torch.rand((P, 2), device=DEV, generator=g) for _ in range(2)]
        ys = [torch.rand((P, 3), device=DEV, generator=g) for _ in range(2)]
        for opt_in in (False, True):
            torch.manual_seed(11)
            net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=4096,
                                                  MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                                  HPD_out_features=T, feature_dim=4, topk_k=4)
            net.return_indices = False
            assert net.dp.persist_ok is False           # the default: like torch
            net.dp.persist_ok = opt_in
            with torch.no_grad():
                net.encoding.packed_tables().mul_(100.0)
            kept, snap = [], []
            for k in range(2):
                net.zero_grad()                          # set_to_none: the parameters let go, the caller's references stay
                with net.fused_mse(ys[k], gloss=1.0):
                    rgb, _p, _i, _c = net(xs[k], 1.0)
                ops.mse_loss(rgb, ys[k]).backward()
                torch.cuda.synchronize()
                kept.append([m.weight.grad for m in net.encoding._hash_tables])
                snap.append([t_.clone() for t_ in kept[-1]])
            same = all(torch.equal(a, b) for a, b in zip(kept[0], snap[0]))
            assert (net.dp.persist_grad is not None) == opt_in
            if opt_in:
                assert not same, "with the opt-in the two steps share one buffer (the aliasing the loop's owner accepted)"
            else:
                assert same, "a kept gradient was overwritten by the next step"
                assert float(torch.stack(kept[0]).abs().max()) > 0
    finally:
        ops.PERSISTENT_MIN_BYTES = prev_min
        models.should_use_hash_function = False


@pytest.mark.parametrize("graph", [False, True])
def test_training_epochs_on_the_step_to_step_buffer_equal_epochs_on_fresh_allocations(graph):
    """train.train_epoch with the optimizer (eager, and graph=True: cross-replay hipGraphs with FusedAdam inside) on a model whose
    table gradient takes the step-to-step buffer — generic pixel stage (four features), three direct levels bucketed — against the
    same epochs with an allocation per step: same losses and outputs up to the order of the float atomics of the staged levels."""
    from collision_handling_in_instantngp_amd import models, ops, train
    side, L, T = 512, 8, 2 ** 16
    g = torch.Generator(device=DEV).manual_seed(3)
    rows, cols = torch.meshgrid(torch.arange(side, device=DEV), torch.arange(side, device=DEV), indexing="ij")
    X = (torch.stack([rows, cols], -1).reshape(-1, 2).float() / (side - 1)).contiguous()
    Y = torch.rand((side * side, 3), device=DEV, generator=g)
    perm = torch.randperm(side * side, generator=torch.Generator().manual_seed(1)).to(DEV)
    models.should_use_hash_function = True
    prev, prev_min = ops.PERSISTENT_TABLE_GRAD, ops.PERSISTENT_MIN_BYTES
    ops.PERSISTENT_MIN_BYTES = 0
    try:
        outs = {}
        for on in (True, False):
            ops.PERSISTENT_TABLE_GRAD = on
            torch.manual_seed(5)
            net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=4096,
                                                  MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                                  HPD_out_features=T, feature_dim=4, topk_k=4)
            net.return_indices = False
            net.dp.persist_ok = True          # (graph=True: train.GraphedStep opts in by itself; the eager epoch loop does not)
            loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
            opt = train.get_optimizer(net, 1e-2, 1e-3, 1e-3, 0.0, 0.0, 1e-6)
            mses = []
            for _epoch in range(2):
                rec = train.train_epoch(net, loss_fn, opt, X, Y, side, side, 1, 1, 1e-3, batch_percentage=1 / 2, should_shuffle=True,
                                        shuffled_indices=perm, graph=graph)
                mses += rec["mse"]
            torch.cuda.synchronize()
            outs[on] = (rec["outputs"].clone(), torch.stack(mses).clone(), torch.stack([m.weight.detach() for m in net.encoding._hash_tables]).clone())
            assert (net.dp.persist_grad is not None) == on
        assert float(outs[False][1][0]) > 0 and float(outs[False][1][-1]) < float(outs[False][1][0])         # it trains
        assert float((outs[True][1] - outs[False][1]).abs().max()) <= 1e-4 * float(outs[False][1].abs().max())
        assert float((outs[True][0] - outs[False][0]).abs().max()) <= 2e-3
        assert float((outs[True][2] - outs[False][2]).abs().max()) <= 1e-3 * float(outs[False][2].abs().max())
    finally:
        ops.PERSISTENT_TABLE_GRAD, ops.PERSISTENT_MIN_BYTES = prev, prev_min
        models.should_use_hash_function = False
