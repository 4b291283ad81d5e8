"""GPU: the cfg4 / cfg5 single-GPU steps EXACTLY as bench.py builds them (bench.build_model + bench.make_batch +
train.GraphedStep(cross_replay=True)), over several DIFFERENT batches, against the C oracle at P = 2^20.

What these shapes run that the headline shape does not: the table gradient lives in ONE buffer from step to step
(ops.PERSISTENT_TABLE_GRAD: only the rows the staged levels can ever touch are cleared), the direct levels' backward is the bucketed
form in WRITE mode (csrc/encode_bucket.hip: every row of those levels is rewritten, none is cleared), the gradient of fp16 tables is
handed over as the fp32 buffer (param.grad_fp32), and the steps replay from the cold / steady cross-replay graphs.  A stale row
(cleared set != touched set, a 32-bit row count at T = 2^24, a second step on another batch) would show here and nowhere else:
  * every row of the (L,T,F) table gradient after the last step vs c_oracle.encode_bwd_f64 (the double-precision sum of the
    reference's fp32 terms, /root/reference/models.py:181-191 + the embedding backward triggered at functions.py:272), level by level;
  * rows the PREVIOUS step's batch touched and this one does not: exactly zero (counted, so the check is not vacuous);
  * rgb of all 2^20 pixels and the MSE of EVERY step, the six decoder gradients of the last;
  * from the dispatch trace: the step-to-step buffer, the bucketed write mode and (cfg5) the fp32 hand-over were taken."""
import numpy as np
import pytest
import torch

from conftest import PARITY, parity_close
from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
P = 2 ** 20


def _poison_allocator(nbytes=3 << 30):
    blocks = []
    for sz in (nbytes // 2, nbytes // 4, nbytes // 8, 64 << 20, 16 << 20, 16 << 20, 4 << 20, 4 << 20):
        blocks.append(torch.full((sz // 4,), float("nan"), dtype=torch.float32, device=DEV))
    del blocks
    torch.cuda.synchronize()


def _numpy_state(net, L):
    sd = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    tables = np.ascontiguousarray(np.stack([sd[f"encoding._hash_tables.{l}.weight"] for l in range(L)]))     # fp16 storage: the rounded values
    dw = [np.ascontiguousarray(sd[f"mlp.{i}.0.weight"]) for i in range(3)]
    db = [np.ascontiguousarray(sd[f"mlp.{i}.0.bias"]) for i in range(3)]
    return tables, dw, db


def _table_grad_levels(net, L):
    """level l's gradient as the step left it: .grad (fp32 tables) or the fp32 hand-over buffer (fp16 tables)"""
    out = []
    for l in range(L):
        w = net.encoding._hash_tables[l].weight
        g = getattr(w, "grad_fp32", None)
        out.append(g if g is not None else w.grad)
    return out


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libgngf_oracle_c.so not built (make -C oracle)")
@pytest.mark.parametrize("unroll", [1, 3])
@pytest.mark.parametrize("mode", ["cfg4_hash", "cfg5_hash_fp16"])
def test_big_shape_bench_steps_over_three_batches_match_the_oracle(mode, unroll):
    import bench
    from collision_handling_in_instantngp_amd import models, ops, train
    cfg = bench.MODES[mode]
    c = bench.SHAPES[cfg]
    L, T, F = c["L"], c["T"], c["F"]
    dev = torch.device(DEV)
    batches = [bench.make_batch(cfg, P, r, dev) for r in range(3)]           # three different pixel draws (seed 65535 + r)
    assert not torch.equal(batches[0][0], batches[1][0]) and not torch.equal(batches[1][0], batches[2][0])
    _poison_allocator()
    trace = []
    prev_trace, ops.STEP_TRACE = ops.STEP_TRACE, trace
    prev_fp32 = ops.FP16_TABLE_GRAD_FP32
    try:
        net, _ = bench.build_model(mode, dev, batches[0][2])
        with torch.no_grad():
            net.encoding.packed_tables().mul_(100.0)       # (see test_gpu_bench_chain: at the init scale every row's gradient is the same few bits)
        loss_fn = train.Loss(delta=1, gamma=-2, epsilon=1)
        gs = train.GraphedStep(net, loss_fn, None, 1, 1, 1e-3, unroll=unroll, cross_replay=True)
        tables, dw, db = _numpy_state(net, L)
        n_ls = np.array(net._n_ls_host, np.int32)
        xs = [np.ascontiguousarray(b[0].cpu().numpy()) for b in batches]
        ys = [np.ascontiguousarray(b[1].cpu().numpy()) for b in batches]

        def oracle_fwd(k):
            enc = c_oracle.encode_fwd(xs[k], tables, n_ls)
            rgb, h1, h2 = c_oracle.decoder_fwd(enc, dw, db)
            return enc, rgb, h1, h2

        # cold graph, then both steady ones, every batch seen, the last step's predecessor on ANOTHER batch
        order = [0, 1, 2, 0, 1, 2]
        if unroll == 1:
            results = []
            for j, k in enumerate(order):
                nxt = batches[order[(j + 1) % len(order)]][0]
                r = gs(batches[k][0], batches[k][1], next_first=nxt)
                torch.cuda.synchronize()
                results.append((k, r.out.clone(), float(r.mse)))
        else:
            pairs = [(b[0], b[1]) for b in batches]
            results = []
            for _rep in range(3):
                rs = gs.run_many(pairs, next_first=batches[0][0])
                torch.cuda.synchronize()
                results = [(k, r.out.clone(), float(r.mse)) for k, r in enumerate(rs)]
        # ---- every step's outputs (rgb of all pixels, MSE value)
        fwd = {}
        for k, out, mse in results[-3:]:
            if k not in fwd:
                fwd[k] = oracle_fwd(k)
            rgb = fwd[k][1]
            parity_close(out, rgb, 0, 1e-5, f"{mode} unroll={unroll} batch {k}: rgb, all 2^20 pixels vs C oracle")
            parity_close(mse, float(np.mean((rgb.astype(np.float64) - ys[k]) ** 2)), 1e-5, 0, f"{mode} unroll={unroll} batch {k}: MSE value")
        # ---- the dispatch the big shapes are about
        srcs = [f["source"] for w, f in trace if w == "table_grad"]
        assert srcs and all(s_ == "persist" for s_ in srcs), srcs
        dbw = [f for w, f in trace if w == "direct_bwd"]
        assert dbw and all(f["bucketed"] and f["write"] and f["levels"][1] == L for f in dbw), dbw
        if c["half"]:
            go = [f for w, f in trace if w == "grad_out"]
            assert go and all(f["fp32_handover"] for f in go), go
        # ---- the last step (batch 2, its predecessor ran batch 1): decoder gradients, every table-gradient row
        k_last, k_prev = 2, 1
        enc, rgb, h1, h2 = fwd[k_last]
        drgb = ((2.0 / rgb.size) * (rgb - ys[k_last])).astype(np.float32)
        genc, gdec = c_oracle.decoder_bwd(enc, h1, h2, rgb, drgb, dw)
        names = ["mlp.0.0.weight", "mlp.0.0.bias", "mlp.1.0.weight", "mlp.1.0.bias", "mlp.2.0.weight", "mlp.2.0.bias"]
        params = dict(net.named_parameters())
        for nm, wg in zip(names, gdec):
            scale = float(np.abs(wg).max()) + 1e-30
            parity_close(params[nm].grad, wg, 1e-3, 2e-5 * scale, f"{mode} unroll={unroll}: grad {nm} vs C oracle")
        del enc, h1, h2, fwd
        want = c_oracle.encode_bwd_f64(xs[k_last], (L, T, F), n_ls, np.ascontiguousarray(genc))
        ones = np.ones_like(genc)
        prev_touch = c_oracle.encode_bwd_f64(xs[k_prev], (L, T, F), n_ls, ones)       # > 0 exactly on the rows batch k_prev reaches
        now_touch = c_oracle.encode_bwd_f64(xs[k_last], (L, T, F), n_ls, ones)
        got = _table_grad_levels(net, L)
        mx = float(np.abs(want).max())
        assert mx > 0
        worst, n_stale, n_untouched_nonzero, n_rows = 0.0, 0, 0, 0
        for l in range(L):
            g = got[l].double()
            assert tuple(g.shape) == (T, F) and bool(torch.isfinite(g).all()), f"level {l}: non-finite table gradient (NaN-poisoned allocator)"
            w_l = torch.from_numpy(want[l]).to(dev)
            worst = max(worst, float((g - w_l).abs().max()))
            now_l = torch.from_numpy(now_touch[l]).to(dev) > 0
            prev_l = torch.from_numpy(prev_touch[l]).to(dev) > 0
            n_untouched_nonzero += int((g[~now_l] != 0).sum())                        # rows this batch does not reach: EXACTLY zero
            n_stale += int((prev_l & ~now_l).sum())                                   # ... among them the ones the previous step wrote
            n_rows += int(now_l.sum())
            del g, w_l, now_l, prev_l
        print(f"[{mode} unroll={unroll}] table gradient: max |err| {worst:.3e} = {worst / mx:.2e} of max; {n_rows} touched entries, "
              f"{n_stale} entries touched by the previous step only, {n_untouched_nonzero} untouched entries non-zero")
        PARITY.rows.append({"test": f"test_gpu_big_shapes.py::{mode}-unroll{unroll}", "quantity": f"{mode} unroll={unroll}: table gradient, all "
                            f"{L} x {T} rows after three different batches vs C oracle (atol 1e-5 of max); {n_stale} entries the previous step "
                            "wrote and this one does not are exactly zero", "n": int(L * T * F), "max_abs_err": worst, "ref_max_abs": mx,
                            "max_err_over_ref_max": worst / mx, "max_rel_err_significant": 0.0, "rtol": 0.0, "atol": 1e-5 * mx})
        assert worst <= 1e-5 * mx, (worst, mx)
        assert n_stale > 100000, n_stale                # the two batches differ: there ARE rows only the previous step touched
        assert n_untouched_nonzero == 0, n_untouched_nonzero
    finally:
        ops.STEP_TRACE = prev_trace
        ops.FP16_TABLE_GRAD_FP32 = prev_fp32
        models.should_use_hash_function = False
