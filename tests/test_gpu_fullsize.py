"""GPU, BASELINE.json full sizes, P = 2^20 pixels per batch:
    cfg2  L=16 F=2 T=2^19 fp32, N 16->512   (strawberry aspect; every level staged)          hash + vertex-table indexing
    cfg4  L=16 F=2 T=2^22 fp32, N 16->4095  (4096^2 image: 12 staged + 4 direct levels)      hash + vertex-table indexing
    cfg5  L=16 F=4 T=2^24 fp16, N 16->8191  (8192^2 image, fp16 table storage)               hash indexing
Size-independent properties of the HIP path (mass conservation, linearity, tiled == direct, run-to-run reproducibility)
plus a sampled comparison with the oracle (the full oracle needs ~30 s per call) and the reference's own hash indices at the
large-table shapes (golden G3b)."""
import numpy as np
import pytest
import torch

from conftest import parity_close
from oracle import gngf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
P = 2 ** 20

SHAPES = {
    #        L   F  T        K  n_min n_max  table dtype     short-axis extent
    "cfg2": (16, 2, 2 ** 19, 4, 16, 512, torch.float32, 338.0 / 507.0),
    "cfg4": (16, 2, 2 ** 22, 4, 16, 4096, torch.float32, 1.0),
    "cfg5": (16, 4, 2 ** 24, 4, 16, 8192, torch.float16, 1.0),
}
CASES = [("cfg2", "hash"), ("cfg2", "gngf"), ("cfg4", "hash"), ("cfg4", "gngf"), ("cfg5", "hash")]


@pytest.fixture(scope="module")
def shapes():
    """One set of device tensors per configuration, built on first use and dropped when the next one is asked for
    (cfg5 alone holds 2 GiB of tables + 4 GiB of gradient buffers)."""
    from collision_handling_in_instantngp_amd import ops
    cache = {}

    def get(name):
        if name not in cache:
            cache.clear()
            torch.cuda.empty_cache()
            L, F, T, K, n_min, n_max, dtype, short = SHAPES[name]
            g = torch.Generator(device=DEV).manual_seed(123)
            xy = torch.rand((P, 2), device=DEV, generator=g)
            xy[:, 1] *= short
            xy[:8] = torch.tensor([[0, 0], [1, short], [0, short], [1, 0], [0.5, 0.25], [1 / 32, 1 / 32], [31 / 32, 0.5], [1, 1 / 512]],
                                  device=DEV)
            n_host = [int(v) for v in orc.level_resolutions(n_min, n_max, L)]
            n_ls = torch.tensor(n_host, dtype=torch.int32, device=DEV)
            amp = 2e-4 if dtype == torch.float32 else 2e-2            # fp16 storage: keep the features well above fp16 subnormals
            tables = ((torch.rand((L, T, F), device=DEV, generator=g) - 0.5) * amp).to(dtype)
            genc = torch.randn((P, L * F), device=DEV, generator=g)
            cache[name] = dict(ops=ops, L=L, F=F, T=T, K=K, n_host=n_host, n_ls=n_ls, xy=xy, tables=tables, genc=genc, dtype=dtype,
                               vs=n_host[-1] + 2, gen=g)
        return cache[name]
    return get


def _vertex_table(s):
    if "vidx" not in s:
        vs, K, T, g = s["vs"], s["K"], s["T"], s["gen"]
        s["vidx"] = torch.randint(0, T, (vs * vs, K), device=DEV, dtype=torch.int32, generator=g)
        s["vw"] = torch.softmax(torch.rand((vs * vs, K), device=DEV, generator=g), -1)
    return s["vidx"], s["vw"]


def _index_source(s, mode):
    if mode == "hash":
        return None, None, 0
    vidx, vw = _vertex_table(s)
    return vidx, vw, s["vs"]


def _oracle_rows(s, mode, sel):
    """oracle encoder output (and per-instance indices / weights) for the sampled pixels"""
    x_np = s["xy"][sel].cpu().numpy()
    n_np = np.array(s["n_host"], np.int32)
    _, grid = orc.scale_to_grid(x_np, n_np)
    L, T = s["L"], s["T"]
    if mode == "hash":
        idx, w = orc.spatial_hash(grid.astype(np.int32), T), None
        rows = torch.from_numpy(idx).to(DEV)                                          # (n,L,4)
        feats = s["tables"][torch.arange(L, device=DEV)[None, :, None], rows].float().cpu().numpy()      # (n,L,4,F): gathered on the device
        feats = np.ascontiguousarray(feats.transpose(0, 3, 1, 2))
    else:
        vidx, vw = _vertex_table(s)
        gi = grid.astype(np.int64)
        vid = torch.from_numpy(gi[:, 1] * s["vs"] + gi[:, 0]).to(DEV)
        idx = vidx[vid].long()                                                         # (n,L,4,K)
        w = vw[vid].cpu().numpy()
        rows = s["tables"][torch.arange(L, device=DEV)[None, :, None, None], idx].float().cpu().numpy()  # (n,L,4,K,F)
        feats = (rows * w[..., None]).astype(np.float32).sum(3, dtype=np.float32)       # models.py:215-219 (weights given: blend None)
        feats = np.ascontiguousarray(feats.transpose(0, 3, 1, 2))
    return orc.bilinear_forward(x_np, n_np, feats)


@pytest.mark.parametrize("name,mode", CASES)
def test_full_size_forward_sample_vs_oracle_and_forms_agree(shapes, name, mode):
    s = shapes(name)
    ops = s["ops"]
    vi, w, vstr = _index_source(s, mode)
    plan = ops.EncodePlan(P, s["n_host"], s["F"], "tiled")
    if name != "cfg2":
        assert 0 < plan.Ls < s["L"], (name, plan.Ls)          # the mixed staged / direct plan is what this shape exercises
    else:
        assert plan.Ls == s["L"]
    enc_t = ops.encode_apply(s["xy"], s["n_ls"], s["n_host"], s["tables"], vi, w, vstr, path="tiled")
    enc_d = ops.encode_apply(s["xy"], s["n_ls"], s["n_host"], s["tables"], vi, w, vstr, path="direct")
    assert torch.equal(enc_t, enc_d)                            # tiled and direct forms: bit-identical forward
    sel = torch.cat([torch.arange(8), torch.randint(0, P, (4096,))]).to(DEV)
    want = _oracle_rows(s, mode, sel)
    parity_close(enc_t[sel], want, 2e-6, 1e-10, f"{name} {mode}: encoder output, 4104 sampled pixels of 2^20 vs oracle")


@pytest.mark.parametrize("name,mode", CASES)
def test_full_size_backward_properties(shapes, name, mode):
    """(1) mass conservation: bilinear (and softmax-blend) weights sum to 1, so per level and feature the table gradient
    sums to the sum of the upstream gradient; (2) linearity: backward(a*g1 + g2) = a*backward(g1) + backward(g2);
    (3) tiled and direct forms agree to fp32 round-off; (4) the pixel stage is bitwise reproducible run to run;
    (5) sampled rows of the table gradient equal the oracle's scatter-add restricted to a pixel subset (linearity in g)."""
    s = shapes(name)
    ops = s["ops"]
    L, F, T = s["L"], s["F"], s["T"]
    vi, w, vstr = _index_source(s, mode)
    xy, n_ls, n_host, tables, genc = s["xy"], s["n_ls"], s["n_host"], s["tables"], s["genc"]
    half = s["dtype"] == torch.float16
    tol = 2e-3 if half else 2e-5                                # fp16 storage: the gradient is rounded to fp16 once

    def grad(g, path):
        t_ = tables.clone().requires_grad_()
        enc = ops.encode_apply(xy, n_ls, n_host, t_, vi, w, vstr, path=path)
        enc.backward(g)
        return t_.grad

    g1 = genc
    d1 = grad(g1, "tiled")
    assert d1.dtype == s["dtype"]
    want = g1.double().reshape(P, L, F).sum(0)                   # (L,F)
    got = d1.double().sum(1)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=(2.0 if half else 1e-2))
    g2 = torch.roll(genc, 1, 0) * 0.5
    d2 = grad(g2, "tiled")
    d12 = grad(3.0 * g1 + g2, "tiled")
    scale = float(d12.float().abs().max())
    assert float((d12.float() - (3.0 * d1.float() + d2.float())).abs().max()) <= tol * scale
    del d2, d12
    dd = grad(g1, "direct")
    err = float((dd.float() - d1.float()).abs().max())
    assert err <= tol * float(d1.float().abs().max())
    del dd
    # (5) pixels outside the sample get zero upstream gradient: the full-size kernels then reproduce the oracle's sum
    sel = torch.cat([torch.arange(8), torch.randint(0, P, (2048,))]).unique().to(DEV)
    gs = torch.zeros_like(genc)
    gs[sel] = genc[sel]
    ds = grad(gs, "tiled").float()
    x_np = xy[sel].cpu().numpy()
    n_np = np.array(n_host, np.int32)
    _, grid = orc.scale_to_grid(x_np, n_np)
    dfe = orc.bilinear_backward(x_np, n_np, genc[sel].cpu().numpy(), F)      # (n,F,L,4)
    gcorner = torch.from_numpy(np.ascontiguousarray(dfe.transpose(0, 2, 3, 1))).to(DEV).double()   # (n,L,4,F)
    want_dt = torch.zeros((L, T, F), dtype=torch.float64, device=DEV)
    lvl = torch.arange(L, device=DEV)[None, :, None]
    if mode == "hash":
        rows = torch.from_numpy(orc.spatial_hash(grid.astype(np.int32), T)).to(DEV)
        want_dt.index_put_((lvl.expand_as(rows), rows), gcorner, accumulate=True)
    else:
        gi = grid.astype(np.int64)
        vid = torch.from_numpy(gi[:, 1] * s["vs"] + gi[:, 0]).to(DEV)
        rows, wk = vi[vid].long(), w[vid].double()                              # (n,L,4,K)
        want_dt.index_put_((lvl[..., None].expand_as(rows), rows), gcorner[:, :, :, None, :] * wk[..., None], accumulate=True)
    nzrows = want_dt.abs().sum(-1) > 0
    parity_close(ds[nzrows], want_dt[nzrows].float(), 1e-3 if half else 1e-4, (1e-3 if half else 1e-6) * float(want_dt.abs().max()),
                 f"{name} {mode}: table gradient rows touched by {sel.numel()} sampled pixels vs oracle scatter-add")
    assert float(ds[~nzrows].abs().max()) == 0.0                  # and nothing anywhere else
    del ds, want_dt
    # (4) reproducibility of the privatised pixel stage (64-bit fixed-point accumulation commutes)
    plan = ops.EncodePlan(P, n_host, F, "tiled")
    ws = ops.TiledWorkspace(plan, xy)
    dG = [torch.zeros((plan.vtot, F), device=DEV) for _ in range(2)]
    for d in dG:
        ops._pixel_bwd(plan, ws, n_ls, g1, d, L, F)
    assert torch.equal(dG[0], dG[1])


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_full_size_hash_indices_checksum(shapes, name):
    """bit-exact index work at full size: the int64 (P,L,4) tensor equals the oracle on a sample, and every index < T."""
    s = shapes(name)
    idx = s["ops"].hash_indices(s["xy"][: 2 ** 18], s["n_ls"], s["T"])
    assert int(idx.min()) >= 0 and int(idx.max()) < s["T"]
    sel = torch.randint(0, 2 ** 18, (2048,), device=DEV)
    _, grid = orc.scale_to_grid(s["xy"][sel].cpu().numpy(), np.array(s["n_host"], np.int32))
    assert np.array_equal(idx[sel].cpu().numpy(), orc.spatial_hash(grid.astype(np.int32), s["T"]))


@pytest.mark.parametrize("tag", ["cfg4", "cfg5"])
def test_hash_indices_large_tables_vs_reference_golden(golden, tag):
    """T = 2^22 / N_max 4096 (-> 4095) and T = 2^24 / N_max 8192 (-> 8191): indices computed by the reference itself (G3b)."""
    from collision_handling_in_instantngp_amd import ops
    g = golden("G3b_hash_large_tables")
    a, b, L, T = (int(v) for v in g[f"{tag}_cfg"])
    n_ls = orc.level_resolutions(a, b, L)
    assert np.array_equal(n_ls, g[f"{tag}_n_ls"]) and int(n_ls[-1]) == b - 1
    idx = ops.hash_indices(torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(n_ls).to(DEV), T)
    assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_hash"])


@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_full_size_model_step_runs_and_matches_oracle_on_a_sample(name):
    """GeneralNeuralGaugeFields at the configuration's own constructor arguments (hash indexing; cfg5 with fp16 table storage):
    one forward + MSE + backward at P = 2^20; rgb of sampled pixels equals oracle encoder + oracle decoder on the same weights."""
    from collision_handling_in_instantngp_amd import models
    L, F, T, K, n_min, n_max, dtype, short = SHAPES[name]
    torch.cuda.empty_cache()
    models.should_use_hash_function = True
    try:
        torch.manual_seed(7)
        net = models.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=n_min, n_max=n_max,
                                              MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                              HPD_out_features=T, feature_dim=F, topk_k=K, table_dtype=dtype)
        net.return_indices = False
        with torch.no_grad():
            for m in net.encoding._hash_tables:
                m.weight.mul_(200.0)                            # 2e-2-scale features (fp16-representable, visible in rgb)
        g = torch.Generator(device=DEV).manual_seed(5)
        xy = torch.rand((P, 2), device=DEV, generator=g)
        target = torch.rand((P, 3), device=DEV, generator=g)
        rgb, probs, idx, counts = net(xy, 1.0)
        loss = torch.nn.functional.mse_loss(rgb, target) * (1024.0 if dtype == torch.float16 else 1.0)
        loss.backward()
        assert probs is None and idx is None and counts == []
        for k_, p_ in net.named_parameters():
            if k_.startswith("encoding") or k_.startswith("mlp"):
                assert p_.grad is not None and torch.isfinite(p_.grad.float()).all(), k_
        assert net.encoding._hash_tables[L - 1].weight.grad.dtype == dtype
        assert float(net.encoding._hash_tables[L - 1].weight.grad.float().abs().max()) > 0
        sel = torch.randint(0, P, (2048,), device=DEV)
        n_host = [int(v) for v in orc.level_resolutions(n_min, n_max, L)]
        x_np = xy[sel].cpu().numpy()
        n_np = np.array(n_host, np.int32)
        _, grid = orc.scale_to_grid(x_np, n_np)
        rows = torch.from_numpy(orc.spatial_hash(grid.astype(np.int32), T)).to(DEV)
        tables = net.encoding.packed_tables()
        feats = tables[torch.arange(L, device=DEV)[None, :, None], rows].float().cpu().numpy()
        enc = orc.bilinear_forward(x_np, n_np, np.ascontiguousarray(feats.transpose(0, 3, 1, 2)))
        Ws = [seq[0].weight.detach().cpu().numpy() for seq in net.mlp]
        Bs = [seq[0].bias.detach().cpu().numpy() for seq in net.mlp]
        want = orc.decoder_forward(enc, Ws, Bs)
        parity_close(rgb[sel], want, 0, 1e-5, f"{name}: model rgb at P=2^20, 2048 sampled pixels vs oracle encoder + decoder")
    finally:
        models.should_use_hash_function = False
