"""GPU, BASELINE.json full sizes (L=16, F=2, T=2^19, N 16->512, P=2^20 strawberry-shaped batch): size-independent
properties of the HIP path plus a sampled comparison with the oracle (the full oracle needs ~30 s per call)."""
import numpy as np
import pytest
import torch

from oracle import gngf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
L, F, T, K, P = 16, 2, 2 ** 19, 4, 2 ** 20


@pytest.fixture(scope="module")
def setup():
    from collision_handling_in_instantngp_amd import ops
    g = torch.Generator(device=DEV).manual_seed(123)
    xy = torch.rand((P, 2), device=DEV, generator=g)
    xy[:, 1] *= 338.0 / 507.0                                  # strawberry aspect: short axis tops out at 0.667
    xy[:8] = torch.tensor([[0, 0], [1, 338 / 507], [0, 338 / 507], [1, 0], [0.5, 0.25], [1 / 32, 1 / 32], [31 / 32, 0.5], [1, 1 / 512]], device=DEV)
    n_host = [int(v) for v in orc.level_resolutions(16, 512, L)]
    n_ls = torch.tensor(n_host, dtype=torch.int32, device=DEV)
    tables = (torch.rand((L, T, F), device=DEV, generator=g) - 0.5) * 2e-4
    vs = 514
    vidx = torch.randint(0, T, (vs * vs, K), device=DEV, dtype=torch.int32, generator=g)
    vw = torch.softmax(torch.rand((vs * vs, K), device=DEV, generator=g), -1)
    genc = torch.randn((P, L * F), device=DEV, generator=g)
    return ops, xy, n_ls, n_host, tables, vidx, vw, vs, genc


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_full_size_forward_sample_vs_oracle_and_forms_agree(setup, mode):
    ops, xy, n_ls, n_host, tables, vidx, vw, vs, genc = setup
    vi, w = (None, None) if mode == "hash" else (vidx, vw)
    enc_t = ops.encode_apply(xy, n_ls, n_host, tables, vi, w, vs if vi is not None else 0, path="tiled")
    enc_d = ops.encode_apply(xy, n_ls, n_host, tables, vi, w, vs if vi is not None else 0, path="direct")
    assert torch.equal(enc_t, enc_d)                            # tiled and direct forms: bit-identical forward
    sel = torch.cat([torch.arange(8), torch.randint(0, P, (4096,))]).to(DEV)
    x_np = xy[sel].cpu().numpy()
    n_np = np.array(n_host, np.int32)
    _, grid = orc.scale_to_grid(x_np, n_np)
    if mode == "hash":
        feats = orc.encoding_forward(tables.cpu().numpy(), orc.spatial_hash(grid.astype(np.int32), T))
    else:
        gi = grid.astype(np.int64)
        vid = gi[:, 1] * vs + gi[:, 0]
        feats = orc.encoding_forward(tables.cpu().numpy(), vidx.cpu().numpy()[vid].astype(np.int64), vw.cpu().numpy()[vid], None)
    want = orc.bilinear_forward(x_np, n_np, feats)
    np.testing.assert_allclose(enc_t[sel].cpu().numpy(), want, rtol=2e-6, atol=1e-10)


@pytest.mark.parametrize("mode", ["hash", "gngf"])
def test_full_size_backward_properties(setup, mode):
    """(1) mass conservation: bilinear (and softmax-blend) weights sum to 1, so per level and feature the table gradient
    sums to the sum of the upstream gradient; (2) linearity: backward(a*g1 + g2) = a*backward(g1) + backward(g2);
    (3) tiled and direct forms agree to fp32 round-off; (4) the pixel stage is bitwise reproducible run to run."""
    ops, xy, n_ls, n_host, tables, vidx, vw, vs, genc = setup
    vi, w = (None, None) if mode == "hash" else (vidx, vw)
    vstr = vs if vi is not None else 0

    def grad(g, path):
        t_ = tables.clone().requires_grad_()
        enc = ops.encode_apply(xy, n_ls, n_host, t_, vi, w, vstr, path=path)
        enc.backward(g)
        return t_.grad

    g1 = genc
    g2 = torch.roll(genc, 1, 0) * 0.5
    d1 = grad(g1, "tiled")
    want = g1.double().reshape(P, L, F).sum(0)                   # (L,F)
    got = d1.double().sum(1)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-2)
    d2 = grad(g2, "tiled")
    d12 = grad(3.0 * g1 + g2, "tiled")
    scale = float(d12.abs().max())
    assert float((d12 - (3.0 * d1 + d2)).abs().max()) <= 2e-5 * scale
    dd = grad(g1, "direct")
    assert float((dd - d1).abs().max()) <= 2e-5 * float(d1.abs().max())
    # reproducibility of the privatised pixel stage (64-bit fixed-point accumulation commutes)
    plan = ops.EncodePlan(P, n_host, F, "tiled")
    ws = ops.TiledWorkspace(plan, xy)
    dG = [torch.zeros((plan.vtot, F), device=DEV) for _ in range(2)]
    for d in dG:
        ops._pixel_bwd(plan, ws, n_ls, g1, d, L, F)
    assert torch.equal(dG[0], dG[1])


def test_full_size_hash_indices_checksum(setup):
    """bit-exact index work at full size: the int64 (P,L,4) tensor equals the oracle on a sample, and every index < T."""
    ops, xy, n_ls, n_host, tables, vidx, vw, vs, genc = setup
    idx = ops.hash_indices(xy[: 2 ** 18], n_ls, T)
    assert int(idx.min()) >= 0 and int(idx.max()) < T
    sel = torch.randint(0, 2 ** 18, (2048,), device=DEV)
    _, grid = orc.scale_to_grid(xy[sel].cpu().numpy(), np.array(n_host, np.int32))
    assert np.array_equal(idx[sel].cpu().numpy(), orc.spatial_hash(grid.astype(np.int32), T))
