"""CPU: the C/OpenMP oracle (cpu_baseline port) agrees with the numpy oracle, which is pinned to the reference goldens."""
import numpy as np
import pytest

from oracle import c_oracle, gngf_oracle as orc

pytestmark = pytest.mark.skipif(not c_oracle.available(), reason="oracle/libgngf_oracle_c.so not built (make -C oracle)")


@pytest.mark.parametrize("mode", ["hash", "vt"])
def test_c_encoder_matches_numpy_oracle(mode):
    rng = np.random.default_rng(4)
    L, T, F, K, P, n_max = 6, 1000, 2, 3, 3000, 64
    n_ls = orc.level_resolutions(8, n_max, L)
    x = rng.random((P, 2), dtype=np.float32)
    x[:3] = [[0, 0], [1, 1], [0.5, 1]]
    tables = ((rng.random((L, T, F), dtype=np.float32) - 0.5) * 2e-4).astype(np.float32)
    g = rng.standard_normal((P, L * F)).astype(np.float32)
    _, grid = orc.scale_to_grid(x, n_ls)
    if mode == "hash":
        idx, w, vi, vw, vs = orc.spatial_hash(grid.astype(np.int32), T), None, None, None, 0
    else:
        vs = n_max + 2
        vi = rng.integers(0, T, (vs * vs, K)).astype(np.int32)
        vw = rng.random((vs * vs, K), dtype=np.float32)
        gi = grid.astype(np.int64)
        vid = gi[:, 1] * vs + gi[:, 0]
        idx, w = vi[vid].astype(np.int64), vw[vid]
    want = orc.bilinear_forward(x, n_ls, orc.encoding_forward(tables, idx, w, None))
    got = c_oracle.encode_fwd(x, tables, n_ls, vi, vw, vs)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-10)
    dt, _ = orc.encoding_backward(tables, idx, w, None, orc.bilinear_backward(x, n_ls, g, F))
    got_dt = c_oracle.encode_bwd(x, tables, n_ls, g, vi, vw, vs)
    np.testing.assert_allclose(got_dt, dt, rtol=2e-4, atol=2e-5 * np.abs(dt).max())
    # the wide-accumulation variant: the same fp32 terms, summed in double (the numpy oracle's np.add.at sums in fp32)
    got64 = c_oracle.encode_bwd_f64(x, tables.shape, n_ls, g, vi, vw, vs)
    np.testing.assert_allclose(got64, dt, rtol=2e-4, atol=2e-5 * np.abs(dt).max())
    np.testing.assert_allclose(got64, got_dt, rtol=1e-4, atol=1e-5 * np.abs(dt).max())
    # ... against an independent float64 evaluation of the sum of fp32 terms (numpy, np.add.at in double)
    want64 = np.zeros(tables.shape, np.float64)
    scaled, grid = orc.scale_to_grid(x, n_ls)
    dfe = orc.bilinear_backward(x, n_ls, g, F)                       # (P,F,L,4): the fp32 products genc * c
    for l in range(L):
        for v in range(4):
            if mode == "hash":
                np.add.at(want64[l], idx[:, l, v], dfe[:, :, l, v].astype(np.float64))
            else:
                for k in range(K):
                    term = (dfe[:, :, l, v] * w[:, l, v, k][:, None]).astype(np.float32)
                    np.add.at(want64[l], idx[:, l, v, k], term.astype(np.float64))
    np.testing.assert_allclose(got64, want64, rtol=1e-12, atol=1e-18)
    # exact products: the mathematically exact gradient of the fp32 inputs; differs from the fp32-term sum by <= 1.2e-7 sum |term|
    exact = c_oracle.encode_bwd_f64(x, tables.shape, n_ls, g, vi, vw, vs, exact_products=True)
    np.testing.assert_allclose(exact, got64, rtol=0, atol=2e-6 * np.abs(dt).max())
    assert np.abs(exact - got64).max() > 0


def test_c_decoder_matches_numpy_oracle():
    rng = np.random.default_rng(5)
    P, dims = 500, [32, 64, 64, 3]
    W = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    B = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
    x = rng.standard_normal((P, 32)).astype(np.float32)
    gy = rng.standard_normal((P, 3)).astype(np.float32)
    y, h1, h2 = c_oracle.decoder_fwd(x, W, B)
    np.testing.assert_allclose(y, orc.decoder_forward(x, W, B), rtol=1e-5, atol=1e-6)
    dx, g = c_oracle.decoder_bwd(x, h1, h2, y, gy, W)
    wdx, wdW, wdB = orc.decoder_backward(x, W, B, gy)
    np.testing.assert_allclose(dx, wdx, rtol=1e-4, atol=1e-6)
    for i in range(3):
        np.testing.assert_allclose(g[2 * i], wdW[i], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(g[2 * i + 1], wdB[i], rtol=1e-4, atol=1e-5)
