"""CPU (build container only): the REFERENCE'S OWN DRIVER runs against the drop-in boundary.

`/root/reference/functions.py` (imported untouched, as oracle/ref_harness.py does) is given the classes of
`collision_handling_in_instantngp_amd/reference_shim/models.py` under the name `models`, and its `train_step`
(functions.py:139-355) plus the checkpoint block (functions.py:761-781) are driven for one epoch in both indexing modes.
There is no GPU here, and the product has no CPU path — so, IN THIS TEST ONLY, the handful of `ops` entry points the module
classes call are replaced by plain-torch restatements of the same per-vertex formulation (test infrastructure, like the
oracle; the product never sees them).  What this pins is the CALL CONTRACT that nothing else exercises:
  * the constructor called with the reference's keywords (functions.py:540-556), `get_optimizer` reaching `.encoding`, `.HPD`,
    `.mlp` (functions.py:96-127), `net(batch_x, batch_percentage, should_calc_counts=...)` and its 4-tuple (functions.py:203);
  * `batch_indices_topk` being a `torch.empty` FLOAT buffer with K of 3 K slots written per pixel (functions.py:179,216) that is
    handed to `net.calc_hash_collisions` (functions.py:327);
  * the five `state_dict()` calls of the checkpoint block — and that in hash mode `GGNF_model.HPD` does not exist, in the
    reference itself as here (AttributeError at functions.py:775 on both sides);
  * the numbers: the epoch's loss, MSE, reconstructed image and parameters after the three optimizer steps equal those of the
    reference's own classes from the same initial weights.
The reference's files stay in /root/reference; nothing of them is copied."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "collision_handling_in_instantngp_amd", "reference_shim")
REF = os.environ.get("GNGF_REFERENCE_ROOT", "/root/reference")

SCRIPT = textwrap.dedent(r'''
    import importlib, os, sys, types
    import numpy as np
    import torch
    import matplotlib
    matplotlib.use("Agg")
    MODE, SHIM, ROOT, REF = sys.argv[1:5]
    hash_mode = MODE == "hash"
    for name in ("cv2", "wandb"):
        sys.modules[name] = types.ModuleType(name)
    tv = types.ModuleType("torchvision"); tv.io = types.ModuleType("torchvision.io"); tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.update({"torchvision": tv, "torchvision.io": tv.io, "torchvision.transforms": tv.transforms})
    sys.path[:0] = [ROOT, REF]
    real = torch.set_default_device
    torch.set_default_device = lambda *a, **k: None            # functions.py:52 asks for 'cuda'
    try:
        functions = importlib.import_module("functions")      # the reference's, untouched
        utils = importlib.import_module("utils")
        params = importlib.import_module("params")
        ref_models = importlib.import_module("models")        # the reference's own classes: the numbers to match
    finally:
        torch.set_default_device = real
    cpu = torch.device("cpu")
    for m_ in (functions, utils, ref_models):
        m_.device = cpu
        m_.should_use_hash_function = hash_mode
    params.should_use_hash_function = hash_mode
    # ... and the replacement, under its own module name (the shim reads `params` of the caller at call time)
    del sys.modules["models"]
    sys.path.insert(0, SHIM)
    shim = importlib.import_module("models")
    assert shim.__file__.startswith(SHIM), shim.__file__
    from collision_handling_in_instantngp_amd import models as hip_models, ops

    # ------------------------------------------------------------------ plain-torch stand-ins for the HIP entry points (TEST ONLY)
    def cells(xy, n_ls):
        n = n_ls.to(torch.float32)
        s = xy[:, :, None] * n[None, None, :]                    # (P,2,L)  models.py:492-495, separately rounded ops
        a = torch.floor(s)
        d = a + 1.0
        w0, w1 = d - s, s - a
        c = torch.stack([w0[:, 0] * w0[:, 1], w1[:, 0] * w0[:, 1], w0[:, 0] * w1[:, 1], w1[:, 0] * w1[:, 1]], -1)   # (P,L,4)
        g = a.to(torch.int64)
        gx = torch.stack([g[:, 0], g[:, 0] + 1, g[:, 0], g[:, 0] + 1], -1)      # (P,L,4): corner order v = dx + 2 dy
        gy = torch.stack([g[:, 1], g[:, 1], g[:, 1] + 1, g[:, 1] + 1], -1)
        return c, gx, gy

    def spatial_hash(gx, gy, T):
        h = gx.to(torch.int32) ^ (gy.to(torch.int32) * torch.tensor(2654435761, dtype=torch.int64).to(torch.int32))   # int32 wrap
        return torch.remainder(h.to(torch.int64), T)

    def encode_apply(xy, n_ls, n_ls_host, tables, vert_idx, vert_w, vstride, path=None, order=None, dp=None, link=None, sink=None):
        xy = xy.detach()
        L, T, F = tables.shape
        c, gx, gy = cells(xy, n_ls)
        lev = torch.arange(L)[None, :, None]
        if vert_idx is None:
            feats = tables[lev, spatial_hash(gx, gy, T)]                             # (P,L,4,F)
        else:
            vid = gy * vstride + gx
            rows = tables[lev[..., None], vert_idx.to(torch.int64)[vid]]            # (P,L,4,K,F)
            feats = (rows * vert_w[vid][..., None]).sum(3)
        f = feats
        enc = ((f[:, :, 0] * c[:, :, 0, None] + f[:, :, 1] * c[:, :, 1, None]) + f[:, :, 2] * c[:, :, 2, None]) + f[:, :, 3] * c[:, :, 3, None]
        return enc.reshape(xy.shape[0], L * F)                                          # "p f l -> p (l f)"  models.py:651

    def decoder_apply(enc, acts, params_, fused=None, mse_target=None, mse_gloss=None, link=None):
        h = enc
        for i, a in enumerate(acts):
            h = torch.nn.functional.linear(h, params_[2 * i], params_[2 * i + 1])
            h = {ops.ACT_RELU: torch.relu, ops.ACT_LEAKY: lambda z: torch.nn.functional.leaky_relu(z, 0.01), ops.ACT_SIGMOID: torch.sigmoid,
                 ops.ACT_NONE: lambda z: z}[a](h)
        return h

    class HpdVertex:
        @staticmethod
        def apply(NV, vstride, K, mw, keep_probs, chunk_bytes, aux, *params_):
            u = torch.arange(NV)
            x = torch.stack([u % vstride, u // vstride], 1).to(torch.float32)       # the raw integer vertex (models.py:416-418)
            n = len(params_) // 2
            for i in range(n):
                x = torch.nn.functional.linear(x, params_[2 * i], params_[2 * i + 1])
                if i < n - 1:
                    x = torch.relu(x)
            probs = torch.nan_to_num(torch.softmax(x, -1))                             # models.py:85,111
            tv_, ti_ = torch.topk(probs, K, dim=-1)
            pbar = (mw.T @ probs) if mw is not None else None
            return tv_, ti_.to(torch.int32), pbar, (probs if keep_probs else None)

    class Blend:
        @staticmethod
        def apply(q, code):
            return {0: lambda t: torch.softmax(t, -1), 1: lambda t: t, 2: lambda t: t / t.sum(-1, keepdim=True)}[code](q)

    def vertex_multiplicity_weights(xy, n_ls, vstride, NV):
        c, gx, gy = cells(xy, n_ls)
        L = n_ls.numel()
        vid = (gy * vstride + gx)
        mw = torch.zeros((NV, L))
        for l in range(L):
            mw[:, l] = torch.bincount(vid[:, l].reshape(-1), minlength=NV).to(torch.float32)
        return mw / float(4 * xy.shape[0])

    def expand_vertex_table(xy, n_ls, vstride, NV, src_idx=None, src_val=None, want_vid=False):
        c, gx, gy = cells(xy, n_ls)
        vid = gy * vstride + gx
        return (vid if want_vid else None, src_idx.to(torch.int64)[vid] if src_idx is not None else None,
                src_val[vid] if src_val is not None else None)

    def hash_indices(xy, n_ls, T):
        c, gx, gy = cells(xy, n_ls)
        return spatial_hash(gx, gy, T)

    ops.encode_apply, ops.decoder_apply, ops.HpdVertexFunction, ops.BlendFunction = encode_apply, decoder_apply, HpdVertex, Blend
    ops.vertex_multiplicity_weights, ops.expand_vertex_table, ops.hash_indices = vertex_multiplicity_weights, expand_vertex_table, hash_indices
    ops.mse_loss = lambda pred, label: torch.nn.functional.mse_loss(pred, label)
    ops.slot_order = lambda *a, **k: None

    # ------------------------------------------------------------------ the reference's driver, twice
    img = np.load(os.path.join(ROOT, "tests", "golden", "strawberry_rgb.npz"))["img"][100:136, 60:108]     # a 36 x 48 crop
    h, w = img.shape[:2]
    rows, cols = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    X = torch.tensor(np.stack([rows, cols], -1).reshape(-1, 2)).float() / (max(w, h) - 1)        # utils.py:56-59, main.py:50-51
    Y = torch.tensor(img.reshape(-1, 3) / 255).float()                                           # utils.py:61
    shape = w * h
    torch.manual_seed(3)
    shuffled = torch.randperm(shape).int()                                                       # main.py:54-58
    reordered = torch.zeros((shape,)).int()
    reordered[shuffled] = torch.arange(shape).int()
    K, T, L = 4, 256, 4
    kw = dict(input_dim=X.shape[1], hash_table_size=T, num_levels=L, n_min=8, n_max=32, MLP_hidden_layers_widths=[64, 64],
              HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=T, feature_dim=2, topk_k=K, should_keep_topk_only=False,
              should_bw=False, should_log=False, HPD_weights_path=None, encoding_weights_path=None)      # functions.py:540-556

    def run(cls, init):
        net = cls(**kw)
        if init is not None:
            net.load_state_dict(init)
        loss_fn = utils.Loss(delta=1, gamma=-2, epsilon=1, should_log=False)                     # functions.py:561-566
        opt = functions.get_optimizer(net=net, encoding_lr=1e-4, HPD_lr=1e-3, MLP_lr=1e-3, encoding_weight_decay=0,
                                      HPD_weight_decay=1e-6, MLP_weight_decay=1e-6)              # functions.py:568-576
        out = functions.train_step(net, loss_fn, opt, X.clone(), Y.clone(), w, h, T, K, 1, 1, 1e-3, batch_percentage=params.batch_size,
                                   num_levels=L, should_bw=False, should_calc_counts=False, should_shuffle=True,
                                   shuffled_indices=shuffled, reordered_indices=reordered,
                                   previous_collisions=torch.tensor([]), previous_min_possible_collisions=torch.tensor([]))
        # the checkpoint block (functions.py:767-780)
        saved = {"whole": net.state_dict(), "opt": opt.state_dict(), "encoding": net.encoding.state_dict(), "mlp": net.mlp.state_dict()}
        try:
            saved["HPD"] = net.HPD.state_dict()
        except AttributeError:
            saved["HPD"] = "AttributeError"
        return net, out, saved

    init_net = ref_models.GeneralNeuralGaugeFields(**kw)
    init = {k: v.clone() for k, v in init_net.state_dict().items()}
    ref_net, ref_out, ref_saved = run(ref_models.GeneralNeuralGaugeFields, init)
    our_net, our_out, our_saved = run(shim.GeneralNeuralGaugeFields, init)
    assert isinstance(our_net, hip_models.GeneralNeuralGaugeFields)

    names = ["loss", "image", "collisions", "min_possible_collisions", "counts_per_level", "mse", "kl_div_losses", "collisions_losses", "indices_per_level"]
    assert len(ref_out) == len(our_out) == 9
    np.testing.assert_allclose(our_out[0], ref_out[0], rtol=2e-5, err_msg="loss of the epoch")
    np.testing.assert_allclose(our_out[5], ref_out[5], rtol=2e-5, err_msg="mse of the epoch")
    assert our_out[1].shape == ref_out[1].shape == (h, w, 3) and our_out[1].dtype == ref_out[1].dtype
    assert np.abs(our_out[1].astype(np.int64) - ref_out[1].astype(np.int64)).max() <= 1, "reconstructed image (int, 0..255)"
    if hash_mode:
        assert our_out[6] is None and our_out[7] is None and ref_out[6] is None
        assert our_saved["HPD"] == ref_saved["HPD"] == "AttributeError"          # functions.py:775 fails in the reference itself in hash mode
    else:
        np.testing.assert_allclose(our_out[6], ref_out[6], rtol=2e-4, atol=1e-7, err_msg="JS/KL terms of the epoch")
        assert sorted(our_saved["HPD"]) == sorted(ref_saved["HPD"])
    # collisions: computed from the torch.empty buffer (partly uninitialised in both): contract = type and shape only
    assert tuple(our_out[2].shape) == tuple(ref_out[2].shape) and tuple(our_out[3].shape) == tuple(ref_out[3].shape)
    for part in ("whole", "encoding", "mlp"):
        assert list(our_saved[part]) == list(ref_saved[part]), part                # same keys, same order
    for k, v in ref_saved["whole"].items():
        got = our_saved["whole"][k]
        assert got.shape == v.shape and got.dtype == v.dtype, k
        if v.dtype.is_floating_point:
            scale = float(v.abs().max()) + 1e-12
            assert float((got - v).abs().max()) <= 2e-3 * scale, (k, float((got - v).abs().max()), scale)   # three Adam steps (eps 1e-15)
    assert [g["lr"] for g in our_saved["opt"]["param_groups"]] == [g["lr"] for g in ref_saved["opt"]["param_groups"]]
    assert [len(g["params"]) for g in our_saved["opt"]["param_groups"]] == [len(g["params"]) for g in ref_saved["opt"]["param_groups"]]
    print("DRIVER OK", MODE)
''')


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["gngf", "hash"])
def test_reference_train_step_and_checkpoint_block_run_against_the_boundary(tmp_path, mode):
    if not os.path.isfile(os.path.join(REF, "functions.py")):
        pytest.skip("reference not present (GPU box): this rehearsal runs in the build container only")
    script = tmp_path / "drive.py"
    script.write_text(SCRIPT)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(script), mode, SHIM, ROOT, REF], env=env, capture_output=True, text=True, timeout=580,
                       cwd=str(tmp_path))
    assert r.returncode == 0 and f"DRIVER OK {mode}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
