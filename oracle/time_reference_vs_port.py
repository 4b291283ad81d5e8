"""Times the PYTHON REFERENCE itself next to the C/OpenMP port (oracle/gngf_oracle_c.c) on the same hash-mode input, in the
build container (the only place the reference exists): shows that the port bench.py times as `cpu_baseline` on the GPU box
is a fair stand-in — not a straw man — for the reference's own CPU path.  TEST INFRASTRUCTURE ONLY.

    python oracle/time_reference_vs_port.py [--pixels 262144] [--threads 8]

Workload: hash indexing, L=16, F=2, T=2^19, N 16->512, P strawberry pixels;
timed = forward + loss.backward() (encoder + decoder + MSE gradient), optimizer and data loading excluded,
median of 5 after 2 warm-ups.  Result is printed as one JSON line and quoted in BASELINE.md §3."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402
import c_oracle  # noqa: E402
import gngf_oracle as orc  # noqa: E402


def median_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pixels", type=int, default=2 ** 18)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    a = ap.parse_args()
    assert rh.available(), "the reference is only present in the build container"
    torch.set_num_threads(a.threads)
    os.environ["OMP_NUM_THREADS"] = str(a.threads)
    F_, U_, M, P_ = rh.load_reference()
    rh.set_flag((F_, U_, M), "should_use_hash_function", True)
    L, T, Fd, P = 16, 2 ** 19, 2, a.pixels
    torch.manual_seed(65535)
    net = M.GeneralNeuralGaugeFields(input_dim=2, hash_table_size=T, num_levels=L, n_min=16, n_max=512,
                                     MLP_hidden_layers_widths=[64, 64], HPD_hidden_layers_widths=[32, 64, 128],
                                     HPD_out_features=T, feature_dim=Fd, topk_k=4)
    img = np.load(os.path.join(os.path.dirname(HERE), "tests", "golden", "strawberry_rgb.npz"))["img"]
    h, w = img.shape[:2]
    rng = np.random.default_rng(0)
    sel = np.concatenate([rng.permutation(h * w) for _ in range(-(-P // (h * w)))])[:P]
    x_np = np.ascontiguousarray((np.stack([sel // w, sel % w], 1) / np.float32(max(w, h) - 1)).astype(np.float32))
    y_np = np.ascontiguousarray((img.reshape(-1, 3)[sel] / 255).astype(np.float32))
    x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
    loss_fn = U_.Loss(delta=1, gamma=-2, epsilon=1)

    def ref_step():
        net.zero_grad()
        rgb, _, _, _ = net(x, 1.0, should_calc_counts=False)
        mse, _, _ = loss_fn(rgb, y, None, None, None, None)
        mse.backward()
    t_ref, all_ref = median_time(ref_step)

    # the port on the same weights and pixels
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    tables = np.ascontiguousarray(np.stack([sd[f"encoding._hash_tables.{l}.weight"] for l in range(L)]))
    dw = [np.ascontiguousarray(sd[f"mlp.{i}.0.weight"]) for i in range(3)]
    db = [np.ascontiguousarray(sd[f"mlp.{i}.0.bias"]) for i in range(3)]
    n_ls = orc.level_resolutions(16, 512, L)

    def port_step():
        enc = c_oracle.encode_fwd(x_np, tables, n_ls, None, None, 0)
        rgb, h1, h2 = c_oracle.decoder_fwd(enc, dw, db)
        grgb = ((2.0 / rgb.size) * (rgb - y_np)).astype(np.float32)
        genc, _ = c_oracle.decoder_bwd(enc, h1, h2, rgb, grgb, dw)
        return rgb, c_oracle.encode_bwd(x_np, tables, n_ls, genc, None, None, 0)
    t_port, all_port = median_time(port_step)
    # same answer (so the comparison is like for like)
    rgb_ref = net(x, 1.0)[0].detach().numpy()
    rgb_port, _ = port_step()
    print(json.dumps({
        "workload": f"hash indexing, L=16 F=2 T=2^19 N 16->512, {P} strawberry pixels, fwd + backward (encoder + decoder + MSE grad)",
        "threads": a.threads, "cpu": os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[-1].strip(),
        "reference_s": t_ref, "reference_mpix_s": P / t_ref / 1e6, "port_s": t_port, "port_mpix_s": P / t_port / 1e6,
        "port_over_reference_speed": t_ref / t_port, "max_abs_rgb_diff": float(np.abs(rgb_ref - rgb_port).max()),
        "torch": torch.__version__}))


if __name__ == "__main__":
    main()
