"""CPU: the data formats either side of the hot path follow the reference's conventions
(utils.py:12-75, main.py:50-58, functions.py:308,332-335,761-781)."""
import os

import numpy as np
import torch

from collision_handling_in_instantngp_amd import data, models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strawberry():
    return np.load(os.path.join(ROOT, "tests", "golden", "strawberry_rgb.npz"))["img"]


def test_dataset_pixel_list_is_row_major_row_col():
    img = _strawberry()
    X, Y, height, width = data.MyDataset(image=img)[0]
    assert (height, width) == img.shape[:2] and X.shape == (height * width, 2) and Y.shape == (height * width, 3)
    assert X.dtype == torch.float32 and Y.dtype == torch.float32
    # the reference builds X with a Python loop over meshgrid(range(height), range(width), 'ij')
    want = np.array([(r, c) for r in range(3) for c in range(width)], dtype=np.float32)
    assert np.array_equal(X[:3 * width].numpy(), want)
    assert tuple(X[-1].tolist()) == (height - 1, width - 1)
    p = 5 * width + 7
    assert np.array_equal(Y[p].numpy(), (img[5, 7].astype(np.float64) / 255).astype(np.float32))
    assert float(Y.max()) <= 1.0 and float(Y.min()) >= 0.0


def test_dataset_reads_an_image_file(tmp_path):
    from PIL import Image
    img = _strawberry()[:40, :56]
    os.makedirs(tmp_path / "images")
    Image.fromarray(img).save(tmp_path / "images" / "crop.png")         # lossless
    ds = data.MyDataset(root=str(tmp_path), dir_name="images", image_name="crop.png")
    X, Y, h, w = ds[0]
    assert (h, w) == (40, 56) and len(ds) == 1 and ds.get_image_name() == "crop.png"
    assert np.array_equal(ds.get_image(), img)
    assert np.array_equal((Y.numpy() * 255).round().astype(np.uint8).reshape(40, 56, 3), img)


def test_black_and_white_uses_opencv_luma():
    img = _strawberry()[:16, :16]
    X, Y, h, w = data.MyDataset(image=img, should_bw=True)[0]
    assert Y.shape == (256, 1)
    luma = 0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]
    assert np.abs(Y.numpy().reshape(16, 16) * 255 - luma).max() <= 0.5 + 1e-3
    grey = np.full((2, 2, 3), 200, dtype=np.uint8)
    assert np.array_equal(data.to_grayscale(grey), np.full((2, 2), 200, dtype=np.uint8))


def test_normalise_and_permutation_roundtrip():
    h, w = 339, 508
    X = torch.from_numpy(data.pixel_grid(h, w)).float()
    xn = data.normalise_coordinates(X, w, h)
    assert float(xn.max()) == 1.0 and float(xn[:, 0].max()) == float(np.float32(h - 1) / np.float32(w - 1))
    g = torch.Generator().manual_seed(3)
    shuffled, reordered = data.make_permutation(h * w, g)
    assert shuffled.dtype == torch.int32 and reordered.dtype == torch.int32
    assert torch.equal(shuffled[reordered.long()].long(), torch.arange(h * w))
    assert torch.equal(reordered[shuffled.long()].long(), torch.arange(h * w))


def test_reassemble_inverts_the_shuffle_and_truncates():
    img = _strawberry()[:20, :30]
    h, w = img.shape[:2]
    Y = torch.from_numpy(img.reshape(-1, 3).astype(np.float64) / 255).float()
    shuffled, reordered = data.make_permutation(h * w, torch.Generator().manual_seed(1))
    batch_order = Y[shuffled.long()]                          # what train_step's `outputs` holds
    out = data.reassemble_image(batch_order, reordered, h, w)
    assert out.dtype == np.int32 and out.shape == (h, w, 3)
    # (v / 255) * 255 in fp32 can land just below the integer: the reference truncates, it does not round
    want = (Y * 255).reshape(h, w, 3).int().numpy()
    assert np.array_equal(out, want)
    assert np.abs(out - img.astype(np.int32)).max() <= 1
    t = torch.tensor([[0.9999, 0.5, 1.2], [-0.1, 0.0, 1.0]])
    got = data.reassemble_image(t, None, 1, 2, should_shuffle=False)
    assert got.tolist() == [[[254, 127, 306], [-25, 0, 255]]]
    bw = data.reassemble_image(torch.tensor([[0.5], [1.0]]), None, 2, 1, should_bw=True, should_shuffle=False)
    assert bw.shape == (2, 1) and bw.tolist() == [[127], [255]]


def test_checkpoint_files_roundtrip(tmp_path):
    old = models.should_use_hash_function
    models.should_use_hash_function = False
    try:
        def make(seed):
            torch.manual_seed(seed)
            return models.GeneralNeuralGaugeFields(2, 64, 2, 4, 8, [16, 16], [8], 64, feature_dim=2, topk_k=2)
        net = make(0)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        paths = data.save_checkpoint(net, opt, str(tmp_path / "weights"))
        assert sorted(os.path.basename(p) for p in paths.values()) == sorted(
            ["whole_model.pt", "whole_opt.pt", "encoding_model.pt", "HPD_model.pt", "MLP_model.pt"])
        assert all(os.path.exists(p) for p in paths.values())
        other = make(1)
        assert any(not torch.equal(a, b) for a, b in zip(net.state_dict().values(), other.state_dict().values()))
        data.load_checkpoint(other, str(tmp_path / "weights"), parts=("encoding", "HPD", "mlp"), map_location="cpu")
        for (ka, a), (kb, b) in zip(net.state_dict().items(), other.state_dict().items()):
            assert ka == kb and torch.equal(a.cpu(), b.cpu()), ka
        third = make(2)
        data.load_checkpoint(third, str(tmp_path / "weights"), optimizer=torch.optim.Adam(third.parameters(), lr=1e-3),
                             map_location="cpu")
        assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(net.state_dict().values(), third.state_dict().values()))
        enc_keys = set(torch.load(paths["encoding"]).keys())
        assert enc_keys == {"_hash_tables.0.weight", "_hash_tables.1.weight"}
    finally:
        models.should_use_hash_function = old
