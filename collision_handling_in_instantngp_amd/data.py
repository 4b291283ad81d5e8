"""Data formats either side of the hot path, mirrored from the reference so that a training script
written against it finds the same names and conventions:

  MyDataset            utils.py:12-75      image file -> (X, Y, height, width); X = integer (row, col) pixel list in
                                           row-major order, Y = channels / 255
  normalise / permutation                  main.py:50-58: x / (max(w, h) - 1); shuffled_indices and their inverse
  reassemble_image     functions.py:308,332-335: inverse permutation, (output * 255) -> int32 image (truncation)
  save_checkpoint / load_checkpoint        functions.py:761-781, models.py HPD/encoding weight paths: the five
                                           state-dict files of the reference, loadable in either direction

Host-side numpy/torch only: nothing here touches the GPU kernels."""
import os
from typing import Optional, Tuple

import numpy as np
import torch

CHECKPOINT_FILES = {            # functions.py:768-781
    "model": "whole_model.pt",
    "optimizer": "whole_opt.pt",
    "encoding": "encoding_model.pt",
    "HPD": "HPD_model.pt",
    "mlp": "MLP_model.pt",
}


def pixel_grid(height: int, width: int) -> np.ndarray:
    """(height*width, 2) integer (row, col) of every pixel, rows outermost (utils.py:56-59: meshgrid(range(height),
    range(width), indexing='ij') flattened)."""
    rows = np.repeat(np.arange(height, dtype=np.int64), width)
    cols = np.tile(np.arange(width, dtype=np.int64), height)
    return np.stack([rows, cols], axis=1)


def to_grayscale(rgb: np.ndarray) -> np.ndarray:
    """uint8 luma with OpenCV's COLOR_BGR2GRAY weights (0.299 R + 0.587 G + 0.114 B, 14-bit fixed point, rounded)."""
    r, g, b = (rgb[..., c].astype(np.int64) for c in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


class MyDataset(torch.utils.data.Dataset):
    """reference utils.py:12-75.  One image; `dataset[0]` -> (X float32 (P,2), Y float32 (P,C), height, width).
    The image is decoded with Pillow (cv2 is not a dependency here); `image=` takes an already decoded (H,W,3) uint8
    RGB array instead of a file."""

    def __init__(self, root: str = ".", dir_name: str = "images", image_name: str = "", should_bw: bool = False,
                 image: Optional[np.ndarray] = None) -> None:
        self._root = root
        self._dir_name = dir_name
        self._image_name = image_name
        self._should_bw = should_bw
        self._image_path = os.path.join(os.path.join(self._root, self._dir_name), self._image_name)
        self._source = None if image is None else np.asarray(image)
        self._image = None

    def _load_rgb(self) -> np.ndarray:
        if self._source is not None:
            return self._source[:, :, :3]
        from PIL import Image
        with Image.open(self._image_path) as im:
            return np.asarray(im.convert("RGB"))

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
        rgb = self._load_rgb()
        self._image = to_grayscale(rgb) if self._should_bw else np.ascontiguousarray(rgb)
        height, width = self._image.shape[0], self._image.shape[1]
        X = torch.from_numpy(pixel_grid(height, width)).float()
        Y = torch.from_numpy(self._image.reshape(height * width, -1).astype(np.float64) / 255).float()
        return X, Y, height, width            # height, width in this order, as the reference returns them

    def __len__(self) -> int:
        return 1

    def get_image(self) -> np.ndarray:
        return self._image

    def get_image_name(self) -> str:
        return self._image_name


def normalise_coordinates(x: torch.Tensor, w: int, h: int) -> torch.Tensor:
    """main.py:50-51: pixel indices -> [0, 1] by the longer side."""
    return x / (max(w, h) - 1)


def make_permutation(shape: int, generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """main.py:55-58: (shuffled_indices, reordered_indices) as int32, reordered[shuffled] = arange."""
    shuffled = torch.randperm(shape, generator=generator).int()
    reordered = torch.zeros((shape,)).int()
    reordered[shuffled.long()] = torch.arange(shape).int()
    return shuffled, reordered


def reassemble_image(outputs: torch.Tensor, reordered_indices: Optional[torch.Tensor], h: int, w: int,
                     should_bw: bool = False, should_shuffle: bool = True) -> np.ndarray:
    """functions.py:308,332-335: batch-order outputs -> image order -> (output * 255) as int32 (h,w,3) / (h,w)
    (torch's float -> int conversion truncates toward zero; values are not clamped, as in the reference)."""
    out = outputs[reordered_indices.long()] if should_shuffle else outputs
    img = (out * 255).reshape((h, w, 3) if not should_bw else (h, w))
    return img.int().detach().cpu().numpy()


def save_checkpoint(net, optimizer, folder: str) -> dict:
    """functions.py:764-781: the whole model, the optimizer and the three sub-modules as separate state dicts."""
    os.makedirs(folder, exist_ok=True)
    paths = {k: os.path.join(folder, v) for k, v in CHECKPOINT_FILES.items()}
    torch.save(net.state_dict(), paths["model"])
    if optimizer is not None:
        torch.save(optimizer.state_dict(), paths["optimizer"])
    torch.save(net.encoding.state_dict(), paths["encoding"])
    if getattr(net, "HPD", None) is not None:
        torch.save(net.HPD.state_dict(), paths["HPD"])
    torch.save(net.mlp.state_dict(), paths["mlp"])
    return paths


def load_checkpoint(net, folder: str, optimizer=None, parts=("model",), map_location=None) -> None:
    """Loads `whole_model.pt` (parts=('model',)) or any of the per-module files ('encoding', 'HPD', 'mlp'), and the
    optimizer state when given.  Files written by the reference load as well: parameter names are the same."""
    for part in parts:
        state = torch.load(os.path.join(folder, CHECKPOINT_FILES[part]), map_location=map_location)
        target = net if part == "model" else getattr(net, part)
        target.load_state_dict(state)
    if optimizer is not None:
        optimizer.load_state_dict(torch.load(os.path.join(folder, CHECKPOINT_FILES["optimizer"]), map_location=map_location))
