"""Callers of the hot path, mirrored from the reference so that parity tests read like the reference's own
training loop: Loss (utils.py:78-174), get_optimizer (functions.py:96-127), train_step
(functions.py:139-355, batching + loss assembly only; no wandb / plotting / collision diagnostics)."""
import numpy as np
import torch

from . import models, ops
from .models import VertexDistribution


class Loss(torch.nn.Module):
    """reference utils.py:78-174.  forward(pred, labels, N, prob, collisions, min_possible_collisions)
    -> (mse, kl_divs (L,), collisions_losses).  `prob` is the dense (P,L,4,N) tensor or a VertexDistribution
    (its .pbar is the batch-mean distribution the reference computes at utils.py:138,159)."""

    def __init__(self, delta: float = 1, gamma: float = 1, epsilon: float = 0.5, should_log: bool = False) -> None:
        super().__init__()
        self._delta = delta
        self._gamma = gamma
        self._epsilon = epsilon
        self.mse = torch.nn.MSELoss()

    @staticmethod
    def _kldiv_batchmean(log_input, target):
        # torch.nn.KLDivLoss(reduction='batchmean') on (L,N) rows treated independently as 1-D inputs: /N
        return torch.xlogy(target, target).sub(target * log_input).sum(-1) / log_input.shape[-1]

    def js_kl_rows(self, pbar):
        """pbar (L,N) -> (L,) : -(gamma+eps)*JS + eps*KL per level (utils.py:122-174)."""
        N = pbar.shape[-1]
        q = torch.full_like(pbar, 1.0 / N)
        lp = pbar.log()
        kl = self._kldiv_batchmean(lp, q)
        m = (pbar + q) / 2
        js = (self._kldiv_batchmean(lp, m) + self._kldiv_batchmean(q.log(), m)) / 2
        return -(self._gamma + self._epsilon) * js + self._epsilon * kl

    def _mse(self, pred, labels):
        # device tensors: the two-launch HIP kernels; host tensors (the CPU tests of the loss algebra): the torch module
        if pred.is_cuda and pred.dtype == torch.float32 and labels.dtype == torch.float32 and pred.shape == labels.shape:
            return ops.mse_loss(pred, labels)
        return self.mse(pred, labels)

    def forward(self, pred, labels, N, prob, collisions, min_possible_collisions):
        mse_loss = self._mse(pred, labels)
        if models.should_use_hash_function:
            return mse_loss, None, None
        collisions_losses = collisions / (min_possible_collisions + self._delta)
        if isinstance(prob, VertexDistribution):
            pbar = prob.pbar
        else:
            pbar = prob.sum(0).sum(1) / (prob.shape[0] * prob.shape[2])          # (L,N)
        return mse_loss, self.js_kl_rows(pbar), collisions_losses


def get_optimizer(net, encoding_lr, HPD_lr, MLP_lr, encoding_weight_decay, HPD_weight_decay, MLP_weight_decay,
                  betas=(0.9, 0.99), eps=1e-15):
    """reference functions.py:96-127."""
    groups = [{"params": net.encoding.parameters(), "lr": encoding_lr, "weight_decay": encoding_weight_decay}]
    if not models.should_use_hash_function:
        groups.append({"params": net.HPD.parameters(), "lr": HPD_lr, "weight_decay": HPD_weight_decay})
    groups.append({"params": net.mlp.parameters(), "lr": MLP_lr, "weight_decay": MLP_weight_decay})
    # same update rule as the reference's torch.optim.Adam; the single-kernel ("fused") implementation moves the
    # 64 MiB of tables + moments in ~0.1 ms per step at cfg2 instead of ~0.3 ms for the per-op default
    on_gpu = all(p.is_cuda for p in net.parameters())
    return torch.optim.Adam(groups, betas=betas, eps=eps, **({"fused": True} if on_gpu else {}))


def assemble_loss(mse, kls, colls, l_mse, l_js_kl, l_collisions):
    """reference functions.py:243-245, including the '+1 per level while previous_collisions is empty' quirk."""
    loss = l_mse * mse
    if not models.should_use_hash_function:
        loss = loss + ((l_js_kl * kls) + (l_collisions * colls if colls.nelement() != 0 else 1)).sum(0)
    return loss


def train_step(net, loss_fn, optimizer, x, target, w, h, l_mse, l_js_kl, l_collisions, batch_percentage=1.0,
               should_shuffle=True, shuffled_indices=None, previous_collisions=None, previous_min_possible_collisions=None):
    """One epoch = ceil(1/batch_percentage) mini-batches of zero_grad -> net -> Loss -> weighted sum -> backward -> step
    (reference functions.py:183-281).  Returns (mean loss, mean mse, outputs (P,3) in batch order)."""
    net.train()
    shape = w * h
    num_batches = int(np.ceil(shape / (shape * batch_percentage)))
    step = int(batch_percentage * shape)
    dev = x.device
    empty = torch.tensor([], device=dev)
    previous_collisions = empty if previous_collisions is None else previous_collisions
    previous_min_possible_collisions = empty if previous_min_possible_collisions is None else previous_min_possible_collisions
    outputs = torch.empty((shape, target.shape[1]), device=dev)
    losses, mses = [], []
    for b in range(num_batches):
        lo, hi = b * step, (b + 1) * step
        sel = shuffled_indices[lo:hi].long() if should_shuffle else slice(lo, hi)
        bx, by = x[sel], target[sel]
        optimizer.zero_grad()
        out, probs, _idx, _counts = net(bx, batch_percentage, should_calc_counts=False)
        outputs[lo:hi] = out.detach()
        mse, kls, colls = loss_fn(out, by, None if probs is None else probs.shape[-1], probs,
                                  previous_collisions, previous_min_possible_collisions)
        loss = assemble_loss(mse, kls, colls, l_mse, l_js_kl, l_collisions)
        loss.backward()
        optimizer.step()
        losses.append(loss.detach())
        mses.append(mse.detach())
    return torch.stack(losses).mean().item(), torch.stack(mses).mean().item(), outputs


def calc_psnr(pred: np.ndarray, target: np.ndarray) -> float:
    """reference functions.py:134-136 (peak = max(target))."""
    mse = np.square(pred - target).mean()
    return 20 * np.log10(np.max(target)) - 10 * np.log10(mse)
