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
        """pbar (L,N) -> (L,) : -(gamma+eps)*JS + eps*KL per level (utils.py:122-174).  Device fp32 distributions run
        on the HIP kernels (ops.JsKlFunction); the torch expression below is the same formula for host tensors."""
        if pbar.is_cuda and pbar.dtype == torch.float32 and pbar.dim() == 2:
            return ops.js_kl_rows(pbar, self._gamma, self._epsilon)
        return self.js_kl_rows_torch(pbar)

    def js_kl_rows_torch(self, pbar):
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


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (amsgrad off; the reference's get_optimizer, functions.py:96-127) for every tensor of
    every parameter group in ONE launch of the HIP kernel behind `gngf_adam_step` (csrc/optim.hip).  State keeps
    torch.optim.Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so optimizer state dicts written by the
    reference's torch.optim.Adam load here and vice versa.  The step count is a device tensor advanced by the kernel:
    a step needs no host synchronisation and can be captured in a hipGraph together with forward and backward."""

    DTYPES = (torch.float32,)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 4:
            raise ValueError("at most 4 parameter groups (GNGF_ADAM_MAX_GROUPS)")
        if len({(g["betas"], g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("betas and eps are shared by all groups (as in the reference's get_optimizer)")
        self._step = None            # device float: steps taken
        self._table = None           # device byte tensor: the packed segment list
        self._host = None            # ring of [pinned staging buffer, event of the upload that last read it]
        self._slot = 0
        self._captured = []          # staging buffers owned by captured steps (read again at every replay)
        self._spares = []            # pinned buffers set aside for captures (no pinned allocation while capturing)
        self._table_key = None
        self._total_blocks = 0

    def _segments(self):
        segs = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdam handles dense fp32 device parameters")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self._step is None:
                    prev = st.get("step", 0.0)
                    self._step = torch.full((), float(prev), dtype=torch.float32, device=p.device)
                st["step"] = self._step          # one shared counter (torch keeps one equal copy per parameter)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                for t_ in (p, st["exp_avg"], st["exp_avg_sq"]):
                    if not t_.is_contiguous():
                        raise RuntimeError("FusedAdam needs contiguous parameters and moments")
                segs.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi, g))
        return segs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        import ctypes
        from ._lib import call, ptr, query, stream_ptr
        segs = self._segments()
        if not segs:
            return loss
        key = tuple(s_[:6] for s_ in segs)
        capturing = torch.cuda.is_current_stream_capturing()
        table, total_blocks = self._table, self._total_blocks
        if capturing or key != self._table_key:    # pointers moved (first step, new gradient buffers, loaded state)
            blk = query("gngf_adam_block_elems")
            rec = np.zeros(len(segs), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"),
                                                      ("first", "<i8"), ("group", "<i4"), ("pad", "<i4")]))
            total_blocks = 0
            for i, (pp, gp, mp, vp, n, gi, _keep) in enumerate(segs):
                rec[i] = (pp, gp, mp, vp, n, total_blocks, gi, 0)
                total_blocks += -(-n // blk)
            raw = rec.view(np.uint8).reshape(-1)
            dev = self._step.device
            # Pinned staging + asynchronous copy.  Inside a hipGraph capture (gradients allocated from the graph's pool have
            # their own addresses) the copy becomes a node of the graph, so the captured step owns a private staging buffer
            # and device table that eager steps never touch.
            if capturing:
                # pinned memory cannot be allocated while a stream captures: take one of the buffers set aside by an eager step
                spare = [h for h in self._spares if h.numel() == raw.size]
                if not spare:
                    raise RuntimeError("FusedAdam: run one eager step() before capturing a step in a graph")
                host = spare[0]
                self._spares.remove(host)
                host.numpy()[:] = raw
                table = torch.empty((raw.size,), dtype=torch.uint8, device=dev)
                table.copy_(host, non_blocking=True)
                self._captured.append(host)
            else:
                if self._host is None or self._host[0][0].numel() != raw.size:
                    # a ring of staging buffers: gradient buffers that alternate between two addresses re-upload the table
                    # every step, and waiting for the PREVIOUS upload would tie the host to the device each step
                    self._host = [[torch.empty((raw.size,), dtype=torch.uint8, pin_memory=True), None] for _ in range(4)]
                    self._table = torch.empty((raw.size,), dtype=torch.uint8, device=dev)
                    self._slot = 0
                    self._spares = [torch.empty((raw.size,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
                slot = self._host[self._slot]
                self._slot = (self._slot + 1) % len(self._host)
                if slot[1] is not None:
                    slot[1].synchronize()          # the upload issued four table changes ago has read this buffer
                slot[0].numpy()[:] = raw
                self._table.copy_(slot[0], non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record()
                self._table_key, self._total_blocks, table = key, total_blocks, self._table
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        b1, b2 = self.param_groups[0]["betas"]
        call("gngf_adam_step", ptr(table), len(segs), total_blocks, ptr(self._step), lr, wd, ng, float(b1), float(b2),
             float(self.param_groups[0]["eps"]), stream_ptr())
        return loss

    def state_dict(self):
        # torch.optim.Adam advances every parameter's `step` tensor in place: hand out one copy per parameter, not the
        # shared counter (a shared tensor would be advanced once per parameter after loading into torch's optimizer)
        sd = super().state_dict()
        sd["state"] = {k: {**v, **({"step": v["step"].detach().clone()} if "step" in v else {})} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step = None                          # re-read from the loaded per-parameter `step`
        self._table_key = None


def get_optimizer(net, encoding_lr, HPD_lr, MLP_lr, encoding_weight_decay, HPD_weight_decay, MLP_weight_decay,
                  betas=(0.9, 0.99), eps=1e-15, fused_kernel=None):
    """reference functions.py:96-127.  On the GPU the update runs as one launch of this package's Adam kernel
    (FusedAdam: same rule and state layout as torch.optim.Adam); host tensors (CPU tests of the training algebra) and
    fused_kernel=False use torch.optim.Adam itself."""
    groups = [{"params": list(net.encoding.parameters()), "lr": encoding_lr, "weight_decay": encoding_weight_decay}]
    if not models.should_use_hash_function:
        groups.append({"params": list(net.HPD.parameters()), "lr": HPD_lr, "weight_decay": HPD_weight_decay})
    groups.append({"params": list(net.mlp.parameters()), "lr": MLP_lr, "weight_decay": MLP_weight_decay})
    # decided from the parameters the optimizer will actually update (hash mode also owns the int64, non-trainable
    # `_prime_numbers` Parameter, which is in no group)
    grouped = [p for g_ in groups for p in g_["params"]]
    on_gpu = bool(grouped) and all(p.is_cuda and p.dtype in FusedAdam.DTYPES for p in grouped)
    if fused_kernel is None:
        fused_kernel = on_gpu
    if fused_kernel:
        return FusedAdam(groups, betas=betas, eps=eps)
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
    one = torch.ones((), device=dev)                 # seed of backward(): the same d loss / d loss = 1, without a fill per batch
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
        loss.backward(gradient=one.to(loss.dtype))
        optimizer.step()
        losses.append(loss.detach())
        mses.append(mse.detach())
    return torch.stack(losses).mean().item(), torch.stack(mses).mean().item(), outputs


def calc_psnr(pred: np.ndarray, target: np.ndarray) -> float:
    """reference functions.py:134-136 (peak = max(target))."""
    mse = np.square(pred - target).mean()
    return 20 * np.log10(np.max(target)) - 10 * np.log10(mse)
