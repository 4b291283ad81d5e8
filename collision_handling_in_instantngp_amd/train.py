"""Callers of the hot path, mirrored from the reference so that parity tests read like the reference's own
training loop: Loss (utils.py:78-174), get_optimizer (functions.py:96-127), train_step
(functions.py:139-355, batching + loss assembly only; no wandb / plotting / collision diagnostics)."""
import contextlib

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
            if pbar is None:
                # net.compute_pbar = False: the caller opted out of the batch-mean distribution (with a frozen HPD the
                # distribution term is a constant without gradient) — the term is not evaluated, not faked
                return mse_loss, None, collisions_losses
        else:
            pbar = prob.sum(0).sum(1) / (prob.shape[0] * prob.shape[2])          # (L,N)
        return mse_loss, self.js_kl_rows(pbar), collisions_losses


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (amsgrad off; the reference's get_optimizer, functions.py:96-127) for every tensor of
    every parameter group in ONE launch of the HIP kernel behind `gngf_adam_step` (csrc/optim.hip).  State keeps
    torch.optim.Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so optimizer state dicts written by the
    reference's torch.optim.Adam load here and vice versa.  The step count is a device tensor advanced by the kernel:
    a step needs no host synchronisation and can be captured in a hipGraph together with forward and backward."""

    DTYPES = (torch.float32, torch.float16)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 4:
            raise ValueError("at most 4 parameter groups (GNGF_ADAM_MAX_GROUPS)")
        if len({(g["betas"], g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("betas and eps are shared by all groups (as in the reference's get_optimizer)")
        self.grad_scale = 1.0        # loss scale of fp16 training: every gradient is divided by it inside the kernel
        self._step = None            # device float: steps taken
        self._table = None           # device byte tensor: the packed segment list
        self._host = None            # ring of [pinned staging buffer, event of the upload that last read it]
        self._slot = 0
        self._captured = []          # staging buffers owned by captured steps (read again at every replay)
        self._spares = []            # pinned buffers set aside for captures (no pinned allocation while capturing)
        self._table_key = None
        self._total_blocks = 0

    _RECORD = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("w", "<u8"), ("n", "<i8"), ("first", "<i8"),
                        ("group", "<i4"), ("flags", "<i4")])          # csrc/optim.hip: AdamSegment (64 bytes)

    def _segments(self):
        """fp32 parameters: torch.optim.Adam's state layout.  fp16 parameters (fp16 level tables, BASELINE config 5): fp32
        moments plus `master`, the fp32 copy the update is applied to; the fp16 parameter is its rounding."""
        segs = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                grad = p.grad
                g32 = getattr(p, "grad_fp32", None) if grad is None else None      # fp16 tables: the fp32 buffer itself (ops._grad_out)
                if grad is None and g32 is None:
                    continue
                if g32 is not None:
                    if p.dtype != torch.float16 or g32.dtype != torch.float32 or g32.shape != p.shape or not p.is_cuda:
                        raise RuntimeError("grad_fp32 is the fp32 gradient of an fp16 device parameter, same shape")
                    grad = g32
                elif grad.is_sparse or p.dtype not in self.DTYPES or not p.is_cuda or grad.dtype != p.dtype:
                    raise RuntimeError("FusedAdam handles dense fp32 / fp16 device parameters (gradient of the same type)")
                half = p.dtype == torch.float16
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
                    st["exp_avg_sq"] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
                if half and "master" not in st:
                    st["master"] = p.detach().float().contiguous()
                if self._step is None:
                    prev = st.get("step", 0.0)
                    self._step = torch.full((), float(prev), dtype=torch.float32, device=p.device)
                st["step"] = self._step          # one shared counter (torch keeps one equal copy per parameter)
                g = grad if grad.is_contiguous() else grad.contiguous()
                for t_ in (p, st["exp_avg"], st["exp_avg_sq"]):
                    if not t_.is_contiguous():
                        raise RuntimeError("FusedAdam needs contiguous parameters and moments")
                # data parallel, sharded optimizer state (parallel.shard_direct_levels): this rank updates elements [lo, hi) of the
                # parameter only — the others' rows arrive with the all-gather that follows the step
                lo, hi = getattr(p, "_adam_range", None) or (0, p.numel())
                if hi <= lo:
                    continue
                segs.append((p.data_ptr() + lo * p.element_size(), g.data_ptr() + lo * g.element_size(), st["exp_avg"].data_ptr() + lo * 4,
                             st["exp_avg_sq"].data_ptr() + lo * 4, (st["master"].data_ptr() + lo * 4) if half else 0, hi - lo, gi,
                             int(half) | (2 if g32 is not None else 0), g))
        return segs

    def zero_grad(self, set_to_none=True):
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "grad_fp32", None) is not None:
                    p.grad_fp32 = None
        return super().zero_grad(set_to_none=set_to_none)

    def prepare_capture(self, steps=1):
        """Sets aside the pinned staging buffers a hipGraph capture of `steps` step() calls needs (pinned memory cannot be
        allocated while a stream captures) WITHOUT taking a step: call after one backward pass, before capturing."""
        segs = self._segments()
        size = len(segs) * self._RECORD.itemsize
        have = [h for h in self._spares if h.numel() == size]
        for _ in range(max(0, 1 + int(steps) - len(have))):
            self._spares.append(torch.empty((size,), dtype=torch.uint8, pin_memory=True))

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
        key = tuple(s_[:8] for s_ in segs)
        capturing = torch.cuda.is_current_stream_capturing()
        table, total_blocks = self._table, self._total_blocks
        if capturing or key != self._table_key:    # pointers moved (first step, new gradient buffers, loaded state)
            blk = query("gngf_adam_block_elems")
            rec = np.zeros(len(segs), dtype=self._RECORD)
            total_blocks = 0
            for i, (pp, gp, mp, vp, wp, n, gi, flags, _keep) in enumerate(segs):
                rec[i] = (pp, gp, mp, vp, wp, n, total_blocks, gi, flags)
                total_blocks += -(-n // blk)
            raw = rec.view(np.uint8).reshape(-1)
            dev = self._step.device
            # Pinned staging + asynchronous copy.  Inside a hipGraph capture (gradients allocated from the graph's pool have
            # their own addresses) the copy becomes a node of the graph, so the captured step owns a private staging buffer
            # and device table that eager steps never touch.
            if capturing:
                # pinned memory cannot be allocated while a stream captures: take one of the buffers set aside beforehand
                spare = [h for h in self._spares if h.numel() == raw.size]
                if not spare:
                    raise RuntimeError("FusedAdam: call prepare_capture() (or run one eager step()) before capturing a step in a graph")
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
             float(self.param_groups[0]["eps"]), 1.0 / float(self.grad_scale), stream_ptr())
        return loss

    def state_dict(self):
        # torch.optim.Adam advances every parameter's `step` tensor in place: hand out one copy per parameter, not the
        # shared counter (a shared tensor would be advanced once per parameter after loading into torch's optimizer)
        sd = super().state_dict()
        sd["state"] = {k: {**v, **({"step": v["step"].detach().clone()} if "step" in v else {})} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch casts loaded state to the parameter's dtype: the fp32 moments / master copy of an fp16 parameter are put back
        from itertools import chain
        ids = list(chain.from_iterable(g["params"] for g in state_dict["param_groups"]))
        params = list(chain.from_iterable(g["params"] for g in self.param_groups))
        for i, p in zip(ids, params):
            src = state_dict["state"].get(i)
            if src is None or p.dtype != torch.float16:
                continue
            for k in ("exp_avg", "exp_avg_sq", "master"):
                if k in src:
                    self.state[p][k] = src[k].detach().to(device=p.device, dtype=torch.float32).clone()
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
    enc = getattr(net, "encoding", None)
    handover = getattr(enc, "grad_fp32_handover", None)
    handover = ops.FP16_TABLE_GRAD_FP32 if handover is None else handover
    if handover and any(p.dtype == torch.float16 for p in groups[0]["params"]):
        raise ValueError("fp16 level tables with the fp32 gradient hand-over (grad_fp32, no .grad) need train.FusedAdam: "
                         "torch.optim.Adam would never see a gradient for them — pass fused_kernel=True or turn the hand-over off")
    return torch.optim.Adam(groups, betas=betas, eps=eps, **({"fused": True} if on_gpu else {}))


def promised_gloss(loss_fn, l_mse):
    """The gradient that will reach the pixel-loss value when the step's total is assemble_loss(...) and backward() is seeded
    with 1: l_mse — but only when loss_fn is this package's Loss (its mse term IS MSELoss(rgb, target)) and l_mse is a plain
    number; anything else makes no promise (the decoder then runs its two-kernel path, correct for any gradient)."""
    if type(loss_fn) is Loss and isinstance(l_mse, (int, float)) and not isinstance(l_mse, bool):
        return float(l_mse)
    return None


def assemble_loss(mse, kls, colls, l_mse, l_js_kl, l_collisions):
    """reference functions.py:243-245, including the '+1 per level while previous_collisions is empty' quirk."""
    loss = mse if (isinstance(l_mse, (int, float)) and l_mse == 1) else l_mse * mse     # (x * 1 is a kernel launch of its own)
    if not models.should_use_hash_function and kls is not None:
        loss = loss + ((l_js_kl * kls) + (l_collisions * colls if colls.nelement() != 0 else 1)).sum(0)
    return loss


class GraphedStep:
    """One optimisation step — zero_grad, net(batch), Loss, the weighted sum of functions.py:243-245, backward and
    optimizer.step() — captured ONCE into a hipGraph per batch shape and replayed for every further batch: the step is
    ~15 short kernels, so eager launch gaps and Python dispatch are a visible fraction of it (0.72 ms eager vs 0.61 ms
    replayed at 2^20 pixels).  Batches are copied into static device buffers; results come back in static buffers that
    the next replay overwrites.

        gs = GraphedStep(net, loss_fn, optimizer, l_mse, l_js_kl, l_collisions, coord_bounds=(max_row, max_col))
        r = gs(batch_x, batch_target)            # r.out (P,C), r.loss, r.mse, r.kls, r.colls, r.idx

    optimizer: a FusedAdam (its step is capturable and needs no host synchronisation) or None (forward + backward only).
    GNGF indexing with a trainable HPD, or with the dense distribution returned, needs `coord_bounds` (an upper bound of
    the coordinates of EVERY batch) so that the vertex rectangle is fixed and no device->host read happens in the step."""

    class Result:
        __slots__ = ("out", "probs", "idx", "loss", "mse", "kls", "colls")

    def __init__(self, net, loss_fn, optimizer, l_mse=1.0, l_js_kl=1.0, l_collisions=1e-3, batch_percentage=1.0,
                 coord_bounds=None, warm=2, unroll=1, cross_replay=False):
        """unroll = k > 1: ONE graph holds k consecutive steps on k batches (run_many): a replay costs ~9 us of launch latency
        on top of its kernels whatever it contains, so k steps per replay spread it over k steps.
        cross_replay: the LAST step of a replay also bins the first batch of the NEXT replay (ops.BinPipeline riders) when the caller
        names it in advance — run_many(batches, next_first=x) / gs(x, y, next_first=x) — so that no step of a steady sequence bins
        at its head, at any unroll (with unroll = 1, the data-parallel case, every step's binning rides on the step before).
        Three graphs instead of one per batch shape: `cold` (its first step bins itself) and two `steady` ones that read one of
        two binning workspaces and fill the other."""
        self.unroll = int(unroll)
        self.cross_replay = bool(cross_replay)
        if optimizer is not None and not isinstance(optimizer, FusedAdam):
            raise TypeError("GraphedStep captures FusedAdam.step(); torch.optim.Adam's step is not capturable here")
        self.net, self.loss_fn, self.optimizer = net, loss_fn, optimizer
        self.weights = (l_mse, l_js_kl, l_collisions)
        self.batch_percentage = batch_percentage
        self.warm = int(warm)
        if coord_bounds is not None:
            net.coord_bounds = (float(coord_bounds[0]), float(coord_bounds[1]))
        self._graphs = {}
        # A captured step works on static buffers anyway (every replay overwrites the previous replay's results), and _body lets
        # go of all gradients before the forward pass: the encoder may keep the table gradient in one buffer from step to step
        # (ops.PERSISTENT_TABLE_GRAD is opt-in per model: an eager loop that may keep old .grad tensors does not get it)
        if getattr(net, "dp", None) is not None:
            net.dp.persist_ok = True

    def _body(self, st, next_x=None, next_ws=None, binned_in=None):
        """next_x: the coordinate buffer of the step that FOLLOWS this one (inside the same graph, or — cross_replay — the first step
        of the next replay) — announced to the encoder, whose pixel-stage launches then carry that batch's binning
        (ops.BinPipeline), into workspace next_ws if given.  binned_in: the workspace that holds THIS step's binned pixels (filled by
        the previous replay's last step); None: found through the pipeline, or binned at the head of the step."""
        net, (l_mse, l_js_kl, l_collisions) = self.net, self.weights
        pipe = getattr(getattr(net, "dp", None), "pipeline", None)
        if pipe is not None:
            pipe.announce(next_x, next_ws)
            if binned_in is not None:
                pipe.seed(st["x"], binned_in)
        shadow = st["shadow"]
        for s_ in shadow.values():
            s_.grad = None
            if getattr(s_, "grad_fp32", None) is not None:
                s_.grad_fp32 = None
        # the step begins with zero_grad (functions.py:199): the parameters let go of the previous step's gradients as well
        # (they get this step's at the end) — a step-to-step gradient buffer (ops.PERSISTENT_TABLE_GRAD) is free to be reused
        for _name, p in st["named"]:
            p.grad = None
            if getattr(p, "grad_fp32", None) is not None:
                p.grad_fp32 = None
        # The step runs on SHADOW leaves (detached aliases of the parameters: same storage, fresh autograd identity).  An
        # autograd leaf's AccumulateGrad node is bound to the stream that was current when it was first created, and it stays
        # alive as long as any earlier graph does (a kept `loss` from an eager step on the default stream is enough): the
        # captured backward would then hop onto the legacy default stream, which cannot take part in a capture
        # (hipStreamEndCapture crashes; tools/dbg_graphed.py).  Fresh leaves get their nodes on the capturing stream.
        # the pixel loss is fused into the decoder: its gradient is formed inside the backward kernel, and its VALUE — which no
        # kernel of the step reads when it enters the total with weight 1 — runs on a parallel branch, joined at the end
        aside = isinstance(l_mse, (int, float)) and l_mse == 1 and (models.should_use_hash_function or getattr(net, "compute_pbar", True) is False)
        prev_aside, net.loss_value_aside = net.loss_value_aside, bool(aside)
        try:
            # backward() below is seeded with 1 and the total is l_mse * mse + (terms without mse): the gradient that reaches the
            # pixel loss IS l_mse — promised to the decoder, which then runs forward and backward in one launch (the promise
            # is only made for this package's own Loss: a custom loss_fn may weigh the pixel term differently)
            with net.fused_mse(st["y"], gloss=promised_gloss(self.loss_fn, l_mse)):
                out, probs, idx, _counts = torch.func.functional_call(net, shadow, (st["x"], self.batch_percentage), {"should_calc_counts": False})
        finally:
            net.loss_value_aside = prev_aside
        mse, kls, colls = self.loss_fn(out, st["y"], None if probs is None else probs.shape[-1], probs, st["pc"], st["pm"])
        loss = assemble_loss(mse, kls, colls, l_mse, l_js_kl, l_collisions)
        if loss is not mse:
            net.join_loss_value()                      # the total is computed from the value: it has to be there
        loss.backward(gradient=st["one"].to(loss.dtype))
        net.join_loss_value()
        for name, p in st["named"]:
            p.grad = shadow[name].grad             # the parameters' own .grad: what the optimizer and the caller read
            g32 = getattr(shadow[name], "grad_fp32", None)
            if g32 is not None or getattr(p, "grad_fp32", None) is not None:
                p.grad_fp32 = g32
        r = GraphedStep.Result()
        r.out, r.probs, r.idx, r.loss, r.mse, r.kls, r.colls = out.detach(), probs, idx, loss.detach(), mse.detach(), kls, colls
        return r

    def _build(self, key, bx, by, pc, pm):
        net = self.net
        if not models.should_use_hash_function and getattr(net, "coord_bounds", None) is None:
            frozen_fast = (net.hpd_is_frozen() and net.dense_probs is False and not net.compute_pbar
                           and not net._should_keep_topk_only)
            if not frozen_fast:
                raise RuntimeError("GraphedStep: GNGF indexing reads the batch's coordinate bounds on the host; pass coord_bounds= "
                                   "(an upper bound over every batch) so that the step has no device->host read")
        named = [(n, p) for n, p in net.named_parameters()]
        st = {"x": bx.detach().clone().contiguous(), "y": by.detach().clone().contiguous(), "pc": pc.detach().clone(),
              "pm": pm.detach().clone(), "one": torch.ones((), device=bx.device),
              "named": [(n, p) for n, p in named if p.requires_grad],
              "shadow": {n: p.detach().requires_grad_(p.requires_grad) for n, p in named}}
        # further batches of an unrolled graph: their own input buffers, everything else shared
        st["more"] = [dict(st, x=torch.empty_like(st["x"]), y=torch.empty_like(st["y"])) for _ in range(self.unroll - 1)]
        for m in st["more"]:
            m["x"].copy_(st["x"])
            m["y"].copy_(st["y"])
        pipe = getattr(getattr(net, "dp", None), "pipeline", None)
        if pipe is not None:
            pipe.reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, self.warm)):        # lazy initialisation (workspaces, cached tables) happens here, not in the capture
                self._body(st)
            if self.optimizer is not None:
                self.optimizer.prepare_capture(self.unroll)      # no optimizer step is taken before the first real batch
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        st["variants"] = {}
        st["pre"] = None                 # (index of the workspace that holds the next replay's first batch, identity of that batch)
        if self.cross_replay and pipe is not None and ops.BIN_PIPELINE:
            plan = ops.EncodePlan(st["x"].shape[0], net._n_ls_host, net._feature_dim)
            if plan.Ls > 0 and plan.interleaved(backward=True) and st["x"].shape[0] > 0:
                # two binning workspaces, allocated outside every capture: each steady graph reads one and fills the other
                st["W"] = [ops.TiledWorkspace(plan, st["x"], launch=False) for _ in range(2)]
                st["x_next"] = st["x"].clone()
        st["graph"], st["results"] = self._capture(st, "cold")
        st["result"] = st["results"][0]
        self._graphs[key] = st
        return st

    _VARIANTS = {"cold": (None, 0), "01": (0, 1), "10": (1, 0)}      # name -> (workspace read by the first step, filled by the last)

    def _capture(self, st, name):
        net = self.net
        pipe = getattr(getattr(net, "dp", None), "pipeline", None)
        reads, fills = self._VARIANTS[name]
        W = st.get("W")
        if self.optimizer is not None and st["variants"]:
            self.optimizer.prepare_capture(self.unroll)
        if pipe is not None:
            pipe.reset()
        g = torch.cuda.CUDAGraph()
        results = []
        # No garbage collection while the stream captures: a collection that happens to run inside the body finalises whatever
        # cyclic garbage earlier work left behind (other models' graphs, events, blocks last used on other streams), and a
        # finaliser that touches the runtime during a capture aborts the process (seen once in round 5: "Fatal Python error:
        # Aborted ... Garbage-collecting" in the middle of a captured forward pass).  torch.cuda.graph() collects once on entry;
        # what the body itself leaves behind is collected after the capture has ended.
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # thread_local: other threads of the process (the RCCL watchdog at world > 1) may query events while we capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                subs = [st] + st["more"]
                for j, sub in enumerate(subs):
                    last = j + 1 == len(subs)
                    # step j's pixel-stage launches bin step j + 1's batch (its own static buffer: refilled before every replay);
                    # the last step's bin the first batch of the NEXT replay (cross_replay: the caller puts it into x_next beforehand)
                    nx = subs[j + 1]["x"] if not last else (st["x_next"] if W is not None else None)
                    results.append(self._body(sub, next_x=nx, next_ws=(W[fills] if (last and W is not None) else None),
                                              binned_in=(W[reads] if (j == 0 and W is not None and reads is not None) else None)))
                    if self.optimizer is not None:
                        self.optimizer.step()
        finally:
            if gc_was_on:
                gc.enable()
        if pipe is not None:
            pipe.reset()
        st["variants"][name] = (g, results, self._handover(st))
        return st["variants"][name][:2]

    def _handover(self, st):
        """What a captured step leaves behind on the HOST side for its readers — the parameters' .grad / .grad_fp32, the
        encoder's one-buffer table gradient (parallel.py all-reduces it), the data-parallel bookkeeping of the deferred vertex
        stage — points into the memory pool of the graph that was captured LAST.  Each variant keeps its own set and _adopt()
        puts it back before that variant is replayed: whoever reads p.grad after a replay reads what THAT replay wrote."""
        net = self.net
        dp = getattr(net, "dp", None)
        owners = [m for m in net.modules() if hasattr(m, "_grad_base")]
        return {"grads": [(p, p.grad, getattr(p, "grad_fp32", None)) for _n, p in st["named"]],
                "bases": [(m, m._grad_base, getattr(m, "_grad_base_fp32", None)) for m in owners],
                "dp": None if dp is None else (dp.deferred, dp.tables_reduced)}

    def _adopt(self, handover):
        for p, g, g32 in handover["grads"]:
            p.grad = g
            if g32 is not None or getattr(p, "grad_fp32", None) is not None:
                p.grad_fp32 = g32
        for m, base, base32 in handover["bases"]:
            m._grad_base = base
            if base32 is not None or getattr(m, "_grad_base_fp32", None) is not None:
                m._grad_base_fp32 = base32
        dp = getattr(self.net, "dp", None)
        if dp is not None and handover["dp"] is not None:
            dp.deferred, dp.tables_reduced = handover["dp"]

    def _variant(self, st, use):
        if use not in st["variants"]:
            self._capture(st, use)
        g, results, handover = st["variants"][use]
        self._adopt(handover)
        return g, results

    @staticmethod
    def _ident(t):
        return (id(t), t.data_ptr(), t._version, tuple(t.shape))

    def _replay(self, st, batches, pc, pm, next_first):
        """copies the batches in, picks the graph (steady when the previous call announced exactly batches[0][0]), announces the
        next replay's first batch, replays"""
        use = "cold"
        pre = st.get("pre")
        if "W" in st and pre is not None and batches is not None and pre[2] is batches[0][0] and pre[1] == self._ident(batches[0][0]):
            use = "01" if pre[0] == 0 else "10"
        if batches is not None:
            for sub, (bx, by) in zip([st] + st["more"], batches):
                sub["x"].copy_(bx)
                sub["y"].copy_(by)
        if pc is not None and pc.numel():
            st["pc"].copy_(pc)
            st["pm"].copy_(pm)
        st["pre"] = None
        if "W" in st and next_first is not None and tuple(next_first.shape) == tuple(st["x"].shape):
            st["x_next"].copy_(next_first)
            # (the tensor itself is kept: while it is alive no other tensor can take its id and allocator block — ADVICE r4)
            st["pre"] = (self._VARIANTS[use][1], self._ident(next_first), next_first)
        g, results = self._variant(st, use)
        g.replay()
        return results

    def run_many(self, batches, previous_collisions=None, previous_min_possible_collisions=None, next_first=None):
        """unroll consecutive steps in ONE replay: batches = [(x, target)] * unroll (same shapes).  Returns the list of results.
        next_first (cross_replay): the coordinates of the first batch of the NEXT call — pass that very tensor, unmodified, as its
        batches[0][0] and no step of that replay bins at its head."""
        if len(batches) != self.unroll:
            raise ValueError(f"run_many needs exactly unroll = {self.unroll} batches")
        (x0, y0) = batches[0]
        dev = x0.device
        empty = torch.tensor([], device=dev)
        pc = empty if previous_collisions is None else previous_collisions.to(dev)
        pm = empty if previous_min_possible_collisions is None else previous_min_possible_collisions.to(dev)
        key = (tuple(x0.shape), tuple(y0.shape), tuple(pc.shape), tuple(pm.shape), bool(models.should_use_hash_function))
        st = self._graphs.get(key)
        if st is None:
            st = self._build(key, x0, y0, pc, pm)
        return self._replay(st, batches, pc, pm, next_first)

    def replay_only(self, key=None):
        """Replays the (only, or the named) captured step on the batch already in its static buffers (benchmarks)."""
        st = self._graphs[key] if key is not None else next(iter(self._graphs.values()))
        g, results = self._variant(st, "cold")
        g.replay()
        return results[0]

    def replay_steady(self, key=None):
        """Benchmarks, cross_replay: replays the steady graph on the batches already in the static buffers (x_next holds the first
        batch again), alternating between the two binning workspaces.  Needs one run_many(..., next_first=batches[0][0]) before."""
        st = self._graphs[key] if key is not None else next(iter(self._graphs.values()))
        pre = st.get("pre")
        if "W" not in st or pre is None:
            return self.replay_only(key)
        use = "01" if pre[0] == 0 else "10"
        g, results = self._variant(st, use)
        st["pre"] = (self._VARIANTS[use][1], *pre[1:])
        g.replay()
        return results[0]

    def __call__(self, batch_x, batch_target, previous_collisions=None, previous_min_possible_collisions=None, next_first=None):
        dev = batch_x.device
        empty = torch.tensor([], device=dev)
        pc = empty if previous_collisions is None else previous_collisions.to(dev)
        pm = empty if previous_min_possible_collisions is None else previous_min_possible_collisions.to(dev)
        key = (tuple(batch_x.shape), tuple(batch_target.shape), tuple(pc.shape), tuple(pm.shape), bool(models.should_use_hash_function))
        if self.unroll != 1:
            raise ValueError("an unrolled GraphedStep takes its batches through run_many()")
        st = self._graphs.get(key)
        if st is None:
            st = self._build(key, batch_x, batch_target, pc, pm)
        return self._replay(st, [(batch_x, batch_target)], pc, pm, next_first)[0]


def train_epoch(net, loss_fn, optimizer, x, target, w, h, l_mse, l_js_kl, l_collisions, batch_percentage=1.0,
                should_shuffle=True, shuffled_indices=None, previous_collisions=None, previous_min_possible_collisions=None,
                should_calc_counts=False, graph=False):
    """The batch loop of the reference's train_step (functions.py:183-281): ceil(1/batch_percentage) mini-batches of
    zero_grad -> net -> Loss -> weighted sum -> backward -> step.  graph=True replays each step from a hipGraph
    (GraphedStep, cached on the net per batch shape).  Returns a dict of per-batch device tensors and the outputs /
    indices in batch order; nothing is read back to the host inside the loop."""
    net.train()
    shape = w * h
    num_batches = int(np.ceil(shape / (shape * batch_percentage)))
    step = int(batch_percentage * shape)
    dev = x.device
    empty = torch.tensor([], device=dev)
    previous_collisions = empty if previous_collisions is None else previous_collisions.to(dev)
    previous_min_possible_collisions = empty if previous_min_possible_collisions is None else previous_min_possible_collisions.to(dev)
    outputs = torch.zeros((shape, target.shape[1]), device=dev)     # (the reference leaves unvisited rows uninitialised)
    indices = None
    rec = {"loss": [], "mse": [], "kls": [], "colls": [], "counts": []}
    one = torch.ones((), device=dev)                 # seed of backward(): the same d loss / d loss = 1, without a fill per batch
    # the reference's collision statistic (functions.py:327) without the index tensor: when the model does not return indices the
    # forward passes mark the slots their batches use (models.start_collision_tracking) and train_step reads the maps
    tracked = (hasattr(net, "start_collision_tracking") and not getattr(net, "return_indices", True) and not should_calc_counts
               and not models.should_batchnorm_data)
    if tracked:
        net.start_collision_tracking()
    elif hasattr(net, "stop_collision_tracking"):
        net.stop_collision_tracking()
    rec["tracked"] = tracked
    gs = None
    if graph:
        gs = getattr(net, "_graphed_step", None)
        cfg = (id(loss_fn), id(optimizer), l_mse, l_js_kl, l_collisions, batch_percentage, tracked)
        if gs is None or gs._cfg != cfg:
            bounds = None if models.should_use_hash_function else (float(x[:, 0].max()), float(x[:, 1].max()))
            gs = GraphedStep(net, loss_fn, optimizer, l_mse, l_js_kl, l_collisions, batch_percentage, coord_bounds=bounds,
                             cross_replay=True)
            gs._cfg = cfg
            net._graphed_step = gs
    def batch(b):
        lo_, hi_ = b * step, (b + 1) * step
        sel = shuffled_indices[lo_:hi_].long() if should_shuffle else slice(lo_, hi_)
        return x[sel].contiguous(), target[sel]

    pipe = getattr(getattr(net, "dp", None), "pipeline", None)
    upcoming = batch(0) if num_batches > 0 else None
    for b in range(num_batches):
        lo, hi = b * step, (b + 1) * step
        bx, by = upcoming
        # the batches are fixed slices of one permutation (functions.py:186-194): the next one is materialised a step ahead and
        # announced, so that this step's pixel-stage launches bin it (ops.BinPipeline; eager steps only — a replayed step
        # copies its batch into static buffers)
        upcoming = batch(b + 1) if b + 1 < num_batches else None
        if pipe is not None:
            pipe.announce(upcoming[0] if (upcoming is not None and gs is None and upcoming[0].shape == bx.shape and bx.shape[0] > 0) else None)
        if bx.shape[0] == 0:
            continue
        if gs is not None and not should_calc_counts:
            # (the next slice is named a step ahead: this replay's launches bin it, the next replay starts on binned pixels)
            nf = upcoming[0] if (upcoming is not None and upcoming[0].shape == bx.shape) else None
            r = gs(bx, by, previous_collisions, previous_min_possible_collisions, next_first=nf)
            out, idx, loss, mse, kls, colls, counts = r.out, r.idx, r.loss, r.mse, r.kls, r.colls, []
        else:
            optimizer.zero_grad()
            by = by.contiguous()
            with (net.fused_mse(by, gloss=promised_gloss(loss_fn, l_mse)) if hasattr(net, "fused_mse") else contextlib.nullcontext()):
                out, probs, idx, counts = net(bx, batch_percentage, should_calc_counts=should_calc_counts)
            mse, kls, colls = loss_fn(out, by, None if probs is None else probs.shape[-1], probs,
                                      previous_collisions, previous_min_possible_collisions)
            loss = assemble_loss(mse, kls, colls, l_mse, l_js_kl, l_collisions)
            loss.backward(gradient=one.to(loss.dtype))
            optimizer.step()
        outputs[lo:lo + out.shape[0]] = out.detach()
        if idx is not None:
            if indices is None:
                indices = torch.zeros((shape, *idx.shape[1:]), dtype=idx.dtype, device=dev)
            indices[lo:lo + idx.shape[0]] = idx
        rec["loss"].append(loss.detach().clone())
        rec["mse"].append(mse.detach().clone())
        if kls is not None:
            rec["kls"].append(kls.detach().clone())
            rec["colls"].append(colls.detach().clone() if colls.nelement() != 0 else torch.ones_like(kls.detach()))
        rec["counts"].append(counts)
    rec["outputs"], rec["indices"] = outputs, indices
    if tracked:
        rec["collisions"] = net.tracked_hash_collisions()
        net.stop_collision_tracking()
    return rec


def train_step(net, loss_fn, optimizer, x, target, w, h, hash_table_size, topk_k, l_mse, l_js_kl, l_collisions,
               batch_percentage=1, num_levels=4, should_bw=False, should_calc_counts=False, should_shuffle=True,
               shuffled_indices=None, reordered_indices=None, previous_collisions=None,
               previous_min_possible_collisions=None, *, graph=False):
    """reference functions.py:139-355 — same positional signature and the same 9-tuple:
        (loss, to_show_img (h,w,3) int32 numpy, collisions, min_possible_collisions, counts_per_level, mse,
         kl_div_losses (L,) | None, collisions_losses (L,) | None, indices_per_level)
    Differences, all in host-side bookkeeping: values are read back once per epoch instead of per batch; pixels that no
    batch visits (int(batch_percentage * w * h) * num_batches < w * h) show as zeros where the reference shows
    uninitialised memory; collision statistics are taken over each pixel's own top-K indices (the reference allocates
    topk_k / batch_size index slots per pixel and fills topk_k of them, so its statistic is partly uninitialised memory —
    SURVEY.md §8f.3).  graph=True (keyword-only, extension) replays every step from a hipGraph."""
    import collections
    import functools
    import operator
    rec = train_epoch(net, loss_fn, optimizer, x, target, w, h, l_mse, l_js_kl, l_collisions, batch_percentage=batch_percentage,
                      should_shuffle=should_shuffle, shuffled_indices=shuffled_indices, previous_collisions=previous_collisions,
                      previous_min_possible_collisions=previous_min_possible_collisions, should_calc_counts=should_calc_counts,
                      graph=graph)
    hash_mode = bool(models.should_use_hash_function)
    loss_item = float(torch.stack(rec["loss"]).double().mean().item())            # np.mean over batches (functions.py:286)
    mse_loss = float(torch.stack(rec["mse"]).double().mean().item())
    no_dist = hash_mode or not rec["kls"]
    kl_div_losses = None if no_dist else torch.stack(rec["kls"]).double().mean(0).cpu().numpy()
    collisions_losses = None if no_dist else torch.stack(rec["colls"]).double().mean(0).cpu().numpy()
    outputs, indices = rec["outputs"], rec["indices"]
    if should_shuffle:
        ro = reordered_indices
        if ro is None:                                  # inverse permutation (main.py:55-58)
            ro = torch.empty_like(shuffled_indices)
            ro[shuffled_indices.long()] = torch.arange(shuffled_indices.numel(), device=shuffled_indices.device, dtype=ro.dtype)
        ro = ro.long().to(outputs.device)
        outputs = outputs[ro]
        indices = indices[ro] if indices is not None else None
    indices_per_level = []
    if should_calc_counts and indices is not None:
        flat = indices.permute(1, 0, *range(2, indices.dim())).reshape(indices.shape[1], -1).cpu().numpy()
        indices_per_level = [dict(zip(*np.unique(level, return_counts=True))) for level in flat]
    if indices is not None:
        collisions, min_possible_collisions = net.calc_hash_collisions(indices)
    elif rec.get("collisions") is not None:             # return_indices = False: the slot maps the forward passes marked
        collisions, min_possible_collisions = rec["collisions"]
    else:
        collisions, min_possible_collisions = torch.tensor([]), torch.tensor([])
    to_show_img = (outputs * 255).reshape((h, w, 3) if not should_bw else (h, w)).int().detach().cpu().numpy()
    counts_per_level = [
        dict(functools.reduce(operator.add, map(collections.Counter, [c[i] for c in rec["counts"]])))
        for i in range(num_levels) if should_calc_counts
    ]
    return (loss_item, to_show_img, collisions, min_possible_collisions, counts_per_level, mse_loss, kl_div_losses,
            collisions_losses, indices_per_level)


def calc_psnr(pred: np.ndarray, target: np.ndarray) -> float:
    """reference functions.py:134-136 (peak = max(target))."""
    mse = np.square(pred - target).mean()
    return 20 * np.log10(np.max(target)) - 10 * np.log10(mse)
