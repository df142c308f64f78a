"""Training step of the hot path: forward -> loss -> backward -> gradient all-reduce -> AdamW, one rank per GPU.

Mirrors the order of the reference's hot loop (yogo/train.py:309-325: zero_grad, forward, loss, backward with the
per-parameter clamp, DDP gradient averaging, ``optimizer.step()``, ``scheduler.step()``) and its optimiser set-up
(yogo/train.py:206-223: AdamW(lr, weight_decay) over ALL parameters, CosineAnnealingLR stepped every iteration with
``eta_min = lr / decay_factor``) without going through autograd or torch DDP:

* all parameters live in ONE flat fp32 buffer (the modules' ``.data`` are views of it), so do the gradients and the
  two Adam moments: the optimiser is one fused kernel, the data-parallel exchange is one RCCL all-reduce of
  2.17 MB (``torch.distributed`` backend "nccl" is RCCL on ROCm; over xGMI the message is latency-bound, so a single
  bucket is the right granularity -- SURVEY.md section 8e);
* gradients are clamped per rank BEFORE the all-reduce (the reference's hooks fire before DDP averages);
* BatchNorm statistics stay per GPU (the reference has no SyncBN).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from yogo_amd import _hip
from yogo_amd.engine import get_engine
from yogo_amd.model import YOGO
from yogo_amd.yogo_loss import YOGOLoss


def cosine_lr(step: int, base_lr: float, t_max: int, eta_min: float) -> float:
    """closed form of torch.optim.lr_scheduler.CosineAnnealingLR after ``step`` scheduler steps"""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


class FlatParams:
    """re-homes every parameter of a module into one contiguous fp32 buffer (+ same-shaped grad / Adam buffers)"""

    def __init__(self, module: torch.nn.Module):
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters()]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.total = sum(sizes)
        self.flat = torch.empty(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad_views: Dict[int, torch.Tensor] = {}
        off = 0
        for p, n in zip(self.params, sizes):
            view = self.flat[off : off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.grad_views[id(p)] = self.grad[off : off + n].view(p.shape)
            off += n

    def publish_grads(self) -> None:
        """expose the flat gradient through ``param.grad`` (views, no copy)"""
        for p in self.params:
            p.grad = self.grad_views[id(p)]

    def check_homed(self) -> None:
        """every parameter must still be a view of the flat buffer: ``model.to(...)``, ``.half()`` or ``.float()`` after the
        trainer was built re-allocate ``param.data`` and the fused AdamW kernel would then update memory the model no longer reads"""
        lo = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.data_ptr() != lo + 4 * off or p.dtype != torch.float32:
                raise RuntimeError("yogo_amd: a model parameter no longer lives in the trainer's flat buffer (was the model moved or "
                                   "cast after HipTrainer was built?) -- build a new HipTrainer")
            off += p.numel()


class HipTrainer:
    def __init__(
        self,
        model: YOGO,
        loss: Optional[YOGOLoss] = None,
        learning_rate: float = 3e-4,
        weight_decay: float = 5e-2,
        betas: Tuple[float, float] = (0.9, 0.999),
        eps: float = 1e-8,
        total_steps: int = 1000,
        decay_factor: float = 10.0,
        process_group=None,
        half: bool = False,
    ):
        self.model = model
        self.loss = loss if loss is not None else YOGOLoss()
        self.lr = learning_rate
        self.wd = weight_decay
        self.betas = betas
        self.eps = eps
        self.t_max = max(1, int(total_steps))
        self.eta_min = learning_rate / decay_factor
        self.global_step = 0
        self.half = bool(half)   # bf16 activations / activation gradients (the reference's --half is fp16 autocast)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.flat = FlatParams(model)
        self.engine = get_engine(model.model)
        self.engine.invalidate_packed()
        self.last_loss: Optional[torch.Tensor] = None   # 4 device floats: total, iou, objectness, classification

    def current_lr(self) -> float:
        return cosine_lr(self.global_step, self.lr, self.t_max, self.eta_min)

    # ---- checkpointing: the layout of torch.optim.AdamW.state_dict() (what the reference's Trainer.checkpoint stores under
    #      "optimizer_state_dict", yogo/train.py:280-293), so reference checkpoints resume here and vice versa ----------------
    def state_dict(self) -> Dict:
        state = {}
        off = 0
        step = torch.tensor(float(self.global_step))
        for i, p in enumerate(self.flat.params):
            n = p.numel()
            if self.global_step > 0:   # torch creates the per-parameter state lazily at the first step
                state[i] = {"step": step.clone(),
                            "exp_avg": self.flat.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.flat.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "initial_lr": self.lr, "params": list(range(len(self.flat.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError("yogo_amd: optimizer state does not match the model (one parameter group over all parameters expected)")
        g = groups[0]
        self.betas = tuple(g.get("betas", self.betas))
        self.eps = float(g.get("eps", self.eps))
        self.wd = float(g.get("weight_decay", self.wd))
        # the cosine schedule (base lr, eta_min, T_max) stays what this trainer was built with: the schedule is a closed form of
        # global_step here, and the reference restarts its scheduler on --from-pretrained as well (yogo/train.py:136-148)
        off, steps = 0, []
        for i, p in zip(g["params"], self.flat.params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is None:
                self.flat.exp_avg[off:off + n].zero_()
                self.flat.exp_avg_sq[off:off + n].zero_()
            else:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"yogo_amd: optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
                self.flat.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.flat.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st["step"])))
            off += n
        if steps:
            if len(set(steps)) != 1:
                raise ValueError("yogo_amd: per-parameter step counts differ; the fused AdamW kernel keeps one step count")
            self.global_step = steps[0]
        else:
            self.global_step = 0

    def broadcast_parameters(self, src: int = 0) -> None:
        """rank-0 weights and BatchNorm buffers to every rank (what DDP does at construction)"""
        if self.world > 1:
            dist.broadcast(self.flat.flat, src=src, group=self.pg)
            for b in self.model.buffers():
                if b.is_floating_point() or b.dtype == torch.long:
                    dist.broadcast(b, src=src, group=self.pg)

    @torch.no_grad()
    def step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """one optimisation step on this rank's shard; returns the 4-float device loss record"""
        m = self.model
        _hip.require_cuda(imgs, "imgs")
        _hip.require_cuda(labels, "labels")
        with torch.cuda.device(imgs.device):
            st = _hip.stream_ptr()
            eng = self.engine
            eng.clip = m._clip
            self.flat.check_homed()
            if imgs.ndim == 3:
                imgs = imgs[None]
            if not imgs.is_floating_point() and imgs.dtype != torch.uint8:
                imgs = imgs.float()
            # ---- forward: backbone + decode ------------------------------------------------------------------
            if self.half:
                from yogo_amd.engine import backward_bf16_train, forward_bf16_train

                raw, saved = forward_bf16_train(eng, imgs)
            else:
                raw, saved = eng.forward(imgs, need_grad=True)
            B, P, Sy, Sx = raw.shape
            aw, ah, wm, hm = m._decode_scalars()
            pred = torch.empty_like(raw)
            _hip.call("yogo_decode_fwd", raw, pred, m._Cxs, m._Cys, B, P, Sy, Sx, aw, ah, wm, hm, int(bool(m.inference)), st)
            # ---- loss forward + backward (one kernel) -------------------------------------------------------------
            L = self.loss
            gpred = torch.empty_like(raw)
            out = torch.empty(4, dtype=torch.float32, device=raw.device)
            ws = torch.empty(_hip.query_size("yogo_loss_workspace_bytes", B, Sy, Sx) // 4, dtype=torch.float32, device=raw.device)
            lab = labels if (labels.dtype == torch.float32 and labels.is_contiguous()) else labels.contiguous().float()
            _hip.call("yogo_loss_fwd_bwd", pred, lab, gpred, out, ws, B, P, Sy, Sx, float(L.no_obj_weight), float(L.iou_weight),
                      float(L.classify_weight), float(L.label_smoothing), st)
            # ---- backward: decode, then the backbone (clamp fused into the gradient kernels) -------------------------
            if self.half:   # the head's gradient goes straight to bf16 NCHW8c
                g8 = torch.empty(B, ((P + 15) // 16) * 2, Sy, Sx, 8, dtype=torch.bfloat16, device=raw.device)
                _hip.call("yogo_decode_bwd_bf16", raw, pred, gpred, g8, B, P, Sy, Sx, int(bool(m.inference)), st)
                backward_bf16_train(eng, saved, g8, grad_out=self.flat.grad_views)
            else:
                graw = torch.empty_like(raw)
                _hip.call("yogo_decode_bwd", raw, pred, gpred, graw, B, P, Sy, Sx, int(bool(m.inference)), st)
                eng.backward(saved, graw, grad_out=self.flat.grad_views)
            # ---- data-parallel exchange: one RCCL all-reduce of the flat gradient --------------------------------------
            scale = 1.0
            if self.world > 1:
                dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.pg)
                scale = 1.0 / self.world
            # ---- AdamW + cosine LR (scheduler stepped every iteration) ----------------------------------------------------
            lr = self.current_lr()
            self.global_step += 1
            _hip.call("yogo_adamw_step", self.flat.flat, self.flat.grad, self.flat.exp_avg, self.flat.exp_avg_sq, self.flat.total,
                      self.global_step, lr, self.betas[0], self.betas[1], self.eps, self.wd, scale, st)
            eng.invalidate_packed()
            self.last_loss = out
            return out

    def loss_components(self) -> Dict[str, float]:
        v = self.last_loss.cpu().tolist()
        return {"loss": v[0], "iou_loss": v[1], "objectness_loss": v[2], "classification_loss": v[3]}
