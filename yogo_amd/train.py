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
* the exchange is OVERLAPPED with backward like DDP's buckets (yogo/train.py:155-159; torch fills buckets from the head side):
  the flat gradient is cut once, in front of the middle layer -- the head-side part (82 % of the bytes in base_model) is
  all-reduced on the communication stream as soon as its last gradient kernel is enqueued and travels under the backward
  pass of the input-side layers; the small remainder follows at the end of backward;
* two transports: ``comm="torch"`` (``torch.distributed``; backend "nccl" = RCCL on ROCm, gloo in the CPU tests) and
  ``comm="rccl"`` (librccl called directly through the C ABI, ``yogo_comm_*`` in include/yogo_hip.h, on a side HIP stream);
* BatchNorm statistics stay per GPU (the reference has no SyncBN).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from yogo_amd import _hip
from yogo_amd.engine import get_engine
from yogo_amd.model import YOGO
from yogo_amd.yogo_loss import YOGOLoss

_FUSED_DECODE_LOSS = True   # bf16 training: decode + loss + decode backward as one kernel (yogo_decode_loss_bwd_bf16)


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
        comm: str = "torch",
        overlap: bool = True,
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
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.flat = FlatParams(model)
        self.engine = get_engine(model.model)
        self.engine.invalidate_packed()
        self.last_loss: Optional[torch.Tensor] = None   # 4 device floats: total, iou, objectness, classification
        # tests / probes only: set to a dict and the next bf16 step leaves its forward records ("saved"), the head output ("raw")
        # and the activation gradients between the kernels (engine.backward_bf16_train) in it
        self.trace: Optional[dict] = None
        # ---- data-parallel exchange -----------------------------------------------------------------------------------
        if comm not in ("torch", "rccl"):
            raise ValueError("comm must be 'torch' (torch.distributed) or 'rccl' (librccl through the C ABI)")
        self.comm = comm
        self.overlap = bool(overlap)
        nl = len(self.engine.layers)
        self.split_layer = nl // 2      # gradients of layers >= split_layer travel while layers < split_layer are still in backward
        first = {}                      # layer index -> offset of its first parameter in the flat buffer
        off = 0
        for name, p in model.named_parameters():
            parts = name.split(".")
            if len(parts) > 1 and parts[0] == "model" and parts[1].isdigit():
                first.setdefault(int(parts[1]), off)
            off += p.numel()
        self.split_off = first.get(self.split_layer, 0)
        self._pending: list = []
        self._rccl = None
        self._comm_stream = None
        if self.world > 1 and comm == "rccl":
            self._init_rccl()

    # ---- exchange: one (overlapped: two-part) all-reduce of the flat gradient ------------------------------------------------
    def _init_rccl(self) -> None:
        """create the RCCL communicator through the C ABI; the 128-byte unique id travels through torch.distributed"""
        dev = self.flat.flat.device
        n = _hip.lib().yogo_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        if self.rank == 0:
            _hip.call("yogo_comm_unique_id", ctypes.addressof(buf))
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0, group=self.pg)
        idb = ctypes.create_string_buffer(box[0], n)
        handle = ctypes.c_void_p(0)
        with torch.cuda.device(dev):
            _hip.call("yogo_comm_init", self.rank, self.world, ctypes.addressof(idb), ctypes.addressof(handle))
            self._comm_stream = torch.cuda.Stream(device=dev)
        self._rccl = handle.value

    def close(self) -> None:
        if self._rccl is not None:
            torch.cuda.synchronize()
            _hip.call("yogo_comm_destroy", self._rccl)
            self._rccl = None

    def exchange_begin(self, lo: int, hi: int) -> None:
        """start the SUM all-reduce of flat.grad[lo:hi] over the ranks; everything enqueued on the current stream so far is
        ordered before it, later kernels of the current stream run beside it"""
        if self.world <= 1 or hi <= lo:
            return
        view = self.flat.grad[lo:hi]
        if self._rccl is not None:
            main = torch.cuda.current_stream()
            self._comm_stream.wait_stream(main)
            _hip.call("yogo_comm_allreduce_flat", self._rccl, view, hi - lo, self._comm_stream.cuda_stream)
            self._pending.append(None)
        else:
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def exchange_end(self) -> float:
        """make the current stream wait for the outstanding all-reduces; returns the factor that turns the SUM into the
        mean (folded into the AdamW kernel: yogo_adamw_step's grad_scale)"""
        if self.world <= 1:
            return 1.0
        for w in self._pending:
            if w is not None:
                w.wait()
        if self._rccl is not None and self._pending:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._pending.clear()
        return 1.0 / self.world

    def _on_layer_done(self, i: int) -> None:
        """backward hook: every gradient kernel of layers >= i has been enqueued"""
        if self.overlap and i == self.split_layer and self.split_off > 0:
            self.exchange_begin(self.split_off, self.flat.total)

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
        if self.world <= 1:
            return
        bufs = [b for b in self.model.buffers() if b.is_floating_point() or b.dtype == torch.long]
        if self._rccl is not None:
            st = torch.cuda.current_stream().cuda_stream
            _hip.call("yogo_comm_broadcast_flat", self._rccl, self.flat.flat, self.flat.total * 4, src, st)
            for b in bufs:
                if b.is_cuda and b.numel() > 0:
                    _hip.call("yogo_comm_broadcast_flat", self._rccl, b, b.numel() * b.element_size(), src, st)
        else:
            dist.broadcast(self.flat.flat, src=src, group=self.pg)
            for b in bufs:
                dist.broadcast(b, src=src, group=self.pg)

    def _bn_buffers(self) -> List[torch.Tensor]:
        """running_mean / running_var / num_batches_tracked of every BatchNorm, in module order"""
        out: List[torch.Tensor] = []
        for mod in self.model.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.running_mean is not None:
                out += [mod.running_mean, mod.running_var, mod.num_batches_tracked]
        return out

    def broadcast_buffers(self, src: int = 0) -> None:
        """rank-0 BatchNorm statistics to every rank.  torch DDP's default ``broadcast_buffers=True`` does this in front of every
        forward (yogo/train.py:155-159), so in the reference every rank validates -- and averages its validation loss,
        yogo/train.py:400 -- with RANK 0's running statistics.  Training-mode forwards use batch statistics and never read the
        buffers, so ONE broadcast in front of evaluation gives the same numbers as the reference's per-step broadcast; the
        Trainer calls this before ``_validate`` / ``test``.  The float buffers travel as one flat tensor (one collective)."""
        if self.world <= 1:
            return
        bufs = self._bn_buffers()
        fl = [b for b in bufs if b.is_floating_point()]
        cnt = [b for b in bufs if not b.is_floating_point()]
        if not fl:
            return
        flat = torch.cat([b.reshape(-1).float() for b in fl])
        # the integer counters (num_batches_tracked: the averaging factor of a momentum=None BatchNorm) travel as int64, bit for bit
        icnt = torch.cat([b.reshape(-1).to(torch.int64) for b in cnt]) if cnt else None
        if self._rccl is not None:
            st = torch.cuda.current_stream().cuda_stream
            _hip.call("yogo_comm_broadcast_flat", self._rccl, flat, flat.numel() * 4, src, st)
            if icnt is not None:
                _hip.call("yogo_comm_broadcast_flat", self._rccl, icnt, icnt.numel() * 8, src, st)
        else:
            dist.broadcast(flat, src=src, group=self.pg)
            if icnt is not None:
                dist.broadcast(icnt, src=src, group=self.pg)
        off = 0
        for b in fl:
            n = b.numel()
            b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
            off += n
        off = 0
        for b in cnt:
            n = b.numel()
            b.copy_(icnt[off:off + n].view(b.shape).to(b.dtype))
            off += n
        self.engine.generation += 1   # folded inference weights derive from the running statistics

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
                if self.trace is not None:
                    self.trace["saved"], self.trace["raw"] = saved, raw
            else:
                raw, saved = eng.forward(imgs, need_grad=True)
            B, P, Sy, Sx = raw.shape
            aw, ah, wm, hm = m._decode_scalars()
            L = self.loss
            out = torch.empty(4, dtype=torch.float32, device=raw.device)
            ws = torch.empty(_hip.query_size("yogo_loss_workspace_bytes", B, Sy, Sx) // 4, dtype=torch.float32, device=raw.device)
            lab = labels if (labels.dtype == torch.float32 and labels.is_contiguous()) else labels.contiguous().float()
            hook = self._on_layer_done if self.world > 1 else None
            # the hook needs complete gradients at split_layer only: the deferred split-K reductions are flushed there and at the end
            flush = (self.split_layer,) if hook is not None else None
            fused = self.half and not m.inference and _FUSED_DECODE_LOSS
            if fused:
                # ---- decode + loss forward/backward + decode backward in one pass over the cells (bit-identical to the three calls
                #      below; the decoded prediction and its gradient never go to memory) ------------------------------------------
                g8 = torch.empty(B, ((P + 15) // 16) * 2, Sy, Sx, 8, dtype=torch.bfloat16, device=raw.device)
                _hip.call("yogo_decode_loss_bwd_bf16", raw, lab, m._Cxs, m._Cys, g8, out, ws, B, P, Sy, Sx, aw, ah, wm, hm,
                          float(L.no_obj_weight), float(L.iou_weight), float(L.classify_weight), float(L.label_smoothing), st)
                backward_bf16_train(eng, saved, g8, grad_out=self.flat.grad_views, on_layer=hook, trace=self.trace, flush_layers=flush)
            else:
                pred = torch.empty_like(raw)
                _hip.call("yogo_decode_fwd", raw, pred, m._Cxs, m._Cys, B, P, Sy, Sx, aw, ah, wm, hm, int(bool(m.inference)), st)
                # ---- loss forward + backward (one kernel) -------------------------------------------------------------
                gpred = torch.empty_like(raw)
                _hip.call("yogo_loss_fwd_bwd", pred, lab, gpred, out, ws, B, P, Sy, Sx, float(L.no_obj_weight), float(L.iou_weight),
                          float(L.classify_weight), float(L.label_smoothing), st)
            # ---- backward: decode, then the backbone (clamp fused into the gradient kernels) -------------------------
            if fused:
                pass
            elif self.half:   # the head's gradient goes straight to bf16 NCHW8c
                g8 = torch.empty(B, ((P + 15) // 16) * 2, Sy, Sx, 8, dtype=torch.bfloat16, device=raw.device)
                _hip.call("yogo_decode_bwd_bf16", raw, pred, gpred, g8, B, P, Sy, Sx, int(bool(m.inference)), st)
                backward_bf16_train(eng, saved, g8, grad_out=self.flat.grad_views, on_layer=hook, trace=self.trace, flush_layers=flush)
            else:
                graw = torch.empty_like(raw)
                _hip.call("yogo_decode_bwd", raw, pred, gpred, graw, B, P, Sy, Sx, int(bool(m.inference)), st)
                eng.backward(saved, graw, grad_out=self.flat.grad_views, on_layer=hook)
            # ---- data-parallel exchange: the rest of the flat gradient (all of it without overlap), then the join ---------------
            started = self.split_off if (self.overlap and self._pending) else self.flat.total
            self.exchange_begin(0, started)
            scale = self.exchange_end()
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
