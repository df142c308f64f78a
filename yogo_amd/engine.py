"""Host-side executor of the backbone on the HIP kernels (libyogo_hip.so).

Walks a :class:`~yogo_amd.model_defns.HipBackbone` (block grammar of yogo/model_defns.py: Conv2d, [BatchNorm2d],
[LeakyReLU|SiLU], [Dropout2d]) into a layer table and runs forward / backward by calling the C ABI with raw device
pointers on the current HIP stream.  PyTorch supplies device memory, the stream and the autograd edge only.

Fusion plan (fp32):
  block without BN :  conv + bias + activation + Dropout2d channel mask            -> 1 kernel
  block with BN    :  conv (+bias) + per-workgroup BN partial sums | finalize | normalise + activation
  backward         :  [BN backward] -> wgrad (+bias grad, +clamp) -> dgrad whose epilogue applies the previous block's
                      activation derivative and dropout mask, so no separate element-wise backward passes exist.
The per-parameter gradient clamp of yogo/model.py:76-77 is fused into the gradient-finishing kernels.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from yogo_amd import _hip

ACT_NONE, ACT_LEAKY, ACT_SILU = 0, 1, 2


@dataclass
class Layer:
    conv: nn.Conv2d
    bn: Optional[nn.BatchNorm2d] = None
    act: int = ACT_NONE
    drop: Optional[nn.Dropout2d] = None

    @property
    def cin(self) -> int:
        return self.conv.in_channels

    @property
    def cout(self) -> int:
        return self.conv.out_channels

    @property
    def k(self) -> int:
        return self.conv.kernel_size[0]

    @property
    def s(self) -> int:
        return self.conv.stride[0]

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        p = 1 if self.k == 3 else 0
        return (h + 2 * p - self.k) // self.s + 1, (w + 2 * p - self.k) // self.s + 1


@dataclass
class Saved:
    x_in: torch.Tensor
    y: Optional[torch.Tensor] = None
    z: Optional[torch.Tensor] = None       # conv output of a BN block
    pre: Optional[torch.Tensor] = None     # pre-activation (SiLU blocks)
    signs: Optional[torch.Tensor] = None   # sign map of y (LeakyReLU blocks without BatchNorm, bf16 path)
    signs0: Optional[torch.Tensor] = None  # layer 0 on the matrix cores: sign map of its BatchNorm output -- kept INSTEAD of z
    w_used: Optional[torch.Tensor] = None  # layer 0 on the matrix cores: the bf16-rounded weights the forward multiplied with
    gram: Optional[torch.Tensor] = None    # layer 0: float[90] patch sums P and Gram matrix G of the batch (reused by backward)
    mask: Optional[torch.Tensor] = None    # Dropout2d channel mask, already scaled
    mean: Optional[torch.Tensor] = None
    invstd: Optional[torch.Tensor] = None
    bn_train: bool = False


def parse_backbone(backbone: nn.Sequential) -> List[Layer]:
    layers: List[Layer] = []
    for blk in backbone:
        mods = [blk] if isinstance(blk, nn.Conv2d) else list(blk)
        if not mods or not isinstance(mods[0], nn.Conv2d):
            raise RuntimeError(f"yogo_amd: unsupported block {blk!r}: every block must start with nn.Conv2d")
        conv = mods[0]
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        ok = k[0] == k[1] and s[0] == s[1] and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
        ok = ok and ((k[0] == 3 and p == (1, 1) and s[0] in (1, 2)) or (k[0] == 1 and p == (0, 0) and s[0] == 1))
        if not ok:
            raise RuntimeError(f"yogo_amd: unsupported convolution {conv!r} (3x3 pad 1 stride 1|2, or 1x1)")
        L = Layer(conv=conv)
        for m in mods[1:]:
            if isinstance(m, nn.BatchNorm2d):
                L.bn = m
            elif isinstance(m, nn.LeakyReLU):
                if abs(m.negative_slope - 0.01) > 1e-12:
                    raise RuntimeError("yogo_amd: LeakyReLU slope must be 0.01")
                L.act = ACT_LEAKY
            elif isinstance(m, nn.SiLU):
                L.act = ACT_SILU
            elif isinstance(m, nn.Dropout2d):
                L.drop = m
            else:
                raise RuntimeError(f"yogo_amd: unsupported module {m!r} in block")
        if L.bn is not None and L.drop is not None:
            raise RuntimeError("yogo_amd: BatchNorm + Dropout2d in one block is not part of the block grammar")
        layers.append(L)
    return layers


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


class Engine:
    def __init__(self, backbone: nn.Sequential):
        self.backbone_ref = weakref.ref(backbone)
        self.layers = parse_backbone(backbone)
        self._pack: Dict[Tuple[int, int], Tuple[int, int, torch.Tensor]] = {}
        self.clip: float = 0.0  # > 0: clamp parameter gradients to +-clip (set by YOGO from clip_value)
        # optional per-launch HIP-event timing (bench.py): list of (kind, layer, mw, flops, start_event, end_event)
        self.prof: Optional[list] = None
        self.prof_only: Optional[set] = None   # bracket only the launches whose `mw` tag is in this set (bench.py's timed region)
        self._open = None
        # bumped whenever a kernel writes parameters or BatchNorm buffers through raw pointers (AdamW on the flat buffer,
        # bn_finalize / bn_stats_from_gram on the running statistics): torch's _version does not see those writes, so every
        # cache of derived tensors (packed weights, folded inference weights) keys on this counter as well
        self.generation: int = 0

    def _tick(self, kind: str, layer: int, flops: float, mw: int = 0, nbytes: float = 0.0) -> None:
        """open a HIP-event bracket around one kernel call (bench.py): algorithmic FLOPs and bytes of that call"""
        if self.prof is not None and (self.prof_only is None or mw in self.prof_only):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._open = (kind, layer, mw, flops, e0, e1, nbytes)

    def _tock(self) -> None:
        if self._open is not None:
            self._open[5].record()
            self.prof.append(self._open)
            self._open = None

    # ------------------------------------------------------------------------------------------------------
    def invalidate_packed(self) -> None:
        """parameters / BatchNorm buffers were rewritten behind torch's back: drop every derived tensor"""
        self._pack.clear()
        self.generation += 1

    def _packed(self, i: int, mode: int) -> torch.Tensor:
        L = self.layers[i]
        w = L.conv.weight
        key = (i, mode)
        hit = self._pack.get(key)
        if hit is not None and hit[0] == w._version and hit[1] == w.data_ptr():
            return hit[2]
        nbytes = _hip.query_size("yogo_conv_packed_bytes", L.cin, L.cout, L.k, L.s, mode)
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        _hip.call("yogo_conv_pack_f32", _f32(w.detach()), buf, L.cin, L.cout, L.k, L.s, mode, _hip.stream_ptr())
        self._pack[key] = (w._version, w.data_ptr(), buf)
        return buf

    def _first_direct(self, i: int) -> bool:
        L = self.layers[i]
        return i == 0 and L.cin in (1, 3) and L.k == 3

    # ------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, need_grad: bool) -> Tuple[torch.Tensor, List[Saved]]:
        _hip.require_cuda(x, "the input batch")
        if x.ndim != 4:
            raise RuntimeError(f"yogo_amd: expected a [B,C,H,W] batch, got {tuple(x.shape)}")
        dev = x.device
        st = _hip.stream_ptr()
        B = x.shape[0]
        if x.dtype == torch.uint8:
            cur = x.contiguous()
        else:
            cur = _f32(x)
        saved: List[Saved] = []
        for i, L in enumerate(self.layers):
            if cur.shape[1] != L.cin:
                raise RuntimeError(f"yogo_amd: layer {i} expects {L.cin} channels, got {cur.shape[1]}")
            IH, IW = int(cur.shape[2]), int(cur.shape[3])
            OH, OW = L.out_hw(IH, IW)
            if OH <= 0 or OW <= 0:
                raise RuntimeError(f"yogo_amd: image too small at layer {i}")
            w = L.conv.weight
            _hip.require_cuda(w, "the model parameters")
            bias = _f32(L.conv.bias.detach()) if L.conv.bias is not None else None
            S = Saved(x_in=cur)
            mask = None
            if L.drop is not None and L.drop.training and L.drop.p > 0:
                p = float(L.drop.p)
                mask = (torch.rand(B, L.cout, device=dev) >= p).to(torch.float32) / (1.0 - p)
                S.mask = mask
            has_bn = L.bn is not None
            bn_train = has_bn and (L.bn.training or L.bn.running_mean is None)
            out = torch.empty(B, L.cout, OH, OW, dtype=torch.float32, device=dev)
            pre = None
            if (not has_bn) and need_grad and L.act == ACT_SILU:
                pre = torch.empty_like(out)
            stats = None
            rows = mpad = 0
            direct = self._first_direct(i)
            if bn_train:
                if direct:
                    rows = _hip.query_ints("yogo_conv_first_stats_rows", 1, B, IH, IW, L.s)[0]
                    mpad = L.cout
                else:
                    rows, mpad = _hip.query_ints("yogo_conv2d_fwd_stats_shape", 2, B, L.cin, L.cout, IH, IW, L.k, L.s)
                stats = torch.empty(rows * mpad * 2, dtype=torch.float32, device=dev)
            fused_act = ACT_NONE if has_bn else L.act
            if direct:
                _hip.call("yogo_conv_first_fwd", cur, 0 if cur.dtype == torch.uint8 else 1, _f32(w.detach()), bias, out, pre,
                          mask, stats, B, L.cin, L.cout, IH, IW, L.s, fused_act, st)
            else:
                if cur.dtype != torch.float32:
                    cur = cur.float()
                    S.x_in = cur
                pk = self._packed(i, 0)
                self._tick("fwd", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW, mw=4 if L.cout > 64 else (2 if L.cout > 32 else 1))
                _hip.call("yogo_conv2d_fwd_f32", cur, pk, bias, out, pre, mask, stats, B, L.cin, L.cout, IH, IW, L.k, L.s,
                          fused_act, st)
                self._tock()
            if has_bn:
                bn = L.bn
                gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
                beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
                y = torch.empty_like(out) if need_grad else out
                if bn_train:
                    mean = torch.empty(L.cout, dtype=torch.float32, device=dev)
                    invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                    track = bn.track_running_stats and bn.running_mean is not None
                    if track and bn.momentum is None:
                        raise RuntimeError("yogo_amd: BatchNorm2d(momentum=None) is not supported")
                    _hip.call("yogo_bn_finalize", stats, rows, mpad, L.cout, B * OH * OW, float(bn.eps),
                              float(bn.momentum if bn.momentum is not None else 0.0), mean, invstd,
                              bn.running_mean if track else None, bn.running_var if track else None,
                              bn.num_batches_tracked if track else None, st)
                    if track:
                        self.generation += 1   # running statistics rewritten by raw pointer
                    _hip.call("yogo_bn_apply_act", out, y, mean, invstd, 0, float(bn.eps), gamma, beta, B, L.cout, OH * OW,
                              L.act, st)
                    S.mean, S.invstd = mean, invstd
                else:
                    _hip.call("yogo_bn_apply_act", out, y, bn.running_mean, bn.running_var, 1, float(bn.eps), gamma, beta, B,
                              L.cout, OH * OW, L.act, st)
                    if need_grad:
                        invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                        _hip.call("yogo_bn_invstd", bn.running_var, float(bn.eps), invstd, L.cout, st)
                        S.mean, S.invstd = bn.running_mean, invstd
                S.bn_train = bn_train
                S.z = out if need_grad else None
                S.y = y
                cur = y
            else:
                S.y = out
                S.pre = pre
                cur = out
            saved.append(S if need_grad else Saved(x_in=cur))
        return cur, saved

    # ------------------------------------------------------------------------------------------------------
    def backward(self, saved: List[Saved], graw: torch.Tensor,
                 grad_out: Optional[Dict[int, torch.Tensor]] = None, on_layer=None) -> List[Optional[torch.Tensor]]:
        """returns gradients in the order of ``backbone.parameters()``.  ``grad_out`` maps id(param) to a preallocated
        contiguous destination (views of one flat gradient buffer in the trainer).  ``on_layer(i)`` is called once every
        gradient kernel of layers >= i has been enqueued (the trainer starts the all-reduce of that part there)."""
        st = _hip.stream_ptr()
        dev = graw.device
        clip = float(self.clip)
        grads: Dict[int, torch.Tensor] = {}

        def dst(param: torch.Tensor) -> torch.Tensor:
            if grad_out is not None and id(param) in grad_out:
                return grad_out[id(param)]
            return torch.empty(param.shape, dtype=torch.float32, device=dev)
        g = _f32(graw)
        n = len(self.layers)
        if g is graw and self.layers[-1].bn is not None:
            g = g.clone()   # yogo_bn_bwd writes dz in place: never into the gradient autograd handed us
        for i in range(n - 1, -1, -1):
            L, S = self.layers[i], saved[i]
            B, _, OH, OW = g.shape
            IH, IW = int(S.x_in.shape[2]), int(S.x_in.shape[3])
            if L.bn is not None:
                bn = L.bn
                gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
                beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
                dgamma = dst(bn.weight) if bn.weight is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
                dbeta = dst(bn.bias) if bn.bias is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
                rows = _hip.query_ints("yogo_bn_bwd_rows", 1, B, OH * OW)[0]
                part = torch.empty(rows * L.cout * 2, dtype=torch.float32, device=dev)
                sums = torch.empty(2 * L.cout, dtype=torch.float32, device=dev)
                # g arrives as dL/d(block output): the BN-backward kernels apply the activation derivative themselves
                _hip.call("yogo_bn_bwd", g, S.z, g, S.mean, S.invstd, gamma, beta, L.act, dgamma, dbeta, part, sums, B, L.cout,
                          OH * OW, 1 if S.bn_train else 0, clip, st)
                if bn.weight is not None:
                    grads[id(bn.weight)] = dgamma
                    grads[id(bn.bias)] = dbeta
            # ---- weight / bias gradient -------------------------------------------------------------------------
            dw = dst(L.conv.weight)
            has_bias = L.conv.bias is not None
            if self._first_direct(i):
                rows = _hip.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, L.s)[0]
                nj = L.cin * 9 + 1
                part = torch.empty(rows * L.cout * nj, dtype=torch.float32, device=dev)
                x_in = S.x_in
                _hip.call("yogo_conv_first_wgrad", x_in, 0 if x_in.dtype == torch.uint8 else 1, g, part, B, L.cin, L.cout, IH,
                          IW, L.s, st)
                red = torch.empty(L.cout, nj, dtype=torch.float32, device=dev)
                _hip.call("yogo_partials_reduce", part, rows, L.cout * nj, clip, red, st)
                dw.copy_(red[:, : nj - 1].reshape(dw.shape))
                if has_bias:
                    db = dst(L.conv.bias)
                    db.copy_(red[:, nj - 1])
                    grads[id(L.conv.bias)] = db
            else:
                wsb = _hip.query_size("yogo_conv2d_wgrad_workspace_bytes", B, L.cin, L.cout, IH, IW, L.k, L.s)
                ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
                db = dst(L.conv.bias) if has_bias else None
                self._tick("wgrad", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW)
                _hip.call("yogo_conv2d_wgrad_f32", S.x_in, g, dw, db, ws, B, L.cin, L.cout, IH, IW, L.k, L.s, clip, st)
                self._tock()
                if has_bias:
                    grads[id(L.conv.bias)] = db
            grads[id(L.conv.weight)] = dw
            if on_layer is not None:
                on_layer(i)
            # ---- data gradient, with the previous block's activation derivative and dropout mask fused -------------
            if i > 0:
                Lp, Sp = self.layers[i - 1], saved[i - 1]
                ref_act = Lp.act
                if Lp.bn is not None:
                    act_ref, ref_act = None, ACT_NONE   # BatchNorm backward applies the activation derivative
                elif Lp.act == ACT_SILU:
                    act_ref = Sp.pre
                    if act_ref is None:
                        raise RuntimeError("yogo_amd: missing saved pre-activation for a SiLU block")
                elif Lp.act == ACT_LEAKY:
                    act_ref = Sp.y
                else:
                    act_ref = None
                dx = torch.empty(B, L.cin, IH, IW, dtype=torch.float32, device=dev)
                pk = self._packed(i, 1)
                # mw tags the kernel instantiation for bench.py's roofline: stride-2 dgrad runs the fused <.,4,true> kernel
                mw_tag = 20 if L.s == 2 else (4 if L.cin > 64 else (2 if L.cin > 32 else 1))
                self._tick("dgrad", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW, mw=mw_tag)
                _hip.call("yogo_conv2d_dgrad_f32", g, pk, dx, act_ref, ref_act, Sp.mask, B, L.cin, L.cout, IH, IW, L.k, L.s, st)
                self._tock()
                g = dx
        bb = self.backbone_ref()
        return [grads.get(id(p)) for p in bb.parameters()]



# ---------------------------------------------------------------------------------------------------------------------
# bf16 training: activations / activation gradients in NCHW8c bf16, fp32 weights, statistics and parameter gradients
# (the reference's `--half` training is fp16 autocast, yogo/train.py:315-318; bf16 needs no loss scaling)
# ---------------------------------------------------------------------------------------------------------------------
def _blocks(c: int) -> int:
    return ((c + 15) // 16) * 2


# Execution plan of the bf16 training path.  Plain module constants (no environment switches): the alternatives they once
# selected were measured in round 1 ("measured and dropped") and only the winners stayed; the tests flip
# _FUSE_LAYER0_BWD / _L0_GRAM to compare the fused kernels with the separate passes they replace.
_WGRAD_BF16_MFMA = True     # weight gradients on the bf16 matrix cores (False: exact fp32 MFMA on the widened inputs)
# weight gradients on a second HIP stream, beside the data-gradient / BatchNorm-backward chain they do not feed (-2 % step
# time).  bench.py switches it off while it times single kernels with HIP events (two kernels sharing the chip are not
# attributable).
_WGRAD_SIDE_STREAM = False
_FUSE_LAYER0_BWD = True     # BatchNorm backward + activation derivative + first-conv weight gradient in one sweep
_LEAKY_SIGNS = True         # LeakyReLU blocks without BatchNorm hand the next data gradient a 1-bit sign map, not the bf16 output
_L0_MFMA = True             # layer 0 (uint8 image, 1 -> <=16 channels, stride 2, BatchNorm) on the matrix cores
_L0_GRAM = True             # ... with the batch statistics from the exact integer patch Gram matrix (backward reuses it)
_WGRAD_DEFER_REDUCE = True  # the per-layer split-K reductions of the weight gradients in one launch at the end of the backward pass
_L0_NO_Z = True             # ... and without its conv output in memory: sign map + derived sums (yogo_conv_first_*_xs)
_PACK_MULTI = True          # all weight packings of a step in one launch
_L01_FUSE_BWD = True        # layer 1's data gradient folded into layer 0's backward sums (yogo_conv2d_dgrad_bf16_first_bwd): no dy of layer 0 in memory
_HEAD_BN_FUSE = True        # the 1x1 head's data gradient computed inside the BatchNorm backward of the block under it (yogo_bn_bwd_bf16_head)
_BN_STATS_PASS = True       # BatchNorm statistics of layers > 0 by a sweep over the stored bf16 output (not the conv epilogue)
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _packed_bf16(eng: Engine, i: int, mode: int) -> torch.Tensor:
    L = eng.layers[i]
    w = L.conv.weight
    key = (i, 10 + mode)
    hit = eng._pack.get(key)
    if hit is not None and hit[0] == w._version and hit[1] == w.data_ptr():
        return hit[2]
    nbytes = _hip.query_size("yogo_conv_bf16_packed_bytes", L.cin, L.cout, L.k, mode)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _hip.call("yogo_conv_bf16_pack", _f32(w.detach()), None, buf, L.cin, L.cout, L.k, mode, _hip.stream_ptr())
    eng._pack[key] = (w._version, w.data_ptr(), buf)
    return buf


def _pack_all_bf16(eng: Engine) -> None:
    """Forward and data-gradient packings of every matrix-core layer in ONE launch (the weights change every step)."""
    jobs = [(i, mode) for i in range(1, len(eng.layers)) for mode in (0, 2 if (eng.layers[i].s == 2 and eng.layers[i].k == 3) else 1)]
    ws = [eng.layers[i].conv.weight for i, _ in jobs]
    if any(w.dtype != torch.float32 or not w.is_contiguous() for w in ws) or not jobs or not _PACK_MULTI:
        return   # (the per-layer path converts)
    if all((h := eng._pack.get((i, 10 + m))) is not None and h[0] == w._version and h[1] == w.data_ptr() for (i, m), w in zip(jobs, ws)):
        return
    dev = ws[0].device
    key = tuple(w.data_ptr() for w in ws)
    st = getattr(eng, "_pack_multi", None)
    if st is None or st[0] != key:
        bufs, rows, blk = [], [], 0
        for (i, mode), w in zip(jobs, ws):
            L = eng.layers[i]
            buf = torch.empty(_hip.query_size("yogo_conv_bf16_packed_bytes", L.cin, L.cout, L.k, mode), dtype=torch.uint8, device=dev)
            bufs.append(buf)
            rows.append([w.data_ptr(), 0, buf.data_ptr(), L.cin, L.cout, L.k, mode, blk])
            blk += _hip.query_ints("yogo_conv_bf16_pack_blocks", 1, L.cin, L.cout, L.k, mode)[0]
        st = (key, bufs, torch.tensor(rows, dtype=torch.int64, device=dev), blk)
        eng._pack_multi = st
    _, bufs, table, blk = st
    _hip.call("yogo_conv_bf16_pack_multi", table, len(jobs), blk, _hip.stream_ptr())
    for (i, mode), w, buf in zip(jobs, ws, bufs):
        eng._pack[(i, 10 + mode)] = (w._version, w.data_ptr(), buf)


def _dropout_masks(eng: Engine, B: int, dev) -> Dict[int, torch.Tensor]:
    """Dropout2d channel masks (already scaled by 1 / (1 - p)) of all layers from ONE torch.rand call."""
    act = [(i, float(L.drop.p)) for i, L in enumerate(eng.layers) if L.drop is not None and L.drop.training and L.drop.p > 0]
    if not act:
        return {}
    key = (B, tuple(act), str(dev))
    cached = getattr(eng, "_drop_consts", None)
    if cached is None or cached[0] != key:
        sizes = [B * eng.layers[i].cout for i, _ in act]
        pvec = torch.cat([torch.full((n,), p, dtype=torch.float32) for n, (_, p) in zip(sizes, act)]).to(dev)
        cached = (key, sizes, pvec, 1.0 / (1.0 - pvec))
        eng._drop_consts = cached
    _, sizes, pvec, inv_keep = cached
    m = (torch.rand(pvec.numel(), device=dev) >= pvec).to(torch.float32) * inv_keep
    out, off = {}, 0
    for n, (i, _) in zip(sizes, act):
        out[i] = m[off:off + n].view(B, eng.layers[i].cout)
        off += n
    return out


def forward_bf16_train(eng: Engine, x: torch.Tensor) -> Tuple[torch.Tensor, List[Saved]]:
    _hip.require_cuda(x, "the input batch")
    if x.ndim != 4:
        raise RuntimeError(f"yogo_amd: expected a [B,C,H,W] batch, got {tuple(x.shape)}")
    if x.shape[1] != eng.layers[0].cin:
        raise RuntimeError(f"yogo_amd: layer 0 expects {eng.layers[0].cin} channels, got {x.shape[1]}")
    for j in range(1, len(eng.layers)):
        if eng.layers[j].cin != eng.layers[j - 1].cout:
            raise RuntimeError(f"yogo_amd: layer {j} expects {eng.layers[j].cin} channels, layer {j - 1} produces {eng.layers[j - 1].cout}")
    for j, Lj in enumerate(eng.layers):
        bnj = Lj.bn
        if (bnj is not None and (bnj.training or bnj.running_mean is None) and bnj.track_running_stats
                and bnj.running_mean is not None and bnj.momentum is None):
            raise RuntimeError("yogo_amd: BatchNorm2d(momentum=None) is not supported")
    dev, st, B = x.device, _hip.stream_ptr(), x.shape[0]
    if not eng._first_direct(0):
        raise RuntimeError("yogo_amd: bf16 training needs a 1- or 3-channel 3x3 first convolution")
    _pack_all_bf16(eng)
    masks = _dropout_masks(eng, B, dev)
    cur = x.contiguous() if x.dtype == torch.uint8 else _f32(x)
    H, W = int(cur.shape[2]), int(cur.shape[3])
    saved: List[Saved] = []
    n = len(eng.layers)
    for i, L in enumerate(eng.layers):
        OH, OW = L.out_hw(H, W)
        if OH <= 0 or OW <= 0:
            raise RuntimeError(f"yogo_amd: image too small at layer {i}")
        last = i == n - 1
        has_bn = L.bn is not None
        silu_pre = L.act == ACT_SILU and not has_bn   # silu'(z) needs the pre-activation: the conv writes it next to the output
        if silu_pre and (i == 0 or last):
            raise RuntimeError("yogo_amd: bf16 training of a first / last SiLU block without BatchNorm is not implemented (use fp32)")
        bias = _f32(L.conv.bias.detach()) if L.conv.bias is not None else None
        S = Saved(x_in=cur)
        mask = None
        if i in masks:
            mask = masks[i]
            S.mask = mask
        bn_train = has_bn and (L.bn.training or L.bn.running_mean is None)
        if (i == 0 and _L0_MFMA and has_bn and mask is None and not last and cur.dtype == torch.uint8
                and _hip.lib().yogo_conv_first_mfma_supported(0, L.cin, L.cout, H, W, L.s)):
            bn = L.bn
            gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
            beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
            w32 = _f32(L.conv.weight.detach())
            y = torch.empty(B, _blocks(L.cout), OH, OW, 8, dtype=torch.bfloat16, device=dev)
            if bn_train and _L0_GRAM:   # sweep 1: exact integer patch sums -> statistics of all channels (and backward's G)
                rows = _hip.query_ints("yogo_conv_first_gram_rows", 1, B, H, W)[0]
                gpart = torch.empty(rows * 54, dtype=torch.int32, device=dev)
                gram64 = torch.empty(90, dtype=torch.float64, device=dev)
                S.gram = torch.empty(90, dtype=torch.float32, device=dev)
                _hip.call("yogo_conv_first_gram", cur, gpart, gram64, S.gram, B, H, W, st)
                mean = torch.empty(L.cout, dtype=torch.float32, device=dev)
                invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                track = bn.track_running_stats and bn.running_mean is not None
                _hip.call("yogo_bn_stats_from_gram", gram64, w32, bias, L.cout, B * OH * OW, float(bn.eps),
                          float(bn.momentum if bn.momentum is not None else 0.0), mean, invstd,
                          bn.running_mean if track else None, bn.running_var if track else None,
                          bn.num_batches_tracked if track else None, st)
                if track:
                    eng.generation += 1
            elif bn_train:   # sweep 1: batch statistics, nothing written
                rows = _hip.query_ints("yogo_conv_first_mfma_stats_rows", 1, B, H, W)[0]
                stats = torch.empty(rows * 16 * 2, dtype=torch.float32, device=dev)
                _hip.call("yogo_conv_first_mfma", cur, w32, bias, None, None, None, None, None, None, stats, B, L.cout, H, W, L.act, st)
                mean = torch.empty(L.cout, dtype=torch.float32, device=dev)
                invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                track = bn.track_running_stats and bn.running_mean is not None
                _hip.call("yogo_bn_finalize", stats, rows, 16, L.cout, B * OH * OW, float(bn.eps),
                          float(bn.momentum if bn.momentum is not None else 0.0), mean, invstd,
                          bn.running_mean if track else None, bn.running_var if track else None,
                          bn.num_batches_tracked if track else None, st)
                if track:
                    eng.generation += 1
            else:
                invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                _hip.call("yogo_bn_invstd", bn.running_var, float(bn.eps), invstd, L.cout, st)
                mean = bn.running_mean
            # sweep 2: the convolution again, y = act(BatchNorm(z)) and what backward needs of z: its sign map when the fused backward
            # sweep can take that (it derives everything else from its own sums), else z itself
            no_z = (_L0_NO_Z and _FUSE_LAYER0_BWD and S.gram is not None and L.conv.bias is None
                    and _hip.lib().yogo_conv_first_bn_wgrad_xs_supported(0, L.cin, L.cout, H, W, L.s, L.act))
            if no_z:
                out8 = None
                S.signs0 = torch.empty(B, OH * OW * 2, dtype=torch.uint8, device=dev)
                _hip.call("yogo_conv_first_mfma_signs", cur, w32, bias, None, y, S.signs0, mean, invstd, gamma, beta, B, L.cout, H, W, L.act, st)
            else:
                out8 = torch.empty_like(y)
                _hip.call("yogo_conv_first_mfma", cur, w32, bias, out8, y, mean, invstd, gamma, beta, None, B, L.cout, H, W, L.act, st)
            S.mean, S.invstd, S.bn_train, S.z, S.y = mean, invstd, bn_train, out8, y
            S.w_used = w32.to(torch.bfloat16).to(torch.float32)
            cur = y
            saved.append(S)
            H, W = OH, OW
            continue
        fused_act = ACT_NONE if has_bn else L.act
        stats = None
        rows = mpad = 0
        stats_pass = bn_train and i > 0 and _BN_STATS_PASS   # statistics by a separate sweep over the stored output
        if bn_train and not stats_pass:
            if i == 0:
                rows, mpad = _hip.query_ints("yogo_conv_first_stats_rows", 1, B, H, W, L.s)[0], L.cout
            else:
                rows, mpad = _hip.query_ints("yogo_conv2d_fwd_bf16_stats_shape", 2, B, L.cin, L.cout, H, W, L.k, L.s)
            stats = torch.empty(rows * mpad * 2, dtype=torch.float32, device=dev)
        out8 = None if last else torch.empty(B, _blocks(L.cout), OH, OW, 8, dtype=torch.bfloat16, device=dev)
        out32 = torch.empty(B, L.cout, OH, OW, dtype=torch.float32, device=dev) if last else None
        if last and (has_bn or mask is not None):
            raise RuntimeError("yogo_amd: BatchNorm / Dropout on the last layer is not supported by the bf16 path")
        if i == 0:
            _hip.call("yogo_conv_first_fwd_train_bf16", cur, 0 if cur.dtype == torch.uint8 else 1, _f32(L.conv.weight.detach()), bias,
                      out8, mask, stats, B, L.cin, L.cout, H, W, L.s, fused_act, st)
        else:
            pk = _packed_bf16(eng, i, 0)
            # algorithmic bytes: input + output once at storage precision (bf16 NCHW8c, channels padded to 16; fp32 head)
            nbytes = B * (_blocks(L.cin) * 8 * H * W * 2 + (L.cout * OH * OW * 4 if last else _blocks(L.cout) * 8 * OH * OW * 2))
            # mw tags the kernel variant for bench.py: 34 = conv_bf16_kernel<4,2,8,...> (128 output channels, stride 1)
            eng._tick("fwd", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW, mw=34 if (L.cout > 64 and L.s == 1) else (35 if (L.cout > 64 and L.cin > 64 and L.k == 3) else 30), nbytes=nbytes)
            if silu_pre:
                S.pre = torch.empty_like(out8)
                _hip.call("yogo_conv2d_fwd_bf16_pre", cur, pk, bias, out8, S.pre, mask, B, L.cin, L.cout, H, W, L.k, L.s, fused_act, st)
            elif _LEAKY_SIGNS and L.act == ACT_LEAKY and not has_bn and not last:
                S.signs = torch.empty(_hip.query_size("yogo_bf16_signs_bytes", B, L.cout, OH, OW), dtype=torch.uint8, device=dev)
                _hip.call("yogo_conv2d_fwd_bf16_signs", cur, pk, bias, out8, S.signs, mask, B, L.cin, L.cout, H, W, L.k, L.s, fused_act, st)
            else:
                _hip.call("yogo_conv2d_fwd_bf16", cur, pk, bias, out8, out32, mask, stats, B, L.cin, L.cout, H, W, L.k, L.s, fused_act, st)
            eng._tock()
        if has_bn:
            bn = L.bn
            gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
            beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
            y = torch.empty_like(out8)
            if bn_train:
                if stats_pass:
                    rows, mpad = _hip.query_ints("yogo_bn_bwd_bf16_rows", 1, B, OH * OW)[0], L.cout
                    stats = torch.empty(rows * mpad * 2, dtype=torch.float32, device=dev)
                    _hip.call("yogo_bn_stats_bf16", out8, stats, B, L.cout, OH * OW, st)
                mean = torch.empty(L.cout, dtype=torch.float32, device=dev)
                invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                track = bn.track_running_stats and bn.running_mean is not None
                _hip.call("yogo_bn_finalize", stats, rows, mpad, L.cout, B * OH * OW, float(bn.eps),
                          float(bn.momentum if bn.momentum is not None else 0.0), mean, invstd,
                          bn.running_mean if track else None, bn.running_var if track else None,
                          bn.num_batches_tracked if track else None, st)
                if track:
                    eng.generation += 1
                _hip.call("yogo_bn_apply_act_bf16", out8, y, mean, invstd, 0, float(bn.eps), gamma, beta, B, L.cout, OH * OW, L.act, st)
            else:
                invstd = torch.empty(L.cout, dtype=torch.float32, device=dev)
                _hip.call("yogo_bn_invstd", bn.running_var, float(bn.eps), invstd, L.cout, st)
                mean = bn.running_mean
                _hip.call("yogo_bn_apply_act_bf16", out8, y, mean, invstd, 0, float(bn.eps), gamma, beta, B, L.cout, OH * OW, L.act, st)
            S.mean, S.invstd, S.bn_train, S.z, S.y = mean, invstd, bn_train, out8, y
            cur = y
        else:
            S.y = out32 if last else out8
            cur = S.y
        saved.append(S)
        H, W = OH, OW
    return cur, saved


def backward_bf16_train(eng: Engine, saved: List[Saved], graw: torch.Tensor,
                        grad_out: Optional[Dict[int, torch.Tensor]] = None, on_layer=None,
                        trace: Optional[dict] = None, flush_layers=None) -> List[Optional[torch.Tensor]]:
    """``on_layer(i)`` is called once every gradient kernel of layer i has been enqueued.  With ``flush_layers`` (layer indices) the
    hook only needs COMPLETE parameter gradients at those layers: the split-K reductions stay deferred and are flushed in front of
    the hook there (and behind the last layer) -- two launches for the data-parallel trainer's two-part exchange instead of one
    reduction per layer, the same bits.  Without it a hook gets every layer's gradients reduced as soon as the layer is through.
    ``trace`` (tests / probes only): a dict that receives clones of the activation gradients as they exist between the
    kernels -- ("g", i): gradient w.r.t. block i's output, ("dz", i): BatchNorm-backward output of block i -- so that every
    kernel of a real step can be checked against the oracle given its ACTUAL inputs (tests/_util.py, teacher-forced check)."""
    st, dev, clip = _hip.stream_ptr(), graw.device, float(eng.clip)
    grads: Dict[int, torch.Tensor] = {}

    def dst(param: torch.Tensor) -> torch.Tensor:
        if grad_out is not None and id(param) in grad_out:
            return grad_out[id(param)]
        return torch.empty(param.shape, dtype=torch.float32, device=dev)

    keep: list = []   # tensors in use on the weight-gradient stream
    # the split-K reductions of the weight gradients: deferred to ONE launch behind the last layer -- unless a per-layer hook wants
    # each layer's gradients as soon as the layer is through
    wq = _wgrad_queue() if (_WGRAD_DEFER_REDUCE and (on_layer is None or flush_layers is not None)) else None
    flush_at = frozenset(flush_layers) if flush_layers is not None else frozenset()
    if wq is not None:
        _hip.call("yogo_wgrad_reduce_queue_reset", wq)   # (a pass that raised half way must not leave its reductions behind)
    if graw.dtype == torch.bfloat16 and graw.ndim == 5:   # already NCHW8c (yogo_decode_bwd_bf16)
        g = graw
        B = g.shape[0]
    else:
        graw = _f32(graw)
        B, P, Sy, Sx = graw.shape
        g = torch.empty(B, _blocks(P), Sy, Sx, 8, dtype=torch.bfloat16, device=dev)
        _hip.call("yogo_nchw_f32_to_bf16_8c", graw, g, B, P, Sy * Sx, st)
    n = len(eng.layers)
    head_g = None    # (head gradient, head weights) when the head's data gradient is left to the BatchNorm backward of the block under it
    fused01 = None   # (part, rows) of layer 0's backward sums when layer 1's data gradient produced them (g is then None at layer 0)
    for i in range(n - 1, -1, -1):
        L, S = eng.layers[i], saved[i]
        OH, OW = (int(g.shape[2]), int(g.shape[3])) if g is not None else fused01[2:]
        if i == 0:
            IH, IW = int(S.x_in.shape[2]), int(S.x_in.shape[3])
        else:
            IH, IW = int(S.x_in.shape[2]), int(S.x_in.shape[3])   # NCHW8c: [B, Cb, H, W, 8]
        # layer 0 with BatchNorm and no conv bias: BatchNorm backward, activation derivative and the weight gradient share ONE
        # sweep over (image, g, z) -- dz is never written (see conv_first_bn_wgrad_kernel)
        fuse0 = _FUSE_LAYER0_BWD and i == 0 and L.bn is not None and L.conv.bias is None and L.act in (ACT_NONE, ACT_LEAKY)
        # layer 1 above a matrix-core layer 0 that kept its sign map: ONE sweep over g does layer 1's weight gradient and folds its data
        # gradient straight into layer 0's backward sums (layer 0 has no data gradient of its own, so its dy need not exist; a trace wants
        # to see it)
        fuse01 = False
        if _L01_FUSE_BWD and i == 1 and trace is None and _FUSE_LAYER0_BWD and _WGRAD_BF16_MFMA:
            L0, S0 = eng.layers[0], saved[0]
            fuse01 = bool(S0.signs0 is not None and S0.x_in.dtype == torch.uint8 and L0.bn is not None and L0.conv.bias is None and L0.cin == 1
                          and L0.s == 2 and L0.act in (ACT_NONE, ACT_LEAKY) and L.k == 3 and L.s == 1 and S0.mask is None
                          and _hip.lib().yogo_conv2d_dgrad_first_bwd_supported(L.cin, L.cout, IH, IW, B, L0.act))
        if trace is not None and g is not None:
            trace[("g", i)] = g.clone()
        if L.bn is not None and not fuse0:
            bn = L.bn
            gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
            beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
            dgamma = dst(bn.weight) if bn.weight is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
            dbeta = dst(bn.bias) if bn.bias is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
            rows = _hip.query_ints("yogo_bn_bwd_bf16_rows", 1, B, OH * OW)[0]
            part = torch.empty(rows * L.cout * 2, dtype=torch.float32, device=dev)
            sums = torch.empty(2 * L.cout, dtype=torch.float32, device=dev)
            if head_g is not None:   # (g is still the head's output gradient: its data gradient is computed inside both sweeps)
                dz = torch.empty(B, _blocks(L.cout), OH, OW, 8, dtype=torch.bfloat16, device=dev)
                _hip.call("yogo_bn_bwd_bf16_head", head_g[0], head_g[1], head_g[2], S.z, dz, S.mean, S.invstd, gamma, beta, L.act, dgamma, dbeta,
                          part, sums, B, L.cout, OH * OW, 1 if S.bn_train else 0, clip, st)
                keep.extend(head_g[:2])
                g, head_g = dz, None
            else:
                _hip.call("yogo_bn_bwd_bf16", g, S.z, g, S.mean, S.invstd, gamma, beta, L.act, dgamma, dbeta, part, sums, B, L.cout, OH * OW,
                          1 if S.bn_train else 0, clip, st)
            if bn.weight is not None:
                grads[id(bn.weight)] = dgamma
                grads[id(bn.bias)] = dbeta
            if trace is not None:
                trace[("dz", i)] = g.clone()
        # ---- weight / bias gradient: independent of everything downstream -> second stream -------------------------------
        # Everything the side stream touches is allocated here, from the MAIN stream's pool, and kept alive until the main
        # stream has waited for the side stream (end of this function): no record_stream bookkeeping, no allocator stalls.
        main = torch.cuda.current_stream()
        wstream = _side_stream(dev) if _WGRAD_SIDE_STREAM else main
        dw = dst(L.conv.weight)
        has_bias = L.conv.bias is not None
        db = dst(L.conv.bias) if has_bias else None
        if fuse0:
            bn = L.bn
            gamma = _f32(bn.weight.detach()) if bn.weight is not None else torch.ones(L.cout, device=dev)
            beta = _f32(bn.bias.detach()) if bn.bias is not None else torch.zeros(L.cout, device=dev)
            dgamma = dst(bn.weight) if bn.weight is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
            dbeta = dst(bn.bias) if bn.bias is not None else torch.empty(L.cout, dtype=torch.float32, device=dev)
            cols = _hip.query_ints("yogo_conv_first_bn_wgrad_cols", 1, L.cin, L.cout)[0]
            if fused01 is not None:
                part, rows = fused01[:2]
            else:
                rows = _hip.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, L.s)[0]
                part = torch.empty(rows * cols, dtype=torch.float32, device=dev)
            sums = torch.empty(cols, dtype=torch.float32, device=dev)
            keep.extend((gamma, beta, dgamma, dbeta, part, sums))
        elif i == 0:
            rows = _hip.query_ints("yogo_conv_first_wgrad_rows", 1, B, IH, IW, L.s)[0]
            nj = L.cin * 9 + 1
            part = torch.empty(rows * L.cout * nj, dtype=torch.float32, device=dev)
            red = torch.empty(L.cout, nj, dtype=torch.float32, device=dev)
            keep.extend((part, red))
        elif fuse01:
            ws = torch.empty(_hip.query_size("yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes", B, IH, IW) // 4, dtype=torch.float32, device=dev)
            rows01 = _hip.query_ints("yogo_conv2d_dgrad_first_bwd_rows", 1, B, IH, IW, 1)[0]
            cols01 = _hip.query_ints("yogo_conv_first_bn_wgrad_cols", 1, L0.cin, L0.cout)[0]
            part01 = torch.empty(rows01 * cols01, dtype=torch.float32, device=dev)
            keep.extend((ws, part01))
        else:
            wname = "yogo_conv2d_wgrad_bf16_workspace_bytes" if _WGRAD_BF16_MFMA else "yogo_conv2d_wgrad_workspace_bytes"
            ws = torch.empty(_hip.query_size(wname, B, L.cin, L.cout, IH, IW, L.k, L.s) // 4, dtype=torch.float32, device=dev)
            keep.append(ws)
        keep.extend((g, S.x_in, dw))
        if wstream is not main:
            wstream.wait_stream(main)           # g (= dz of this layer) is complete
        with torch.cuda.stream(wstream):
            wst = _hip.stream_ptr()
            xdt = 0 if S.x_in.dtype == torch.uint8 else 1
            if fuse0:
                xg = "_xs" if S.signs0 is not None else "_xg" if S.gram is not None else ""
                zs = S.signs0 if S.signs0 is not None else S.z
                keep.append(zs)
                if fused01 is None:   # (else: layer 1's data gradient has filled `part`)
                    _hip.call("yogo_conv_first_bn_wgrad_bf16" + xg, S.x_in, xdt, g, zs, S.mean, S.invstd, gamma, beta, part, B, L.cin, L.cout,
                              IH, IW, L.s, L.act, wst)
                _hip.call("yogo_partials_reduce", part, rows, cols, 0.0, sums, wst)
                _hip.call("yogo_conv_first_bn_wgrad_finalize" + xg, sums, *((S.gram,) if S.gram is not None else ()), S.mean, S.invstd, gamma,
                          S.w_used if S.w_used is not None else _f32(L.conv.weight.detach()), dw, dgamma,
                          dbeta, B, L.cin, L.cout, IH, IW, L.s, 1 if S.bn_train else 0, clip, wst)
                if bn.weight is not None:
                    grads[id(bn.weight)] = dgamma
                    grads[id(bn.bias)] = dbeta
            elif fuse01:
                eng._tick("wgrad", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW * 2, mw=30,
                          nbytes=B * (16 * (_blocks(L.cout) + _blocks(L.cin)) * OH * OW + 2 * IH * IW + 4 * IH * IW))
                _hip.call("yogo_conv2d_dgrad_wgrad_bf16_first_bwd", g, _packed_bf16(eng, i, 1), S.x_in, S0.x_in, S0.signs0, part01, dw, db, ws,
                          B, L.cin, L.cout, IH, IW, L0.act, clip, wq, wst)
                eng._tock()
                fused01 = (part01, rows01, IH, IW)
            elif i == 0:
                _hip.call("yogo_conv_first_wgrad_bf16g", S.x_in, xdt, g, part, B, L.cin, L.cout, IH, IW, L.s, wst)
                _hip.call("yogo_partials_reduce", part, rows, L.cout * nj, clip, red, wst)
                dw.view(L.cout, nj - 1).copy_(red[:, : nj - 1])
                if has_bias:
                    db.copy_(red[:, nj - 1])
            else:
                eng._tick("wgrad", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW, mw=30,
                          nbytes=B * 2 * 8 * (_blocks(L.cout) * OH * OW + _blocks(L.cin) * IH * IW))
                if _WGRAD_BF16_MFMA and wq is not None:   # the split-K reduction waits for the flush behind the last layer
                    _hip.call("yogo_conv2d_wgrad_bf16_deferred", S.x_in, g, dw, db, ws, B, L.cin, L.cout, IH, IW, L.k, L.s, clip, wq, wst)
                elif _WGRAD_BF16_MFMA:
                    _hip.call("yogo_conv2d_wgrad_bf16", S.x_in, g, dw, db, ws, B, L.cin, L.cout, IH, IW, L.k, L.s, clip, wst)
                else:   # exact fp32 MFMA on the widened bf16 inputs
                    _hip.call("yogo_conv2d_wgrad_bf16in", S.x_in, g, dw, db, ws, B, L.cin, L.cout, IH, IW, L.k, L.s, clip, wst)
                eng._tock()
        if has_bias:
            grads[id(L.conv.bias)] = db
        grads[id(L.conv.weight)] = dw
        if on_layer is not None:
            if wq is not None and i in flush_at:   # the hook takes the gradients of layers >= i: their reductions run now, as one launch
                with torch.cuda.stream(wstream):
                    _hip.call("yogo_wgrad_reduce_flush", wq, _hip.stream_ptr())
            if wstream is not main:
                main.wait_stream(wstream)   # the weight gradient of this layer is part of what the hook hands over
            on_layer(i)
        if fuse01:   # (layer 1's data gradient went into layer 0's sums above)
            g = None
            continue
        if i > 0:
            Lp, Sp = eng.layers[i - 1], saved[i - 1]
            # the 1x1 head above a BatchNorm block: 12 multiply-adds per element inside that block's BatchNorm backward are cheaper than
            # writing and twice reading its 128-channel data gradient (a trace wants to see the tensor)
            if (_HEAD_BN_FUSE and i == n - 1 and trace is None and L.k == 1 and L.s == 1 and L.cout <= 16 and Lp.bn is not None and i - 1 > 0
                    and Lp.cout % 16 == 0 and g.shape[1] == 2):
                head_g = (g, _f32(L.conv.weight.detach()).reshape(L.cout, L.cin), L.cout)
                continue
            ref_act = Lp.act
            if Lp.bn is not None or Lp.act == ACT_NONE:
                act_ref, ref_act = None, ACT_NONE
            elif Lp.act == ACT_LEAKY:
                act_ref = Sp.y        # sign of the output = sign of the pre-activation
            else:
                act_ref = Sp.pre      # SiLU: the pre-activation saved by yogo_conv2d_fwd_bf16_pre
            pk = _packed_bf16(eng, i, 2 if (L.s == 2 and L.k == 3) else 1)
            dx = torch.empty(B, _blocks(L.cin), IH, IW, 8, dtype=torch.bfloat16, device=dev)
            nbytes = B * 2 * 8 * (_blocks(L.cout) * OH * OW + _blocks(L.cin) * IH * IW * (2 if act_ref is not None else 1))
            if ref_act == ACT_LEAKY and Sp.signs is not None:   # one byte per 16-byte unit in place of the reference
                nbytes = B * (16 * _blocks(L.cout) * OH * OW + 17 * _blocks(L.cin) * IH * IW)
            eng._tick("dgrad", i, 2.0 * B * L.cout * L.cin * L.k * L.k * OH * OW, mw=34 if (L.cin > 64 and L.s == 1 and act_ref is None) else 30, nbytes=nbytes)
            if ref_act == ACT_LEAKY and Sp.signs is not None:
                _hip.call("yogo_conv2d_dgrad_bf16_signs", g, pk, dx, Sp.signs, Sp.mask, B, L.cin, L.cout, IH, IW, L.k, L.s, st)
            else:
                _hip.call("yogo_conv2d_dgrad_bf16", g, pk, dx, act_ref, ref_act, Sp.mask, B, L.cin, L.cout, IH, IW, L.k, L.s, st)
            eng._tock()
            g = dx
    if wq is not None:   # every layer's split-K reduction in ONE launch (they are ~21 us each, mostly launch and tail latency)
        with torch.cuda.stream(_side_stream(dev) if _WGRAD_SIDE_STREAM else torch.cuda.current_stream()):
            _hip.call("yogo_wgrad_reduce_flush", wq, _hip.stream_ptr())
    if _WGRAD_SIDE_STREAM:
        torch.cuda.current_stream().wait_stream(_side_stream(dev))
    keep.clear()
    bb = eng.backbone_ref()
    return [grads.get(id(p)) for p in bb.parameters()]


_WGRAD_QUEUE = None


def _wgrad_queue() -> int:
    """the process's queue of deferred weight-gradient reductions (yogo_wgrad_reduce_queue_create; one host thread drives a GPU)"""
    global _WGRAD_QUEUE
    if _WGRAD_QUEUE is None:
        import ctypes
        h = ctypes.c_void_p(0)
        _hip.call("yogo_wgrad_reduce_queue_create", ctypes.addressof(h))
        _WGRAD_QUEUE = int(h.value)
    return _WGRAD_QUEUE


_ENGINES: "weakref.WeakKeyDictionary[nn.Module, Engine]" = weakref.WeakKeyDictionary()


def get_engine(backbone: nn.Sequential) -> Engine:
    eng = _ENGINES.get(backbone)
    if eng is None:
        eng = Engine(backbone)
        _ENGINES[backbone] = eng
    return eng


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, backbone, *params):  # type: ignore[override]
        eng = get_engine(backbone)
        raw, saved = eng.forward(x, need_grad=True)
        ctx.eng = eng
        ctx.saved = saved
        return raw

    @staticmethod
    def backward(ctx, graw):  # type: ignore[override]
        with torch.cuda.device(graw.device):
            grads = ctx.eng.backward(ctx.saved, graw)
        ctx.saved = None
        return (None, None, *grads)


def backbone_apply(backbone: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    _hip.require_cuda(x, "the input batch")
    params = list(backbone.parameters())
    if not backbone.training and not torch.is_grad_enabled() and (
            bf16_inference_requested() or getattr(backbone, "bf16_inference", False)):
        raw = backbone_infer_bf16(backbone, x)
        if raw is not None:
            return raw
    with torch.cuda.device(x.device):
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _BackboneFn.apply(x, backbone, *params)
        raw, _ = get_engine(backbone).forward(x, need_grad=False)
        return raw

class InferEngineBF16:
    """Eval-mode forward on the bf16 matrix cores (the reference runs `yogo infer` under bf16 autocast, yogo/infer.py:313-317).

    Activations live in NCHW8c bf16 ([B][C/8][H][W][8]); eval-mode BatchNorm is folded into the packed weights (per-output-
    channel scale) and the bias; bias + activation are fused into the conv epilogue; the last layer writes fp32 NCHW for the
    decode / NMS kernels.  Folded tensors and packed weights are cached and rebuilt when any parameter or buffer changes.
    """

    def __init__(self, engine: Engine):
        self.engine = engine
        self._key = None
        self._prep: List[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = []

    def _state_key(self):
        k = [self.engine.generation]
        for L in self.engine.layers:
            ts = [L.conv.weight, L.conv.bias]
            if L.bn is not None:
                ts += [L.bn.weight, L.bn.bias, L.bn.running_mean, L.bn.running_var]
            for t in ts:
                k.append(None if t is None else (t.data_ptr(), t._version))
        return tuple(k)

    def supported(self) -> bool:
        L0 = self.engine.layers[0]
        if not (L0.cin in (1, 3) and L0.k == 3):
            return False
        return all(L.bn is None or (not L.bn.training and L.bn.running_mean is not None) for L in self.engine.layers)

    @torch.no_grad()
    def _prepare(self) -> None:
        key = self._state_key()
        if key == self._key:
            return
        st = _hip.stream_ptr()
        prep = []
        for i, L in enumerate(self.engine.layers):
            w = _f32(L.conv.weight.detach())
            bias = L.conv.bias.detach().float() if L.conv.bias is not None else None
            scale = None
            if L.bn is not None:
                bn = L.bn
                gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(bn.running_var)
                beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(bn.running_var)
                scale = gamma / torch.sqrt(bn.running_var.float() + bn.eps)
                base = bias if bias is not None else torch.zeros_like(scale)
                bias = beta + (base - bn.running_mean.float()) * scale
            if i == 0:
                wf = w * scale[:, None, None, None] if scale is not None else w
                prep.append((wf.contiguous(), bias.contiguous() if bias is not None else None))
            else:
                nbytes = _hip.query_size("yogo_conv_bf16_packed_bytes", L.cin, L.cout, L.k, 0)
                packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
                _hip.call("yogo_conv_bf16_pack", w, scale.contiguous() if scale is not None else None, packed, L.cin, L.cout, L.k, 0, st)
                prep.append((packed, bias.contiguous() if bias is not None else None))
        self._prep = prep
        self._key = key

    @torch.no_grad()
    def forward(self, x: torch.Tensor, decode=None) -> torch.Tensor:
        """raw head output [B, 5 + C, Sy, Sx] -- or, given the decode's operands ``(cxs, cys, anchor_w, anchor_h, width_mult,
        height_mult, inference)`` and a 1x1 head the fused kernel takes, the DECODED tensor of yogo/model.py:277-313 (one launch for
        head + decode, bit-identical to the two)"""
        _hip.require_cuda(x, "the input batch")
        if x.ndim != 4:
            raise RuntimeError(f"yogo_amd: expected a [B,C,H,W] batch, got {tuple(x.shape)}")
        self._prepare()
        st = _hip.stream_ptr()
        dev = x.device
        B = x.shape[0]
        cur = x.contiguous() if x.dtype == torch.uint8 else _f32(x)
        H, W = int(cur.shape[2]), int(cur.shape[3])
        n = len(self.engine.layers)
        for i, L in enumerate(self.engine.layers):
            OH, OW = L.out_hw(H, W)
            if OH <= 0 or OW <= 0:
                raise RuntimeError(f"yogo_amd: image too small at layer {i}")
            wq, bias = self._prep[i]
            last = i == n - 1
            if i == 0:
                if cur.shape[1] != L.cin:
                    raise RuntimeError(f"yogo_amd: layer 0 expects {L.cin} channels, got {cur.shape[1]}")
                mb = _hip.lib().yogo_bf16_channel_blocks(L.cout)
                out = torch.empty(B, mb, OH, OW, 8, dtype=torch.bfloat16, device=dev)
                if _L0_MFMA and cur.dtype == torch.uint8 and _hip.lib().yogo_conv_first_mfma_supported(0, L.cin, L.cout, H, W, L.s):
                    _hip.call("yogo_conv_first_mfma", cur, wq, bias, out, None, None, None, None, None, None, B, L.cout, H, W, L.act, st)
                else:
                    _hip.call("yogo_conv_first_fwd_bf16", cur, 0 if cur.dtype == torch.uint8 else 1, wq, bias, out, B, L.cin, L.cout, H, W,
                              L.s, L.act, st)
                if last:
                    raise RuntimeError("yogo_amd: a one-layer network is not supported by the bf16 inference path")
            elif last:
                out = torch.empty(B, L.cout, OH, OW, dtype=torch.float32, device=dev)
                if decode is not None and head_decode_fusable(L):
                    cxs, cys, aw, ah, wm, hm, inference = decode
                    _hip.call("yogo_head1x1_decode_fwd_bf16", cur, wq, bias, out, cxs, cys, B, L.cin, L.cout, OH, OW, aw, ah, wm, hm, 1 if inference else 0, st)
                    self.decoded = True
                else:
                    _hip.call("yogo_conv2d_fwd_bf16", cur, wq, bias, None, out, None, None, B, L.cin, L.cout, H, W, L.k, L.s, L.act, st)
                    self.decoded = False
            else:
                mb = _hip.lib().yogo_bf16_channel_blocks(L.cout)
                out = torch.empty(B, mb, OH, OW, 8, dtype=torch.bfloat16, device=dev)
                _hip.call("yogo_conv2d_fwd_bf16", cur, wq, bias, out, None, None, None, B, L.cin, L.cout, H, W, L.k, L.s, L.act, st)
            cur, H, W = out, OH, OW
        return cur


def head_decode_fusable(L) -> bool:
    """the last layer is a plain 1x1 head the fused head + decode kernel takes (yogo_head1x1_decode_fwd_bf16)"""
    return L.k == 1 and L.s == 1 and L.bn is None and L.act == 0 and 6 <= L.cout <= 16 and L.cin % 16 == 0 and L.cin <= 128


def bf16_inference_requested() -> bool:
    """True inside `torch.autocast("cuda", dtype=torch.bfloat16)` -- how the reference's predict() asks for half inference"""
    try:
        return bool(torch.is_autocast_enabled()) and torch.get_autocast_gpu_dtype() == torch.bfloat16
    except Exception:
        return False


def backbone_infer_bf16(backbone: nn.Sequential, x: torch.Tensor, decode=None):
    """raw head output via the bf16 path, or None when the model state does not allow it (train-mode BatchNorm, ...).  With ``decode``
    (the decode's operands, see InferEngineBF16.forward) the result is ``(tensor, decoded)``: decoded = True when head + decode ran as one
    launch and ``tensor`` already is the decoded prediction"""
    eng = get_engine(backbone)
    inf = getattr(eng, "_infer_bf16", None)
    if inf is None:
        inf = InferEngineBF16(eng)
        eng._infer_bf16 = inf
    if not inf.supported():
        return None
    with torch.cuda.device(x.device):
        if decode is None:
            return inf.forward(x)
        out = inf.forward(x, decode=decode)
        return out, bool(inf.decoded)
