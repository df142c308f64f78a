"""``YOGO`` module -- same constructor, buffers, state_dict keys, checkpoint loader and forward contract as the
reference (yogo/model.py:13-313), with the backbone, the box decode and their backward running as hand-written
HIP kernels (libyogo_hip.so) on an MI355X.  There is no CPU compute path: forward on a CPU tensor raises.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn

from yogo_amd import _hip
from yogo_amd.engine import get_engine
from yogo_amd.model_defns import ModelDefn, base_model, get_model_func

PathLike = Union[Path, str]


class _DecodeFn(torch.autograd.Function):
    """box decode of yogo/model.py:277-313 (forward) and its derivative."""

    @staticmethod
    def forward(ctx, raw, cxs, cys, anchor_w, anchor_h, wmul, hmul, inference):  # type: ignore[override]
        out = _decode(raw, cxs, cys, anchor_w, anchor_h, wmul, hmul, inference)
        ctx.save_for_backward(raw, out)
        ctx.inference = inference
        return out

    @staticmethod
    def backward(ctx, gout):  # type: ignore[override]
        raw, out = ctx.saved_tensors
        B, P, Sy, Sx = raw.shape
        gout = gout.contiguous().float()
        graw = torch.empty_like(raw)
        with torch.cuda.device(raw.device):
            _hip.call("yogo_decode_bwd", raw, out, gout, graw, B, P, Sy, Sx, int(ctx.inference), _hip.stream_ptr())
        return graw, None, None, None, None, None, None, None


def _decode(raw, cxs, cys, anchor_w, anchor_h, wmul, hmul, inference) -> torch.Tensor:
    B, P, Sy, Sx = raw.shape
    if tuple(cxs.shape) != (Sy, Sx) or tuple(cys.shape) != (Sy, Sx):
        raise RuntimeError(f"yogo_amd: grid buffers {tuple(cxs.shape)} do not match the network output grid ({Sy}, {Sx})")
    out = torch.empty_like(raw)
    with torch.cuda.device(raw.device):
        _hip.call("yogo_decode_fwd", raw, out, cxs.contiguous(), cys.contiguous(), B, P, Sy, Sx, float(anchor_w), float(anchor_h),
                  float(wmul), float(hmul), int(inference), _hip.stream_ptr())
    return out


class YOGO(nn.Module):
    def __init__(
        self,
        img_size: Tuple[int, int],
        anchor_w: float,
        anchor_h: float,
        num_classes: int,
        is_rgb: bool = False,
        normalize_images: bool = False,
        inference: bool = False,
        tuning: bool = False,
        model_func: ModelDefn = base_model,
        clip_value: float = 1.0,
        device: Union[torch.device, str] = "cpu",
    ):
        super().__init__()
        self.device = device
        self.inference = inference
        self.model = model_func(num_classes, is_rgb).to(device)
        self.model_version = model_func.__name__

        given = {"img_size": img_size, "anchor_w": anchor_w, "anchor_h": anchor_h, "num_classes": num_classes,
                 "clip_value": clip_value, "is_rgb": is_rgb, "normalize_images": normalize_images}
        for name in self._BUFFERS_BEFORE_GRID:
            self.register_buffer(name, torch.tensor(given[name]))
        self.Sx, self.Sy = self.get_grid_size()
        for name, grid in zip(("_Cxs", "_Cys"), self._cell_grids(self.Sx, self.Sy, self.device)):
            self.register_buffer(name, grid)
        for name, value in self._BUFFERS_AFTER_GRID.items():
            self.register_buffer(name, torch.tensor(value))

        self.model.apply(self.set_bn_eval if tuning else self.init_network_weights)

        # gradient clipping: the reference registers one clamp hook per parameter (yogo/model.py:76-77); here the
        # clamp is fused into the kernels that finish each parameter gradient.
        self._clip = float(clip_value)
        self._scalars: Optional[Tuple[float, float, float, float]] = None

    # The eleven buffers of a YOGO checkpoint, in state_dict order (= the order the reference registers them in,
    # yogo/model.py:24-65; tests/golden/ckpt_keys.json): seven constructor scalars, the two cell-origin grids, then the two
    # crop multipliers that only resize_model changes.
    _BUFFERS_BEFORE_GRID = ("img_size", "anchor_w", "anchor_h", "num_classes", "clip_value", "is_rgb", "normalize_images")
    _BUFFERS_AFTER_GRID = {"height_multiplier": 1.0, "width_multiplier": 1.0}
    # buffers that checkpoints written by older versions lack, with the values they implied (yogo/model.py:100-109)
    _LEGACY_BUFFER_DEFAULTS = {"is_rgb": False, "clip_value": 1.0, "height_multiplier": 1.0, "width_multiplier": 1.0}

    @staticmethod
    def _cell_grids(Sx: int, Sy: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """x / y origin of every grid cell as [Sy, Sx] tensors, from linspace as the reference builds them (yogo/model.py:48-61) --
        not k / S, which differs in the last bit for some k"""
        xs = torch.linspace(0, 1 - 1 / Sx, Sx)    # on the host, like the reference: a device linspace may round differently
        ys = torch.linspace(0, 1 - 1 / Sy, Sy)
        return xs[None, :].expand(Sy, Sx).clone().to(device), ys[:, None].expand(Sy, Sx).clone().to(device)

    @staticmethod
    def init_network_weights(module: nn.Module):
        if isinstance(module, nn.Conv2d):
            torch.nn.init.kaiming_normal_(module.weight, a=0.01, mode="fan_out", nonlinearity="leaky_relu")
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)

    @staticmethod
    def set_bn_eval(module: nn.Module):
        if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
            module.eval()

    @classmethod
    def from_pth(cls, pth_path: PathLike, inference: bool = False) -> Tuple["YOGO", Dict[str, Any]]:
        ckpt = torch.load(Path(pth_path), map_location="cpu", weights_only=False)
        state = ckpt["model_state_dict"]
        for name, value in cls._LEGACY_BUFFER_DEFAULTS.items():
            state.setdefault(name, torch.tensor(value))
        # (normalize_images used to live beside the state dict)
        state.setdefault("normalize_images", torch.tensor(ckpt.get("normalize_images", False)))

        h, w = (int(v) for v in state["img_size"])
        model = cls(
            (h, w), state["anchor_w"].item(), state["anchor_h"].item(), num_classes=int(state["num_classes"].item()),
            is_rgb=bool(state["is_rgb"].item()), inference=inference, tuning=True,
            model_func=get_model_func(ckpt.get("model_version", None)), clip_value=float(state["clip_value"].item()),
        )
        model.load_state_dict(state)
        if inference:
            model.eval()
        return model, {"step": ckpt.get("step", 0), "class_names": ckpt.get("class_names", None),
                       "normalize_images": state["normalize_images"]}

    def to(self, device, *args, **kwargs):
        self.device = device
        super().to(device, *args, **kwargs)
        self._scalars = None
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._scalars = None
        self._clip = float(self.clip_value)
        get_engine(self.model).invalidate_packed()
        return out

    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def grad_norm(self) -> float:
        total = 0.0
        for p in self.parameters():
            if p.grad is not None and p.requires_grad:
                total += p.grad.detach().data.norm(2).item() ** 2
        return total ** 0.5

    def param_norm(self) -> float:
        total = 0.0
        for p in self.parameters():
            if p.grad is not None and p.requires_grad:
                total += p.detach().data.norm(2).item() ** 2
        return total ** 0.5

    def get_img_size(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if isinstance(self.img_size, torch.Tensor):
            h, w = self.img_size
            return h, w
        raise ValueError(f"self.img_size is not a tensor: {type(self.img_size)}")

    def get_grid_size(self, img_size: Optional[Tuple[int, int]] = None) -> Tuple[int, int]:
        """return Sx, Sy -- floor formula per convolution, walking the modules as yogo/model.py:189-234 does"""
        if img_size is not None:
            h, w = int(img_size[0]), int(img_size[1])
        else:
            hh, ww = self.get_img_size()
            h, w = int(hh), int(ww)

        def pair(v):
            return v if isinstance(v, tuple) else (v, v)

        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                p0, p1 = pair(mod.padding)
                d0, d1 = pair(mod.dilation)
                k0, k1 = pair(mod.kernel_size)
                s0, s1 = pair(mod.stride)
                h = (h + 2 * p0 - d0 * (k0 - 1) - 1) // s0 + 1
                w = (w + 2 * p1 - d1 * (k1 - 1) - 1) // s1 + 1
            elif isinstance(mod, nn.ConvTranspose2d):
                p0, p1 = pair(mod.padding)
                d0, d1 = pair(mod.dilation)
                k0, k1 = pair(mod.kernel_size)
                s0, s1 = pair(mod.stride)
                o0, o1 = pair(mod.output_padding)
                h = (h - 1) * s0 - 2 * p0 + d0 * (k0 - 1) + o0 + 1
                w = (w - 1) * s1 - 2 * p1 + d1 * (k1 - 1) + o1 + 1
        return int(w), int(h)

    def resize_model(self, img_height: Optional[int] = None, img_width: Optional[int] = None) -> None:
        """crop-resize: new grid buffers + size multipliers (yogo/model.py:236-265)"""
        org_img_height, org_img_width = (int(d) for d in self.get_img_size())
        crop_size = (img_height or org_img_height, img_width or org_img_width)
        Sx, Sy = self.get_grid_size(crop_size)
        self.Sx, self.Sy = Sx, Sy
        # the new grids live where the old ones do (``self.device`` goes stale under ``.cuda()`` -- only ``.to()`` updates it,
        # in the reference too; a host grid handed to the decode kernel would be a GPU memory fault, not an error)
        dev = self._Cxs.device
        grids = self._cell_grids(Sx, Sy, dev)
        self.register_buffer("height_multiplier", torch.tensor(org_img_height / crop_size[0], device=dev))
        self.register_buffer("width_multiplier", torch.tensor(org_img_width / crop_size[1], device=dev))
        self.register_buffer("img_size", torch.tensor(crop_size, device=dev))
        self.register_buffer("_Cxs", grids[0])
        self.register_buffer("_Cys", grids[1])
        self._scalars = None

    def _decode_scalars(self) -> Tuple[float, float, float, float]:
        # the four scalar buffers are read back once (host copy) and cached; they only change through
        # load_state_dict / resize_model / to, which reset the cache
        if self._scalars is None:
            self._scalars = (float(self.anchor_w), float(self.anchor_h), float(self.width_multiplier), float(self.height_multiplier))
        return self._scalars

    def _backbone(self, x: torch.Tensor) -> torch.Tensor:
        # we get either raw uint8 tensors or float tensors
        if x.ndim == 3:
            x.unsqueeze_(0)
        _hip.require_cuda(x, "the input batch")
        if not x.is_floating_point() and x.dtype != torch.uint8:
            x = x.float()
        eng = get_engine(self.model)
        eng.clip = self._clip
        return self.model(x)

    @torch.no_grad()
    def forward_raw(self, x: torch.Tensor):
        """the backbone + head WITHOUT the box decode, wrapped with the decode's operands: what ``format_preds_batched`` /
        ``save_predictions`` / ``get_prediction_class_counts`` / ``format_to_numpy_batched`` take in place of ``model(x)`` to run the
        decode inside the threshold + NMS kernel (the `yogo infer` path, yogo/infer.py:311-380, with no decoded tensor in memory)"""
        from yogo_amd.utils.prediction_formatting import RawPredictions

        raw = self._backbone(x)
        return RawPredictions(raw, self._Cxs, self._Cys, *self._decode_scalars(), bool(self.inference))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        aw, ah, wm, hm = self._decode_scalars()
        if not self.model.training and not torch.is_grad_enabled():
            # eval-mode `model(x)` on the bf16 path (yogo/infer.py:313-317's autocast): the 1x1 head and the decode below in ONE launch
            # (yogo_head1x1_decode_fwd_bf16, bit-identical to the two) where the head is one the fused kernel takes
            from yogo_amd.engine import backbone_infer_bf16, bf16_inference_requested

            if bf16_inference_requested() or getattr(self.model, "bf16_inference", False):
                if x.ndim == 3:
                    x.unsqueeze_(0)
                _hip.require_cuda(x, "the input batch")
                xin = x if (x.is_floating_point() or x.dtype == torch.uint8) else x.float()
                res = backbone_infer_bf16(self.model, xin, decode=(self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference)))
                if res is not None:
                    out, decoded = res
                    return out if decoded else _decode(out, self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference))
        raw = self._backbone(x)
        if torch.is_grad_enabled() and raw.requires_grad:
            return _DecodeFn.apply(raw, self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference))
        return _decode(raw, self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference))
