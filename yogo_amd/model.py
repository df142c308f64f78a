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

        self.model = model_func(num_classes, is_rgb).to(device)
        self.model_version = model_func.__name__

        self.register_buffer("img_size", torch.tensor(img_size))
        self.register_buffer("anchor_w", torch.tensor(anchor_w))
        self.register_buffer("anchor_h", torch.tensor(anchor_h))
        self.register_buffer("num_classes", torch.tensor(num_classes))
        self.register_buffer("clip_value", torch.tensor(clip_value))
        self.register_buffer("is_rgb", torch.tensor(is_rgb))
        self.register_buffer("normalize_images", torch.tensor(normalize_images))

        self.inference = inference

        Sx, Sy = self.get_grid_size()
        self.Sx, self.Sy = Sx, Sy

        # grids as the reference builds them: linspace, not k/S (yogo/model.py:48-61)
        _Cxs = torch.linspace(0, 1 - 1 / Sx, Sx).expand(Sy, -1).to(self.device)
        _Cys = torch.linspace(0, 1 - 1 / Sy, Sy).expand(1, -1).transpose(0, 1).expand(Sy, Sx).to(self.device)
        self.register_buffer("_Cxs", _Cxs.clone())
        self.register_buffer("_Cys", _Cys.clone())

        self.register_buffer("height_multiplier", torch.tensor(1.0))
        self.register_buffer("width_multiplier", torch.tensor(1.0))

        if tuning:
            self.model.apply(self.set_bn_eval)
        else:
            self.model.apply(self.init_network_weights)

        # gradient clipping: the reference registers one clamp hook per parameter (yogo/model.py:76-77); here the
        # clamp is fused into the kernels that finish each parameter gradient.
        self._clip = float(clip_value)
        self._scalars: Optional[Tuple[float, float, float, float]] = None

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
        pth_path = Path(pth_path)
        loaded_pth = torch.load(pth_path, map_location="cpu", weights_only=False)

        global_step = loaded_pth.get("step", 0)
        model_version = loaded_pth.get("model_version", None)
        class_names = loaded_pth.get("class_names", None)

        params = loaded_pth["model_state_dict"]
        img_size = params["img_size"]
        anchor_w = params["anchor_w"]
        anchor_h = params["anchor_h"]
        num_classes = params["num_classes"]

        # be permissive of older pth files
        params.setdefault("is_rgb", torch.tensor(False))
        params.setdefault("clip_value", torch.tensor(1.0))
        params.setdefault("height_multiplier", torch.tensor(1.0))
        params.setdefault("width_multiplier", torch.tensor(1.0))
        if "normalize_images" not in params:
            params["normalize_images"] = torch.tensor(loaded_pth.get("normalize_images", False))

        model = cls(
            (int(img_size[0]), int(img_size[1])),
            anchor_w.item(),
            anchor_h.item(),
            num_classes=int(num_classes.item()),
            is_rgb=bool(params["is_rgb"].item()),
            inference=inference,
            tuning=True,
            model_func=get_model_func(model_version),
            clip_value=float(params["clip_value"].item()),
        )
        model.load_state_dict(params)
        if inference:
            model.eval()
        return model, {
            "step": global_step,
            "class_names": class_names,
            "normalize_images": params["normalize_images"],
        }

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
        _Cxs = torch.linspace(0, 1 - 1 / Sx, Sx, device=dev).expand(Sy, -1)
        _Cys = torch.linspace(0, 1 - 1 / Sy, Sy, device=dev).expand(1, -1).transpose(0, 1).expand(Sy, Sx)
        self.register_buffer("height_multiplier", torch.tensor(org_img_height / crop_size[0], device=dev))
        self.register_buffer("width_multiplier", torch.tensor(org_img_width / crop_size[1], device=dev))
        self.register_buffer("img_size", torch.tensor(crop_size, device=dev))
        self.register_buffer("_Cxs", _Cxs.clone())
        self.register_buffer("_Cys", _Cys.clone())
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
        raw = self._backbone(x)
        aw, ah, wm, hm = self._decode_scalars()
        if torch.is_grad_enabled() and raw.requires_grad:
            return _DecodeFn.apply(raw, self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference))
        return _decode(raw, self._Cxs, self._Cys, aw, ah, wm, hm, bool(self.inference))
