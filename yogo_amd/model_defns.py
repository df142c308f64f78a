"""Model registry -- the reference's plugin API for architectures (yogo/model_defns.py:6-27).

``ModelDefn = Callable[[int, bool], nn.Module]``, ``MODELS``, ``register_model`` and ``get_model_func`` keep the
reference's names, semantics and fallback behaviour.  Architectures are generated from a width/stride table (data
read off yogo/model_defns.py:30-529); every definition returns a :class:`HipBackbone`, an ``nn.Sequential`` of
ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` / activation / ``nn.Dropout2d`` modules -- so ``state_dict`` keys
(``model.{i}.{j}.*``), the ``isinstance`` scans of ``YOGO.get_grid_size`` / ``init_network_weights`` /
``set_bn_eval`` (yogo/model.py:81,91,213-230) and checkpoints stay interchangeable with the reference -- whose
``forward`` runs the hand-written HIP kernels of ``libyogo_hip.so`` instead of the modules' own ATen ops.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

ModelDefn = Callable[[int, bool], nn.Module]

MODELS: Dict[str, ModelDefn] = {}

# (cout, ksize, stride, bias, batchnorm, activation, dropout_p); cout None = 5 + num_classes (bare head conv)
LayerSpec = Tuple[Optional[int], int, int, bool, bool, Optional[str], float]


def get_model_func(model_name: Optional[str]) -> ModelDefn:
    if model_name is None:
        return base_model
    try:
        return MODELS[model_name]
    except KeyError:
        return base_model


def register_model(model_defn: ModelDefn) -> ModelDefn:
    """put model in MODELS. When adding a new model, make sure to `@register_model`!"""
    MODELS[model_defn.__name__] = model_defn
    return model_defn


class HipBackbone(nn.Sequential):
    """nn.Sequential whose forward is executed by the HIP engine (yogo_amd/engine.py)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        from yogo_amd.engine import backbone_apply

        return backbone_apply(self, x)


def _activation(act: Optional[str]) -> Optional[nn.Module]:
    if act == "leaky":
        return nn.LeakyReLU()
    if act == "silu":
        return nn.SiLU(inplace=True)
    return None


def build_backbone(spec: Sequence[LayerSpec], num_classes: int, rgb_input: bool) -> HipBackbone:
    cin = 3 if rgb_input else 1
    blocks: List[nn.Module] = []
    for cout, k, s, bias, bn, act, p in spec:
        co = 5 + num_classes if cout is None else cout
        conv = nn.Conv2d(cin, co, k, stride=s, padding=1 if k == 3 else 0, bias=bias)
        if cout is None and not bn and act is None:
            blocks.append(conv)  # bare head: keys model.{i}.weight / model.{i}.bias
        else:
            mods: List[nn.Module] = [conv]
            if bn:
                mods.append(nn.BatchNorm2d(co))
            a = _activation(act)
            if a is not None:
                mods.append(a)
            if p > 0:
                mods.append(nn.Dropout2d(p=p))
            blocks.append(nn.Sequential(*mods))
        cin = co
    return HipBackbone(*blocks)


def _std(widths: Sequence[int], act: str) -> List[LayerSpec]:
    a, b, c, d = widths
    return [
        (a, 3, 2, False, True, act, 0.0),
        (b, 3, 1, True, False, act, 0.05),
        (c, 3, 2, True, False, act, 0.1),
        (d, 3, 1, True, False, act, 0.15),
        (d, 3, 2, False, True, act, 0.0),
        (d, 3, 1, True, True, act, 0.0),
        (d, 3, 1, True, False, act, 0.0),
        (None, 1, 1, True, False, None, 0.0),
    ]


L = "leaky"
SPECS: Dict[str, List[LayerSpec]] = {
    "base_model": _std((16, 32, 64, 128), L),
    "silu_model": _std((16, 32, 64, 128), "silu"),
    "double_filters": _std((32, 64, 128, 256), L),
    "triple_filters": _std((48, 96, 192, 384), L),
    "half_filters": _std((8, 16, 32, 64), L),
    "quarter_filters": _std((4, 8, 16, 32), L),
    "depth_ver_0": [
        (32, 3, 2, False, True, L, 0.0), (128, 3, 2, True, False, L, 0.1),
        (128, 3, 2, False, True, L, 0.0), (None, 1, 1, True, False, None, 0.0)],
    "depth_ver_1": [
        (16, 3, 2, False, True, L, 0.0), (64, 3, 2, True, False, L, 0.1), (128, 3, 1, True, False, L, 0.15),
        (128, 3, 2, False, True, L, 0.0), (128, 3, 1, True, False, L, 0.0), (None, 1, 1, True, False, None, 0.0)],
    "depth_ver_2": _std((16, 32, 64, 128), L),
    "depth_ver_3": [
        (16, 3, 2, False, True, L, 0.0), (32, 3, 1, True, False, L, 0.05), (32, 3, 1, True, False, L, 0.05),
        (64, 3, 2, True, False, L, 0.1), (128, 3, 1, True, False, L, 0.15), (128, 3, 1, True, True, L, 0.0),
        (128, 3, 2, False, False, L, 0.0), (128, 3, 1, True, True, L, 0.0), (128, 3, 1, True, False, L, 0.0),
        (None, 1, 1, True, False, None, 0.0)],
    "depth_ver_4": [
        (16, 3, 2, False, True, L, 0.0), (16, 3, 1, True, False, L, 0.0), (32, 3, 1, True, False, L, 0.05),
        (32, 3, 1, True, False, L, 0.05), (64, 3, 2, True, False, L, 0.1), (64, 3, 1, True, False, L, 0.0),
        (128, 3, 1, True, False, L, 0.15), (128, 3, 1, True, True, L, 0.0), (128, 3, 2, True, False, L, 0.0),
        (128, 3, 1, True, True, L, 0.0), (128, 3, 1, True, False, L, 0.0), (None, 1, 1, True, False, None, 0.0)],
}


def _make(name: str) -> ModelDefn:
    def defn(num_classes: int, rgb_input: bool = False) -> nn.Module:
        return build_backbone(SPECS[name], num_classes, rgb_input)

    defn.__name__ = name
    defn.__qualname__ = name
    defn.__doc__ = f"{name}: see SPECS['{name}'] (reference: yogo/model_defns.py)"
    return register_model(defn)


base_model = _make("base_model")
silu_model = _make("silu_model")
double_filters = _make("double_filters")
triple_filters = _make("triple_filters")
half_filters = _make("half_filters")
quarter_filters = _make("quarter_filters")
depth_ver_0 = _make("depth_ver_0")
depth_ver_1 = _make("depth_ver_1")
depth_ver_2 = _make("depth_ver_2")
depth_ver_3 = _make("depth_ver_3")
depth_ver_4 = _make("depth_ver_4")


@register_model
def convnext_small(num_classes: int, rgb_input: bool = False) -> nn.Module:
    """The reference builds this one from ``timm`` (yogo/model_defns.py:532-558); not part of the HIP hot path."""
    raise NotImplementedError("convnext_small needs timm and is not implemented by the MI355X hot path")
