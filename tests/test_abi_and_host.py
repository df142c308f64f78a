"""CPU-only checks: the C-ABI library loads and exports every symbol include/yogo_hip.h declares, the drop-in surface
keeps the reference's names / state_dict keys / error behaviour, and the product refuses to compute without a GPU."""
import json
import os

import pytest
import torch

from _util import GOLDEN


def test_library_exports_every_declared_symbol():
    from yogo_amd import _hip

    protos = _hip.prototypes()   # raises AttributeError if a declared symbol is missing from the .so
    assert len(protos) >= 26
    assert _hip.lib().yogo_hip_abi_version() == 7
    for name in ("yogo_conv2d_fwd_f32", "yogo_conv2d_dgrad_f32", "yogo_conv2d_wgrad_f32", "yogo_conv_first_fwd", "yogo_bn_finalize",
                 "yogo_decode_fwd", "yogo_loss_fwd_bwd", "yogo_format_preds_batched", "yogo_decode_format_preds_batched", "yogo_adamw_step"):
        assert name in protos


def test_argument_validation_needs_no_gpu():
    from yogo_amd import _hip

    with pytest.raises(RuntimeError, match="only 3x3"):
        _hip.query_size("yogo_conv_packed_bytes", 16, 32, 5, 1, 0)
    assert _hip.query_size("yogo_conv_packed_bytes", 128, 128, 3, 1, 0) == 9 * 128 * 128 * 4
    assert _hip.query_size("yogo_conv_packed_bytes", 128, 128, 3, 2, 1) == 9 * 128 * 128 * 4
    assert _hip.query_size("yogo_loss_workspace_bytes", 64, 97, 129) > 0
    assert _hip.query_size("yogo_format_preds_workspace_bytes", 4, 97, 129) >= 4 * 97 * 129 * 24


def test_registry_surface():
    from yogo_amd.model_defns import MODELS, base_model, get_model_func, register_model

    keys = json.load(open(os.path.join(GOLDEN, "ckpt_keys.json")))
    assert set(keys) | {"convnext_small"} == set(MODELS)
    assert get_model_func(None) is base_model and get_model_func("nope") is base_model
    assert get_model_func("silu_model").__name__ == "silu_model"

    @register_model
    def my_tiny(num_classes, rgb_input=False):
        return base_model(num_classes, rgb_input)

    assert MODELS["my_tiny"] is my_tiny
    del MODELS["my_tiny"]


def test_state_dict_keys_shapes_dtypes_match_reference():
    from yogo_amd.model import YOGO
    from yogo_amd.model_defns import get_model_func

    keys = json.load(open(os.path.join(GOLDEN, "ckpt_keys.json")))
    for name, info in keys.items():
        m = YOGO((772, 1032), 0.0425, 0.0555, 7, model_func=get_model_func(name))
        got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert got == info["keys"], name
        assert m.num_params() == info["num_params"]
        assert (m.Sx, m.Sy) == (info["Sx"], info["Sy"])
        assert m.model_version == name


def test_checkpoint_round_trip(tmp_path):
    # tests/test_model_io.py of the reference
    from copy import deepcopy

    from yogo_amd.model import YOGO
    from yogo_amd.model_defns import get_model_func

    for mf in ("base_model", "silu_model"):
        y = YOGO(img_size=(772, 1032), anchor_w=0.05, anchor_h=0.05, num_classes=7, model_func=get_model_func(mf))
        torch.save({"epoch": 0, "step": 0, "model_state_dict": deepcopy(y.state_dict()), "model_version": y.model_version},
                   str(tmp_path / "test.pth"))
        z, meta = YOGO.from_pth(tmp_path / "test.pth")
        assert all(a == b for a, b in zip(y.img_size, z.img_size))
        for attr in ("anchor_w", "anchor_h", "num_classes", "is_rgb", "normalize_images", "clip_value", "height_multiplier", "width_multiplier"):
            assert getattr(y, attr) == getattr(z, attr)
        assert y.model_version == z.model_version == mf
        for p1, p2 in zip(y.parameters(), z.parameters()):
            assert p1.data.ne(p2.data).sum() == 0
        assert meta["step"] == 0


def test_reference_checkpoint_loads(tmp_path):
    # a checkpoint written from the REFERENCE's state dict (golden fixture) loads into the drop-in module
    from _util import load_net_fixture
    from yogo_amd.model import YOGO

    meta, x, sd, *_ = load_net_fixture("net_base_64x96.npz")
    torch.save({"epoch": 3, "step": 77, "model_state_dict": sd, "model_version": "base_model", "normalize_images": False},
               str(tmp_path / "ref.pth"))
    m, info = YOGO.from_pth(tmp_path / "ref.pth", inference=True)
    assert info["step"] == 77 and m.inference and not m.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_grid_and_resize():
    from _util import as_t, load_npz
    from yogo_amd.model import YOGO

    z = load_npz("grid.npz")
    m = YOGO((772, 1032), 0.0425, 0.0555, 7)
    assert torch.equal(m._Cxs, as_t(z["Cxs"])) and torch.equal(m._Cys, as_t(z["Cys"]))
    for k, (sx, sy) in json.loads(str(z["sizes"])).items():
        h, w = (int(v) for v in k.split("x"))
        assert m.get_grid_size((h, w)) == (sx, sy)
    m.resize_model(193)
    assert (m.Sx, m.Sy) == (129, 25) and m._Cys.shape == (25, 129)
    assert abs(float(m.height_multiplier) - 772 / 193) < 1e-6 and float(m.width_multiplier) == 1.0


def test_no_cpu_fallback():
    from yogo_amd.model import YOGO
    from yogo_amd.utils import format_preds
    from yogo_amd.yogo_loss import YOGOLoss

    m = YOGO((64, 96), 0.0425, 0.0555, 7)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 64, 96))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        YOGOLoss()(torch.zeros(1, 12, 8, 12), torch.zeros(1, 6, 8, 12))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        format_preds(torch.zeros(12, 4, 4))
    with pytest.raises(ValueError):
        format_preds(torch.zeros(1, 12, 4, 4))
    with pytest.raises(ValueError):
        format_preds(torch.zeros(12, 4, 4), box_format="xywh")


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys

    code = "import sys; import yogo_amd, yogo_amd.engine; assert 'yogo_oracle' not in sys.modules; print('ok')"
    root = os.path.dirname(os.path.dirname(GOLDEN))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    for dirpath, _, files in os.walk(os.path.join(root, "yogo_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "yogo_oracle" not in open(os.path.join(dirpath, f)).read(), f
