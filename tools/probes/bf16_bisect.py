"""layer-by-layer comparison of one bf16 HipTrainer step with the oracle's bf16-storage emulation (O.bf16_train_step taps):
forward tensors (z, y of every block), the head gradient, every data gradient / BatchNorm-backward output.  Intercepts the
C-ABI calls of the step (yogo_amd._hip.call) and clones their outputs on the stream.
usage: bf16_bisect.py [model] [H] [W] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import yogo_oracle as O
from yogo_amd import _hip
from yogo_amd.model import YOGO
from yogo_amd.model_defns import MODELS
from yogo_amd.train import HipTrainer
from yogo_amd.yogo_loss import YOGOLoss

name = sys.argv[1] if len(sys.argv) > 1 else "base_model"
H, W, B = (int(a) for a in (sys.argv[2:5] + ["193", "258", "3"][len(sys.argv[2:5]):]))
C = 5
x = torch.randint(0, 256, (B, 1, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
torch.manual_seed(1)
m = YOGO((H, W), 0.0425, 0.0555, C, model_func=MODELS[name], clip_value=1e9).cuda()
m.train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout2d):
        mod.p = 0.0
sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
lab = O.synthetic_labels(B, m.Sx, m.Sy, K=4, num_classes=C, seed=3)
tr = HipTrainer(m, YOGOLoss().cuda(), total_steps=4, half=True)

rec = []
orig = _hip.call
OUT = {"yogo_conv2d_dgrad_bf16_signs": 2, "yogo_conv2d_dgrad_bf16": 2, "yogo_bn_bwd_bf16": 2, "yogo_decode_loss_bwd_bf16": 4,
       "yogo_conv2d_fwd_bf16_signs": 3, "yogo_conv2d_fwd_bf16": None, "yogo_bn_apply_act_bf16": 1, "yogo_conv_first_mfma": None}


def spy(fn, *args):
    r = orig(fn, *args)
    if fn in OUT:
        if fn == "yogo_conv2d_fwd_bf16":
            out = args[3] if args[3] is not None else args[4]
            rec.append((fn, out.clone()))
        elif fn == "yogo_conv_first_mfma":
            if args[3] is not None:
                rec.append((fn + ":z", args[3].clone()))
                rec.append((fn + ":y", args[4].clone()))
        else:
            rec.append((fn, args[OUT[fn]].clone()))
    return r


_hip.call = spy
import yogo_amd.engine as E
import yogo_amd.train as T
E._hip.call = spy
tr.step(x.cuda(), lab.cuda())
torch.cuda.synchronize()
_hip.call = orig

spec = O.arch(name, C)
taps = {}
loss_ref, _, grads_ref, _ = O.bf16_train_step(x, sd0, spec, lab, 0.0425, 0.0555, taps=taps)
print("loss", tr.loss_components()["loss"], loss_ref)


def from8c(t, Cc):
    Bq, cb, Hh, Ww, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(Bq, cb * 8, Hh, Ww)[:, :Cc].cpu()


def cmp(tag, got, want):
    d = (got - want).abs()
    scale = float(want.abs().max()) + 1e-30
    ulp = want.abs().clamp_min(1e-30) * 2.0 ** -7
    nbad = int((d > 2 * ulp + 1e-6 * scale).sum())
    idx = int(d.reshape(-1).argmax())
    pos = torch.unravel_index(torch.tensor(idx), d.shape)
    print(f"{tag:34s} shape {tuple(want.shape)} max|d| {float(d.max()):.3e} (of max {scale:.3e}) at {[int(p) for p in pos]} got {float(got.reshape(-1)[idx]):.5g} want {float(want.reshape(-1)[idx]):.5g}; "
          f"{nbad} of {d.numel()} beyond 2 bf16 ulps")


# ---- forward: walk the records in call order against the layer table
n = len(spec)
fi = 0
bi = n - 1
fwd_layer = 0
pending_bn = None
for fn, t in rec:
    if fn == "yogo_conv_first_mfma:z":
        cmp("L0 z", from8c(t, spec[0][0]), taps["z0"])
    elif fn == "yogo_conv_first_mfma:y":
        cmp("L0 y", from8c(t, spec[0][0]), taps["y0"])
        fwd_layer = 1
    elif fn in ("yogo_conv2d_fwd_bf16_signs", "yogo_conv2d_fwd_bf16"):
        i = fwd_layer
        co, k, s, hb, hbn, act, dp = spec[i]
        if t.dtype == torch.float32:
            cmp(f"L{i} raw (fp32 head)", t.cpu(), taps[f"y{i}"])
            fwd_layer += 1
        elif hbn:
            cmp(f"L{i} z", from8c(t, co), taps[f"z{i}"])
        else:
            cmp(f"L{i} y", from8c(t, co), taps[f"y{i}"])
            fwd_layer += 1
    elif fn == "yogo_bn_apply_act_bf16":
        i = fwd_layer
        cmp(f"L{i} y (BN)", from8c(t, spec[i][0]), taps[f"y{i}"])
        fwd_layer += 1
    elif fn == "yogo_decode_loss_bwd_bf16":
        cmp("head gradient g", from8c(t, spec[-1][0]), taps[f"g{n - 1}"])
        gf = taps["graw_f32"]
        print("   (fp32 autograd head gradient: max", float(gf.abs().max()), ")")
        bi = n - 1
    elif fn == "yogo_bn_bwd_bf16":
        cmp(f"L{bi} dz (BN backward)", from8c(t, spec[bi][0]), taps[f"dz{bi}"])
    elif fn in ("yogo_conv2d_dgrad_bf16_signs", "yogo_conv2d_dgrad_bf16"):
        bi -= 1
        cmp(f"L{bi} g (dgrad of L{bi + 1})", from8c(t, spec[bi][0]), taps[f"g{bi}"])
off = 0
for pname, p in m.named_parameters():
    a = tr.flat.grad[off:off + p.numel()].view(p.shape).cpu()
    off += p.numel()
    cmp("grad " + pname, a, grads_ref[pname])
