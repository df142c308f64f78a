"""The drivers around the hot path, with the real kernels (SURVEY.md 8(f) rows + the `yogo train | infer | test` CLI):
* `Trainer` on the fake dataset (BASELINE configs[0] layout) with the HIP backend: per-step training losses equal the oracle's
  steps on the same batches; checkpoint round trip;
* the device loader: batches on the MI355X, labels rasterised by one launch == the oracle's per-image rasteriser;
* `predict()`: .txt / .npy / .json / counts / drawn boxes from batched launches == the oracle's per-image loops;
* `Metrics` on the batched matching == the same statistics on the oracle's matching;
* crop path: `resize_model(193)` inference (Sy = 25) == the oracle with the size multipliers;
* the console entry point end to end in child processes: train -> infer -> test."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import yogo_oracle as O

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
HW = (64, 96)
CLASSES = ["you", "only", "glance", "once"]


@pytest.fixture()
def in_repo_root(monkeypatch):
    monkeypatch.chdir(ROOT)


def _config(tmp_path, **over):
    from yogo_amd.trainer import build_config
    from yogo_amd.utils.argparsers import global_parser

    args = global_parser().parse_args(["train", "tests/fake-data/defns/train_val_test.yml", "--epochs", "1", "-bs", "2", "--image-hw", "64", "96",
                                       "--name", "gpu_run"])
    cfg = build_config(args)
    cfg["trained_models_dir"] = str(tmp_path / "trained_models")
    cfg.update(over)
    return cfg


def test_device_loader_matches_oracle_rasteriser(in_repo_root):
    from yogo_amd.dataset_definition_file import DatasetDefinition
    from yogo_amd.yogo_dataloader import get_dataloader

    d = DatasetDefinition.from_yaml(Path("tests/fake-data/defns/literal_tests_123.yml"))
    loaders = get_dataloader(d, 4, 12, 8, training=False, image_hw=HW)
    assert set(loaders) == {"train", "test"}
    seen = 0
    for imgs, labels in loaders["test"]:
        assert imgs.is_cuda and labels.is_cuda and imgs.dtype == torch.uint8 and tuple(labels.shape[1:]) == (6, 8, 12)
        seen += imgs.shape[0]
    assert seen == 3
    ds = loaders["test"].dataset
    order = list(iter(loaders["test"].sampler))       # DistributedSampler shuffles (torch defaults, seed 0): same order per epoch
    imgs, labels = next(iter(loaders["test"]))
    want = torch.stack([O.label_rows_to_tensor(ds[i][1], 12, 8) for i in order])
    assert torch.equal(labels.cpu(), want)
    assert torch.equal(imgs.cpu(), torch.stack([ds[i][0] for i in order]))
    # training loaders flip whole batches, labels included: a flipped batch is still a valid label tensor of the flipped image
    tl = get_dataloader(d, 3, 12, 8, training=True, image_hw=HW)["train"]
    for imgs, labels in tl:
        assert set(labels[:, 0].unique().tolist()) <= {0.0, 1.0}


def test_trainer_losses_match_oracle_steps(in_repo_root, tmp_path):
    from yogo_amd.trainer import Trainer
    from yogo_amd.yogo_dataloader import get_dataloader

    class T(Trainer):
        def _init_model(self):
            torch.manual_seed(5)
            super()._init_model()
            for m in self.net.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0.0                      # Dropout2d cannot match torch's RNG in the oracle

    def loaders(defn, config, Sx, Sy):             # the product's loaders, augmentation off (flips are random draws)
        return get_dataloader(defn, config["batch_size"], Sx, Sy, training=False, image_hw=tuple(config["image_hw"]))

    cfg = _config(tmp_path, epochs=2)
    tr = T(cfg, loader_factory=loaders)
    tr.init()
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.net.state_dict().items()}
    batches = [(i.cpu(), l.cpu()) for i, l in tr.train_dataloader] * 2     # (sampler order is fixed per epoch only when set_epoch is not shuffling: see below)
    tr.train_dataloader.sampler.set_epoch(0)
    e0 = [(i.cpu(), l.cpu()) for i, l in tr.train_dataloader]
    tr.train_dataloader.sampler.set_epoch(1)
    e1 = [(i.cpu(), l.cpu()) for i, l in tr.train_dataloader]
    tr.train()
    run = tmp_path / "trained_models" / "gpu_run"
    got = [json.loads(l)["train loss"] for l in open(run / "log.jsonl") if "train loss" in l]
    # oracle: the same steps on the CPU
    spec = O.arch("base_model", 4)
    names = [k for k, v in sd0.items() if k.startswith("model.") and v.is_floating_point() and "running" not in k]
    sd = dict(sd0)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    total = 2 * len(e0)
    want = []
    for step, (x, lab) in enumerate(e0 + e1, start=1):
        leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
        ns = {}
        pred = O.yogo_forward(x, {**sd, **leaf}, spec, 0.0425, 0.0555, train=True, new_stats=ns)
        loss, _ = O.yogo_loss(pred, lab)
        loss.backward()
        g = O.clamp_grads({k: t.grad for k, t in leaf.items()})
        lr = O.cosine_lr(step - 1, 3e-4, total, 3e-5)
        for k in names:
            sd[k], m[k], v[k] = O.adamw_step(sd[k], g[k], m[k], v[k], step, lr)
            sd[k] = sd[k].detach()
        sd.update(ns)
        want.append(float(loss.detach()))
    assert len(got) == len(want) == total
    np.testing.assert_allclose(got, want, rtol=2e-3)
    # checkpoint of epoch 0's validation: reference keys, loads through from_pth, optimiser state in AdamW's layout
    from yogo_amd.model import YOGO

    ck = torch.load(run / "best.pth", map_location="cpu", weights_only=False)
    assert ck["classes"] == CLASSES and ck["optimizer_state_dict"]["param_groups"][0]["weight_decay"] == 0.05
    net, info = YOGO.from_pth(run / "best.pth")
    assert info["step"] == ck["step"] > 0
    assert (run / "test_metrics.json").exists()


def _make_checkpoint(tmp_path, seed=3):
    """a checkpoint with 'trained-like' weights: a few steps on the fake data, so that some predictions pass the thresholds"""
    from yogo_amd.model import YOGO

    torch.manual_seed(seed)
    net = YOGO(HW, 0.0425, 0.0555, 4).cuda()
    net.eval()
    with torch.no_grad():
        for m in net.modules():               # trained-like running statistics keep the eval-mode activations in range
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 50.0)
                m.running_var.uniform_(2000.0, 9000.0)
        net.model[7].bias[4] += 1.5           # ... and the objectness logit up so that a handful of cells fire
    p = tmp_path / "m.pth"
    torch.save({"epoch": 0, "step": 7, "normalize_images": False, "classes": CLASSES, "model_name": "fake_model",
                "model_state_dict": {k: v.cpu() for k, v in net.state_dict().items()}, "model_version": "base_model"}, p)
    return p, net


@pytest.mark.parametrize("half", [False, True])
def test_predict_outputs_match_oracle(in_repo_root, tmp_path, half, capsys):
    from yogo_amd.infer import predict
    from yogo_amd.yogo_dataset import read_image

    pth, net = _make_checkpoint(tmp_path)
    imgdir = ROOT / "tests/fake-data/data/images1"
    out = tmp_path / "out"
    full = predict(str(pth), path_to_images=imgdir, output_dir=str(out), save_preds=True, save_npy=True, count_predictions=True, batch_size=2,
                   obj_thresh=0.4, iou_thresh=0.5, half=half, return_full_predictions=True, class_names=CLASSES)
    files = sorted(imgdir.glob("*.png"))
    assert full.shape == (3, 9, 8, 12)
    # the forward itself: fp32 == the oracle to fp32 tolerance; bf16 within bf16's (3e-2 of the output range, as test_gpu_parity)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    x = torch.stack([read_image(f) for f in files])
    ref = O.yogo_forward(x, sd, O.arch("base_model", 4), 0.0425, 0.0555, inference=True)
    tol = 3e-2 * float(ref.abs().max()) if half else 1e-4 * float(ref.abs().max())
    assert float((full - ref).abs().max()) < tol
    # every output file is the oracle's per-image post-processing OF THE DEVICE'S predictions
    for k, f in enumerate(files):
        want = O.save_predictions_text(O.format_preds(full[k], obj_thresh=0.4, iou_thresh=0.5))
        assert (out / f.with_suffix(".txt").name).read_text() == want
    npy = np.load(out / "data.npy")
    want_np = np.hstack([O.format_to_numpy(k, full[k].numpy(), 64, 96) for k in range(3)])
    assert npy.dtype == want_np.dtype and np.array_equal(npy, want_np)
    meta = json.load(open(out / "data.json"))
    assert set(meta) == {"run_name", "model_name", "obj_thresh", "iou_thresh", "vertical_crop_height_px", "write_date"}
    assert meta["model_name"] == "fake_model" and meta["vertical_crop_height_px"] == 64 and meta["run_name"] == "data"
    counts = O.get_prediction_class_counts(full, obj_thresh=0.4, iou_thresh=0.5)
    assert str(list(zip(CLASSES, map(int, counts)))) in capsys.readouterr().out
    # drawn boxes: one image per input, same size, something drawn when something was predicted
    out2 = tmp_path / "boxes"
    predict(str(pth), path_to_images=imgdir, output_dir=str(out2), draw_boxes=True, batch_size=3, obj_thresh=0.4, half=half)
    from PIL import Image

    for f in files:
        im = Image.open(out2 / f.name)
        assert im.size == (96, 64) and im.mode == "RGBA"
    with pytest.raises(ValueError):
        predict(str(pth), path_to_images=imgdir, output_dir=str(out2), draw_boxes=True, save_preds=True)
    with pytest.raises(ValueError):
        predict(str(pth), path_to_images=imgdir, class_names=["a", "b"])


def test_predict_with_vertical_crop(in_repo_root, tmp_path):
    """--crop-height: CenterCrop + YOGO.resize_model (yogo/infer.py:221-226, yogo/model.py:236-265)"""
    from yogo_amd.infer import predict
    from yogo_amd.image_path_dataset import CenterCrop
    from yogo_amd.yogo_dataset import read_image

    pth, net = _make_checkpoint(tmp_path)
    imgdir = ROOT / "tests/fake-data/data/images2"
    full = predict(str(pth), path_to_images=imgdir, vertical_crop_height=0.5, batch_size=3, return_full_predictions=True)
    assert full.shape == (3, 9, 4, 12)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    x = torch.stack([CenterCrop((32, 96))(read_image(f)) for f in sorted(imgdir.glob("*.png"))])
    sd = {k: v for k, v in sd.items() if k not in ("_Cxs", "_Cys")}      # the crop has its own grid
    ref = O.yogo_forward(x, sd, O.arch("base_model", 4), 0.0425, 0.0555, inference=True, height_multiplier=2.0)
    assert float((full - ref).abs().max()) < 1e-4 * float(ref.abs().max())


def test_resized_model_at_production_width():
    """the crop path at the reference's operating point: 193 x 1032 (Sy = 25, Sx = 129) through resize_model"""
    from yogo_amd.model import YOGO

    torch.manual_seed(2)
    net = YOGO((772, 1032), 0.0425, 0.0555, 7, inference=True).cuda()
    net.eval()
    with torch.no_grad():
        for m in net.modules():                  # trained-like statistics keep eval-mode activations in range
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 50.0)
                m.running_var.uniform_(2000.0, 9000.0)
    net.resize_model(193)
    assert (net.Sx, net.Sy) == (129, 25) and float(net.height_multiplier) == 4.0
    x = O.synthetic_images(2, 193, 1032, seed=8)
    with torch.no_grad():
        got = net(x.cuda()).cpu()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = O.yogo_forward(x, sd, O.arch("base_model", 7), 0.0425, 0.0555, inference=True, height_multiplier=4.0)
    assert got.shape == ref.shape == (2, 12, 25, 129)
    assert float((got - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        got16 = net(x.cuda()).cpu()
    assert float((got16 - ref).abs().max()) < 3e-2 * float(ref.abs().max())


def test_metrics_on_batched_matching_equal_oracle_matching():
    import yogo_amd.metrics as M
    from yogo_amd.utils.prediction_formatting import PredictionLabelMatch

    B, C, Sy, Sx = 6, 4, 24, 33
    preds = O.synthetic_predictions(B, Sx, Sy, num_classes=C, K=12, seed=4)
    labels = O.synthetic_labels(B, Sx, Sy, K=12, num_classes=C, seed=5)
    a = M.Metrics(CLASSES, include_mAP=True, include_background=False, min_class_confidence_threshold=0.3)
    a.update(preds.cuda(), labels.cuda())
    ra = a.compute()
    b = M.Metrics(CLASSES, include_mAP=True, include_background=False, min_class_confidence_threshold=0.3)
    real = M.format_preds_and_labels_v2_batched
    try:
        M.format_preds_and_labels_v2_batched = lambda p, l, min_class_confidence_threshold=0.0: [
            PredictionLabelMatch(*O.format_preds_and_labels_v2(pi.cpu(), li.cpu(), 0.5, min_class_confidence_threshold)) for pi, li in zip(p, l)]
        b.update(preds, labels)
    finally:
        M.format_preds_and_labels_v2_batched = real
    rb = b.compute()
    assert float(ra[0]["map"]) == float(rb[0]["map"]) and float(ra[0]["map"]) >= 0
    assert torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]) and torch.equal(ra[4], rb[4]) and torch.equal(ra[5], rb[5])
    assert ra[6] == rb[6] and torch.equal(ra[7], rb[7]) and torch.equal(ra[8], rb[8]) and int(ra[9]) == int(rb[9]) > 0
    # with the background class (the reference's default for Metrics): missed labels / extra predictions become rows
    c = M.Metrics(CLASSES, include_mAP=False, include_background=True, min_class_confidence_threshold=0.3)
    rc = c.forward(preds.cuda(), labels.cuda())
    assert int(rc[1].sum()) == int(ra[1].sum()) + int(ra[7].sum()) + int(ra[8].sum())


def test_console_entry_point_train_infer_test(tmp_path):
    """`python -m yogo_amd train | infer | test` in child processes, on a copy of the fake data (absolute paths)"""
    defn = tmp_path / "defn.yml"
    data = ROOT / "tests" / "fake-data" / "data"
    defn.write_text(
        "class_names: [you, only, glance, once]\n"
        "dataset_split_fractions: {train: 0.75, val: 0.25}\n"
        f"dataset_paths:\n  a: {{image_path: {data}/images1, label_path: {data}/labels1}}\n  b: {{image_path: {data}/images2, label_path: {data}/labels2}}\n"
        f"test_paths:\n  c: {{image_path: {data}/images3, label_path: {data}/labels3}}\n")
    env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(*argv):
        r = subprocess.run([sys.executable, "-m", "yogo_amd", *argv], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    run("train", str(defn), "--epochs", "1", "-bs", "3", "--image-hw", "64", "96", "--name", "cli", "--half")
    best = tmp_path / "trained_models" / "cli" / "best.pth"
    assert best.exists() and (tmp_path / "trained_models" / "cli" / "test_metrics.json").exists()
    out = run("infer", str(best), "--path-to-images", str(data / "images3"), "--count", "--save-preds", "--output-dir", str(tmp_path / "preds"),
              "--obj-thresh", "0.2", "--half", "--no-use-tqdm")
    assert "[(0," in out or "[('" in out or "[(" in out
    assert len(list((tmp_path / "preds").glob("*.txt"))) == 3
    out = run("test", str(best), str(defn), "--include-mAP")
    assert "test loss" in out
    r = subprocess.run([sys.executable, "-m", "yogo_amd", "export", str(best)], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "not part of this MI355X build" in r.stdout
