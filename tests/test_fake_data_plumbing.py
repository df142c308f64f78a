"""BASELINE configs[0]: the fake-data dataset through the `yogo train` plumbing on the CPU -- dataset definition -> datasets ->
split -> host DataLoader -> batches -> Trainer epoch loop -> validation -> checkpoint (reference keys) -> YOGO.from_pth ->
final test with Metrics -> run log.  The product's compute is HIP-only; here the TEST plugs the CPU oracle into the two seams of
``yogo_amd.trainer.Trainer`` (``backend_factory`` / ``loader_factory``), so every host-side piece of the driver runs without a
GPU.  The same driver with the real kernels: tests/test_gpu_cli.py."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

import yogo_oracle as O
from yogo_amd.dataset_definition_file import DatasetDefinition
from yogo_amd.model import YOGO
from yogo_amd.trainer import Trainer, build_config
from yogo_amd.utils.argparsers import global_parser
from yogo_amd.yogo_dataloader import collate_rows, get_datasets, split_dataset
from yogo_amd.yogo_dataset import ObjectDetectionDataset, read_image, resize_image

ROOT = Path(__file__).resolve().parent.parent
HW = (64, 96)


@pytest.fixture()
def in_repo_root(monkeypatch):
    monkeypatch.chdir(ROOT)   # dataset definitions hold paths relative to the working directory, like the reference's


def test_fake_dataset_files_and_dataset_class(in_repo_root):
    ds = ObjectDetectionDataset("tests/fake-data/data/images1", "tests/fake-data/data/labels1", 12, 8, ["you", "only", "glance", "once"], image_hw=HW)
    assert len(ds) == 3
    img, rows = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (1, 64, 96)
    assert rows.ndim == 2 and rows.shape[1] == 5 and rows.shape[0] >= 2
    want = np.loadtxt("tests/fake-data/data/labels1/img_1.txt", dtype=np.float32).reshape(-1, 5)
    np.testing.assert_allclose(rows.numpy(), want, rtol=0, atol=0)
    assert int(ds.calc_class_counts().sum()) == sum(len(open(p).read().strip().splitlines()) for p in Path("tests/fake-data/data/labels1").glob("*.txt"))
    # the reader and the resize: uint8 in, uint8 out; identity at the native size
    im = read_image("tests/fake-data/data/images1/img_1.png")
    assert resize_image(im, HW) is im and tuple(resize_image(im, (32, 48)).shape) == (1, 32, 48)
    rgb = read_image("tests/fake-data/data/images1/img_1.png", rgb=True)
    assert tuple(rgb.shape) == (3, 64, 96) and torch.equal(rgb[0], im[0])
    with pytest.raises(FileNotFoundError):
        ObjectDetectionDataset("tests/fake-data/data/images2", "tests/fake-data/data/labels1", 12, 8, ["a"], image_hw=HW,
                               extensions=("jpg",))   # the .png files are not valid files under this filter -> images "missing"
    # unreadable samples are dropped by the collate function, an all-bad batch yields None
    assert collate_rows([None, ds[1], None])[0].shape[0] == 1 and collate_rows([None]) is None


def test_splits_follow_the_definition(in_repo_root):
    d = DatasetDefinition.from_yaml(Path("tests/fake-data/defns/literal_tests_123.yml"))
    parts = get_datasets(d, 12, 8, image_hw=HW)
    assert {k: len(v) for k, v in parts.items()} == {"train": 6, "val": 0, "test": 3}
    d2 = DatasetDefinition.from_yaml(Path("tests/fake-data/defns/train_val_test.yml"))
    sizes = {k: len(v) for k, v in get_datasets(d2, 12, 8, image_hw=HW).items()}
    assert sum(sizes.values()) == 9 and sizes["train"] >= 4 and sizes["val"] >= 2 and sizes["test"] >= 2
    # the split is the reference's random_split under manual_seed(7271978): reproducible
    a = split_dataset(list(range(9)), d2.split_fractions)
    b = split_dataset(list(range(9)), d2.split_fractions)
    assert [list(a[k].indices) for k in a] == [list(b[k].indices) for k in b]


class OracleBackend:
    """test-only compute for the Trainer seam: the CPU oracle's step / evaluation on the model's own state dict"""

    def __init__(self, net, config, total_steps, device):
        self.net, self.cfg, self.total, self.step_no = net, config, max(1, total_steps), 0
        self.spec = O.arch(net.model_version, int(net.num_classes))
        self.names = [k for k, _ in net.named_parameters()]
        self.m = {k: torch.zeros_like(v) for k, v in net.named_parameters()}
        self.v = {k: torch.zeros_like(v) for k, v in net.named_parameters()}

    def current_lr(self):
        return O.cosine_lr(self.step_no, self.cfg["learning_rate"], self.total, self.cfg["learning_rate"] / self.cfg["decay_factor"])

    def train_step(self, imgs, labels):
        sd = {k: v.detach().clone() for k, v in self.net.state_dict().items()}
        leaf = {k: sd[k].clone().requires_grad_(True) for k in self.names}
        ns = {}
        pred = O.yogo_forward(imgs, {**sd, **leaf}, self.spec, self.cfg["anchor_w"], self.cfg["anchor_h"], train=True, new_stats=ns)
        loss, comps = O.yogo_loss(pred, labels, self.cfg["no_obj_weight"], self.cfg["iou_weight"], 1.0, self.cfg["label_smoothing"])
        loss.backward()
        g = O.clamp_grads({k: v.grad for k, v in leaf.items()})
        lr = self.current_lr()
        self.step_no += 1
        for k in self.names:
            sd[k], self.m[k], self.v[k] = O.adamw_step(sd[k], g[k], self.m[k], self.v[k], self.step_no, lr, weight_decay=self.cfg["weight_decay"])
        sd.update(ns)
        self.net.load_state_dict(sd)
        return {"loss": float(loss.detach()), **comps}

    @torch.no_grad()
    def eval_batch(self, imgs, labels):
        sd = self.net.state_dict()
        pred = O.yogo_forward(imgs, sd, self.spec, self.cfg["anchor_w"], self.cfg["anchor_h"], train=False)
        loss, _ = O.yogo_loss(pred, labels, self.cfg["no_obj_weight"], self.cfg["iou_weight"], 1.0, self.cfg["label_smoothing"])
        return pred, loss

    def optimizer_state_dict(self):
        return {"state": {}, "param_groups": []}

    def set_global_step(self, step):
        self.step_no = int(step)


class HostLoader:
    """test-only loader for the Trainer seam: the product's datasets / sampler / collate, labels rasterised by the oracle"""

    def __init__(self, dataset, batch_size, Sx, Sy):
        self.dataset, self.batch_size, self.Sx, self.Sy, self.sampler = dataset, batch_size, Sx, Sy, None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.dataset), self.batch_size):
            imgs, rows = collate_rows([self.dataset[j] for j in range(i, min(len(self.dataset), i + self.batch_size))])
            yield imgs, torch.stack([O.label_rows_to_tensor(r, self.Sx, self.Sy) for r in rows])


def _loader_factory(defn, config, Sx, Sy):
    parts = get_datasets(defn, Sx, Sy, image_hw=tuple(config["image_hw"]), split_fraction_override=config["dataset_split_override"])
    return {k: HostLoader(v, config["batch_size"], Sx, Sy) for k, v in parts.items() if len(v) > 0}


def test_yogo_train_plumbing_one_epoch_on_fake_data(in_repo_root, tmp_path, monkeypatch):
    args = global_parser().parse_args(["train", "tests/fake-data/defns/train_val_test.yml", "--epochs", "5", "-bs", "2", "--image-hw", "64", "96",
                                       "--name", "plumbing", "--lr", "0.003"])
    config = build_config(args)
    config["trained_models_dir"] = str(tmp_path / "trained_models")
    config["compute_device"] = "cpu"            # test seam only: the product's HipBackend refuses CPU tensors
    import yogo_amd.metrics as M

    # Metrics.update matches predictions with the batched HIP kernel; on the CPU the TEST swaps in the oracle's matching
    def cpu_matching(preds, labels, objectness_thresh=0.5, min_class_confidence_threshold=0.0):
        from yogo_amd.utils.prediction_formatting import PredictionLabelMatch

        out = []
        for p, l in zip(preds, labels):
            r = O.format_preds_and_labels_v2(p, l, objectness_thresh, min_class_confidence_threshold)
            out.append(PredictionLabelMatch(*r) if isinstance(r, tuple) else PredictionLabelMatch(r.preds, r.labels, r.missed_labels, r.extra_predictions))
        return out

    monkeypatch.setattr(M, "format_preds_and_labels_v2_batched", cpu_matching)
    tr = Trainer(config, backend_factory=lambda net, cfg, steps, dev: OracleBackend(net, cfg, steps, dev), loader_factory=_loader_factory)
    tr.init()
    assert (tr.Sx, tr.Sy) == (12, 8) and config["class_names"] == ["you", "only", "glance", "once"]
    tr.train()
    run = tmp_path / "trained_models" / "plumbing"
    recs = [json.loads(l) for l in open(run / "log.jsonl")]
    losses = [r["train loss"] for r in recs if "train loss" in r]
    n_train = len(tr.train_dataloader.dataset)
    assert len(losses) == 5 * ((n_train + 1) // 2) and all(np.isfinite(losses))
    assert np.mean(losses[-2:]) < np.mean(losses[:2])                 # it learns
    assert any("val loss" in r for r in recs) and (run / "best.pth").exists()
    ck = torch.load(run / "best.pth", map_location="cpu", weights_only=False)
    assert set(ck) >= {"epoch", "step", "normalize_images", "classes", "model_name", "model_state_dict", "optimizer_state_dict", "model_version"}
    assert ck["classes"] == ["you", "only", "glance", "once"] and ck["model_version"] == "base_model"
    net, cfg = YOGO.from_pth(run / "best.pth")                        # the checkpoint loads through the reference's loader surface
    assert cfg["step"] == ck["step"] and int(net.num_classes) == 4
    tm = json.load(open(run / "test_metrics.json"))
    assert np.isfinite(tm["test loss"]) and "test mAP" in tm and len(tm["confusion"]) == 16
