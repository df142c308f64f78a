"""`yogo test` (yogo/utils/test_model.py:23-116): load a checkpoint with ``inference=False``, run the test split of a dataset
definition through ``Trainer.test`` (loss + Metrics), optionally dump the result.  The hard-coded evaluation settings of the
reference are kept (iou_weight 1, label_smoothing 1e-4, half precision)."""
from __future__ import annotations

import os
import pickle

import torch

from yogo_amd.dataset_definition_file import DatasetDefinition
from yogo_amd.model import YOGO


def test_model(args) -> None:
    from yogo_amd.trainer import Trainer
    from yogo_amd.yogo_dataloader import get_dataloader

    device = "cuda"
    y, cfg = YOGO.from_pth(args.pth_path, inference=False)
    y.to(device)
    data_defn = DatasetDefinition.from_yaml(args.dataset_defn_path)
    config = {
        "class_names": data_defn.classes,
        "no_classify": False,
        "iou_weight": 1,
        "no_obj_weight": 0.5,
        "label_smoothing": 0.0001,
        "half": True,
        "model": str(args.pth_path),
        "test_set": str(args.dataset_defn_path),
        "slurm-job-id": os.getenv("SLURM_JOB_ID", default=None),
    }
    Sx, Sy = y.get_grid_size()
    loaders = get_dataloader(data_defn, 64, Sx, Sy, training=False, image_hw=tuple(int(v) for v in y.get_img_size()),
                             normalize_images=bool(cfg["normalize_images"]), device=device)
    if "test" not in loaders:
        raise RuntimeError(f"{args.dataset_defn_path} defines no test split")
    test_metrics = Trainer.test(loaders["test"], device, config, y, include_mAP=args.include_mAP, include_background=args.include_background)
    if test_metrics is not None:
        mean_loss, mAP = test_metrics[0], test_metrics[1]
        print(f"test loss {mean_loss:.6f}  mAP {float(mAP['map']):.4f}")
    if args.dump_to_disk:
        pickle.dump(test_metrics, open("test_metrics.pkl", "wb"))
test_model.__test__ = False   # (not a pytest test)


def do_model_test(args) -> None:
    if torch.cuda.device_count() == 0:
        raise RuntimeError("at least 1 gpu is required for testing; the hot path is HIP-only (no CPU compute path)")
    test_model(args)
