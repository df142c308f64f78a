"""``predict`` / ``do_infer`` -- the driver behind `yogo infer` (yogo/infer.py:139-451) on the HIP path.

Same arguments, output files and return value as the reference.  Per batch: ONE forward (fp32, or the bf16 matrix-core path
under ``half`` -- the reference's bf16 autocast, infer.py:313-317) and, per requested output, ONE batched threshold + NMS
launch for the whole batch (``save_predictions`` / ``format_to_numpy_batched`` / ``get_prediction_class_counts``) instead of the
reference's per-image Python loops over ``format_preds`` (infer.py:45,73); unless the decoded tensor itself is asked for, the box
decode runs inside that launch's loads (``YOGO.forward_raw``).  No ``torch.compile``: there is no graph to trace,
the model is already a fixed sequence of hand-written kernels.
"""
from __future__ import annotations

import datetime
import json
import warnings
from pathlib import Path
from typing import List, Optional, Union

import numpy as np
import torch
from torch.utils.data import DataLoader

from yogo_amd.image_path_dataset import CenterCrop, collate_fn, get_dataset
from yogo_amd.model import YOGO
from yogo_amd.utils import format_to_numpy_batched, get_prediction_class_counts, save_predictions  # noqa: F401
from yogo_amd.utils.utils import choose_device, draw_yogo_prediction
from yogo_amd.yogo_dataloader import choose_dataloader_num_workers


def get_model_name_from_pth(path_to_pth: Union[str, Path]) -> Optional[str]:
    return torch.load(Path(path_to_pth), map_location="cpu", weights_only=False).get("model_name", None)


def write_metadata(metadata_path: Path, **kwargs) -> None:
    """a json file with the kwargs, next to the .npy (yogo/infer.py:129-135)"""
    with open(Path(metadata_path).with_suffix(".json"), "w") as f:
        json.dump(kwargs, f, indent=4)


@torch.no_grad()
def predict(
    path_to_pth: str,
    *,
    path_to_images: Optional[Path] = None,
    path_to_zarr: Optional[Path] = None,
    output_dir: Optional[str] = None,
    draw_boxes: bool = False,
    save_preds: bool = False,
    save_npy: bool = False,
    class_names: Optional[List[str]] = None,
    count_predictions: bool = False,
    batch_size: int = 64,
    obj_thresh: float = 0.5,
    iou_thresh: float = 0.5,
    vertical_crop_height: Optional[float] = None,
    use_tqdm: bool = False,
    device: Optional[Union[str, torch.device]] = None,
    output_img_ftype: str = ".png",
    requested_num_workers: Optional[int] = None,
    min_class_confidence_threshold: float = 0.0,
    half: bool = False,
    return_full_predictions: bool = False,
) -> Optional[torch.Tensor]:
    if save_preds and draw_boxes:
        raise ValueError("cannot save predictions in YOGO format and draw_boxes at the same time")
    elif output_dir is not None and not (save_preds or draw_boxes or save_npy):
        warnings.warn(f"output dir is not None (is {output_dir}), but it will not be used since save_preds and draw_boxes are both false")
    elif output_dir is not None:
        Path(output_dir).mkdir(exist_ok=True, parents=False)
    elif save_preds:
        raise ValueError("output_dir must not be None if save_preds is True")
    elif output_img_ftype not in [".png", ".tif", ".tiff"]:
        raise ValueError(f"only .png, .tif, and .tiff are supported for output img filetype; got {output_img_ftype}")

    device = torch.device(device or choose_device())
    if device.type != "cuda":
        raise RuntimeError(f"yogo_amd: inference runs on an MI355X (got device {device}); there is no CPU compute path")
    model, cfg = YOGO.from_pth(Path(path_to_pth), inference=True)
    model.eval()
    model.to(device)

    transforms = []
    img_h, img_w = (int(v) for v in model.get_img_size())
    if vertical_crop_height:
        crop_px = int(round(vertical_crop_height * img_h))
        transforms.append(CenterCrop((crop_px, img_w)))
        model.resize_model(crop_px)
        img_h = crop_px
    assert model.img_size.numel() == 2, f"YOGO model must be 2D, is {model.img_size}"
    num_classes = int(model.num_classes)
    if class_names is not None and len(class_names) != num_classes:
        raise ValueError(f"expected {num_classes} class names, got {len(class_names)}")

    image_dataset = get_dataset(path_to_images=path_to_images, path_to_zarr=path_to_zarr, image_transforms=transforms,
                                normalize_images=bool(model.normalize_images))
    num_workers = choose_dataloader_num_workers(len(image_dataset), requested_num_workers=requested_num_workers)
    loader = DataLoader(image_dataset, batch_size=batch_size, shuffle=False, drop_last=False, pin_memory=True, collate_fn=collate_fn,
                        num_workers=num_workers)
    try:
        from tqdm import tqdm

        pbar = tqdm(disable=not use_tqdm, unit="images", total=len(image_dataset))
    except ImportError:   # pragma: no cover
        pbar = None

    results = torch.zeros((len(image_dataset), 5 + num_classes, model.Sy, model.Sx)) if return_full_predictions else None
    np_results: list = []
    tot_counts = torch.zeros((num_classes,)) if count_predictions else None

    file_iterator = enumerate(loader)
    while True:
        # forgiving to malformed images, as the reference (infer.py:300-309)
        try:
            i, (img_batch, fnames) = next(file_iterator)
        except StopIteration:
            break
        except RuntimeError as e:
            warnings.warn(f"got error {e}; continuing")
            continue
        x = img_batch.to(device, non_blocking=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(half)):
            # the decoded tensor itself is only needed for drawing and for return_full_predictions; every other output goes
            # through the threshold + NMS kernel, which decodes the head's raw output as it loads it
            res = model(x) if (draw_boxes or return_full_predictions) else model.forward_raw(x)
        # the fused launch re-runs the decode per consumer: with more than one post-process output of this batch, decode ONCE
        if sum(map(bool, (save_preds, save_npy, count_predictions))) > 1 and hasattr(res, "decoded"):
            res = res.decoded()
        if draw_boxes:
            for k in range(img_batch.shape[0]):
                bbox_img = draw_yogo_prediction(img=img_batch[k, ...], prediction=res[k, ...], obj_thresh=obj_thresh, iou_thresh=iou_thresh,
                                                min_class_confidence_threshold=min_class_confidence_threshold, labels=class_names,
                                                images_are_normalized=bool(model.normalize_images))
                if output_dir is not None:
                    bbox_img.save(Path(output_dir) / Path(fnames[k]).with_suffix(output_img_ftype).name, compress_level=1)
                else:   # pragma: no cover  (interactive display)
                    import matplotlib.pyplot as plt

                    fig, ax = plt.subplots()
                    ax.set_axis_off()
                    ax.imshow(bbox_img)
                    plt.show()
                    plt.close()
        if save_preds:
            assert output_dir is not None, "output_dir must not be None if save_preds is True"
            save_predictions([Path(output_dir) / Path(f).with_suffix(".txt").name for f in fnames], res, obj_thresh=obj_thresh, iou_thresh=iou_thresh)
        if save_npy:
            ids = [i * batch_size + j for j in range(res.shape[0])]
            np_results.extend(format_to_numpy_batched(ids, res, img_h, img_w))
        if count_predictions:
            tot_counts += get_prediction_class_counts(res, obj_thresh=obj_thresh, iou_thresh=iou_thresh,
                                                      min_class_confidence_threshold=min_class_confidence_threshold)
        if return_full_predictions:
            results[i * batch_size: i * batch_size + res.shape[0], ...] = res.cpu()
        if pbar is not None:
            pbar.update(res.shape[0])
    if pbar is not None:
        pbar.close()

    if count_predictions:
        print(list(zip(class_names or range(num_classes), map(int, tot_counts))))
    if save_npy:
        pred_tensors = np.hstack(np_results) if np_results else np.zeros((8 + num_classes, 0), dtype=np.float32)
        filename = Path(path_to_images).resolve().parent.stem if path_to_images else Path(path_to_zarr).resolve().stem
        base = Path(output_dir).resolve() if output_dir is not None else Path.cwd().resolve()
        fp = base / Path(filename).with_suffix(".npy")
        np.save(fp, pred_tensors)
        write_metadata(fp.with_suffix(".json"), run_name=fp.with_suffix("").name, model_name=get_model_name_from_pth(path_to_pth),
                       obj_thresh=obj_thresh, iou_thresh=iou_thresh, vertical_crop_height_px=img_h,
                       write_date=datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"))
    return results if return_full_predictions else None


def do_infer(args) -> None:
    predict(
        args.pth_path, path_to_images=args.path_to_images, path_to_zarr=args.path_to_zarr, output_dir=args.output_dir,
        draw_boxes=args.draw_boxes, save_preds=args.save_preds, save_npy=args.save_npy, class_names=args.class_names,
        obj_thresh=args.obj_thresh, iou_thresh=args.iou_thresh, batch_size=args.batch_size, device=args.device, use_tqdm=args.use_tqdm,
        vertical_crop_height=args.crop_height, count_predictions=args.count, output_img_ftype=args.output_img_filetype,
        min_class_confidence_threshold=args.min_class_confidence_threshold, half=args.half,
    )
