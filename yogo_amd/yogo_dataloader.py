"""Datasets -> batches on the device (yogo/data/yogo_dataloader.py:69-324).

``get_datasets`` / ``split_dataset`` / ``get_dataloader`` keep the reference's names, arguments and split behaviour
(``random_split`` under ``manual_seed(7271978)``, ``DistributedSampler`` per rank with torch's defaults, batch-level flip
augmentations on the training split only).  What differs is WHERE the work happens: workers decode images and parse label
files, the loader then moves the stacked uint8 images to the MI355X, rasterises all label tensors of the batch with one HIP
launch and applies both flips in one fused pass (yogo_amd/data.py) -- a ``DeviceLoader`` yields ``(imgs [B,C,H,W] uint8,
labels [B,6,Sy,Sx] fp32)`` already resident in HBM, which is what ``HipTrainer.step`` consumes.
"""
from __future__ import annotations

import os
import warnings
from typing import Any, Dict, Iterable, List, MutableMapping, Optional, Tuple

import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset, Subset, random_split
from torch.utils.data.distributed import DistributedSampler

from yogo_amd.data import MultiArgSequential, RandomHorizontalFlipWithBBs, RandomVerticalFlipWithBBs, format_labels_batch
from yogo_amd.dataset_definition_file import DatasetDefinition, SplitFractions
from yogo_amd.yogo_dataset import ObjectDetectionDataset

SPLIT_SEED = 7271978   # yogo/data/yogo_dataloader.py:176


def guess_suggested_num_workers() -> Optional[int]:
    if hasattr(os, "sched_getaffinity"):
        try:
            return len(os.sched_getaffinity(0))
        except Exception:
            pass
    n = os.cpu_count()
    if n is None:
        warnings.warn("could not figure out the number of cpus on this machine")
    return n


def choose_dataloader_num_workers(dataset_size: int, requested_num_workers: Optional[int] = None) -> int:
    if dataset_size < 1000:
        return 0
    if requested_num_workers is not None:
        return requested_num_workers
    return min(guess_suggested_num_workers() or 32, 64)


def _concat(dataset_paths, Sx, Sy, classes, image_hw, rgb, normalize_images) -> ConcatDataset:
    return ConcatDataset([ObjectDetectionDataset(dsp.image_path, dsp.label_path, Sx, Sy, image_hw=image_hw, rgb=rgb, classes=classes,
                                                 normalize_images=normalize_images) for dsp in dataset_paths])


def get_datasets(dataset_definition: DatasetDefinition, Sx: int, Sy: int, rgb: bool = False, image_hw: Tuple[int, int] = (772, 1032),
                 normalize_images: bool = False, split_fraction_override: Optional[SplitFractions] = None) -> MutableMapping[str, Dataset]:
    """dataset definition -> {"train": ..., "val": ..., "test": ...} (yogo_dataloader.py:69-151).  The thumbnail ("blob")
    augmentation of the reference is not part of this build: a definition that asks for it is refused, not silently ignored."""
    if getattr(dataset_definition, "thumbnail_augmentation", None):
        raise NotImplementedError("yogo_amd: thumbnail_augmentation (yogo/data/blobgen.py) is out of scope of this build")
    classes = dataset_definition.classes
    full = _concat(dataset_definition.dataset_paths, Sx, Sy, classes, image_hw, rgb, normalize_images)
    test_paths = dataset_definition.test_dataset_paths
    if test_paths is not None and len(test_paths) > 0:
        test = _concat(test_paths, Sx, Sy, classes, image_hw, rgb, normalize_images)
        if split_fraction_override is not None:
            return split_dataset(ConcatDataset([full, test]), split_fraction_override)
        assert "test" not in dataset_definition.split_fractions
        return {**split_dataset(full, dataset_definition.split_fractions), "test": test}
    return split_dataset(full, split_fraction_override if split_fraction_override is not None else dataset_definition.split_fractions)


def split_dataset(dataset: Dataset, split_fractions: SplitFractions) -> MutableMapping[str, Dataset]:
    if not hasattr(dataset, "__len__"):
        raise ValueError(f"dataset {dataset} must have a length (specifically, `__len__` must be defined)")
    keys = split_fractions.keys()
    sizes = split_fractions.partition_sizes(len(dataset))   # type: ignore[arg-type]
    return dict(zip(keys, random_split(dataset, [sizes[k] for k in keys], generator=torch.Generator().manual_seed(SPLIT_SEED))))


def collate_rows(batch: List[Optional[Tuple[torch.Tensor, torch.Tensor]]]) -> Optional[Tuple[torch.Tensor, List[torch.Tensor]]]:
    """host side of yogo/data/utils.py:49-63 (collate_batch_robust): drop unreadable samples, stack the images; the label rows stay
    a list (ragged) until the device rasterises them"""
    pairs = [pair for pair in batch if pair is not None]
    if not pairs:
        return None
    imgs, rows = zip(*pairs)
    return torch.stack(imgs), list(rows)


class DeviceLoader:
    """wraps the host DataLoader: batch -> device, label rows -> [B, 6, Sy, Sx] with one launch, flips fused.  Keeps the
    attributes the training loop touches (``dataset``, ``sampler``, ``batch_size``, ``__len__``)."""

    def __init__(self, loader: DataLoader, Sx: int, Sy: int, transforms: MultiArgSequential, device=None, strict_steps: bool = False):
        self.loader, self.Sx, self.Sy, self.transforms, self.device = loader, Sx, Sy, transforms, device
        # strict_steps (the TRAINING split only): every batch is one gradient all-reduce, so a skipped batch must fail loudly
        # under data parallelism; validation / test loaders issue no per-step collective and may skip, as the reference does
        self.strict_steps = strict_steps
        self.dataset, self.sampler, self.batch_size = loader.dataset, loader.sampler, loader.batch_size

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self):
        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        for item in self.loader:
            if item is None:
                # every sample of the batch was unreadable (the robust collate of yogo/data/utils.py:49-63 returned nothing).  A
                # single process just skips it; under data parallelism a rank that skips a step issues one gradient all-reduce
                # fewer than its peers and the job hangs -- fail loudly instead
                if multi and self.strict_steps:
                    raise RuntimeError("yogo_amd: a whole batch of this rank was unreadable; in a data-parallel run every rank must "
                                       "take the same number of steps (fix or remove the unreadable files)")
                continue
            imgs, rows = item
            imgs = imgs.to(dev, non_blocking=True)
            labels = format_labels_batch(rows, self.Sx, self.Sy, "cxcywh", device=dev)
            yield self.transforms(imgs, labels)


def get_dataloader(dataset_definition: DatasetDefinition, batch_size: int, Sx: int, Sy: int, training: bool = True,
                   image_hw: Tuple[int, int] = (772, 1032), rgb: bool = False, normalize_images: bool = False,
                   split_fraction_override: Optional[SplitFractions] = None, device=None) -> Dict[str, DeviceLoader]:
    split_datasets = get_datasets(dataset_definition, Sx, Sy, rgb=rgb, image_hw=image_hw, normalize_images=normalize_images,
                                  split_fraction_override=split_fraction_override)
    augmentations = [RandomHorizontalFlipWithBBs(0.5), RandomVerticalFlipWithBBs(0.5)] if training else []
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world_size = torch.distributed.get_rank(), torch.distributed.get_world_size()
    else:
        rank, world_size = 0, 1
    d: Dict[str, DeviceLoader] = {}
    for designation, dataset in split_datasets.items():
        if len(dataset) == 0:   # type: ignore[arg-type]
            continue
        augs = augmentations if designation == "train" else []
        d[designation] = _get_dataloader(dataset, batch_size, augs, rank, world_size, Sx, Sy, device, strict_steps=designation == "train")
    return d


def _get_dataloader(dataset: Dataset, batch_size: int, augmentations: list, rank: int, world_size: int, Sx: int, Sy: int,
                    device=None, strict_steps: bool = False) -> DeviceLoader:
    sampler: Iterable = DistributedSampler(dataset, rank=rank, num_replicas=world_size)   # torch defaults: shuffle, seed 0, padded
    num_workers = choose_dataloader_num_workers(len(dataset)) // world_size   # type: ignore[arg-type]
    if len(dataset) >= 1000:   # type: ignore[arg-type]
        num_workers = max(1, num_workers)
    loader = DataLoader(dataset, shuffle=False, sampler=sampler, drop_last=False, pin_memory=torch.cuda.is_available(), batch_size=batch_size,
                        num_workers=num_workers, persistent_workers=num_workers > 0, generator=torch.Generator().manual_seed(SPLIT_SEED),
                        collate_fn=collate_rows, multiprocessing_context="spawn" if num_workers > 0 else None)
    return DeviceLoader(loader, Sx, Sy, MultiArgSequential(*augmentations), device, strict_steps=strict_steps)


def get_class_counts(d, num_classes: int, verbose: bool = True) -> torch.Tensor:
    """class histogram of the objects a loader yields (yogo_dataloader.py:284-311)"""
    class_counts = torch.zeros(num_classes, dtype=torch.long)
    for _, labels in d:
        bs, pd, Sy, Sx = labels.shape
        flat = labels.permute(1, 0, 2, 3).reshape(pd, bs * Sy * Sx)
        flat = flat[:, flat[0, :] == 1].long()
        class_counts += torch.bincount(flat[5, :], minlength=num_classes).cpu()
    return class_counts


def get_image_count(d) -> int:
    if isinstance(d.dataset, ConcatDataset):
        return d.dataset.cumulative_sizes[-1]
    if isinstance(d.dataset, Subset):
        return len(d.dataset)
    raise TypeError(f"unknown type {type(d.dataset)}")
