"""Inference datasets (yogo/data/image_path_dataset.py:17-159): a directory (or one file) of .png images -> (image, path).
The zarr variant of the reference needs the `zarr` package, which this build does not have: asking for it raises."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, List, Optional, Tuple, Union

import numpy as np
import torch
from torch.utils.data import Dataset

from yogo_amd.yogo_dataset import read_image


class ImageAndIdDataset(Dataset):
    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, str]:
        raise NotImplementedError

    def __len__(self) -> int:
        raise NotImplementedError


class ImagePathDataset(ImageAndIdDataset):
    def __init__(self, root: Union[str, Path], image_transforms: Optional[List[Callable]] = None,
                 loader: Callable[[Union[str, Path]], torch.Tensor] = read_image, normalize_images: bool = False):
        self.root = Path(root)
        if not self.root.exists():
            raise FileNotFoundError(f"{self.root} does not exist")
        self.image_paths = self.make_dataset(self.root)
        self.transforms = list(image_transforms or [])
        self.loader = loader
        self.normalize_images = normalize_images

    def make_dataset(self, path_to_data: Path) -> np.ndarray:
        if path_to_data.is_file() and path_to_data.suffix == ".png":
            img_paths = [path_to_data]
        else:
            img_paths = sorted(p for p in path_to_data.glob("*.png") if not p.name.startswith("."))
        if len(img_paths) == 0:
            raise FileNotFoundError(f"{str(path_to_data)} does not contain any images")
        return np.array([str(p) for p in img_paths]).astype(np.str_)

    def __len__(self) -> int:
        return len(self.image_paths)

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, str]:
        image_path = str(self.image_paths[idx])
        image = self.loader(image_path)
        for t in self.transforms:
            image = t(image)
        if self.normalize_images:
            image = image / 255
        return image, image_path


class CenterCrop:
    """torchvision.transforms.CenterCrop((h, w)) for [C, H, W] tensors no smaller than the crop (yogo/infer.py:221-226)"""

    def __init__(self, size: Tuple[int, int]):
        self.size = (int(size[0]), int(size[1]))

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        h, w = img.shape[-2:]
        ch, cw = self.size
        if ch > h or cw > w:
            raise ValueError(f"crop {self.size} larger than the image {(h, w)}")
        top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return img[..., top:top + ch, left:left + cw].contiguous()


def collate_fn(batch: List[Tuple[torch.Tensor, str]]) -> Tuple[torch.Tensor, Tuple[str, ...]]:
    images, fnames = zip(*batch)
    return torch.stack(images), tuple(fnames)


def get_dataset(path_to_images: Optional[Path] = None, path_to_zarr: Optional[Path] = None,
                image_transforms: Optional[List[Callable]] = None, normalize_images: bool = False) -> ImageAndIdDataset:
    if path_to_images is not None and path_to_zarr is not None:
        raise ValueError("can only take one of 'path_to_images' or 'path_to_zarr', but got both")
    if path_to_images is not None:
        return ImagePathDataset(path_to_images, image_transforms=image_transforms, normalize_images=normalize_images)
    if path_to_zarr is not None:
        raise NotImplementedError("yogo_amd: zarr input needs the `zarr` package, which is not part of this build; use --path-to-images")
    raise ValueError("one of 'path_to_images' or 'path_to_zarr' must not be None")
