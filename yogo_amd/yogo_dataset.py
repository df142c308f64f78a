"""``ObjectDetectionDataset`` -- a folder of images + a folder of YOLO label files (yogo/data/yogo_dataset.py:136-301), and the
image readers of yogo/data/utils.py:12-46.

MI355X-first split of the work: DataLoader workers only do what must happen on the host -- find the files, decode the image
(PIL; torchvision.io is not needed) and PARSE the label file into rows.  A sample is ``(uint8 image [C, H, W], label rows
[N, 5] = (class, xc, yc, w, h))``; the ``[6, Sy, Sx]`` label tensors of a whole batch are rasterised on the device by one HIP
launch (``yogo_labels_rasterize``) inside ``yogo_amd.yogo_dataloader`` -- the reference rasterises per image in Python inside
the workers (yogo_dataset.py:24-46).  ``dataset.label_tensor(i)`` gives the reference's per-sample tensor (on the device).
"""
from __future__ import annotations

import json
import time
from functools import partial
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from yogo_amd.data import LABEL_TENSOR_PRED_DIM_SIZE, label_file_to_tensor, load_labels  # noqa: F401

IMG_EXTENSIONS = ("png", "jpg", "jpeg", "tif")


def read_image(img_path: Union[str, Path], rgb: bool = False) -> torch.Tensor:
    """uint8 [1, H, W] (gray) or [3, H, W] (rgb) -- yogo/data/utils.py:12-21 (torchvision.io.read_image there, PIL here)"""
    from PIL import Image

    with Image.open(str(img_path)) as im:
        im = im.convert("RGB" if rgb else "L")
        arr = np.asarray(im, dtype=np.uint8)
    t = torch.from_numpy(arr.copy())
    return t.permute(2, 0, 1).contiguous() if rgb else t[None]


def read_image_robust(img_path: Union[str, Path], retries: int = 3, min_duration: float = 0.1, rgb: bool = False) -> Optional[torch.Tensor]:
    """read_image with retries and exponential back-off; None when the file stays unreadable (yogo/data/utils.py:24-46)"""
    for i in range(retries):
        try:
            return read_image(img_path, rgb=rgb)
        except Exception as e:   # a truncated / half-written file: try again, like the reference
            if i == retries - 1:
                import warnings

                warnings.warn(f"could not read {img_path} after {retries} tries: {e}")
                return None
            time.sleep(min_duration * 2 ** i)
    return None


def resize_image(img: torch.Tensor, image_hw: Tuple[int, int]) -> torch.Tensor:
    """torchvision.transforms.Resize(image_hw, antialias=True) on a uint8 [C, H, W] tensor: identity when the size matches
    (the normal case), anti-aliased bilinear otherwise"""
    if tuple(img.shape[-2:]) == tuple(image_hw):
        return img
    out = torch.nn.functional.interpolate(img[None].float(), size=tuple(image_hw), mode="bilinear", antialias=True, align_corners=False)[0]
    return out.round().clamp(0, 255).to(torch.uint8)


class ObjectDetectionDataset(torch.utils.data.Dataset):
    def __init__(
        self,
        image_folder_path: Union[str, Path],
        label_folder_path: Union[str, Path],
        Sx: int,
        Sy: int,
        classes: List[str],
        image_hw: Tuple[int, int] = (772, 1032),
        rgb: bool = False,
        normalize_images: bool = False,
        extensions: Tuple[str, ...] = IMG_EXTENSIONS,
        is_valid_file: Optional[Callable[[str], bool]] = None,
    ):
        self.classes = classes
        self.image_folder_path = Path(image_folder_path)
        self.label_folder_path = Path(label_folder_path)
        self.loader = partial(read_image_robust, retries=3, min_duration=0.1, rgb=rgb)
        self.image_hw = tuple(image_hw)
        self.normalize_images = normalize_images
        self.notes_data: Optional[Dict[str, Any]] = None
        image_paths, label_paths = self.make_dataset(Sx, Sy, is_valid_file=is_valid_file, extensions=extensions)
        self.Sx, self.Sy = Sx, Sy
        # numpy string arrays, not lists: DataLoader workers would copy lists page by page (yogo_dataset.py:163-167)
        self._image_paths = np.array(image_paths).astype(np.str_)
        self._label_paths = np.array(label_paths).astype(np.str_)

    def make_dataset(self, Sx: int, Sy: int, extensions=None, is_valid_file=None) -> Tuple[List[str], List[str]]:
        """pairs every label file (*.txt, hidden files skipped) with its .png / .jpg image (yogo_dataset.py:182-262)"""
        both_none = extensions is None and is_valid_file is None
        both_something = extensions is not None and is_valid_file is not None
        if both_none or both_something:
            raise ValueError("Both extensions and is_valid_file cannot be None or not None at the same time")
        if extensions is not None:
            exts = tuple(e.lower() if e.startswith(".") else "." + e.lower() for e in ((extensions,) if isinstance(extensions, str) else extensions))

            def is_valid_file(x: str) -> bool:   # noqa: F811
                return x.lower().endswith(exts)

        if (self.label_folder_path.parent / "notes.json").exists():
            with open(str(self.label_folder_path.parent / "notes.json"), "r") as notes:
                self.notes_data = json.load(notes)
        image_paths: List[str] = []
        label_paths: List[str] = []
        missing_images: List[str] = []
        for label_file_path in sorted(self.label_folder_path.glob("*.txt")):
            if label_file_path.name.startswith("."):
                continue
            candidates = [self.image_folder_path / label_file_path.with_suffix(sfx).name for sfx in (".png", ".jpg")]
            found = next((ip for ip in candidates if ip.exists() and is_valid_file(str(ip))), None)
            if found is not None:
                image_paths.append(str(found))
                label_paths.append(str(label_file_path))
            else:
                missing_images.append(str(label_file_path))
                if len(image_paths) > 10:
                    break
        if len(missing_images) > 0:
            subset, msg = (missing_images, " ") if len(missing_images) < 5 else (missing_images[:3], " a sample of ")
            raise FileNotFoundError(
                f"{'at least ' if len(missing_images) == 10 else ' '}{len(missing_images)}"
                f"images not found in {self.image_folder_path}; ({len(image_paths)} images were found). Here's{msg}the list:\n{subset}")
        return image_paths, label_paths

    def label_rows(self, index: int) -> torch.Tensor:
        """[N, 5] rows (class, xc, yc, w, h) of sample `index` (the area filter of load_labels applied)"""
        path = str(self._label_paths[index])
        try:
            rows = load_labels(path, classes=self.classes, notes_data=self.notes_data)
        except Exception as e:
            raise RuntimeError(f"exception from {path}") from e
        return torch.tensor(rows, dtype=torch.float32).reshape(-1, 5)

    def label_tensor(self, index: int, device=None) -> torch.Tensor:
        """the reference's per-sample label tensor [6, Sy, Sx] (rasterised on the device)"""
        return label_file_to_tensor(str(self._label_paths[index]), self.Sx, self.Sy, self.classes, self.notes_data, device=device)

    def __getitem__(self, index: int) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        maybe_image = self.loader(str(self._image_paths[index]))
        if maybe_image is None:
            return None
        image = resize_image(maybe_image, self.image_hw)
        if self.normalize_images:
            image = image / 255
        return image, self.label_rows(index)

    def __len__(self) -> int:
        return len(self._image_paths)

    def calc_class_counts(self) -> torch.Tensor:
        counts = torch.zeros(len(self.classes), dtype=torch.long)
        for label_path in self._label_paths:
            for label in load_labels(str(label_path), classes=self.classes, notes_data=self.notes_data):
                counts[int(label[0])] += 1
        return counts
