"""Generates the tiny REAL fake dataset of this build (BASELINE configs[0]): the reference's tests/fake-data layout
(data/images{1,2,3}/img_{1,2,3}.png + data/labels{1,2,3}/img_{1,2,3}.txt, 4 classes you / only / glance / once), but with
non-empty files -- the reference's own 18 data files are 0-byte placeholders, so its `yogo train` cannot read them (SURVEY.md
section 4).  64 x 96 grayscale PNGs with a few bright ellipses, YOLO label rows "class xc yc w h" (normalised).  Deterministic:
    python tests/fake-data/make_fake_data.py"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
H, W = 64, 96


def main():
    rng = np.random.RandomState(20240807)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in (1, 2, 3):
        os.makedirs(os.path.join(HERE, "data", f"images{k}"), exist_ok=True)
        os.makedirs(os.path.join(HERE, "data", f"labels{k}"), exist_ok=True)
        for i in (1, 2, 3):
            img = rng.randint(20, 60, size=(H, W)).astype(np.float32)
            rows = []
            for _ in range(rng.randint(2, 5)):
                cls = rng.randint(0, 4)
                w, h = rng.uniform(0.12, 0.2), rng.uniform(0.15, 0.25)
                xc, yc = rng.uniform(w / 2 + 0.02, 1 - w / 2 - 0.02), rng.uniform(h / 2 + 0.02, 1 - h / 2 - 0.02)
                m = ((xx - xc * W) / (w * W / 2)) ** 2 + ((yy - yc * H) / (h * H / 2)) ** 2 <= 1
                img[m] = 120 + 30 * cls + rng.randint(0, 10)
                rows.append(f"{cls} {xc:.6f} {yc:.6f} {w:.6f} {h:.6f}")
            Image.fromarray(img.clip(0, 255).astype(np.uint8), mode="L").save(os.path.join(HERE, "data", f"images{k}", f"img_{i}.png"))
            with open(os.path.join(HERE, "data", f"labels{k}", f"img_{i}.txt"), "w") as f:
                f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
