"""Caller-side image preprocessing (reference utils/image_utils.py:5-23).

torchvision is not installed in this image, so the three transforms are spelled out with PIL and
torch: Resize((S,S)) on a PIL image is `Image.resize((S,S), BILINEAR)`, ToTensor is HWC uint8 → CHW
float /255, Normalize is the ImageNet mean/std.  Non-RGB files become an all-black RGB canvas, as
the reference does (it calls PIL_Image.new, image_utils.py:18-19).
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def preprocess_image(image_path: str, img_size: int) -> torch.Tensor:
    pil = Image.open(image_path)
    if pil.mode != "RGB":
        pil = Image.new("RGB", pil.size)
    pil = pil.resize((img_size, img_size), Image.BILINEAR)
    chw = torch.from_numpy(np.asarray(pil, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
    mean = torch.tensor(_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(_STD, dtype=torch.float32).view(3, 1, 1)
    return ((chw - mean) / std).unsqueeze(0)
