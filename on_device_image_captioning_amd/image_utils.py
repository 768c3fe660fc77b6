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


# =================================================================================================
# Device-side pipeline: resize + ToTensor + Normalize on the GPU, bit-exact with the
# PIL / torch path above; pinned double-buffered uploads so the copy of image i+1 overlaps the
# kernels of image i.  JPEG decoding stays on the host (PIL) — there is no decoder to bind to.
# =================================================================================================
_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Tap windows and fixed-point weights of PIL's BILINEAR resampler for one axis, computed exactly as
    libImaging/Resample.c::precompute_coeffs + normalize_coeffs_8bpc do (double arithmetic, taps summed in
    order): → (bounds int32 [out, 2] = first tap / tap count, coeffs int32 [out, ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                              # bilinear filter support = 1
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation, values >= -0.5
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    ss = 1.0 / filterscale
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w = np.where(taps < n[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for i in range(ksize):                                                    # same summation order as the C loop
        ww = ww + w[:, i]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)
    fixed = np.where(taps < n[:, None], fixed, 0).astype(np.int32)
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, fixed, ksize


class DevicePreprocessor:
    """Batched `preprocess_image` on the GPU: host-decoded RGB arrays → fp32 [B, 3, S, S].

        pre = DevicePreprocessor(384, device)
        batch = pre([np.asarray(PIL.Image.open(f).convert_or_black()) ...])     # or pre.from_files(paths)
    """

    def __init__(self, img_size: int, device, max_pixels: int = 4608 * 3456):
        from . import _hip
        self._hip = _hip
        self.lib = _hip.load()
        self.S, self.device = img_size, torch.device(device)
        self.max_bytes = max_pixels * 3
        self.host = [torch.empty(self.max_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(self.max_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.tmp = [torch.empty(0, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.ev = [torch.cuda.Event() for _ in range(2)]
        self.stream = torch.cuda.Stream(device=self.device)
        self._coef_cache = {}
        import ctypes
        self._mean = (ctypes.c_float * 3)(*_MEAN)
        self._std = (ctypes.c_float * 3)(*_STD)

    def _coeffs(self, n: int):
        c = self._coef_cache.get(n)
        if c is None:
            b, k, ks = pil_bilinear_coeffs(n, self.S)
            c = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
            self._coef_cache[n] = c
        return c

    def __call__(self, images) -> torch.Tensor:
        """images: sequence of HWC uint8 RGB numpy arrays (any sizes) → normalised fp32 [B,3,S,S] on the device,
        ordered after the work on the CURRENT stream."""
        S = self.S
        out = torch.empty(len(images), 3, S, S, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for i, img in enumerate(images):
                img = np.ascontiguousarray(img, dtype=np.uint8)
                if img.ndim != 3 or img.shape[2] != 3:
                    raise RuntimeError("DevicePreprocessor wants HWC RGB uint8 arrays")
                H, W, _ = img.shape
                nbytes = H * W * 3
                if nbytes > self.max_bytes:
                    raise RuntimeError(f"image {H}x{W} exceeds the staging buffers ({self.max_bytes} bytes)")
                slot = i & 1
                self.ev[slot].synchronize()                                   # the kernels that read this slot are done
                self.host[slot][:nbytes].copy_(torch.from_numpy(img).reshape(-1))
                self.dev[slot][:nbytes].copy_(self.host[slot][:nbytes], non_blocking=True)
                if self.tmp[slot].numel() < H * S * 3:
                    self.tmp[slot] = torch.empty(H * S * 3, dtype=torch.uint8, device=self.device)
                bx, kx, ksx = self._coeffs(W)
                by, ky, ksy = self._coeffs(H)
                self._hip.check(self.lib.odic_resize_bilinear_normalize(
                    self.dev[slot].data_ptr(), H, W, 3 * W, bx.data_ptr(), kx.data_ptr(), ksx, by.data_ptr(),
                    ky.data_ptr(), ksy, self.tmp[slot].data_ptr(), out[i].data_ptr(), S, self._mean, self._std,
                    self.stream.cuda_stream), "odic_resize_bilinear_normalize")
                self.ev[slot].record(self.stream)
        torch.cuda.current_stream().wait_stream(self.stream)
        return out

    def from_files(self, paths) -> torch.Tensor:
        """Host JPEG/PNG decode (PIL; non-RGB files become a black canvas as in the reference) + device pipeline."""
        imgs = []
        for p in paths:
            pil = Image.open(p)
            if pil.mode != "RGB":
                pil = Image.new("RGB", pil.size)
            imgs.append(np.asarray(pil, dtype=np.uint8))
        return self(imgs)
