"""F2 — device-side preprocessing against PIL (an independent implementation, not the reference):
the coefficient tables on CPU, the HIP resize + normalise pipeline on the GPU."""
import numpy as np
import pytest
import torch
from PIL import Image

from on_device_image_captioning_amd import image_utils as IU

SIZES = [(480, 640), (589, 880), (100, 77), (384, 384), (1000, 333), (200, 1500), (31, 2000)]


def _pil_resize(img, S):
    return np.asarray(Image.fromarray(img).resize((S, S), Image.BILINEAR))


def _numpy_resample(img, S):
    """Two integer passes with the tables of pil_bilinear_coeffs (what the kernels execute)."""
    H, W, _ = img.shape
    pb = 22
    bx, kx, _ = IU.pil_bilinear_coeffs(W, S)
    by, ky, _ = IU.pil_bilinear_coeffs(H, S)
    tmp = np.zeros((H, S, 3), np.uint8)
    for xx in range(S):
        x0, n = bx[xx]
        acc = (img[:, x0:x0 + n].astype(np.int64) * kx[xx, :n][None, :, None]).sum(1) + (1 << (pb - 1))
        tmp[:, xx] = np.clip(acc >> pb, 0, 255)
    out = np.zeros((S, S, 3), np.uint8)
    for yy in range(S):
        y0, n = by[yy]
        acc = (tmp[y0:y0 + n].astype(np.int64) * ky[yy, :n][:, None, None]).sum(0) + (1 << (pb - 1))
        out[yy] = np.clip(acc >> pb, 0, 255)
    return out


@pytest.mark.parametrize("H,W", SIZES[:5])
def test_coefficient_tables_reproduce_pil_bilinear(H, W):
    img = np.random.default_rng(H * 7 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(_numpy_resample(img, 96), _pil_resize(img, 96))


@pytest.mark.gpu
def test_device_preprocessor_is_bit_exact_with_pil_and_torch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for H, W in SIZES]
    pre = IU.DevicePreprocessor(384, "cuda:0")
    got = pre(imgs).cpu()
    mean = torch.tensor(IU._MEAN).view(3, 1, 1)
    std = torch.tensor(IU._STD).view(3, 1, 1)
    for i, img in enumerate(imgs):
        chw = torch.from_numpy(_pil_resize(img, 384).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
        want = (chw - mean) / std
        assert torch.equal(got[i], want), f"image {i} {img.shape}: {(got[i] - want).abs().max().item()}"
    again = pre(imgs[::-1]).cpu()                      # buffers reused in another order
    assert torch.equal(again, got.flip(0))
