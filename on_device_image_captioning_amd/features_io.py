"""HDF5-free store for precomputed backbone features (BASELINE config 2's input).

The reference keeps one HDF5 dataset `<img_id>_features` of shape (n_tokens <= 144, 1536) fp32 per image
(written by data_generator.py:95-114, read by data/coco_dataloader.py:437-478,510-520 through h5py, which is not
available here).  This module stores the same records in ONE flat file that is memory-mapped on read:

    [ magic "ODICFEA1" | feat_dim u32 | pad u32 ]  records: fp32 [n_i, feat_dim] back to back, 64-byte aligned
    [ index: count x (img_id i64, byte_offset u64, n_tokens u32, pad u32) | count u64 | index_offset u64 | magic ]

`FeatureStore.get_PADDED_bboxes_batch_by_id(img_id_list)` has the name, argument and result of the reference
loader's method: (features [B, S_max, F] zero-padded as torch's pad_sequence does, list of trailing pad counts) —
the operands of `model(enc_x=..., enc_x_num_pads=...)`.  With `device=` the batch is assembled in one of two
pinned host buffers and uploaded without blocking, so the copy of batch i+1 overlaps the kernels of batch i.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

MAGIC = b"ODICFEA1"
_HDR = struct.Struct("<8sII")
_REC = np.dtype([("img_id", "<i8"), ("offset", "<u8"), ("n", "<u4"), ("pad", "<u4")])
_TAIL = struct.Struct("<QQ8s")
_ALIGN = 64


class FeatureStoreWriter:
    def __init__(self, path: str, feat_dim: int):
        self.path, self.feat_dim = path, int(feat_dim)
        self._f = open(path, "wb")
        self._f.write(_HDR.pack(MAGIC, self.feat_dim, 0))
        self._index: List[Tuple[int, int, int]] = []
        self._ids = set()

    def append(self, img_id: int, features) -> None:
        """features: array-like [n_tokens, feat_dim]; stored as fp32 (the dtype data_generator.py writes)."""
        a = np.ascontiguousarray(np.asarray(features.detach().cpu() if isinstance(features, torch.Tensor) else features,
                                            dtype=np.float32))
        if a.ndim != 2 or a.shape[1] != self.feat_dim:
            raise ValueError(f"expected [n, {self.feat_dim}] features, got {a.shape}")
        if int(img_id) in self._ids:
            raise ValueError(f"image id {img_id} stored twice")
        pos = self._f.tell()
        padn = (-pos) % _ALIGN
        self._f.write(b"\0" * padn)
        self._index.append((int(img_id), pos + padn, a.shape[0]))
        self._ids.add(int(img_id))
        self._f.write(a.tobytes())

    def close(self) -> None:
        if self._f is None:
            return
        pos = self._f.tell()
        padn = (-pos) % _ALIGN
        self._f.write(b"\0" * padn)
        idx = np.zeros(len(self._index), dtype=_REC)
        for i, (iid, off, n) in enumerate(self._index):
            idx[i] = (iid, off, n, 0)
        self._f.write(idx.tobytes())
        self._f.write(_TAIL.pack(len(self._index), pos + padn, MAGIC))
        self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class FeatureStore:
    def __init__(self, path: str, device: Optional[torch.device] = None, max_batch: int = 48, max_tokens: int = 144):
        size = os.path.getsize(path)
        if size < _HDR.size + _TAIL.size:
            raise ValueError(f"{path}: not a feature store (too small)")
        with open(path, "rb") as f:
            magic, self.feat_dim, _ = _HDR.unpack(f.read(_HDR.size))
            f.seek(size - _TAIL.size)
            count, index_off, magic2 = _TAIL.unpack(f.read(_TAIL.size))
        if magic != MAGIC or magic2 != MAGIC or index_off + count * _REC.itemsize + _TAIL.size != size:
            raise ValueError(f"{path}: not a feature store (bad magic / truncated)")
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        idx = np.frombuffer(self._mm, dtype=_REC, count=count, offset=index_off)
        self._where: Dict[int, Tuple[int, int]] = {int(r["img_id"]): (int(r["offset"]), int(r["n"])) for r in idx}
        self.img_ids = [int(r["img_id"]) for r in idx]
        self.device = torch.device(device) if device is not None else None
        self._pinned = None
        if self.device is not None and self.device.type == "cuda":
            self._pinned = [torch.zeros(max_batch, max_tokens, self.feat_dim).pin_memory() for _ in range(2)]
            self._ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._turn = 0

    def __len__(self) -> int:
        return len(self.img_ids)

    def __contains__(self, img_id) -> bool:
        return int(img_id) in self._where

    def get_features(self, img_id: int) -> torch.Tensor:
        """fp32 [n_tokens, feat_dim] of one image (a copy; the file stays memory-mapped)."""
        off, n = self._where[int(img_id)]
        a = np.frombuffer(self._mm, dtype="<f4", count=n * self.feat_dim, offset=off).reshape(n, self.feat_dim)
        return torch.from_numpy(np.array(a))

    def get_PADDED_bboxes_batch_by_id(self, img_id_list: Sequence[int], verbose: bool = False
                                      ) -> Tuple[torch.Tensor, List[int]]:
        """data/coco_dataloader.py:437-478: (batch [B, S_max, F] zero-padded at the end of every sample, trailing pad
        count per sample); on `self.device` if one was given."""
        spans = [self._where[int(i)] for i in img_id_list]
        B, smax = len(spans), max(n for _, n in spans)
        pads = [smax - n for _, n in spans]
        use_pinned = self._pinned is not None and B <= self._pinned[0].shape[0] and smax <= self._pinned[0].shape[1]
        if use_pinned:
            self._ev[self._turn].synchronize()                   # the upload that last used this buffer is done
            # a CONTIGUOUS [B, smax, F] region at the head of the pinned slab: a strided view ([:B, :smax] of
            # [max_batch, max_tokens, F]) would be uploaded through a pageable temporary, i.e. synchronously, and the
            # event below would guard a buffer the DMA never read
            host = self._pinned[self._turn].view(-1)[:B * smax * self.feat_dim].view(B, smax, self.feat_dim)
        else:
            host = torch.empty(B, smax, self.feat_dim)
        dst = host.numpy()                                       # shares memory; np.copyto reads the read-only memmap
        for b, (off, n) in enumerate(spans):
            src = np.frombuffer(self._mm, dtype="<f4", count=n * self.feat_dim, offset=off).reshape(n, self.feat_dim)
            np.copyto(dst[b, :n], src)
            if n < smax:
                dst[b, n:] = 0.0
        if self.device is None:
            return host.clone() if use_pinned else host, pads
        out = host.to(self.device, non_blocking=True)
        if use_pinned:
            self._ev[self._turn].record()
            self._turn ^= 1
        return out, pads


def dump_features(store: FeatureStoreWriter, model, images: torch.Tensor, img_ids: Iterable[int]) -> None:
    """The feature dump of data_generator.py:95-114 on the HIP backbone: `model` is an End_ExpansionNet_v2 on a
    GPU; every image's Swin output [144, 1536] goes into the store under its id."""
    swin, _ = model._engines()
    feats = swin.forward(images.to(swin.device, torch.float32))
    for iid, f in zip(img_ids, feats.cpu()):
        store.append(int(iid), f)


def convert_hdf5(hdf5_path: str, out_path: str, feat_dim: int = 1536) -> int:
    """One-off conversion of a reference `features.hdf5` (needs h5py, which this image does not ship)."""
    try:
        import h5py  # type: ignore
    except ImportError as e:                                     # pragma: no cover
        raise RuntimeError("h5py is not installed: convert the file where it is, then ship the flat store") from e
    n = 0
    with h5py.File(hdf5_path, "r") as h, FeatureStoreWriter(out_path, feat_dim) as w:
        for key in h.keys():
            if key.endswith("_features"):
                w.append(int(key[:-len("_features")]), h[key][()])
                n += 1
    return n
