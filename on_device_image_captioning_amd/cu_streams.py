"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask).

The decoder-step kernels are latency-bound launches of 16..625 small blocks; beside the encoder's GEMM blocks they
wait for CU resources and every CU they land on finishes its GEMM tile late.  Giving the decode
lanes a few compute units of their own and keeping the encode stream off those removes both effects: the pipeline
then behaves like two devices — a large one that encodes, a small one that searches.
"""
import ctypes as C
from typing import Iterable, List, Tuple

import torch

_hip = None


def _lib():
    global _hip
    if _hip is None:
        lib = C.CDLL("libamdhip64.so")
        lib.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        lib.hipExtStreamCreateWithCUMask.restype = C.c_int
        _hip = lib
    return _hip


def masked_stream(device: torch.device, cus: Iterable[int], n_cus: int) -> torch.cuda.Stream:
    """A stream of `device` whose kernels run only on the compute units listed in `cus` (indices < n_cus)."""
    nwords = (n_cus + 31) // 32
    words = (C.c_uint32 * nwords)()
    any_bit = False
    for b in cus:
        if not 0 <= b < n_cus:
            raise ValueError(f"compute unit {b} outside 0..{n_cus - 1}")
        words[b // 32] |= 1 << (b % 32)
        any_bit = True
    if not any_bit:
        raise ValueError("empty compute-unit mask")
    h = C.c_void_p()
    with torch.cuda.device(device):
        rc = _lib().hipExtStreamCreateWithCUMask(C.byref(h), nwords, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(h.value, device=device)


def split_streams(device: torch.device, decode_cus: int, n_encode: int, n_decode: int
                  ) -> Tuple[List[torch.cuda.Stream], List[torch.cuda.Stream]]:
    """(encode streams, decode streams): the decode lanes share compute units 0 .. decode_cus-1 of the mask
    numbering, the encode streams get all the others."""
    n_cus = torch.cuda.get_device_properties(device).multi_processor_count
    if not 0 < decode_cus < n_cus:
        raise ValueError(f"decode_cus must be in 1..{n_cus - 1}")
    import os
    which = os.environ.get("ODIC_CU_MASK_WHICH", "both")       # measurement switch: both | enc | dec
    enc = [masked_stream(device, range(decode_cus, n_cus), n_cus) if which in ("both", "enc")
           else torch.cuda.Stream(device=device) for _ in range(n_encode)]
    dec = [masked_stream(device, range(decode_cus), n_cus) if which in ("both", "dec")
           else torch.cuda.Stream(device=device) for _ in range(n_decode)]
    return enc, dec
