"""The steps between render() and the image file (src/main.cpp:232-240, 272-280) on the GPU.

    postprocess(hdr)             (h, w, 3) float32 linear HDR  ->  (h, w, 3) uint8
    postprocess_device(...)      the same on device tensors, returning the intermediate frames too
    write_ppm(path, rgb8)        the reference's P6 file

No arithmetic happens here: both kernels live in librt_amd.so (csrc/rt_post.hip)."""
import ctypes

import numpy as np

from . import _lib


def postprocess(hdr, device=0):
    a = np.ascontiguousarray(hdr, np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("hdr must have shape (height, width, 3)")
    out = np.empty(a.shape, np.uint8)
    _lib.check(_lib.lib().rt_postprocess(ctypes.c_int(device), _lib.ptr(a), ctypes.c_uint32(a.shape[1]),
                                         ctypes.c_uint32(a.shape[0]), _lib.ptr(out)))
    return out


def postprocess_device(hdr, stream=None):
    """hdr: CUDA float32 tensor (h, w, 3).  Returns (srgb, median, rgb8) tensors: the frame before the
    median filter, after it, and the 8-bit image."""
    import torch
    if not (hdr.is_cuda and hdr.dtype == torch.float32 and hdr.is_contiguous() and hdr.dim() == 3 and hdr.shape[2] == 3):
        raise ValueError("hdr must be a contiguous CUDA float32 tensor of shape (height, width, 3)")
    srgb = torch.empty_like(hdr)
    median = torch.empty_like(hdr)
    rgb8 = torch.empty(hdr.shape, dtype=torch.uint8, device=hdr.device)
    st = stream if stream is not None else torch.cuda.current_stream(hdr.device).cuda_stream
    _lib.check(_lib.lib().rt_postprocess_device(ctypes.c_void_p(hdr.data_ptr()), ctypes.c_uint32(hdr.shape[1]),
                                                ctypes.c_uint32(hdr.shape[0]), ctypes.c_void_p(srgb.data_ptr()),
                                                ctypes.c_void_p(median.data_ptr()), ctypes.c_void_p(rgb8.data_ptr()),
                                                ctypes.c_void_p(st)))
    return srgb, median, rgb8


def write_ppm(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("rgb8 must have shape (height, width, 3)")
    if _lib.lib().rt_write_ppm(str(path).encode(), _lib.ptr(a), ctypes.c_uint32(a.shape[1]), ctypes.c_uint32(a.shape[0])) != 0:
        raise OSError(f"cannot write {path}")
