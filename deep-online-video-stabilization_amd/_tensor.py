"""Torch is plumbing here: device memory, streams.  These helpers hand raw pointers to the C ABI."""
from __future__ import annotations

import torch

from ._lib import StabnetError


def dev_f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """Return `t` as a contiguous float32 CUDA(HIP) tensor; raise if it is not on the GPU."""
    if not isinstance(t, torch.Tensor):
        raise StabnetError("%s: expected a torch.Tensor on the GPU" % name)
    if not t.is_cuda:
        raise StabnetError("%s: tensor is on %s; the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream_ptr(device=None):
    """Current torch stream OF `device` (a tensor's device), not of whichever GPU happens to be current."""
    return torch.cuda.current_stream(device).cuda_stream


def empty(shape, like: torch.Tensor, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)
