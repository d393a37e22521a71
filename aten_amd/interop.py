"""torch <-> raw device pointer plumbing (torch is used for device memory and torch.distributed only)."""
import torch


class _Ptr:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_ptr(ptr, shape, device="cuda:0"):
    """Zero-copy float32 tensor over device memory owned by the C-ABI library."""
    return torch.as_tensor(_Ptr(ptr, shape), device=device)
