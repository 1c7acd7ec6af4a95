"""Shared helpers for the parity tests: layout conversion between the oracle's NCHW numpy arrays and the
backend's channels-last device tensors, and error metrics."""
import numpy as np
import torch


def to_dev(a, backend):
    """NCHW numpy (or 1-D) -> device tensor; 4-D tensors become logical NCHW / physical NHWC."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    t = t.to(backend.device)
    if t.dim() == 4:
        t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return t


def to_np(t):
    return t.detach().cpu().contiguous().numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, "%s: max-norm relative error %.3e > %.1e" % (what, e, tol)
