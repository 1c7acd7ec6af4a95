"""util.lua counterpart: the backend-swap hook and checkpoint save/load.

util.cudnn(net) (util.lua:108-131) is where the reference swaps nn.SpatialConvolution for the GPU backend
under opt.gpu > 0 (train.lua:245-258).  In this package the nn mirror classes ARE the gfx950-backed modules,
so `util.hip(net)` only validates that the library is loaded and returns the net — drivers keep the call.
Checkpoints: a flat fp32 dump in the reference's parameter order plus BN running statistics,
stored as .npz; Torch7's .t7 container is a "next" row.  Like util.save (util.lua:72-97), neither
gradients nor Adam state are saved.
"""
import numpy as np
import torch

from . import _lib
from .nn import SpatialBatchNormalization


def hip(net):
    _lib.load()
    return net


cudnn = hip  # drivers written against the reference call util.cudnn(net)


def save(filename, net, gpu=1):
    if net._flat is None:
        net.getParameters()
    arrays = {"parameters": net.reference_flat().cpu().numpy()}
    for i, m in enumerate(net.leaves()):
        if isinstance(m, SpatialBatchNormalization):
            arrays["bn%d_running_mean" % i] = m.running_mean.cpu().numpy()
            arrays["bn%d_running_var" % i] = m.running_var.cpu().numpy()
    with open(filename, "wb") as fh:
        np.savez(fh, **arrays)


def load(filename, net, gpu=1):
    """Fill an already-constructed net (same topology) from a checkpoint written by `save`."""
    if net._flat is None:
        net.getParameters()
    z = np.load(filename)
    dev = net._flat[0].device
    net.load_reference_flat(torch.from_numpy(z["parameters"]).to(dev))
    for i, m in enumerate(net.leaves()):
        if isinstance(m, SpatialBatchNormalization):
            m.running_mean.copy_(torch.from_numpy(z["bn%d_running_mean" % i]).to(dev))
            m.running_var.copy_(torch.from_numpy(z["bn%d_running_var" % i]).to(dev))
    net.apply(lambda m: None)
    return net
