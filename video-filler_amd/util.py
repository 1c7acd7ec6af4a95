"""util.lua counterpart: the backend-swap hook and checkpoint save/load.

util.cudnn(net) (util.lua:108-131) is where the reference swaps nn.SpatialConvolution for the GPU backend
under opt.gpu > 0 (train.lua:245-258).  In this package the nn mirror classes ARE the gfx950-backed modules,
so `util.hip(net)` only validates that the library is loaded and returns the net — drivers keep the call.
Checkpoints: `*.t7` = Torch7's own container, the object tree util.save writes (util.lua:72-97; t7.py), so a
checkpoint moves between this backend and a Torch7 install in either direction; any other extension = a flat fp32
dump in the reference's parameter order plus BN running statistics, stored as .npz.  Like util.save,
neither gradients nor Adam state are saved.
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
    if str(filename).endswith(".t7"):
        from . import t7
        t7.save(filename, t7.net_to_t7(net))
        return
    if net._flat is None:
        net.getParameters()
    arrays = {"parameters": net.reference_flat().cpu().numpy()}
    for i, m in enumerate(net.leaves()):
        if isinstance(m, SpatialBatchNormalization):
            arrays["bn%d_running_mean" % i] = m.running_mean.cpu().numpy()
            arrays["bn%d_running_var" % i] = m.running_var.cpu().numpy()
    with open(filename, "wb") as fh:
        np.savez(fh, **arrays)


def load(filename, net=None, gpu=1):
    """`.t7`: util.load(filename, gpu) (util.lua:99-105) — builds the net the file describes and returns it (with `net`
    given, copies into that net instead).  Otherwise: fill an already-constructed net from a `.npz` written by `save`."""
    if str(filename).endswith(".t7"):
        from . import t7
        loaded = t7.net_from_t7(t7.load(filename))
        if net is None:
            return loaded
        if net._flat is None:
            net.getParameters()
        loaded.getParameters()
        net.load_reference_flat(loaded.reference_flat())
        for a, b in zip(net.leaves(), loaded.leaves()):
            if isinstance(a, SpatialBatchNormalization):
                a.running_mean.copy_(b.running_mean)
                a.running_var.copy_(b.running_var)
        return net
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
