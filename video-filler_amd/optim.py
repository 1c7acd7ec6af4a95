"""optim.adam with the reference's call shape: optim.adam(opfunc, x, state) (train.lua:421-424).

`x` is the flat parameter tensor from net:getParameters(); `opfunc(x)` returns (f(x), df/dx) where df/dx is the
flat gradient tensor.  State (`m`, `v`, device step counter) lives in `state`, as in optim/adam.lua; the update
itself is ONE fused kernel pass (vf_adam_step).  Defaults as optim.adam: beta1 0.9, beta2 0.999, epsilon 1e-8.
"""
import torch

from .backend import bump_param_version, get_backend


def adam(opfunc, x, state):
    fx, dfdx = opfunc(x)
    adam_update(x, dfdx, state)
    return x, [fx]


def adam_init(x, state):
    """The state optim.adam creates on its first call (m, v, t = 0); callable up front so the device step counter
    exists before the first closure (the noise draw of train.lua:319-323 is keyed by it)."""
    if "m" not in state:
        state["t"] = 0
        state["m"] = torch.zeros_like(x)
        state["v"] = torch.zeros_like(x)
        state["t_dev"] = get_backend().zeros(2, dtype=torch.int32)
    return state


def adam_update(x, dfdx, state):
    """The update half of optim.adam (everything after `local fx, dfdx = opfunc(x)`), callable on its own so a
    data-parallel step can put a gradient all-reduce between the closure and the update."""
    B = get_backend()
    lr = state.get("learningRate", 0.001)
    beta1 = state.get("beta1", 0.9)
    beta2 = state.get("beta2", 0.999)
    eps = state.get("epsilon", 1e-8)
    if "m" not in state:
        adam_init(x, state)
    state["t"] += 1          # host mirror of the device counter (informational)
    B.adam_step(x, dfdx, state["m"], state["v"], lr, beta1, beta2, eps, state["t_dev"])
    bump_param_version(x)    # (weight planes split from the old values are stale now: nn.Sequential checks)


def adam_update_split(x, dfdx, state, side_ranges, side):
    """adam_update with the element ranges `side_ranges` ([(lo, hi)], 16-byte aligned) applied on the backend `side`
    (its own stream, ordered after the step-size kernel) and everything else on the current one.  The caller joins
    `side` before anything reads those ranges.  Element for element the same update as adam_update."""
    B = get_backend()
    lr = state.get("learningRate", 0.001)
    beta1 = state.get("beta1", 0.9)
    beta2 = state.get("beta2", 0.999)
    eps = state.get("epsilon", 1e-8)
    adam_init(x, state)
    state["t"] += 1
    m, v, t_dev = state["m"], state["v"], state["t_dev"]
    B.adam_prep(lr, beta1, beta2, t_dev)
    with side.on():
        for lo, hi in side_ranges:
            side.adam_apply(x[lo:hi], dfdx[lo:hi], m[lo:hi], v[lo:hi], beta1, beta2, eps, t_dev)
    pos = 0
    for lo, hi in sorted(side_ranges) + [(x.numel(), x.numel())]:
        if lo > pos:
            B.adam_apply(x[pos:lo], dfdx[pos:lo], m[pos:lo], v[pos:lo], beta1, beta2, eps, t_dev)
        pos = hi
    bump_param_version(x)


def adam_update_fused(x, dfdx, state, net, keep_grad=False, gathered=None, rows=None):
    """adam_update for a net whose bottleneck weight gradients were left to the optimiser (cnet.CNet.set_fused_adam): the plain
    one-pass update everywhere else, vf_wgrad_adam_outer on those slices — their gradient is formed in the matrix-core accumulators
    and consumed there (24 B per weight instead of 32; dfdx receives it only with keep_grad).  Element for element the update of
    adam_update, given the same gradient.  gathered = (buffer, world): data parallel — the buffer holds every rank's packed
    operands (CNet.fused_adam_pack, all-gathered); the fused slices get the gradient of the global batch, the rest of dfdx is
    expected to have been all-reduced.  rows = (rank, ranks) with gathered: the fused slices are updated in this rank's rows only
    (the caller all-gathers them afterwards)."""
    B = get_backend()
    lr = state.get("learningRate", 0.001)
    beta1 = state.get("beta1", 0.9)
    beta2 = state.get("beta2", 0.999)
    eps = state.get("epsilon", 1e-8)
    adam_init(x, state)
    state["t"] += 1
    m, v, t_dev = state["m"], state["v"], state["t_dev"]
    B.adam_prep(lr, beta1, beta2, t_dev)
    pos, plain = 0, []
    for lo, hi in sorted(net.fused_adam_ranges()) + [(x.numel(), x.numel())]:
        if lo > pos:
            plain.append((pos, lo))
        pos = hi
    if hasattr(B, "adam_apply_ranges") and len(plain) <= 8 and all(a % 4 == 0 and b % 4 == 0 for a, b in plain):
        B.adam_apply_ranges(x, dfdx, m, v, plain, beta1, beta2, eps, t_dev)      # everything outside the fused slices: one launch
    else:
        for a, b in plain:
            B.adam_apply(x[a:b], dfdx[a:b], m[a:b], v[a:b], beta1, beta2, eps, t_dev)
    if gathered is not None:
        net.adam_fused_gathered(gathered[0], gathered[1], m, v, beta1, beta2, eps, t_dev, keep_grad, rows=rows)
    else:
        net.adam_fused(m, v, beta1, beta2, eps, t_dev, keep_grad)
    bump_param_version(x)
