"""Generates the committed golden vectors under tests/golden/ (run once, here, from the repo root):

    python tests/golden/make_golden.py

The reference itself cannot be executed (Lua/Torch7 absent), so the vectors come from the CPU oracle, each op
cross-checked against PyTorch-CPU at generation time where PyTorch implements the same maths.  They pin (a) the
oracle against silent regressions and (b) the HIP path on the GPU box, which has neither /root/reference nor any
need to re-derive them.  Fixtures are data only: inputs and expected outputs, fp32 little-endian in .npz.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = dict(nBottleneck=16, nef=4, ngf=4, ndf=4)


def _r(rng, *s):
    return rng.standard_normal(s).astype(np.float32)


def ops():
    rng = np.random.default_rng(20260101)
    out = {}
    # conv 4x4 s2 p1 and the 4x4 -> 1x1 bottleneck form
    for tag, (B, Cin, H, Cout, s, p) in {"conv_s2": (2, 3, 8, 5, 2, 1), "conv_s1": (2, 6, 4, 7, 1, 0)}.items():
        m = O.SpatialConvolution(Cin, Cout, 4, 4, s, s, p, p)
        m.weight[...] = _r(rng, *m.weight.shape) * 0.1
        m.bias[...] = _r(rng, Cout)
        x = _r(rng, B, Cin, H, H)
        y = m.forward(x).copy()
        yt = F.conv2d(torch.from_numpy(x), torch.from_numpy(m.weight), torch.from_numpy(m.bias), stride=s, padding=p).numpy()
        assert np.abs(y - yt).max() < 1e-5
        gy = _r(rng, *y.shape)
        m.backward(x, gy)
        out.update({tag + "_x": x, tag + "_w": m.weight.copy(), tag + "_b": m.bias.copy(), tag + "_y": y, tag + "_gy": gy,
                    tag + "_gx": m.gradInput.copy(), tag + "_gw": m.gradWeight.copy(), tag + "_gb": m.gradBias.copy()})
    for tag, (B, Cin, H, Cout, s, p) in {"full_s2": (2, 6, 4, 3, 2, 1), "full_s1": (2, 7, 1, 4, 1, 0)}.items():
        m = O.SpatialFullConvolution(Cin, Cout, 4, 4, s, s, p, p)
        m.weight[...] = _r(rng, *m.weight.shape) * 0.1
        m.bias[...] = _r(rng, Cout)
        x = _r(rng, B, Cin, H, H)
        y = m.forward(x).copy()
        yt = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(m.weight), torch.from_numpy(m.bias), stride=s, padding=p).numpy()
        assert np.abs(y - yt).max() < 1e-5
        gy = _r(rng, *y.shape)
        m.backward(x, gy)
        out.update({tag + "_x": x, tag + "_w": m.weight.copy(), tag + "_b": m.bias.copy(), tag + "_y": y, tag + "_gy": gy,
                    tag + "_gx": m.gradInput.copy(), tag + "_gw": m.gradWeight.copy(), tag + "_gb": m.gradBias.copy()})
    # batch norm
    bn = O.SpatialBatchNormalization(8)
    bn.weight[...] = 1 + 0.1 * _r(rng, 8)
    bn.bias[...] = 0.1 * _r(rng, 8)
    x = (_r(rng, 3, 8, 4, 4) * 1.5 + 0.3).astype(np.float32)
    y = bn.forward(x).copy()
    rm, rv = torch.zeros(8), torch.ones(8)
    yt = F.batch_norm(torch.from_numpy(x), rm, rv, torch.from_numpy(bn.weight), torch.from_numpy(bn.bias), True, 0.1, 1e-5).numpy()
    assert np.abs(y - yt).max() < 1e-5 and np.abs(bn.running_var - rv.numpy()).max() < 1e-6
    gy = _r(rng, *x.shape)
    bn.backward(x, gy)
    out.update(bn_x=x, bn_gamma=bn.weight.copy(), bn_beta=bn.bias.copy(), bn_y=y, bn_running_mean=bn.running_mean.copy(),
               bn_running_var=bn.running_var.copy(), bn_save_mean=bn.save_mean.copy(), bn_save_invstd=bn.save_std.copy(),
               bn_gy=gy, bn_gx=bn.gradInput.copy(), bn_ggamma=bn.gradWeight.copy(), bn_gbeta=bn.gradBias.copy())
    # criteria
    p = np.concatenate([rng.random(14), [0.0, 1.0]]).astype(np.float32)
    for lab in (0, 1):
        t = np.full(16, float(lab), np.float32)
        out["bce_loss_%d" % lab] = np.float64(O.BCECriterion().forward(p, t))
        out["bce_grad_%d" % lab] = O.BCECriterion().backward(p, t)
    out["bce_x"] = p
    a, b = _r(rng, 2, 3, 8, 8), _r(rng, 2, 3, 8, 8)
    mask = (rng.random(a.shape) > 0.5).astype(np.uint8)
    mm = O.MaskedMSECriterion(0.05)
    mm.setMask(mask)
    out.update(crit_x=a, crit_t=b, crit_mask=mask, mse_loss=np.float64(O.MSECriterion().forward(a, b)),
               mse_grad=O.MSECriterion().backward(a, b), gdl_loss=np.float64(O.GDLCriterion(1).forward(a, b)),
               mmse_loss=np.float64(mm.forward(a, b)), mmse_grad=mm.backward(a, b))
    # adam, three steps
    x, g = _r(rng, 257), (_r(rng, 257) * 1e-3).astype(np.float32)
    st = {"learningRate": 0.002, "beta1": 0.5}
    xs = x.copy()
    for _ in range(3):
        O.adam(lambda _x: (0.0, g), xs, st)
    out.update(adam_x0=x, adam_g=g, adam_x3=xs.copy(), adam_m3=st["m"].copy(), adam_v3=st["v"].copy())
    np.savez(os.path.join(HERE, "ops.npz"), **out)


STRIDE = 37   # fixtures keep every 37th entry of the big vectors plus double-precision sums


def build(kind):
    """Everything is regenerated from seeds (numpy PCG64 streams are stable), so only RESULTS are stored."""
    rng = np.random.default_rng(7 if kind == "center" else 8)
    if kind == "center":
        tr = O.CenterTrainer(dict(TINY, wtl2=0.999, overlapPred=4), rng)
        batches = [(O.synth_center_batch(2, np.random.default_rng(100 + i)),) for i in range(2)]
    else:
        tr = O.VidTrainer(dict(TINY, predLen=2, wtgdl=0.5), rng)
        batches = [O.synth_vid_batch(2, np.random.default_rng(200 + i), 6) for i in range(2)]
    return tr, batches


def summarize(v):
    v = np.asarray(v).reshape(-1)
    return v[::STRIDE].copy(), np.array([v.astype(np.float64).sum(), (v.astype(np.float64) ** 2).sum()])


def iteration(kind):
    tr, batches = build(kind)
    out = {}
    for i, b in enumerate(batches):
        tr.set_batch(*b)
        r = tr.step()
        out["losses%d" % i] = np.array([r["errD"], r["errG"], r["errG_l2"], r.get("errG_gdl") or 0.0], np.float64)
        for name, vec in (("gG", tr.gradParametersG), ("gD", tr.gradParametersD), ("pG", tr.parametersG),
                          ("pD", tr.parametersD), ("fake", tr.netG.output)):
            out["%s%d_sample" % (name, i)], out["%s%d_sums" % (name, i)] = summarize(vec)
    np.savez(os.path.join(HERE, "iter_%s.npz" % kind), **out)


if __name__ == "__main__":
    ops()
    iteration("center")
    iteration("vid")
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
