"""`.t7` checkpoints on the GPU side: nets TRAINED by the HIP path go through util.save /
util.load (util.lua:72-105) and come back as the same function — the round trip test_t7.py pins on the CPU, here behind the
kernels that produce the parameters and the running statistics the file carries — and a checkpoint typed the way the reference's
GPU runs leave them in memory (cudnn.SpatialConvolution, util.lua:33-50 writes those back as nn.*) loads into the same net."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flat_in_a11_order(tree):
    """SURVEY A.11 straight from the file's object tree: depth-first, per module {weight, bias}, each tensor in its native
    (NCHW) order — independent of this package's loader."""
    parts = []

    def walk(o):
        f = o.fields
        if "modules" in f:
            mods = f["modules"]
            for k in sorted(k for k in mods if isinstance(k, (int, float))):
                walk(mods[k])
            return
        if "weight" in f:
            parts.append(np.asarray(f["weight"], np.float32).reshape(-1))
            parts.append(np.asarray(f["bias"], np.float32).reshape(-1))

    walk(tree)
    return np.concatenate(parts)


@pytest.mark.parametrize("kind", ["center", "vid"])
def test_hip_trained_nets_through_a_t7_checkpoint(kind, tmp_path, oracle, hipb, host):
    """two training iterations on the HIP path (train.lua:421-424 / train_vid_weighted.lua:548-551) -> util.save(netG), util.save(netD)
    -> util.load into FRESH nets (test_vid.lua:47-48): the same module list, the flat parameter vector in the reference's order
    (A.11) bit for bit, BatchNorm running statistics preserved, and the evaluate-mode forward of the reloaded generator equal to the
    trained one's bit for bit; then loaded INTO the nets of a second trainer (the drivers' resume path, train_vid_weighted.lua:242-257),
    whose next training iteration equals the first trainer's."""
    from video_filler_amd import nn, t7, util
    from video_filler_amd.trainers import CenterTrainer, VidTrainer
    if kind == "center":
        opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4)
        mk = lambda: CenterTrainer(opt, seed=3)
        batch = lambda it: (torch.from_numpy(oracle.synth_center_batch(3, np.random.default_rng(80 + it))),)
    else:
        opt = dict(predLen=2, nBottleneck=64)
        mk = lambda: VidTrainer(opt, seed=3)
        batch = lambda it: tuple(torch.from_numpy(a) for a in oracle.synth_vid_batch(4, np.random.default_rng(80 + it), 6, 6))
    tr = mk()
    assert tr.host == host
    for it in range(2):
        tr.set_batch(*batch(it))
        tr.step()
    tr.flush()
    paths = {}
    for name, net in (("netG", tr.netG), ("netD", tr.netD)):
        paths[name] = str(tmp_path / ("%s_%s.t7" % (kind, name)))
        util.save(paths[name], net)
        tree = t7.load(paths[name])
        assert tree.cls == "nn.Sequential"
        assert np.array_equal(_flat_in_a11_order(tree), net.reference_flat().cpu().numpy()), "flat order A.11"
        back = util.load(paths[name])
        assert [m.type_name() for m in back.leaves()] == [m.type_name() for m in net.leaves()]
        back.getParameters()
        assert torch.equal(back.reference_flat(), net.reference_flat())
        moved = 0
        for a, b in zip(net.leaves(), back.leaves()):
            if isinstance(a, nn.SpatialBatchNormalization):
                assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
                moved += int(float(a.running_mean.abs().max()) > 0)
        assert moved > 0, "the training iterations did not move any running statistic: nothing was checked"
        if name == "netG":      # test_vid.lua:47-48,102: net:evaluate(); net:forward(input)
            x = tr._g_in()
            net.evaluate()
            back.evaluate()
            y0 = net.forward(x).clone()
            y1 = back.forward(x)
            assert torch.equal(y0, y1)
            net.training()
    # resume: a second trainer takes the checkpoints into its own nets and walks the same next iteration
    tr2 = mk()
    util.load(paths["netG"], tr2.netG)
    util.load(paths["netD"], tr2.netD)
    assert torch.equal(tr2.parametersG, tr.parametersG) and torch.equal(tr2.parametersD, tr.parametersD)
    fresh = mk()                    # Adam restarts at t = 0 on resume: compare with a trainer that does the same
    fresh.netG.load_reference_flat(tr.netG.reference_flat())
    fresh.netD.load_reference_flat(tr.netD.reference_flat())
    for a, b in zip(tr.netG.leaves() + tr.netD.leaves(), fresh.netG.leaves() + fresh.netD.leaves()):
        if isinstance(a, nn.SpatialBatchNormalization):
            b.running_mean.copy_(a.running_mean)
            b.running_var.copy_(a.running_var)
    for t in (tr2, fresh):
        t.set_batch(*batch(2))
        t.step()
        t.flush()
    assert torch.equal(tr2.parametersG, fresh.parametersG) and torch.equal(tr2.parametersD, fresh.parametersD)
    assert float(tr2.errG) == float(fresh.errG) and float(tr2.errD) == float(fresh.errD)


def test_a_cudnn_typed_checkpoint_loads_into_the_hip_nets(tmp_path, oracle, hipb):
    """util.save converts cudnn.SpatialConvolution back to nn.SpatialConvolution before writing (util.lua:33-50), but a net saved
    with plain torch.save from a GPU run — or by an older util — carries the cudnn type names.  A hand-assembled file of that kind
    (cudnn.SpatialConvolution / cudnn.SpatialBatchNormalization, otherwise netD's layout, train.lua:183-199) must load, and the
    loaded net must compute what the oracle's netD computes with the same parameters (evaluate mode)."""
    from video_filler_amd import t7, util
    rng = np.random.default_rng(17)
    ref = oracle.build_netD(3, 64, False)
    oracle.weights_init(ref, rng)
    mods = []
    for m in ref.modules:
        n = type(m).__name__
        if n == "SpatialConvolution":
            mods.append(t7.TorchObject("cudnn.SpatialConvolution", dict(
                nInputPlane=m.nInputPlane, nOutputPlane=m.nOutputPlane, kW=m.kW, kH=m.kH, dW=m.dW, dH=m.dH, padW=m.padW, padH=m.padH,
                groups=1, weight=m.weight.astype(np.float32).copy(), bias=m.bias.astype(np.float32).copy(), train=True)))
        elif n == "SpatialBatchNormalization":
            m.running_mean[...] = rng.standard_normal(m.running_mean.shape).astype(np.float32) * 0.1
            m.running_var[...] = (rng.random(m.running_var.shape) + 0.5).astype(np.float32)
            mods.append(t7.TorchObject("cudnn.SpatialBatchNormalization", dict(
                affine=True, eps=m.eps, momentum=m.momentum, weight=m.weight.copy(), bias=m.bias.copy(),
                running_mean=m.running_mean.copy(), running_var=m.running_var.copy(), train=True)))
        elif n == "LeakyReLU":
            mods.append(t7.TorchObject("nn.LeakyReLU", dict(negval=m.negval, inplace=True, train=True)))
        elif n == "Sigmoid":
            mods.append(t7.TorchObject("nn.Sigmoid", dict(train=True)))
        elif n == "View":
            mods.append(t7.TorchObject("nn.View", dict(size=t7.Storage(np.asarray([1], np.int64)), numElements=1, numInputDims=3, train=True)))
        else:
            raise AssertionError(n)
    path = str(tmp_path / "netD_cudnn.t7")
    t7.save(path, t7.TorchObject("nn.Sequential", dict(modules=mods, train=True)))
    net = util.load(path)
    assert [m.type_name() for m in net.leaves()][0] == "nn.SpatialConvolution"      # (what util.save would write back)
    net.getParameters()
    flat_ref, _ = ref.getParameters()
    assert np.array_equal(net.reference_flat().cpu().numpy(), flat_ref)
    x = (rng.random((4, 3, 64, 64)).astype(np.float32) * 2 - 1)
    ref.evaluate()
    want = ref.forward(x)
    net.evaluate()
    got = net.forward(torch.from_numpy(x).to(hipb.device).contiguous(memory_format=torch.channels_last))
    assert float(np.abs(got.cpu().numpy().reshape(want.shape) - want).max()) <= 2e-5
