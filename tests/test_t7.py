"""Torch7 serialization (video_filler_amd/t7.py): the grammar of torch7's File.lua / Tensor.lua / Storage.c, pinned by
hand-assembled byte strings (known answers worked out from the published format — no .t7 file exists in the reference
repository, so this is the strongest pin available), by reader/writer round trips, and by util.save/util.load of the
nets the drivers checkpoint (util.lua:72-105).  CPU only."""
import struct

import numpy as np
import pytest
import torch

import video_filler_amd  # noqa: F401
from video_filler_amd import backend as vb
from video_filler_amd import t7

from oracle_backend import OracleBackend


@pytest.fixture()
def cpu_backend():
    prev = vb._BACKEND if hasattr(vb, "_BACKEND") else None
    b = vb.set_backend(OracleBackend())
    yield b
    vb.set_backend(prev)


def i32(v):
    return struct.pack("<i", v)


def i64(v):
    return struct.pack("<q", v)


def lstr(s):
    return i32(len(s)) + s.encode()


def test_scalars_known_bytes():
    assert t7.dumps(None) == i32(0)
    assert t7.dumps(2.5) == i32(1) + struct.pack("<d", 2.5)
    assert t7.dumps(7) == i32(1) + struct.pack("<d", 7.0)              # every Lua number is a double
    assert t7.dumps("V 1") == i32(2) + i32(3) + b"V 1"
    assert t7.dumps(True) == i32(5) + i32(1) and t7.dumps(False) == i32(5) + i32(0)
    for v in (None, 2.5, 7, "abc", True, False):
        assert t7.loads(t7.dumps(v)) == v


def test_table_known_bytes_and_back_reference():
    # {1 = "a", x = 2}: type 3, index 1, 2 pairs
    want = i32(3) + i32(1) + i32(2) + (i32(1) + struct.pack("<d", 1.0)) + (i32(2) + lstr("a")) + (i32(2) + lstr("x")) + (
        i32(1) + struct.pack("<d", 2.0))
    assert t7.dumps({1: "a", "x": 2}) == want
    assert t7.loads(want) == {1: "a", "x": 2}
    # the same table twice: the second occurrence is only (type, index)
    inner = {"k": 1}
    data = t7.dumps({1: inner, 2: inner})
    assert data.endswith(i32(3) + i32(2))
    back = t7.loads(data)
    assert back[1] is back[2] and back[1] == {"k": 1}


def test_float_tensor_known_bytes():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    want = (i32(4) + i32(1) + lstr("V 1") + lstr("torch.FloatTensor") + i32(2) + i64(2) + i64(3) + i64(3) + i64(1) + i64(1)
            + i32(4) + i32(2) + lstr("V 1") + lstr("torch.FloatStorage") + i64(6) + a.tobytes())
    assert t7.dumps(a) == want
    np.testing.assert_array_equal(t7.loads(want), a)


def test_reader_honours_strides_offset_and_shared_storage():
    # a 2x2 view (stride 3,1 ; offset 2) of a 9-element storage, and a second tensor sharing that storage by index
    st = np.arange(9, dtype=np.float32)
    storage = i32(4) + i32(2) + lstr("V 1") + lstr("torch.FloatStorage") + i64(9) + st.tobytes()
    t1 = i32(4) + i32(1) + lstr("V 1") + lstr("torch.FloatTensor") + i32(2) + i64(2) + i64(2) + i64(3) + i64(1) + i64(2) + storage
    t2 = i32(4) + i32(3) + lstr("V 1") + lstr("torch.FloatTensor") + i32(1) + i64(3) + i64(1) + i64(7) + i32(4) + i32(2)
    data = i32(3) + i32(4) + i32(2) + (i32(1) + struct.pack("<d", 1.0)) + t1 + (i32(1) + struct.pack("<d", 2.0)) + t2
    back = t7.loads(data)
    np.testing.assert_array_equal(back[1], [[1, 2], [4, 5]])
    np.testing.assert_array_equal(back[2], [6, 7, 8])


def test_empty_tensor_and_other_dtypes_roundtrip():
    e = t7.loads(t7.dumps(np.zeros((0,), np.float32)))
    assert e.size == 0
    for dt in (np.float64, np.int64, np.int32, np.uint8):
        a = (np.arange(24) % 7).astype(dt).reshape(2, 3, 4)
        b = t7.loads(t7.dumps(a))
        assert b.dtype == dt
        np.testing.assert_array_equal(a, b)


def test_module_object_layout():
    """A torch class without its own writer is (type 4, index, "V 1", class name, field table) — File.lua writeObject."""
    o = t7.TorchObject("nn.Tanh", {"train": True})
    want = i32(4) + i32(1) + lstr("V 1") + lstr("nn.Tanh") + i32(3) + i32(2) + i32(1) + (i32(2) + lstr("train")) + (i32(5) + i32(1))
    assert t7.dumps(o) == want
    back = t7.loads(want)
    assert back.cls == "nn.Tanh" and back["train"] is True


def test_unsupported_payloads_fail_loudly():
    with pytest.raises(ValueError, match="functions"):
        t7.loads(i32(6) + i32(1))
    with pytest.raises(ValueError, match="unknown torch type"):
        t7.loads(i32(42))
    with pytest.raises(TypeError):
        t7.dumps(object())


@pytest.mark.parametrize("which", ["netG", "netD"])
def test_checkpoint_roundtrip_through_util(which, tmp_path, cpu_backend):
    """util.save(..., net) / util.load(...) through a .t7 file: same topology, same parameters in the reference's
    NCHW order, same running statistics, buffers cleared and no gradWeight/gradBias in the file (util.lua:72-97)."""
    from video_filler_amd import util, nn
    from video_filler_amd.trainers import build_netG, build_netD, weights_init
    gen = torch.Generator().manual_seed(3)
    net = build_netG(6, 6, 8, 8, 16, True) if which == "netG" else build_netD(6, 8, True)
    weights_init(net, gen)
    net.getParameters()
    for m in net.leaves():
        if isinstance(m, nn.SpatialBatchNormalization):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen))
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    path = str(tmp_path / ("%s.t7" % which))
    util.save(path, net)
    tree = t7.load(path)
    assert tree.cls == "nn.Sequential" and tree["_type"] == "torch.FloatTensor" and tree["output"].size == 0
    mods = [tree["modules"][k] for k in sorted(tree["modules"])]
    convs = [m for m in mods if "Convolution" in m.cls]
    assert convs and all("gradWeight" not in m.fields and "gradBias" not in m.fields for m in convs)
    first = convs[0]
    assert first["weight"].shape == (first["nOutputPlane"], first["nInputPlane"], 4, 4) or which == "netG"
    back = util.load(path)
    assert [m.type_name() for m in back.leaves()] == [m.type_name() for m in net.leaves()]
    back.getParameters()
    assert torch.equal(back.reference_flat(), net.reference_flat())
    for a, b in zip(net.leaves(), back.leaves()):
        if isinstance(a, nn.SpatialBatchNormalization):
            assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
            assert a.eps == b.eps and a.momentum == b.momentum
        if isinstance(a, nn.LeakyReLU):
            assert a.slope == b.slope and a.inplace == b.inplace
    # same function: evaluate-mode forward of the reloaded net equals the original's
    net.evaluate()
    back.evaluate()
    x = torch.rand((2, 6, 128, 128), generator=gen) * 2 - 1
    y0 = net.forward(x.contiguous(memory_format=torch.channels_last)).clone()
    y1 = back.forward(x.contiguous(memory_format=torch.channels_last))
    assert torch.equal(y0, y1)
    # loading INTO an existing net (the drivers' resume path)
    other = build_netG(6, 6, 8, 8, 16, True) if which == "netG" else build_netD(6, 8, True)
    other.getParameters()
    util.load(path, other)
    assert torch.equal(other.reference_flat(), net.reference_flat())


def test_checkpoint_roundtrip_of_the_table_nets(tmp_path, cpu_backend):
    """The conditionAdv netD and the noiseGen netG (nn.ParallelTable + nn.JoinTable, 5x5 and 1x1 convolutions) through
    util.save / util.load: same module tree, same flat parameters, same evaluate-mode function."""
    from video_filler_amd import util, nn
    from video_filler_amd.trainers import build_netG, build_netD, weights_init
    gen = torch.Generator().manual_seed(5)
    chl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for name, net, x in (
            ("netD_cond", build_netD(3, 8, False, conditionAdv=True),
             [chl(torch.rand((2, 3, 128, 128), generator=gen)), chl(torch.rand((2, 3, 64, 64), generator=gen))]),
            ("netG_noise", build_netG(3, 3, 8, 8, 16, False, noise_nz=12),
             [chl(torch.rand((2, 3, 128, 128), generator=gen)), chl(torch.randn((2, 12, 1, 1), generator=gen))])):
        weights_init(net, gen)
        net.getParameters()
        path = str(tmp_path / (name + ".t7"))
        util.save(path, net)
        tree = t7.load(path)
        first = tree["modules"][1]
        assert first.cls == "nn.ParallelTable" and tree["modules"][2].cls == "nn.JoinTable" and tree["modules"][2]["dimension"] == 2
        back = util.load(path)
        assert isinstance(back.modules[0], nn.ParallelTable) and isinstance(back.modules[1], nn.JoinTable)
        assert [m.type_name() for m in back.leaves()] == [m.type_name() for m in net.leaves()]
        back.getParameters()
        assert torch.equal(back.reference_flat(), net.reference_flat())
        net.evaluate()
        back.evaluate()
        y0 = net.forward(x).clone()
        assert torch.equal(y0, back.forward(x))


def test_reads_pre_2016_running_std(cpu_backend):
    """nn before 2016 stored running_std = 1/sqrt(var + eps) (util.lua:46 copies running_std)."""
    var = np.array([0.5, 2.0, 1.0], np.float32)
    eps = 1e-5
    bn = t7.TorchObject("nn.SpatialBatchNormalization", dict(affine=True, eps=eps, momentum=0.1, train=True,
                        weight=np.ones(3, np.float32), bias=np.zeros(3, np.float32), running_mean=np.zeros(3, np.float32),
                        running_std=(1.0 / np.sqrt(var + eps)).astype(np.float32)))
    seq = t7.TorchObject("nn.Sequential", dict(modules=[bn], train=True))
    net = t7.net_from_t7(t7.loads(t7.dumps(seq)))
    np.testing.assert_allclose(net.leaves()[0].running_var.numpy(), var, rtol=1e-5)


def test_bare_long_storage_known_bytes_and_view_size_form():
    """nn.View.size / nn.JoinTable.size are torch.LongStorage objects in Torch7 (View:resetSize, JoinTable.__init), not
    LongTensors: a bare storage is `4, index, "V 1", "torch.LongStorage", int64 n, n x int64`."""
    want = i32(4) + i32(1) + lstr("V 1") + lstr("torch.LongStorage") + i64(2) + i64(1) + i64(-1)
    assert t7.dumps(t7.Storage(np.array([1, -1], np.int64))) == want
    np.testing.assert_array_equal(t7.loads(want), [1, -1])


def test_view_and_jointable_sizes_are_written_as_storages(cpu_backend):
    from video_filler_amd import nn
    from video_filler_amd.trainers import build_netD
    tree = t7.net_to_t7(build_netD(3, 8, False, conditionAdv=True))
    mods = [tree["modules"][i] for i in range(len(tree["modules"]))]
    view = [m for m in mods if m.cls == "nn.View"][0]
    join = [m for m in mods if m.cls == "nn.JoinTable"][0]
    assert isinstance(view["size"], t7.Storage) and list(view["size"].arr) == [1]
    assert isinstance(join["size"], t7.Storage) and len(join["size"]) == 0
    data = t7.dumps(tree)
    assert data.count(b"torch.LongStorage") == 2 and b"torch.LongTensor" not in data
    back = t7.net_from_t7(t7.loads(data))
    assert isinstance(back.modules[-1], nn.View) and back.modules[-1].sizes == (1,)


def test_tensor_views_outside_their_storage_are_rejected():
    """size / stride / offset are read from the file; a view that leaves its storage must raise, not read out of bounds."""
    def tensor(size, stride, off, n):
        b = i32(4) + i32(1) + lstr("V 1") + lstr("torch.FloatTensor") + i32(len(size))
        b += b"".join(i64(v) for v in size) + b"".join(i64(v) for v in stride) + i64(off)
        b += i32(4) + i32(2) + lstr("V 1") + lstr("torch.FloatStorage") + i64(n) + struct.pack("<%df" % n, *range(n))
        return b
    np.testing.assert_array_equal(t7.loads(tensor([2, 3], [3, 1], 1, 6)), np.arange(6, dtype=np.float32).reshape(2, 3))
    np.testing.assert_array_equal(t7.loads(tensor([2, 2], [1, 2], 2, 6)), [[1, 3], [2, 4]])      # offset is 1-based
    for bad in (tensor([2, 3], [3, 1], 2, 6), tensor([2, 3], [4, 1], 1, 6), tensor([7], [1], 1, 6), tensor([2], [1], 0, 6),
                tensor([2], [-1], 1, 6)):
        with pytest.raises(ValueError):
            t7.loads(bad)
