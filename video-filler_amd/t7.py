"""Torch7 binary serialization (`torch.save` / `torch.load`, torch7 File.lua) — reader and writer, host side.

The reference checkpoints with `util.save(filename, net, gpu)` = `torch.save(filename, netsave)` of a float
nn.Sequential whose buffers were cleared and whose gradWeight/gradBias were dropped, and reloads with `util.load`
(util.lua:72-105).  This module reads such files into this package's nn mirror and writes files of the same shape.
Torch7 is absent here, so the format is restated from File.lua / Tensor.lua / Storage.c as published:

  object   := int32 type, then
     0 nil | 1 number: float64 | 2 string: int32 length + bytes | 5 boolean: int32 0/1
     3 table:  int32 index, and on first sight: int32 count + count x (object key, object value)
     4 torch:  int32 index, and on first sight: string "V 1", string className, then the class's own payload
  torch.XTensor payload:  int32 nDim, int64 size[nDim], int64 stride[nDim], int64 storageOffset (1-based),
                          object storage
  torch.XStorage payload: int64 n, then n raw elements
  any other class (nn.*): one object — the table of the instance's fields
  Integers are little-endian; "long" is 8 bytes (64-bit torch builds).  An index seen before refers to that object.

Parity of this file format is unpinned (no .t7 exists in the reference repository); tests/test_t7.py pins the reader
and writer against hand-assembled byte strings of the grammar above and against each other.
"""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5
TYPE_FUNCTION, TYPE_RECUR_FUNCTION, LEGACY_TYPE_RECUR_FUNCTION = 6, 8, 7

_TENSOR = {"torch.FloatTensor": ("torch.FloatStorage", np.float32), "torch.DoubleTensor": ("torch.DoubleStorage", np.float64),
           "torch.LongTensor": ("torch.LongStorage", np.int64), "torch.IntTensor": ("torch.IntStorage", np.int32),
           "torch.ByteTensor": ("torch.ByteStorage", np.uint8), "torch.CudaTensor": ("torch.CudaStorage", np.float32)}
_STORAGE = {s: d for s, d in _TENSOR.values()}


class Storage:
    """A bare torch.XStorage (not wrapped in a tensor): nn.View.size and nn.JoinTable.size are torch.LongStorage in
    Torch7 (View:resetSize, JoinTable.__init), and View:updateOutput calls storage methods on them."""

    def __init__(self, arr):
        self.arr = np.ascontiguousarray(arr).reshape(-1)

    def __len__(self):
        return self.arr.size


class TorchObject:
    """An instance of a torch class without a payload of its own (nn modules): class name + field table."""

    def __init__(self, cls, fields=None):
        self.cls = cls
        self.fields = fields if fields is not None else {}

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default)

    def __repr__(self):
        return "TorchObject(%s, %s)" % (self.cls, sorted(map(str, self.fields)))


# ------------------------------------------------------------------------------------------------ reader
class Reader:
    def __init__(self, data):
        self.b, self.pos, self.memo = memoryview(data), 0, {}

    def _take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.pos)
        self.pos += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def _string(self):
        n = self._take("i")
        s = bytes(self.b[self.pos:self.pos + n])
        self.pos += n
        return s.decode("latin-1")

    def read(self):
        t = self._take("i")
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self._take("d")
            return int(v) if float(v).is_integer() and abs(v) < 2 ** 53 else v
        if t == TYPE_STRING:
            return self._string()
        if t == TYPE_BOOLEAN:
            return self._take("i") != 0
        if t == TYPE_TABLE:
            idx = self._take("i")
            if idx in self.memo:
                return self.memo[idx]
            out = {}
            self.memo[idx] = out
            for _ in range(self._take("i")):
                k = self.read()
                out[k] = self.read()
            return out
        if t == TYPE_TORCH:
            idx = self._take("i")
            if idx in self.memo:
                return self.memo[idx]
            version = self._string()
            cls = self._string() if version.startswith("V ") else version      # pre-versioning files: the class name
            if cls in _TENSOR:
                nd = self._take("i")
                size = [self._take("q") for _ in range(nd)]
                stride = [self._take("q") for _ in range(nd)]
                off = self._take("q") - 1
                storage = self.read()
                if storage is None or nd == 0:
                    arr = np.zeros((0,), _TENSOR[cls][1])
                else:
                    # size / stride / offset come from the file: a view that leaves the storage is a malformed checkpoint
                    if off < 0 or any(n < 0 for n in size) or any(st < 0 for st in stride):
                        raise ValueError("tensor with negative size, stride or offset at byte %d" % self.pos)
                    last = off + sum((n - 1) * st for n, st in zip(size, stride)) if all(n > 0 for n in size) else off
                    if all(n > 0 for n in size) and last >= len(storage):
                        raise ValueError("tensor view [offset %d, last element %d] exceeds its storage of %d elements"
                                         % (off, last, len(storage)))
                    arr = np.lib.stride_tricks.as_strided(storage[off:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.memo[idx] = arr
                return arr
            if cls in _STORAGE:
                n = self._take("q")
                dt = np.dtype(_STORAGE[cls]).newbyteorder("<")
                arr = np.frombuffer(self.b, dt, n, self.pos).astype(_STORAGE[cls])
                self.pos += n * dt.itemsize
                self.memo[idx] = arr
                return arr
            obj = TorchObject(cls)
            self.memo[idx] = obj
            fields = self.read()
            obj.fields = fields if isinstance(fields, dict) else {"_payload": fields}
            return obj
        if t in (TYPE_FUNCTION, TYPE_RECUR_FUNCTION, LEGACY_TYPE_RECUR_FUNCTION):
            raise ValueError("serialized Lua functions are not supported (offset %d)" % self.pos)
        raise ValueError("unknown torch type tag %d at offset %d" % (t, self.pos - 4))


def load(path):
    with open(path, "rb") as f:
        return Reader(f.read()).read()


def loads(data):
    return Reader(data).read()


# ------------------------------------------------------------------------------------------------ writer
class Writer:
    def __init__(self):
        self.out, self.memo, self.keep = [], {}, []

    def _put(self, fmt, *v):
        self.out.append(struct.pack("<" + fmt, *v))

    def _string(self, s):
        b = s.encode("latin-1")
        self._put("i", len(b))
        self.out.append(b)

    def _index(self, obj):
        """(index, first_time) — torch numbers tables and torch objects in order of first appearance, from 1."""
        key = id(obj)
        if key in self.memo:
            return self.memo[key], False
        self.memo[key] = len(self.memo) + 1
        self.keep.append(obj)        # keep ids alive
        return self.memo[key], True

    def write(self, obj):
        if obj is None:
            self._put("i", TYPE_NIL)
        elif isinstance(obj, bool):
            self._put("i", TYPE_BOOLEAN)
            self._put("i", 1 if obj else 0)
        elif isinstance(obj, (int, float, np.integer, np.floating)):
            self._put("i", TYPE_NUMBER)
            self._put("d", float(obj))
        elif isinstance(obj, str):
            self._put("i", TYPE_STRING)
            self._string(obj)
        elif isinstance(obj, dict):
            self._put("i", TYPE_TABLE)
            idx, first = self._index(obj)
            self._put("i", idx)
            if first:
                self._put("i", len(obj))
                for k, v in obj.items():
                    self.write(k)
                    self.write(v)
        elif isinstance(obj, (list, tuple)):
            self.write({i + 1: v for i, v in enumerate(obj)})      # a Lua array
        elif isinstance(obj, np.ndarray):
            cls = {np.dtype(np.float32): "torch.FloatTensor", np.dtype(np.float64): "torch.DoubleTensor",
                   np.dtype(np.int64): "torch.LongTensor", np.dtype(np.int32): "torch.IntTensor",
                   np.dtype(np.uint8): "torch.ByteTensor"}[obj.dtype]
            self._put("i", TYPE_TORCH)
            idx, first = self._index(obj)
            self._put("i", idx)
            if first:
                self._string("V 1")
                self._string(cls)
                a = np.ascontiguousarray(obj)
                if a.size == 0:                      # torch.Tensor(): no dimensions, no storage
                    self._put("i", 0)
                    self._put("q", 1)
                    self._put("i", TYPE_NIL)
                    return
                self._put("i", a.ndim)
                for s in a.shape:
                    self._put("q", s)
                for s in a.strides:
                    self._put("q", s // a.itemsize)
                self._put("q", 1)
                storage = a.reshape(-1)
                self._put("i", TYPE_TORCH)
                sidx, _ = self._index(storage)
                self._put("i", sidx)
                self._string("V 1")
                self._string(_TENSOR[cls][0])
                self._put("q", storage.size)
                self.out.append(storage.astype(storage.dtype.newbyteorder("<")).tobytes())
        elif isinstance(obj, Storage):
            cls = {np.dtype(np.float32): "torch.FloatStorage", np.dtype(np.float64): "torch.DoubleStorage",
                   np.dtype(np.int64): "torch.LongStorage", np.dtype(np.int32): "torch.IntStorage",
                   np.dtype(np.uint8): "torch.ByteStorage"}[obj.arr.dtype]
            self._put("i", TYPE_TORCH)
            idx, first = self._index(obj)
            self._put("i", idx)
            if first:
                self._string("V 1")
                self._string(cls)
                self._put("q", obj.arr.size)
                self.out.append(obj.arr.astype(obj.arr.dtype.newbyteorder("<")).tobytes())
        elif isinstance(obj, TorchObject):
            self._put("i", TYPE_TORCH)
            idx, first = self._index(obj)
            self._put("i", idx)
            if first:
                self._string("V 1")
                self._string(obj.cls)
                self.write(obj.fields)
        else:
            raise TypeError("cannot serialize %r" % type(obj))

    def bytes(self):
        return b"".join(self.out)


def dumps(obj):
    w = Writer()
    w.write(obj)
    return w.bytes()


def save(path, obj):
    with open(path, "wb") as f:
        f.write(dumps(obj))


# ------------------------------------------------------------------------------------------------ nn <-> t7
_EMPTY = lambda: np.zeros((0,), np.float32)      # m.output = m.output.new()  (util.lua:54-55)


def net_to_t7(net):
    """This package's nn.Sequential -> the object tree util.save writes (util.lua:72-97): float tensors in the
    reference's NCHW order, cleared buffers, no gradWeight/gradBias."""
    from . import nn
    from .nn import Sequential

    def base(m, cls, **fields):
        f = dict(_type="torch.FloatTensor", output=_EMPTY(), gradInput=_EMPTY(), train=bool(m.train))
        f.update(fields)
        return TorchObject(cls, f)

    def host(t):
        return np.ascontiguousarray(t.detach().float().contiguous().cpu().numpy())

    def conv(m):
        if isinstance(m, Sequential):
            return base(m, "nn.Sequential", modules=[conv(c) for c in m.modules])
        if isinstance(m, nn.ParallelTable):         # train.lua:115,168 (noiseGen / conditionAdv nets)
            return base(m, "nn.ParallelTable", modules=[conv(c) for c in m.modules])
        if isinstance(m, nn.JoinTable):
            return base(m, "nn.JoinTable", dimension=m.dimension, size=Storage(np.zeros((0,), np.int64)))
        if isinstance(m, nn.SpatialFullConvolution) or isinstance(m, nn.SpatialConvolution):
            f = dict(nInputPlane=m.nInputPlane, nOutputPlane=m.nOutputPlane, kW=m.kW, kH=m.kH, dW=m.dW, dH=m.dH, padW=m.padW,
                     padH=m.padH, weight=host(m.weight), bias=host(m.bias))
            if isinstance(m, nn.SpatialFullConvolution):
                f.update(adjW=0, adjH=0)
            return base(m, m.type_name(), **f)
        if isinstance(m, nn.SpatialBatchNormalization):
            return base(m, "nn.SpatialBatchNormalization", affine=True, eps=m.eps, momentum=m.momentum, nDim=4,
                        weight=host(m.weight), bias=host(m.bias), running_mean=host(m.running_mean),
                        running_var=host(m.running_var))
        if isinstance(m, nn.LeakyReLU):
            return base(m, "nn.LeakyReLU", negval=m.slope, inplace=bool(m.inplace))
        if isinstance(m, nn.ReLU):
            return base(m, "nn.ReLU", threshold=0, val=0, inplace=bool(m.inplace))
        if isinstance(m, nn.Tanh):
            return base(m, "nn.Tanh")
        if isinstance(m, nn.Sigmoid):
            return base(m, "nn.Sigmoid")
        if isinstance(m, nn.View):
            return base(m, "nn.View", size=Storage(np.asarray(m.sizes, np.int64)), numElements=int(np.prod([s for s in m.sizes if s > 0])),
                        numInputDims=getattr(m, "numInputDims", None))
        raise TypeError("no Torch7 form for %r" % m)

    return conv(net)


def net_from_t7(obj, fuse=True, lazy_zero=True):
    """The object tree of a util.save checkpoint -> this package's nn.Sequential (parameters and running statistics
    on the device; call getParameters() afterwards as the drivers do)."""
    import torch
    from . import nn
    from .backend import get_backend
    B = get_backend()

    def dev(a):
        return B.from_host(torch.from_numpy(np.ascontiguousarray(a, np.float32)))

    def lua_list(t):
        return [t[k] for k in sorted(k for k in t if isinstance(k, (int, float)))]

    def conv(o):
        c, f = o.cls, o.fields
        if c == "nn.Sequential":
            s = nn.Sequential(fuse=fuse, lazy_zero=lazy_zero)
            for m in lua_list(f["modules"]):
                s.add(conv(m))
            m = s
        elif c == "nn.ParallelTable":
            m = nn.ParallelTable()
            for sub in lua_list(f["modules"]):
                m.add(conv(sub))
        elif c == "nn.JoinTable":
            m = nn.JoinTable(int(f["dimension"]))
        elif c in ("nn.SpatialConvolution", "nn.SpatialFullConvolution", "cudnn.SpatialConvolution", "cudnn.SpatialFullConvolution"):
            cls = nn.SpatialFullConvolution if "Full" in c else nn.SpatialConvolution
            m = cls(f["nInputPlane"], f["nOutputPlane"], f["kW"], f["kH"], f["dW"], f["dH"], f.get("padW", 0), f.get("padH", 0))
            m.weight.copy_(dev(f["weight"]).reshape(m.weight.shape))
            m.bias.copy_(dev(f["bias"]))
        elif c in ("nn.SpatialBatchNormalization", "cudnn.SpatialBatchNormalization"):
            n = int(f["running_mean"].shape[0])
            m = nn.SpatialBatchNormalization(n, f.get("eps", 1e-5), f.get("momentum", 0.1), True)
            m.weight.copy_(dev(f["weight"]))
            m.bias.copy_(dev(f["bias"]))
            m.running_mean.copy_(dev(f["running_mean"]))
            if "running_var" in f:
                m.running_var.copy_(dev(f["running_var"]))
            else:                                  # nn before 2016: running_std = 1/sqrt(var + eps)
                rs = np.asarray(f["running_std"], np.float64)
                m.running_var.copy_(dev((1.0 / (rs * rs) - f.get("eps", 1e-5)).astype(np.float32)))
        elif c == "nn.LeakyReLU":
            m = nn.LeakyReLU(f.get("negval", 0.01), bool(f.get("inplace", False)))
        elif c == "nn.ReLU":
            m = nn.ReLU(bool(f.get("inplace", False)))
        elif c == "nn.Tanh":
            m = nn.Tanh()
        elif c == "nn.Sigmoid":
            m = nn.Sigmoid()
        elif c == "nn.View":
            m = nn.View(*[int(v) for v in np.asarray(f["size"]).reshape(-1)])
            if f.get("numInputDims") is not None:
                m.setNumInputDims(int(f["numInputDims"]))
        else:
            raise TypeError("checkpoint holds a %s, which the hot path does not build" % c)
        if f.get("train") is False:
            m.train = False
        return m

    return conv(obj)
