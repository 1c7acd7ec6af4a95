"""Batch preparation on the device — the loader side of the hot path.

The reference prepares every sample on the host, inside the loader threads (datavid/donkey_folder.lua:135-189) or at
the top of the closure (train.lua:284-298), and then copies the batch to the GPU.  Here the host only decodes and
draws the random numbers; cropping, masking, flipping, the [0,1] -> [-1,1] map and the NCHW -> NHWC conversion are one
kernel per sample, writing straight into the batch buffers the closures read (`vf_clip_prepare`, `vf_center_prepare`).
"""
import math

import numpy as np
import torch

from .backend import get_backend, nhwc_empty

CENTER_FILL = (117.0, 104.0, 123.0)      # train.lua:287-289


def center_prepare(batch, overlapPred, fill=CENTER_FILL, out=None):
    """train.lua:284-298: (input_ctx, real_center) as channels-last device tensors from the loader's batch
    (B x nc x fs x fs in [-1,1], host or device).  out = (input_ctx, real_center): write into these existing tensors
    (a captured HIP graph keeps reading the buffers it was captured with)."""
    B = get_backend()
    x = B.from_host(batch).float().contiguous()
    nB, nc, fs, _ = x.shape
    dev = x.device
    fillv = [2 * m / 255.0 - 1.0 for m in fill]
    fillv = (fillv * ((nc + len(fillv) - 1) // len(fillv)))[:nc]
    if out is not None and tuple(out[0].shape) == (nB, nc, fs, fs) and tuple(out[1].shape) == (nB, nc, fs // 2, fs // 2):
        ctx, center = out
    else:
        ctx = nhwc_empty(nB, nc, fs, fs, dev)
        center = nhwc_empty(nB, nc, fs // 2, fs // 2, dev)
    B.center_prepare(x, ctx, center, B.from_host(torch.tensor(fillv, dtype=torch.float32)), overlapPred)
    return ctx, center


class ClipBatcher:
    """trainHook of datavid/donkey_folder.lua:135-189 feeding the (ctx, full, mask) batch of datavid/dataset.lua:426.

    add(clip, mask) takes ONE decoded, scaled clip ((predLen*nc) x iH x iW float in [0,1], host numpy) and the scaled
    Byte mask (1 x iH x iW), draws the hook's random numbers from `rng` in the hook's order (crop corner, dark-crop
    rejection, random blocks, flip) and, unless the sample is rejected, launches one kernel that writes row `n` of the
    three batch tensors.  batch() returns them (channels-last, on the device) for VidTrainer.set_batch."""

    def __init__(self, batchSize, channels, fineSize=128, maskValue=110.0 / 255.0, rng=None):
        B = get_backend()
        self.B, self.C, self.fs, self.maskValue = batchSize, channels, fineSize, float(maskValue)
        self.rng = rng or np.random.default_rng()
        dev = B.device
        self.full = nhwc_empty(batchSize, channels, fineSize, fineSize, dev)
        self.masked = nhwc_empty(batchSize, channels, fineSize, fineSize, dev)
        self.mask = nhwc_empty(batchSize, channels, fineSize, fineSize, dev)
        self.n = 0

    def draw(self, clip, mask):
        """The hook's decisions for one sample (host side, no pixels touched beyond two reductions)."""
        fs, rng = self.fs, self.rng
        _, iH, iW = clip.shape
        h1 = int(math.ceil(rng.uniform(1e-2, iH - fs)))          # :145-146 (1-based corner; image.crop takes it 0-based)
        w1 = int(math.ceil(rng.uniform(1e-2, iW - fs)))
        h1, w1 = min(h1, iH - fs), min(w1, iW - fs)
        crop = clip[:, h1:h1 + fs, w1:w1 + fs]
        if float(crop.mean()) < 0.1 and rng.uniform() > 0.05:     # :148-153: dark crops are mostly rejected
            return None
        blocks, bs = None, fs // 6
        if not (mask[:, h1:h1 + fs, w1:w1 + fs].max() > 0.5):     # :165-169
            nBlocks = int(rng.integers(2, 11))                   # torch.random(2, maxBlocks)
            blocks = [(int(rng.integers(3, fs - bs - 1)), int(rng.integers(3, fs - bs - 1))) for _ in range(nBlocks)]
        flip = bool(rng.uniform() > 0.5)                          # :178
        return dict(w1=w1, h1=h1, flip=flip, blocks=blocks, blockSize=bs)

    def add(self, clip, mask, decisions=None):
        B = get_backend()
        assert self.n < self.B, "batch is full"
        d = decisions if decisions is not None else self.draw(clip, mask)
        if d is None:
            return False
        clip_d = B.from_host(torch.from_numpy(np.ascontiguousarray(clip, np.float32)))
        mask_d = B.from_host(torch.from_numpy(np.ascontiguousarray(mask[0], np.float32))) if d["blocks"] is None else None
        n = self.n
        B.clip_prepare(clip_d, mask_d, self.full[n:n + 1], self.masked[n:n + 1], self.mask[n:n + 1], d["w1"], d["h1"],
                       d["flip"], self.maskValue, d["blocks"], d["blockSize"])
        self.n += 1
        return True

    def batch(self):
        """(real_ctx, real_full, real_mask) — the loader contract of datavid/dataset.lua:426."""
        assert self.n == self.B, "batch holds %d of %d samples" % (self.n, self.B)
        self.n = 0
        return self.masked, self.full, self.mask
