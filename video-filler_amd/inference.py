"""Inference drivers: test_vid.lua (one forward of the clip) and test_vid_wholeim.lua (whole frames, tile loop).

The generator runs in evaluate() mode (BatchNorm uses its running statistics), so tiles are independent and the
whole-image loop — one net:forward per 128x128 tile in the reference (test_vid_wholeim.lua:159-205) — becomes ONE
forward over all tiles: `vf_tiles_gather` builds the NHWC batch (incl. the script's vertical-flip rule for the first
three tiles of the top row), `vf_tiles_scatter` writes the output tiles back into the planar frames.
"""
import torch

from .backend import get_backend, nhwc_empty


def predict_clip(net, input_image):
    """test_vid.lua:47-48,100-106: net:evaluate(); pred = net:forward(input_image); returns (input, pred) mapped back
    to [0,1] (`add(1):mul(0.5)`, :112-114).  input_image: predLen x nc x fs x fs in [-1,1]."""
    B = get_backend()
    net.evaluate()
    x = B.from_host(input_image).float()
    if not (x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()):
        x = x.contiguous(memory_format=torch.channels_last)
    pred = net.forward(x)
    out_in, out_pred = x.clone(), pred.clone()
    B.scale_shift(out_in, 0.5, 0.5)
    B.scale_shift(out_pred, 0.5, 0.5)
    return out_in, out_pred


class WholeImageInpainter:
    """test_vid_wholeim.lua:150-226 with every tile in one batch."""

    def __init__(self, net, predLen, inputLen=1, fineSize=128, nc=3, netI=None):
        assert predLen % inputLen == 0, "I don't do padding in time dim (test_vid_wholeim.lua:40)"
        self.net, self.netI = net, netI
        self.predLen, self.inputLen, self.fs, self.nc = predLen, inputLen, fineSize, nc
        net.evaluate()
        if netI is not None:
            netI.evaluate()

    def __call__(self, fullImages, padmask, mid_mask=None):
        """fullImages: (predLen*nc) x outh x outw planar in [-1,1], padded bottom-right to multiples of fineSize
        (:137-141); padmask: nc x outh x outw Byte (:209-212).  Returns (outImages, inpaintImages, fullImages) in
        [0,1] (:222-224), predLen x nc x outh x outw."""
        B = get_backend()
        fs, nc, predLen = self.fs, self.nc, self.predLen
        full = B.from_host(fullImages).float().contiguous()
        C, H, W = full.shape
        assert C == nc * predLen and H % fs == 0 and W % fs == 0
        G = predLen // self.inputLen                  # opt.batchSize = predLen / inputLen (:41)
        ncin = nc * self.inputLen
        TY, TX = H // fs, W // fs
        # :167 — tiles (h == 1, w in {1, fs+1, 2fs+1}) are flipped vertically on the way in and out
        flips = torch.zeros(TY * TX, dtype=torch.uint8)
        flips[:min(3, TX)] = 1
        flips = B.from_host(flips)
        tiles = nhwc_empty(TY * TX * G, ncin, fs, fs, full.device)
        B.tiles_gather(full, tiles, G, flips)
        if self.netI is None:
            out_tiles = self.net.forward(tiles)
        else:                                         # :181-190: initializer net, fillIn, then the generator
            assert self.inputLen == 1, "inpaint_utils.fillIn: the mask must have as many channels as a batch row"
            mid = self.netI.forward(tiles)
            mm = B.from_host(mid_mask).float().contiguous()
            tmask = nhwc_empty(TY * TX, nc, fs, fs, full.device)
            # :183-189 — the mask tile is sliced UN-flipped (`mid_mask[{{}, {h, h+fs-1}, {w, w+fs-1}}]`) and applied to the
            # flipped patch: the script flips the image tile only, and so does this
            B.tiles_gather(mm, tmask, 1, None)
            tmask = tmask.repeat_interleave(G, dim=0).contiguous(memory_format=torch.channels_last)
            filled = torch.empty_like(tiles)
            B.masked_compose(filled, tiles, mid, tmask)
            out_tiles = self.net.forward(filled)
        ncout = out_tiles.shape[1]
        assert G * ncout == predLen * nc, "out_image:view(predLen, nc, fs, fs) (:200) needs %d output channels" % (predLen * nc // G)
        out = B.empty(G * ncout, H, W)
        B.tiles_scatter(out_tiles, out, G, flips)
        outImages = out.view(predLen, nc, H, W)
        pm = B.from_host(padmask).float().contiguous()
        pm = pm.unsqueeze(0).expand(predLen, nc, H, W).contiguous()
        inpaint = torch.empty_like(outImages)
        B.masked_compose(inpaint, full.view(predLen, nc, H, W), outImages, pm)      # :214-220
        fullv = full.clone()
        for t in (outImages, inpaint, fullv):         # :222-224
            B.scale_shift(t, 0.5, 0.5)
        return outImages, inpaint, fullv.view(predLen, nc, H, W)
