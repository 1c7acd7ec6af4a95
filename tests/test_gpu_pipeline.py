"""GPU parity of the data formats either side of the closures: batch preparation kernels and
the whole-image inference tile loop, against the oracle's restatements of train.lua:284-298,
datavid/donkey_folder.lua:135-189 and test_vid_wholeim.lua:150-226.  Pure data movement and one affine map:
bit-exact, except through the generator (fp32 tolerance 5e-5 of the output range, as for the evaluate-mode test)."""
import numpy as np
import pytest
import torch

from helpers import rel_err, to_np

pytestmark = pytest.mark.gpu


def test_center_prepare_bit_exact(oracle, hipb):
    from video_filler_amd import data
    rng = np.random.default_rng(3)
    for B, fs, ov in ((3, 128, 4), (2, 128, 0), (1, 64, 2)):
        batch = oracle.synth_center_batch(B, rng, 3, fs)
        want_ctx, want_center = oracle.center_prepare(batch, ov)
        ctx, center = data.center_prepare(torch.from_numpy(batch), ov)
        assert ctx.permute(0, 2, 3, 1).is_contiguous() and center.permute(0, 2, 3, 1).is_contiguous()
        np.testing.assert_array_equal(to_np(ctx), want_ctx)
        np.testing.assert_array_equal(to_np(center), want_center)


@pytest.mark.parametrize("mode", ["mask", "blocks"])
@pytest.mark.parametrize("flip", [False, True])
def test_clip_prepare_bit_exact(mode, flip, oracle, hipb):
    from video_filler_amd.data import ClipBatcher
    rng = np.random.default_rng(11)
    C, iH, iW, fs = 12, 150, 200, 128
    clip = rng.uniform(0, 1, (C, iH, iW)).astype(np.float32)
    mask = np.zeros((1, iH, iW), np.uint8)
    if mode == "mask":
        mask[:, 40:90, 60:140] = 1
        dec = dict(w1=37, h1=11, flip=flip, blocks=None, blockSize=fs // 6)
    else:
        dec = dict(w1=72, h1=22, flip=flip, blocks=[(3, 3), (50, 77), (105, 105), (60, 9)], blockSize=fs // 6)
    want_out, want_mask, want_masked = oracle.clip_train_hook(clip, mask, fs, dec["w1"], dec["h1"], flip, 110.0 / 255.0,
                                                              dec["blocks"], dec["blockSize"])
    cb = ClipBatcher(2, C, fs)
    assert cb.add(clip, mask, dec) and cb.add(clip, mask, dec)
    ctx, full, m = cb.batch()
    for row in (0, 1):
        np.testing.assert_array_equal(to_np(full)[row], want_out)
        np.testing.assert_array_equal(to_np(ctx)[row], want_masked)
        np.testing.assert_array_equal(to_np(m)[row], want_mask.astype(np.float32))
    assert want_mask.sum() > 0


def test_clip_batcher_draws_feed_the_trainer(oracle, hipb):
    """The batcher's own draws (crop, rejection, blocks, flip) produce a batch VidTrainer accepts, inside the ranges the
    hook allows (datavid/donkey_folder.lua:114-129,145-178)."""
    from video_filler_amd.data import ClipBatcher
    from video_filler_amd.trainers import VidTrainer
    rng = np.random.default_rng(5)
    C, fs = 6, 128
    cb = ClipBatcher(3, C, fs, rng=np.random.default_rng(7))
    mask = np.zeros((1, 140, 180), np.uint8)
    while cb.n < 3:
        clip = rng.uniform(0.2, 1, (C, 140, 180)).astype(np.float32)
        d = cb.draw(clip, mask)
        assert d is not None and 0 <= d["w1"] <= 180 - fs and 0 <= d["h1"] <= 140 - fs
        assert 2 <= len(d["blocks"]) <= 10 and all(3 <= x <= fs - fs // 6 - 2 and 3 <= y <= fs - fs // 6 - 2 for x, y in d["blocks"])
        cb.add(clip, mask, d)
    ctx, full, m = cb.batch()
    mm = to_np(m)
    assert set(np.unique(mm)) == {0.0, 1.0}
    np.testing.assert_array_equal(to_np(ctx)[mm == 1], np.float32(110.0 / 255.0) * np.float32(2) + np.float32(-1))   # FloatTensor arithmetic: fill, then mul(2):add(-1)
    np.testing.assert_array_equal(to_np(ctx)[mm == 0], to_np(full)[mm == 0])
    tr = VidTrainer(dict(nBottleneck=64, predLen=2, nef=8, ngf=8, ndf=8), seed=2)
    tr.set_batch(ctx, full, m)
    tr.step()
    assert np.isfinite(list(v for v in tr.losses().values() if v is not None)).all()


def test_dark_crops_are_rejected(oracle, hipb):
    from video_filler_amd.data import ClipBatcher
    cb = ClipBatcher(1, 3, 128, rng=np.random.default_rng(0))
    dark = np.full((3, 130, 130), 0.01, np.float32)
    res = [cb.draw(dark, np.zeros((1, 130, 130), np.uint8)) for _ in range(200)]
    kept = sum(r is not None for r in res)
    assert 0 < kept < 40          # kept with probability 0.05 (donkey_folder.lua:148-153)


def test_clip_prepare_rejects_bad_geometry(hipb):
    x = torch.zeros(3, 64, 64, device=hipb.device)
    out = [hipb.empty_act(1, 3, 128, 128) for _ in range(3)]
    with pytest.raises(RuntimeError, match="outside"):
        hipb.clip_prepare(x, x[0], *out, 0, 0, False, 0.4)
    big = torch.zeros(3, 200, 200, device=hipb.device)
    with pytest.raises(RuntimeError, match="leaves"):
        hipb.clip_prepare(big, None, *out, 0, 0, False, 0.4, [(120, 3)], 21)
    with pytest.raises(RuntimeError, match="multiples of fineSize"):
        hipb.tiles_gather(torch.zeros(3, 130, 256, device=hipb.device), hipb.empty_act(2, 3, 128, 128), 1)


@pytest.mark.parametrize("groups,nc", [(1, 3), (4, 6), (2, 27)])
def test_tiles_gather_scatter_roundtrip_and_layout(groups, nc, oracle, hipb):
    rng = np.random.default_rng(1)
    fs, H, W = 128, 256, 512
    full = rng.standard_normal((groups * nc, H, W)).astype(np.float32)
    T = (H // fs) * (W // fs)
    flips = np.zeros(T, np.uint8)
    flips[:3] = 1
    f_d, fl_d = torch.from_numpy(full).to(hipb.device), torch.from_numpy(flips).to(hipb.device)
    tiles = hipb.empty_act(T * groups, nc, fs, fs)
    hipb.tiles_gather(f_d, tiles, groups, fl_d)
    got = to_np(tiles)
    for t in range(T):
        ty, tx = divmod(t, W // fs)
        for g in range(groups):
            want = full[g * nc:(g + 1) * nc, ty * fs:(ty + 1) * fs, tx * fs:(tx + 1) * fs]
            if flips[t]:
                want = want[:, ::-1]
            np.testing.assert_array_equal(got[t * groups + g], want)
    back = torch.empty_like(f_d)
    hipb.tiles_scatter(tiles, back, groups, fl_d)
    np.testing.assert_array_equal(to_np(back), full)


def _small_netG(oracle, hipb, nc_in, nc_out, seed):
    """A small generator whose OUTPUT DEPENDS ON ITS INPUT in evaluate() mode: with weights_init's N(0, 0.02) and running
    variances near 1 the signal dies layer by layer and the net returns the same image whatever it is fed (a constant
    function passes any gather / flip / fill-in test).  Convolution weights are therefore scaled to ~1/sqrt(fan_in)."""
    from video_filler_amd.trainers import build_netG
    rng = np.random.default_rng(seed)
    ref = oracle.build_netG(nc_in, nc_out, 8, 8, 32, True)
    oracle.weights_init(ref, rng)

    def widen(m):
        if "Convolution" in m.type_name():
            m.weight *= np.float32(6.0)

    ref.apply(widen)
    pref, _ = ref.getParameters()
    net = build_netG(nc_in, nc_out, 8, 8, 32, True)
    net.getParameters()
    net.load_reference_flat(torch.from_numpy(pref.copy()).to(hipb.device))
    from test_gpu_trainers import _leaves
    rb = [m for m in _leaves(ref) if hasattr(m, "running_mean")]
    hb = [m for m in net.leaves() if hasattr(m, "running_mean")]
    for a, b in zip(rb, hb):
        a.running_mean[...] = 0.1 * rng.standard_normal(a.running_mean.shape).astype(np.float32)
        a.running_var[...] = (1 + 0.3 * rng.random(a.running_var.shape)).astype(np.float32)
        b.running_mean.copy_(torch.from_numpy(a.running_mean).to(hipb.device))
        b.running_var.copy_(torch.from_numpy(a.running_var).to(hipb.device))
    ref.evaluate()
    net.evaluate()
    return ref, net


@pytest.mark.parametrize("inputLen,with_init", [(1, False), (2, False), (1, True)])
def test_whole_image_inpainting_matches_the_tile_loop(inputLen, with_init, oracle, hipb):
    """One batched forward over all tiles == test_vid_wholeim.lua's per-tile loop (incl. the vflip rule, the
    initializer-net path and the masked paste), on a 256 x 512 padded clip (2 x 4 tiles)."""
    from video_filler_amd.inference import WholeImageInpainter
    rng = np.random.default_rng(21)
    predLen, nc, fs, H, W = 4, 3, 128, 256, 512
    ncin = nc * inputLen
    ref, net = _small_netG(oracle, hipb, ncin, ncin, 5)
    refI = netI = None
    if with_init:
        refI, netI = _small_netG(oracle, hipb, ncin, ncin, 6)
    full = rng.uniform(-1, 1, (predLen * nc, H, W)).astype(np.float32)
    padmask = np.zeros((nc, H, W), np.uint8)
    padmask[:, 60:200, 100:420] = 1
    # the fill-in mask is NOT symmetric under a vertical flip inside the three flipped tiles of the top row, so a mask
    # tile gathered with the image's flip (the reference slices it un-flipped, test_vid_wholeim.lua:183) is caught
    mid_mask = np.zeros((nc, H, W), np.uint8)
    mid_mask[:, 8:50, 20:330] = 1
    mid_mask[:, 150:230, 300:480] = 1
    want_out, want_inp, want_full = oracle.whole_image_inpaint(ref, full, padmask, predLen, inputLen, fs, nc, refI, mid_mask)
    # the oracle's answer must react to its inputs by far more than the tolerance below, or this test tests nothing:
    # (a) another image in one tile, (b) with the initializer path, the mask flipped inside the flipped tiles
    full2 = full.copy()
    full2[:, 128:256, 128:256] = rng.uniform(-1, 1, (predLen * nc, 128, 128)).astype(np.float32)
    alt = oracle.whole_image_inpaint(ref, full2, padmask, predLen, inputLen, fs, nc, refI, mid_mask)[0]
    assert rel_err(alt[:, :, 128:256, 128:256], want_out[:, :, 128:256, 128:256]) > 1e-2
    if with_init:
        mm2 = mid_mask.copy()
        mm2[:, 0:128, 0:384] = mid_mask[:, 127::-1, 0:384]
        alt = oracle.whole_image_inpaint(ref, full, padmask, predLen, inputLen, fs, nc, refI, mm2)[0]
        assert rel_err(alt, want_out) > 1e-2
        alt = oracle.whole_image_inpaint(ref, full, padmask, predLen, inputLen, fs, nc, None, None)[0]
        assert rel_err(alt, want_out) > 1e-2
    run = WholeImageInpainter(net, predLen, inputLen, fs, nc, netI)
    out, inp, fullv = run(torch.from_numpy(full), torch.from_numpy(padmask), torch.from_numpy(mid_mask))
    assert tuple(out.shape) == (predLen, nc, H, W)
    assert rel_err(to_np(out), want_out) < 5e-5
    assert rel_err(to_np(inp), want_inp) < 5e-5
    np.testing.assert_array_equal(to_np(fullv).reshape(want_full.shape), want_full)
    # outside the mask the inpainted frames ARE the input frames
    keep = np.broadcast_to(padmask == 0, (predLen,) + padmask.shape)
    np.testing.assert_array_equal(to_np(inp)[keep], want_full.reshape(predLen, nc, H, W)[keep])


def test_predict_clip_matches_oracle(oracle, hipb):
    """test_vid.lua:47-48,100-114."""
    from video_filler_amd.inference import predict_clip
    rng = np.random.default_rng(9)
    ref, net = _small_netG(oracle, hipb, 3, 3, 8)
    x = rng.uniform(-1, 1, (5, 3, 128, 128)).astype(np.float32)
    want = (np.asarray(ref.forward(x.copy())) + np.float32(1)) * np.float32(0.5)
    got_in, got = predict_clip(net, torch.from_numpy(x))
    assert rel_err(to_np(got), want) < 5e-5
    np.testing.assert_array_equal(to_np(got_in), (x + np.float32(1)) * np.float32(0.5))


@pytest.mark.parametrize("normal", [True, False])
def test_noise_fill_matches_the_oracle_restatement(normal, oracle, hipb):
    """vf_noise_fill (train.lua:319-323's noise:normal / noise:uniform, counter-based): same integers, so uniforms are
    bit-exact; the normal draw differs only by the libm's logf/cosf rounding.  Keyed by (seed, counter): a device-side
    counter gives the same stream as the host-side one."""
    import torch
    out = hipb.zeros(7 * 100)
    hipb.noise_fill(out, 99, counter=5, normal=normal)
    want = oracle.noise_fill((700,), 99, 5, normal)
    got = to_np(out)
    if normal:
        assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max())
        assert abs(got.mean()) < 0.15 and 0.85 < got.std() < 1.15
    else:
        assert np.array_equal(got, want) and got.min() >= -1 and got.max() < 1
    ctr = hipb.from_host(torch.tensor([5, 0], dtype=torch.int32))
    out2 = hipb.zeros(700)
    hipb.noise_fill(out2, 99, counter=123, normal=normal, counter_dev=ctr)
    assert torch.equal(out, out2)
    hipb.noise_fill(out2, 99, counter=6, normal=normal)
    assert not torch.equal(out, out2)


def test_channel_copy_is_join_table(hipb):
    import torch
    a, b = hipb.empty_act(3, 5, 4, 4), hipb.empty_act(3, 2, 4, 4)
    a.copy_(torch.randn(3, 5, 4, 4)); b.copy_(torch.randn(3, 2, 4, 4))
    y = hipb.empty_act(3, 7, 4, 4)
    hipb.channel_copy(a, 0, y, 0, 5)
    hipb.channel_copy(b, 0, y, 5, 2)
    assert torch.equal(y, torch.cat([a, b], dim=1))
    back = hipb.empty_act(3, 2, 4, 4)
    hipb.channel_copy(y, 5, back, 0, 2)
    assert torch.equal(back, b)
    with pytest.raises(RuntimeError):
        hipb.channel_copy(y, 6, back, 0, 2)
