// vf_pipeline.hip — the data formats either side of the hot path, as HBM-bound byte movers:
//   * batch preparation the reference does on the host per sample before the closures run
//     (train.lua:284-298 centre hole; datavid/donkey_folder.lua:135-189 crop / mask / fill / hflip / [0,1] -> [-1,1]),
//     written straight into the NHWC buffers the kernels read;
//   * the whole-image inference tile loop of test_vid_wholeim.lua:159-205 (gather fineSize tiles of the padded
//     planar clip into ONE NHWC batch, scatter the net's output tiles back), incl. its per-tile vertical-flip rule.
// Every kernel is one pass: each output element is written once, each input element read at most once per output
// that needs it.  Threads run along the NHWC channel axis (fastest) so the stores are fully coalesced; the planar
// side is read in C strided streams that stay in L2 (the planes are a few MB).
#include "vf_common.h"

namespace {

inline int pgrid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(n, 256), 1 << 20)); }

// ---------------------------------------------------------------------------------------------------------------
// train.lua:284-298.  batch: B x C x fs x fs planar in [-1,1] (the loader's tensor).
//   center = batch[:, :, fs/4 : 3fs/4, fs/4 : 3fs/4]                  (clone BEFORE painting, :285)
//   ctx    = batch with [fs/4+ov, 3fs/4-ov)^2 of channel c set to fill[c]  (:287-289)
__global__ void k_center_prepare(const float* __restrict__ batch, float* __restrict__ ctx_out, float* __restrict__ center,
                                 const float* __restrict__ fill, int B, int C, int fs, int ov) {
  const int64_t n = (int64_t)B * fs * fs * C;
  const int lo = fs / 4, hi = fs / 2 + fs / 4, cs = fs / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int x = (int)(t % fs);
    t /= fs;
    const int y = (int)(t % fs);
    const int b = (int)(t / fs);
    const float v = batch[(((int64_t)b * C + c) * fs + y) * fs + x];
    const bool hole = y >= lo + ov && y < hi - ov && x >= lo + ov && x < hi - ov;
    ctx_out[i] = hole ? fill[c] : v;
    if (y >= lo && y < hi && x >= lo && x < hi) center[(((int64_t)b * cs + (y - lo)) * cs + (x - lo)) * C + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// datavid/donkey_folder.lua:135-189 (trainHook with withMask) for ONE sample.
//   clip: C x iH x iW planar in [0,1] (decoded, scaled);  mask: iH x iW float (the scaled mask image, one channel)
//   crop (w1, h1) .. +fs (image.crop, :147);  maskout = crop(mask) expanded over C (:162-163)
//   mode 0: masked = out with maskout > 0 filled (maskedFill, :166 — Byte semantics: any non-zero value)
//   mode 1: randomBlockMask (:114-129): nblk square blocks of side bs at 1-based top-left (tlx, tly) are filled and
//           maskout = 1 there, 0 elsewhere
//   hflip (:178-183) mirrors all three;  out/masked -> 2x-1 (:185-187); maskout stays 0/1.
struct ClipPrep {
  const float* clip;
  const float* mask;
  float* full;      // fs x fs x C  NHWC, [-1,1]
  float* masked;    // fs x fs x C
  float* maskout;   // fs x fs x C  (0/1 as float: the trainers copy the Byte mask into a Float tensor, :394)
  int C, iH, iW, fs, w1, h1, flip, mode, nblk, bs;
  float mask_value;
  int tlx[10], tly[10];
};
__global__ void k_clip_prepare(const ClipPrep p) {
  const int64_t n = (int64_t)p.fs * p.fs * p.C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % p.C);
    const int64_t t = i / p.C;
    const int x = (int)(t % p.fs), y = (int)(t / p.fs);
    const int sx = p.flip ? p.fs - 1 - x : x;          // column of the un-flipped crop
    const int gy = p.h1 + y, gx = p.w1 + sx;
    const float v = p.clip[((int64_t)c * p.iH + gy) * p.iW + gx];
    bool m;
    if (p.mode == 0) {
      m = p.mask[(int64_t)gy * p.iW + gx] != 0.f;
    } else {
      m = false;
      for (int k = 0; k < p.nblk; ++k) {
        const int bx = p.tlx[k] - 1, by = p.tly[k] - 1;   // 1-based Lua coordinates
        m = m || (sx >= bx && sx < bx + p.bs && y >= by && y < by + p.bs);
      }
    }
    p.full[i] = 2.f * v - 1.f;
    p.masked[i] = 2.f * (m ? p.mask_value : v) - 1.f;
    p.maskout[i] = m ? 1.f : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// test_vid_wholeim.lua:159-205.  full: (G*nc) x H x W planar (H, W multiples of fs).  Tile (ty, tx), group g ->
// batch row (ty*TX + tx)*G + g, NHWC fs x fs x nc.  vflip[t] != 0: the tile is flipped vertically on the way in
// (:167-170) and its output flipped back on the way out (:191-197).
__global__ void k_tiles_gather(const float* __restrict__ full, float* __restrict__ tiles, int G, int nc, int H, int W,
                               int fs, const unsigned char* __restrict__ vflip) {
  const int TX = W / fs;
  const int64_t n = (int64_t)(H / fs) * TX * G * fs * fs * nc;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nc);
    int64_t t = i / nc;
    const int x = (int)(t % fs);
    t /= fs;
    const int y = (int)(t % fs);
    t /= fs;
    const int g = (int)(t % G);
    const int tile = (int)(t / G);
    const int ty = tile / TX, tx = tile - ty * TX;
    const int sy = (vflip && vflip[tile]) ? fs - 1 - y : y;
    tiles[i] = full[((int64_t)(g * nc + c) * H + ty * fs + sy) * W + tx * fs + x];
  }
}
__global__ void k_tiles_scatter(const float* __restrict__ tiles, float* __restrict__ out, int G, int nc, int H, int W,
                                int fs, const unsigned char* __restrict__ vflip) {
  const int TX = W / fs;
  const int64_t n = (int64_t)(H / fs) * TX * G * fs * fs * nc;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nc);
    int64_t t = i / nc;
    const int x = (int)(t % fs);
    t /= fs;
    const int y = (int)(t % fs);
    t /= fs;
    const int g = (int)(t % G);
    const int tile = (int)(t / G);
    const int ty = tile / TX, tx = tile - ty * TX;
    const int dy = (vflip && vflip[tile]) ? fs - 1 - y : y;
    out[((int64_t)(g * nc + c) * H + ty * fs + dy) * W + tx * fs + x] = tiles[i];
  }
}

}  // namespace

VF_API int vf_center_prepare(vf_ctx* ctx, const float* batch_nchw, float* ctx_nhwc, float* center_nhwc,
                             const float* fill, int B, int C, int fs, int overlapPred) {
  VF_REQUIRE(B > 0 && C > 0 && fs >= 4 && fs % 4 == 0, "vf_center_prepare: bad shape B=%d C=%d fineSize=%d", B, C, fs);
  VF_REQUIRE(overlapPred >= 0 && 2 * overlapPred < fs / 2, "vf_center_prepare: overlapPred=%d does not fit", overlapPred);
  const int64_t n = (int64_t)B * C * fs * fs;
  VfProf prof(ctx, "center_prepare", 0.0, 4.0 * (double)n * 2.25);
  hipLaunchKernelGGL(k_center_prepare, dim3(pgrid(n)), dim3(256), 0, ctx->stream, batch_nchw, ctx_nhwc, center_nhwc, fill,
                     B, C, fs, overlapPred);
  VF_LAUNCH_CHECK();
  return 0;
}

VF_API int vf_clip_prepare(vf_ctx* ctx, const float* clip, const float* mask, float* full, float* masked,
                           float* maskout, int C, int iH, int iW, int fs, int w1, int h1, int flip, float mask_value,
                           int nblocks, int block_size, const int* tlx, const int* tly) {
  VF_REQUIRE(C > 0 && fs > 0 && w1 >= 0 && h1 >= 0 && w1 + fs <= iW && h1 + fs <= iH,
             "vf_clip_prepare: crop (%d,%d)+%d outside the %dx%d clip", w1, h1, fs, iW, iH);
  VF_REQUIRE(nblocks >= 0 && nblocks <= 10, "vf_clip_prepare: at most 10 random blocks (donkey_folder.lua:121), got %d", nblocks);
  VF_REQUIRE(nblocks > 0 || mask != nullptr, "vf_clip_prepare: a mask image or random blocks are required");
  ClipPrep p;
  p.clip = clip; p.mask = mask; p.full = full; p.masked = masked; p.maskout = maskout;
  p.C = C; p.iH = iH; p.iW = iW; p.fs = fs; p.w1 = w1; p.h1 = h1; p.flip = flip ? 1 : 0;
  p.mode = nblocks > 0 ? 1 : 0; p.nblk = nblocks; p.bs = block_size; p.mask_value = mask_value;
  for (int k = 0; k < 10; ++k) {
    p.tlx[k] = k < nblocks ? tlx[k] : 0;     // host arrays
    p.tly[k] = k < nblocks ? tly[k] : 0;
    if (k < nblocks)
      VF_REQUIRE(p.tlx[k] >= 1 && p.tly[k] >= 1 && p.tlx[k] - 1 + block_size <= fs && p.tly[k] - 1 + block_size <= fs,
                 "vf_clip_prepare: block %d at (%d,%d) size %d leaves the %d crop", k, p.tlx[k], p.tly[k], block_size, fs);
  }
  const int64_t n = (int64_t)C * fs * fs;
  VfProf prof(ctx, "clip_prepare", 0.0, 4.0 * (double)n * 4);
  hipLaunchKernelGGL(k_clip_prepare, dim3(pgrid(n)), dim3(256), 0, ctx->stream, p);
  VF_LAUNCH_CHECK();
  return 0;
}

static int tiles_check(int G, int nc, int H, int W, int fs) {
  VF_REQUIRE(G > 0 && nc > 0 && fs > 0 && H > 0 && W > 0 && H % fs == 0 && W % fs == 0,
             "tile loop: the %dx%d clip must be padded to multiples of fineSize=%d (test_vid_wholeim.lua:108)", H, W, fs);
  return 0;
}
VF_API int vf_tiles_gather(vf_ctx* ctx, const float* full, float* tiles, int groups, int nc, int H, int W, int fs,
                           const unsigned char* vflip) {
  if (int e = tiles_check(groups, nc, H, W, fs)) return e;
  const int64_t n = (int64_t)groups * nc * H * W;
  VfProf prof(ctx, "tiles_gather", 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(k_tiles_gather, dim3(pgrid(n)), dim3(256), 0, ctx->stream, full, tiles, groups, nc, H, W, fs, vflip);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_tiles_scatter(vf_ctx* ctx, const float* tiles, float* out, int groups, int nc, int H, int W, int fs,
                            const unsigned char* vflip) {
  if (int e = tiles_check(groups, nc, H, W, fs)) return e;
  const int64_t n = (int64_t)groups * nc * H * W;
  VfProf prof(ctx, "tiles_scatter", 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(k_tiles_scatter, dim3(pgrid(n)), dim3(256), 0, ctx->stream, tiles, out, groups, nc, H, W, fs, vflip);
  VF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ table modules, noise
// nn.JoinTable(2) forward / backward in the NHWC layout: dst[p][c_dst + c] = src[p][c_src + c], c < Cc.
__global__ void k_channel_copy(const float* __restrict__ src, int Cs, int cs0, float* __restrict__ dst, int Cd, int cd0, int Cc,
                               int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Cc;
    const int c = (int)(i - p * Cc);
    dst[p * Cd + cd0 + c] = src[p * Cs + cs0 + c];
  }
}
VF_API int vf_channel_copy(vf_ctx* ctx, const float* src, int Csrc, int c_src, float* dst, int Cdst, int c_dst, int Ccopy,
                           int64_t npix) {
  VF_REQUIRE(Ccopy > 0 && c_src >= 0 && c_dst >= 0 && c_src + Ccopy <= Csrc && c_dst + Ccopy <= Cdst && npix > 0,
             "vf_channel_copy: channels [%d,+%d) of %d -> [%d,+%d) of %d", c_src, Ccopy, Csrc, c_dst, Ccopy, Cdst);
  const int64_t n = npix * Ccopy;
  VfProf prof(ctx, "channel_copy", 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(k_channel_copy, dim3(pgrid(n)), dim3(256), 0, ctx->stream, src, Csrc, c_src, dst, Cdst, c_dst, Ccopy, n);
  VF_LAUNCH_CHECK();
  return 0;
}

// noise:uniform(-1, 1) / noise:normal(0, 1) (train.lua:319-323).  Torch7's Mersenne-Twister stream cannot be reproduced;
// element i of draw `counter` is a pure function of (seed, counter, i): splitmix64 of the 64-bit index, 24-bit
// uniforms, Box-Muller for the normal.  The oracle restates it (oracle.noise_fill).
__device__ __forceinline__ uint64_t vf_splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void k_noise_fill(float* __restrict__ out, int64_t n, uint64_t seed, const int32_t* __restrict__ counter_dev,
                             uint64_t counter, int normal) {
  const uint64_t ctr = counter_dev ? (uint64_t)(uint32_t)counter_dev[0] : counter;
  const uint64_t base = vf_splitmix64(seed ^ vf_splitmix64(ctr));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t z = vf_splitmix64(base + (uint64_t)i * 0xD1342543DE82EF95ull);
    const float u1 = (float)((z >> 40) + 1) * 5.9604644775390625e-8f;            // (0, 1]
    const float u2 = (float)((z >> 16) & 0xFFFFFF) * 5.9604644775390625e-8f;     // [0, 1)
    out[i] = normal ? sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2) : 2.f * u2 - 1.f;
  }
}
VF_API int vf_noise_fill(vf_ctx* ctx, float* out, int64_t n, uint64_t seed, const int32_t* counter_dev, uint64_t counter,
                         int normal) {
  VF_REQUIRE(n > 0, "vf_noise_fill: empty tensor");
  VfProf prof(ctx, "noise_fill", 0.0, 4.0 * (double)n);
  hipLaunchKernelGGL(k_noise_fill, dim3(pgrid(n)), dim3(256), 0, ctx->stream, out, n, seed, counter_dev, counter, normal);
  VF_LAUNCH_CHECK();
  return 0;
}
