--[[ hipnn.lua — LuaJIT/Torch7 binding of the gfx950 backend (include/vf_hip.h).

  STATUS: written against the Torch7 API, NOT executed — this build environment has no Lua/LuaJIT/Torch7 and no
  network (SURVEY.md D7).  The same C-ABI is exercised end to end by the Python mirror in video-filler_amd/nn.py,
  which implements exactly the protocol below; INTEGRATION.md walks through the mapping.

  What it provides (the protocol the reference drivers use, SURVEY.md 8(b)):
    hipnn.SpatialConvolution / SpatialFullConvolution / SpatialBatchNormalization / LeakyReLU / ReLU / Tanh / Sigmoid
      with :updateOutput, :updateGradInput, :accGradParameters, fields weight/bias/gradWeight/gradBias/output/gradInput
    hipnn.BCECriterion / MSECriterion / GDLCriterion / MaskedMSECriterion
    hipnn.adam(opfunc, x, state)                     -- optim.adam call shape
    hipnn.convert(net)  (exported as util.hip)       -- the util.cudnn(net) analogue (util.lua:108-131)

  Device memory: tensors are torch.HipTensor-like userdata holding a device pointer + sizes; in a Torch7 tree this
  role is played by cutorch's CudaTensor built for ROCm.  The binding only needs `:data()` (device pointer) and
  sizes, so any tensor type whose storage lives in HBM works.  Layout is channels-last (see vf_hip.h); `toNHWC` /
  `toNCHW` wrap vf_nchw_to_nhwc / vf_nhwc_to_nchw for tensors that arrive in Torch's NCHW.
]]--
local ffi = require 'ffi'

ffi.cdef[[
typedef struct vf_ctx vf_ctx;
const char* vf_last_error(void);
int vf_ctx_create(vf_ctx** out, int device, void* stream);
int vf_ctx_destroy(vf_ctx* ctx);
int vf_ctx_set_workspace(vf_ctx* ctx, void* ptr, size_t bytes);
size_t vf_workspace_bytes_hint(void);
int vf_stream_synchronize(vf_ctx* ctx);
int vf_malloc(void** out, size_t bytes);
int vf_free(void* ptr);
int vf_memcpy_h2d(vf_ctx* ctx, void* dst, const void* src, size_t bytes);
int vf_memcpy_d2h(vf_ctx* ctx, void* dst, const void* src, size_t bytes);
int vf_zero(vf_ctx* ctx, void* ptr, size_t bytes);
int vf_zero_segments(vf_ctx* ctx, float* base, const int64_t* offs, const int64_t* lens, int nseg);
int vf_nchw_to_nhwc(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W);
int vf_nhwc_to_nchw(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W);
int vf_conv2d_fwd(vf_ctx*, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
int vf_conv2d_bwd_data(vf_ctx*, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int vf_conv2d_bwd_weight(vf_ctx*, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_deconv2d_fwd(vf_ctx*, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
int vf_deconv2d_bwd_data(vf_ctx*, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int vf_deconv2d_bwd_weight(vf_ctx*, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_bn_train_fwd(vf_ctx*, const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums, int64_t npix, int C, float momentum, float eps, int act, float slope);
int vf_bn_eval_fwd(vf_ctx*, const float* x, float* y, const float* gamma, const float* beta, const float* running_mean, const float* running_var, int64_t npix, int C, float eps, int act, float slope);
int vf_bn_bwd(vf_ctx*, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma, float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums, int64_t npix, int C, int act, float slope, float pbeta);
int vf_act_fwd(vf_ctx*, const float* x, float* y, int64_t n, int act, float slope);
int vf_act_bwd(vf_ctx*, const float* y, const float* gy, float* gx, int64_t n, int act, float slope);
int vf_axpby(vf_ctx*, float a, const float* x, float b, float* y, int64_t n);
int vf_cmul(vf_ctx*, const float* x, float* y, int64_t n);
int vf_scale_shift(vf_ctx*, float* y, float a, float b, int64_t n);
int vf_masked_compose(vf_ctx*, float* out, const float* real, const float* fake, const float* mask, int64_t n);
int vf_bce_fwd(vf_ctx*, const float* x, float label, int n, double* loss);
int vf_bce_bwd(vf_ctx*, const float* x, float label, float* gx, int n);
int vf_mse_fwd(vf_ctx*, const float* x, const float* t, int64_t n, double* loss);
int vf_mse_bwd(vf_ctx*, const float* x, const float* t, float* gx, int64_t n);
int vf_gdl_fwd(vf_ctx*, const float* yhat, const float* y, int B, int H, int W, int C, double* loss);
int vf_masked_mse_fwd(vf_ctx*, const float* x, const float* xhat, const uint8_t* mask, float w, int64_t n, double* loss);
int vf_masked_mse_bwd(vf_ctx*, const float* x, const float* xhat, const uint8_t* mask, float w, float* gx, int64_t n);
int vf_adam_step(vf_ctx*, float* x, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int32_t* t_dev);
]]

local C = ffi.load(os.getenv('VF_HIP_LIB') or 'libvf_hip.so')
local hipnn = { C = C }
local ACT = { none = 0, lrelu = 1, relu = 2, tanh = 3, sigmoid = 4 }

-- Torch7 natives raise THError -> error(); every vf_* returns non-zero on failure.
local function check(rc) if rc ~= 0 then error(ffi.string(C.vf_last_error()), 2) end end

local ctxp = ffi.new('vf_ctx*[1]')
function hipnn.init(device)          -- cutorch.setDevice(opt.gpu)  (train.lua:249; Torch counts devices from 1)
   check(C.vf_ctx_create(ctxp, (device or 1) - 1, nil))
   hipnn.ctx = ctxp[0]
   local ws = ffi.new('void*[1]')
   local n = C.vf_workspace_bytes_hint()
   check(C.vf_malloc(ws, n))
   check(C.vf_ctx_set_workspace(hipnn.ctx, ws[0], n))
end

local function fptr(t) return ffi.cast('float*', t:data()) end

---------------------------------------------------------------------------------------------------------------
-- nn.SpatialConvolution replacement.  Constructor signature identical to nn.SpatialConvolution so that
-- hipnn.convert can rebuild a module from (nInputPlane, nOutputPlane, kW, kH, dW, dH, padW, padH) exactly as
-- util.cudnn does for cudnn.SpatialConvolution (util.lua:117-119).
---------------------------------------------------------------------------------------------------------------
local Conv, parent = torch.class('hipnn.SpatialConvolution', 'nn.Module')
function Conv:__init(nIn, nOut, kW, kH, dW, dH, padW, padH)
   parent.__init(self)
   self.nInputPlane, self.nOutputPlane = nIn, nOut
   self.kW, self.kH, self.dW, self.dH = kW, kH, dW or 1, dH or 1
   self.padW, self.padH = padW or 0, padH or 0
   -- logical nOut x nIn x kH x kW like nn.SpatialConvolution; physical channels-last (strides permuted)
   self.weight = hipnn.Tensor(nOut, kH, kW, nIn):permute(1, 4, 2, 3)
   self.gradWeight = hipnn.Tensor(nOut, kH, kW, nIn):permute(1, 4, 2, 3)
   self.bias, self.gradBias = hipnn.Tensor(nOut), hipnn.Tensor(nOut)
   self.fusedAct, self.fusedSlope = 'none', 0      -- set by hipnn.fuse(net) for a following in-place activation
end
function Conv:updateOutput(input)                    -- input: B x C x H x W, channels-last storage
   local B, H, W = input:size(1), input:size(3), input:size(4)
   local Ho = math.floor((H + 2 * self.padH - self.kH) / self.dH) + 1
   local Wo = math.floor((W + 2 * self.padW - self.kW) / self.dW) + 1
   self.output = hipnn.resizeNHWC(self.output, B, self.nOutputPlane, Ho, Wo)
   check(C.vf_conv2d_fwd(hipnn.ctx, fptr(input), fptr(self.weight), fptr(self.bias), fptr(self.output), B, H, W,
                         self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH, ACT[self.fusedAct], self.fusedSlope))
   return self.output
end
function Conv:updateGradInput(input, gradOutput)
   local B, H, W = input:size(1), input:size(3), input:size(4)
   self.gradInput = hipnn.resizeNHWC(self.gradInput, B, self.nInputPlane, H, W)
   check(C.vf_conv2d_bwd_data(hipnn.ctx, fptr(gradOutput), fptr(self.weight), fptr(self.gradInput), B, H, W,
                              self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH))
   return self.gradInput
end
function Conv:accGradParameters(input, gradOutput, scale)
   assert((scale or 1) == 1)
   local B, H, W = input:size(1), input:size(3), input:size(4)
   check(C.vf_conv2d_bwd_weight(hipnn.ctx, fptr(input), fptr(gradOutput), fptr(self.gradWeight), fptr(self.gradBias), B, H, W,
                                self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH, 1.0))   -- beta = 1: Torch accumulates
end

-- hipnn.SpatialFullConvolution, hipnn.SpatialBatchNormalization, activations and criteria follow the same pattern over
-- vf_deconv2d_*, vf_bn_train_fwd / vf_bn_eval_fwd / vf_bn_bwd, vf_act_* and vf_bce_* / vf_mse_* / vf_gdl_fwd /
-- vf_masked_mse_* — one C call per Module method; video-filler_amd/nn.py is the executable statement of each.

---------------------------------------------------------------------------------------------------------------
-- util.hip(net): the backend swap.  Same walk as util.cudnn (util.lua:108-125): recurse into containers, replace by
-- exact type string, construct with the same 8 arguments, copy weight/bias (through vf_nchw_to_nhwc, since
-- nn weights are NCHW-contiguous).
---------------------------------------------------------------------------------------------------------------
local REPLACE = {
   ['nn.SpatialConvolution'] = function(l) return hipnn.SpatialConvolution(l.nInputPlane, l.nOutputPlane, l.kW, l.kH, l.dW, l.dH, l.padW, l.padH) end,
   ['nn.SpatialFullConvolution'] = function(l) return hipnn.SpatialFullConvolution(l.nInputPlane, l.nOutputPlane, l.kW, l.kH, l.dW, l.dH, l.padW, l.padH) end,
   ['nn.SpatialBatchNormalization'] = function(l) return hipnn.SpatialBatchNormalization(l.running_mean:size(1), l.eps, l.momentum, l.affine) end,
}
local function recursiveHip(net)
   for k, l in ipairs(net.modules) do
      if net.modules[k].modules ~= nil then recursiveHip(net.modules[k]) end
      local make = REPLACE[torch.type(l)]
      if make then
         local new = make(l)
         hipnn.copyParameterNCHW(new.weight, l.weight)
         new.bias:copy(l.bias)
         if l.running_mean then new.running_mean:copy(l.running_mean); new.running_var:copy(l.running_var) end
         net.modules[k] = new
      end
   end
end
function hipnn.convert(net) recursiveHip(net); return net end

-- optim.adam(opfunc, x, state) with the fused device update (vf_adam_step); state.t lives on the device.
function hipnn.adam(opfunc, x, state)
   local fx, dfdx = opfunc(x)
   state.m = state.m or x.new(dfdx:size()):zero()
   state.v = state.v or x.new(dfdx:size()):zero()
   state.t_dev = state.t_dev or hipnn.IntTensor(2):zero()
   check(C.vf_adam_step(hipnn.ctx, fptr(x), fptr(dfdx), fptr(state.m), fptr(state.v), x:nElement(),
                        state.learningRate or 0.001, state.beta1 or 0.9, state.beta2 or 0.999, state.epsilon or 1e-8,
                        ffi.cast('int32_t*', state.t_dev:data())))
   return x, { fx }
end

return hipnn
