--[[ hipnn.lua — LuaJIT/Torch7 binding of the gfx950 backend (include/vf_hip.h).

  STATUS: complete against the Torch7 API as published, NOT executed — this build environment has no
  Lua/LuaJIT/Torch7 and no network (SURVEY.md D7), so expect the usual first-run typos.  The same C-ABI is exercised
  end to end by the Python mirror in video-filler_amd/nn.py, which implements exactly the protocol below and is what the
  tests and the bench run; INTEGRATION.md walks through the mapping.

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
int vf_ctx_set_mfma_mode(vf_ctx* ctx, int mode);
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
int vf_conv2d_bwd_data_act(vf_ctx*, const float* gy, const float* w, float* gx, const float* x_act, int act, float slope, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int vf_conv2d_bwd_weight(vf_ctx*, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_deconv2d_fwd(vf_ctx*, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
int vf_deconv2d_bwd_data(vf_ctx*, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int vf_deconv2d_bwd_weight(vf_ctx*, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_bn_train_fwd(vf_ctx*, const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums, int64_t npix, int C, float momentum, float eps, int act, float slope);
int vf_bn_eval_fwd(vf_ctx*, const float* x, float* y, const float* gamma, const float* beta, const float* running_mean, const float* running_var, int64_t npix, int C, float eps, int act, float slope);
int vf_bn_bwd(vf_ctx*, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma, float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums, int64_t npix, int C, int act, float slope, float pbeta);
int vf_bn_train_fwd_groups(vf_ctx*, const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums, int64_t npix_per_group, int C, int groups, float momentum, float eps, int act, float slope);
int vf_bn_bwd_groups(vf_ctx*, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma, float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums, int64_t npix_per_group, int C, int groups, int act, float slope, float pbeta);
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
int vf_gdl_bwd(vf_ctx*, const float* yhat, const float* y, float* gyhat, int B, int H, int W, int C);
int vf_masked_mse_fwd(vf_ctx*, const float* x, const float* xhat, const uint8_t* mask, float w, int64_t n, double* loss);
int vf_masked_mse_bwd(vf_ctx*, const float* x, const float* xhat, const uint8_t* mask, float w, float* gx, int64_t n);
int vf_adam_step(vf_ctx*, float* x, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int32_t* t_dev);
int vf_adam_prep(vf_ctx*, double lr, double beta1, double beta2, int32_t* t_dev);
int vf_adam_apply(vf_ctx*, float* x, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps, const int32_t* t_dev);
int vf_adam_apply_ranges(vf_ctx* ctx, float* x, const float* g, float* m, float* v, const int64_t* offsets, const int64_t* lengths, int nranges, double beta1, double beta2, double eps, const int32_t* t_dev);
int vf_conv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_deconv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes, float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
typedef struct vf_layer_desc { int kind; int nin, nout; int k, stride, pad; int act; float slope; float eps, momentum; } vf_layer_desc;
typedef struct vf_net vf_net;
typedef struct vf_comm vf_comm;
int vf_net_create(vf_ctx* ctx, vf_net** out, const vf_layer_desc* layers, int nlayers, int B, int C, int H, int W);
int vf_net_destroy(vf_net* net);
int vf_net_reshape(vf_net* net, int B, int C, int H, int W);
int vf_net_parameters(vf_net* net, float** params, float** grads, int64_t* count);
int vf_net_bind_parameters(vf_net* net, float* params, float* grads, int64_t count);
int64_t vf_net_param_offset(const vf_net* net, int layer, int which, int64_t* length);
int vf_net_bn_running(vf_net* net, int layer, float** running_mean, float** running_var);
int vf_net_bind_bn_running(vf_net* net, int layer, float* running_mean, float* running_var);
int vf_net_training(vf_net* net, int train);
int vf_net_zero_grad(vf_net* net);
int vf_net_zero_conv_biases(vf_net* net, vf_net* other);
int vf_net_forward(vf_net* net, const float* x, const float** y);
int vf_net_backward(vf_net* net, const float* x, const float* gy, const float** gx);
int vf_net_update_grad_input(vf_net* net, const float* x, const float* gy, const float** gx);
int vf_net_set_skip_input_grad(vf_net* net, int on);
int vf_net_set_batch_groups(vf_net* net, int G);
int vf_net_update_grad_input_group(vf_net* net, const float* x, const float* gy, int g, int G, const float** gx);
int vf_net_plan_size(const vf_net* net);
int vf_net_bucket_split(const vf_net* net, double frac, int* plan_index, int64_t* flat_offset);
int vf_net_backward_range(vf_net* net, const float* x, const float* gy, int hi, int lo, int need_input_grad, const float** gx);
int vf_net_set_fused_adam(vf_net* net, int on, int* count);
int vf_net_fused_adam_range(const vf_net* net, int i, int64_t* offset, int64_t* length);
int vf_net_adam_fused(vf_net* net, float* m, float* v, double beta1, double beta2, double eps, const int32_t* t_dev, int keep_grad);
int vf_wgrad_adam_outer_supported(int K, int Nu, int Ncols);
int vf_wgrad_adam_outer_gathered(vf_ctx* ctx, const float* U, const float* V, int K, int rows_per_seg, int64_t seg_stride, int Nu, int Ncols, float* x, float* m, float* v, float* g, float gscale, double beta1, double beta2, double eps, const int32_t* t_dev);
int vf_net_fused_adam_pack_size(const vf_net* net, int64_t* floats);
int vf_net_fused_adam_pack(vf_net* net, float* segment);
int vf_net_adam_fused_gathered(vf_net* net, const float* all_segments, int world, int64_t seg_stride, float* m, float* v, double beta1, double beta2, double eps, const int32_t* t_dev, int keep_grad);
int vf_net_fused_adam_rows_ok(const vf_net* net, int row_world);
int vf_net_adam_fused_gathered_rows(vf_net* net, const float* all_segments, int world, int64_t seg_stride, float* m, float* v, double beta1, double beta2, double eps, const int32_t* t_dev, int keep_grad, int row_rank, int row_world);
int vf_wgrad_adam_outer(vf_ctx* ctx, const float* U, const float* V, int K, int Nu, int Ncols, float* x, float* m, float* v, float* g, double beta1, double beta2, double eps, const int32_t* t_dev);
int vf_net_set_sync_bn(vf_net* net, vf_comm* comm, int world, int force);
int vf_net_set_weight_planes_managed(vf_net* net, int on);
int vf_net_refresh_weight_planes(vf_net* net);
int vf_net_layer_output(vf_net* net, int layer, const float** y);
int vf_net_layer_shape(const vf_net* net, int layer, int* B, int* C, int* H, int* W, int* Co, int* Ho, int* Wo);
int vf_net_bind_output(vf_net* net, int layer, float* y);
int vf_bce_fwd_bwd(vf_ctx* ctx, const float* x, float label0, float label1, int n_per_group, int groups, double* loss0, double* loss1, float* gx);
int vf_trace_available(void);
int vf_trace_enable(int on);
int vf_range_push(const char* name);
int vf_range_pop(void);
int vf_mark(const char* message);
int vf_range_depth(void);
int vf_wgrad_group_begin(vf_ctx* ctx);
int vf_wgrad_group_end(vf_ctx* ctx);
int vf_wgrad_group_abort(vf_ctx* ctx);
int vf_comm_available(void);
int vf_comm_unique_id(void* id128);
int vf_comm_init(vf_comm** out, const void* id128, int world, int rank);
int vf_comm_allreduce_async(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op, int* ticket);
int vf_comm_allreduce_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t n, int* ticket);
int vf_comm_reduce_scatter_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket);
int vf_comm_allgather_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket);
int vf_comm_wait(vf_comm* c, vf_ctx* ctx, int ticket);
int vf_comm_allreduce_inline(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op);
int vf_comm_broadcast(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int root);
int vf_comm_barrier(vf_comm* c, vf_ctx* ctx);
int vf_comm_destroy(vf_comm* c);
int vf_bias_grad_plan(int64_t P, int C, int* cq, int* rows_per_block, int* gx, int* gy);
int vf_bias_grad_multi(vf_ctx*, const void* desc_dev, int n, int blocks1, int blocks2);
int vf_center_prepare(vf_ctx*, const float* batch_nchw, float* ctx_nhwc, float* center_nhwc, const float* fill, int B, int C, int fs, int overlapPred);
int vf_clip_prepare(vf_ctx*, const float* clip, const float* mask, float* full, float* masked, float* maskout, int C, int iH, int iW, int fs, int w1, int h1, int flip, float mask_value, int nblocks, int block_size, const int* tlx, const int* tly);
int vf_tiles_gather(vf_ctx*, const float* full, float* tiles, int groups, int nc, int H, int W, int fs, const unsigned char* vflip);
int vf_tiles_scatter(vf_ctx*, const float* tiles, float* out, int groups, int nc, int H, int W, int fs, const unsigned char* vflip);
int vf_channel_copy(vf_ctx*, const float* src, int Csrc, int c_src, float* dst, int Cdst, int c_dst, int Ccopy, int64_t npix);
int vf_noise_fill(vf_ctx*, float* out, int64_t n, uint64_t seed, const int32_t* counter_dev, uint64_t counter, int normal);
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

-- vf_ctx_set_mfma_mode: 3 (default) fp32 operands as three exact bf16 planes on the bf16 pipe; 0 native f32 MFMA (the
-- reference's fmaf chain bit for bit); 1 operands rounded to bf16 (opt-in)
function hipnn.setMfmaMode(mode) check(C.vf_ctx_set_mfma_mode(hipnn.ctx, mode)) end
-- every weight gradient recorded between these two runs as one grouped launch (wrap net:backward with them)
----------------------------------------------------------------------------------------------------------------
-- hipnn.Net: the documented host path.  util.cudnn(net) is where the reference puts a net on its GPU backend
-- (util.lua:108-131, train.lua:245-258); hipnn.hip(net, B, C, H, W) is its counterpart: the chain of modules netG:add(...) /
-- netD:add(...) built (train.lua:87-199) becomes ONE library object (vf_net_*) that runs the whole fast path inside
-- libvf_hip.so — activations in their producer's epilogue, BatchNorm statistics out of the neighbouring GEMMs, operand planes
-- handed from producer to consumer, grouped weight / bias gradients, netD's real + fake passes as one 2B batch — and every
-- method the drivers call on a net is one C call.  The object keeps Torch7's protocol: :forward / :backward / :updateGradInput
-- return tensors, .output / .gradInput are set, :getParameters() returns the two flat tensors (host-owned storage the library
-- is bound to, so optim.adam and the checkpoint code keep working on them), :apply visits the original modules, whose
-- weight / bias / running_mean / running_var become views of that storage.
-- (The module-by-module classes further down remain for nets with table modules: train.lua's conditionAdv / noiseGen branches.)
----------------------------------------------------------------------------------------------------------------
local KIND = { ['nn.SpatialConvolution'] = 1, ['nn.SpatialFullConvolution'] = 2, ['nn.SpatialBatchNormalization'] = 3,
               ['nn.LeakyReLU'] = 4, ['nn.ReLU'] = 4, ['nn.Tanh'] = 4, ['nn.Sigmoid'] = 4, ['nn.View'] = 5 }
local ACTCODE = { ['nn.LeakyReLU'] = 1, ['nn.ReLU'] = 2, ['nn.Tanh'] = 3, ['nn.Sigmoid'] = 4 }
local fptr          -- (defined below)
local function flatten(seq, out)      -- nested nn.Sequential -> one module list, depth first (the order of getParameters())
   for _, m in ipairs(seq.modules) do
      if torch.type(m) == 'nn.Sequential' then flatten(m, out) else out[#out + 1] = m end
   end
   return out
end
local Net = {}
Net.__index = Net
function hipnn.Net(seq, B, Cc, H, W)
   local mods = flatten(seq, {})
   local n = #mods
   local d = ffi.new('vf_layer_desc[?]', n)
   for i, m in ipairs(mods) do
      local t = torch.type(m)
      local e = d[i - 1]
      e.kind = assert(KIND[t], 'hipnn.Net hosts chains of convolutions, BatchNorms, activations and views; got ' .. t)
      if e.kind <= 2 then e.nin, e.nout, e.k, e.stride, e.pad = m.nInputPlane, m.nOutputPlane, m.kW, m.dW, m.padW
      elseif e.kind == 3 then e.nout, e.eps, e.momentum = m.running_mean:size(1), m.eps, m.momentum
      elseif e.kind == 4 then e.act, e.slope = ACTCODE[t], m.negval or 0 end
   end
   local out = ffi.new('vf_net*[1]')
   check(C.vf_net_create(hipnn.ctx, out, d, n, B, Cc, H, W))
   local self = setmetatable({ h = ffi.gc(out[0], C.vf_net_destroy), modules = mods, seq = seq, B = B, train = true }, Net)
   -- host-owned flat storage in the library's layout; the modules' parameters move into it (channels-last inside a 4-D weight)
   local p, g, c = ffi.new('float*[1]'), ffi.new('float*[1]'), ffi.new('int64_t[1]')
   check(C.vf_net_parameters(self.h, p, g, c))
   self.count = tonumber(c[0])
   self.flat, self.gflat = hipnn.Tensor(self.count):zero(), hipnn.Tensor(self.count):zero()
   check(C.vf_net_bind_parameters(self.h, fptr(self.flat), fptr(self.gflat), self.count))
   local len = ffi.new('int64_t[1]')
   for i, m in ipairs(mods) do
      if m.weight then
         for which, name in ipairs({ 'weight', 'bias' }) do
            local off = tonumber(C.vf_net_param_offset(self.h, i - 1, which - 1, len))
            local view = hipnn.viewOf(self.flat, off, m[name])          -- same logical sizes, channels-last strides for 4-D
            if m[name]:dim() == 4 then hipnn.copyParameterNCHW(view, m[name]) else view:copy(m[name]) end
            m[name] = view
            m['grad' .. name:sub(1, 1):upper() .. name:sub(2)] = hipnn.viewOf(self.gflat, off, m[name])
         end
      end
      if m.running_mean then
         local rm, rv = hipnn.Tensor(m.running_mean:size(1)):copy(m.running_mean), hipnn.Tensor(m.running_var:size(1)):copy(m.running_var)
         check(C.vf_net_bind_bn_running(self.h, i - 1, fptr(rm), fptr(rv)))
         m.running_mean, m.running_var = rm, rv
      end
   end
   check(C.vf_net_set_weight_planes_managed(self.h, 0))     -- refreshed at every call unless the driver takes over (:manageWeightPlanes)
   return self
end
-- util.cudnn's counterpart: `netD = hipnn.hip(netD, opt.batchSize, nc, opt.fineSize, opt.fineSize)` (train.lua:250-251)
function hipnn.hip(net, B, Cc, H, W) return hipnn.Net(net, B, Cc, H, W) end
local function wrap(self, ptr, layer, isInput)     -- a device pointer of the library as a tensor of the right logical shape
   if ptr == nil then return nil end
   local d = {}
   for i = 1, 7 do d[i] = ffi.new('int[1]') end
   check(C.vf_net_layer_shape(self.h, layer, d[1], d[2], d[3], d[4], d[5], d[6], d[7]))
   local Bn = d[1][0]
   if isInput then return hipnn.wrapNHWC(ptr, Bn, d[2][0], d[3][0], d[4][0]) end
   return hipnn.wrapNHWC(ptr, Bn, d[5][0], d[6][0], d[7][0])
end
function Net:lastComputing()       -- index (0-based) of the last module that is not an nn.View
   for i = #self.modules, 1, -1 do if torch.type(self.modules[i]) ~= 'nn.View' then return i - 1 end end
   return 0
end
function Net:forward(x)
   if x:size(1) ~= self.B then check(C.vf_net_reshape(self.h, x:size(1), x:size(2), x:size(3), x:size(4))); self.B = x:size(1) end
   local y = ffi.new('const float*[1]')
   check(C.vf_net_forward(self.h, fptr(x), y))
   self.output = wrap(self, y[0], self:lastComputing(), false)
   local tail = self.modules[#self.modules]
   if torch.type(tail) == 'nn.View' then self.output = self.output:view(self.B, -1) end       -- nn.View(1):setNumInputDims(3)
   return self.output
end
Net.updateOutput = Net.forward
function Net:backward(x, gy)
   local g = ffi.new('const float*[1]')
   check(C.vf_net_backward(self.h, fptr(x), fptr(gy), g))
   self.gradInput = wrap(self, g[0], 0, true)         -- nil after :skipInputGrad(true)
   return self.gradInput
end
function Net:updateGradInput(x, gy)                   -- train.lua:366
   local g = ffi.new('const float*[1]')
   check(C.vf_net_update_grad_input(self.h, fptr(x), fptr(gy), g))
   self.gradInput = wrap(self, g[0], 0, true)
   return self.gradInput
end
-- netD's two passes of fDx as ONE batch [real; fake] (train.lua:331-349): net:setBatchGroups(2), forward / backward on the 2B
-- tensor, and in fGx the pass over the fake half only: net:updateGradInputGroup(fake, df_do, 1, 2)  (groups count from 0)
function Net:setBatchGroups(G) check(C.vf_net_set_batch_groups(self.h, G)); return self end
function Net:updateGradInputGroup(x, gy, g, G)
   local o = ffi.new('const float*[1]')
   check(C.vf_net_update_grad_input_group(self.h, fptr(x), fptr(gy), g, G, o))
   local t = wrap(self, o[0], 0, true)
   return t and t:narrow(1, 1, self.B / G) or nil
end
function Net:skipInputGrad(on) check(C.vf_net_set_skip_input_grad(self.h, on and 1 or 0)); return self end
function Net:zeroGradParameters() check(C.vf_net_zero_grad(self.h)) end
function Net:training() self.train = true; check(C.vf_net_training(self.h, 1)); return self end
function Net:evaluate() self.train = false; check(C.vf_net_training(self.h, 0)); return self end
function Net:getParameters() return self.flat, self.gflat end            -- train.lua:262-263
function Net:apply(fn) fn(self); for _, m in ipairs(self.modules) do fn(m) end; return self end      -- train.lua:150,201
-- `netD:apply(function(m) if torch.type(m):find('Convolution') then m.bias:zero() end end)` of both closures (train.lua:279-280)
-- as one launch; `other`: the second net of the sweep
function Net:zeroConvBiases(other) check(C.vf_net_zero_conv_biases(self.h, other and other.h or nil)) end
-- weight planes: by default refreshed at the start of every call; a driver that calls :refreshWeightPlanes() after each
-- optim.adam (and after loading a checkpoint) switches that off with :manageWeightPlanes(true): one launch per update
function Net:manageWeightPlanes(on) check(C.vf_net_set_weight_planes_managed(self.h, on and 1 or 0)); return self end
function Net:refreshWeightPlanes() check(C.vf_net_refresh_weight_planes(self.h)) end
-- data parallel: the walk cut where the big gradient bucket is complete (hipnn.allreduceAvgAsync on gflat[offset, end) meanwhile)
function Net:bucketSplit(frac)
   local k, off = ffi.new('int[1]'), ffi.new('int64_t[1]')
   check(C.vf_net_bucket_split(self.h, frac or 0.9, k, off))
   return k[0], tonumber(off[0])
end
function Net:backwardRange(x, gyPtr, hi, lo, needInputGrad)     -- gyPtr: a tensor, or the pointer a previous range returned
   local g = ffi.new('const float*[1]')
   local gp = type(gyPtr) == 'cdata' and gyPtr or fptr(gyPtr)
   check(C.vf_net_backward_range(self.h, fptr(x), gp, hi, lo, needInputGrad and 1 or 0, g))
   return g[0]
end
function Net:syncBatchNorm(world) check(C.vf_net_set_sync_bn(self.h, hipnn.comm, world, 0)); return self end
function Net:bindOutput(layer, tensor) check(C.vf_net_bind_output(self.h, layer, tensor and fptr(tensor) or nil)) end
function Net:layerOutput(i)          -- net.modules[i].output (1-based like Torch7)
   local y = ffi.new('const float*[1]')
   check(C.vf_net_layer_output(self.h, i - 1, y))
   local j = i
   while j > 1 and (KIND[torch.type(self.modules[j])] or 0) >= 4 do j = j - 1 end
   return wrap(self, y[0], j - 1, false)
end
-- roctx ranges (rocprofv3 --marker-trace): hipnn.range('fDx', function() ... end)
function hipnn.range(name, fn) C.vf_range_push(name); local ok, err = pcall(fn); C.vf_range_pop(); if not ok then error(err, 0) end end
function hipnn.beginBackward() check(C.vf_wgrad_group_begin(hipnn.ctx)) end
function hipnn.endBackward() check(C.vf_wgrad_group_end(hipnn.ctx)) end
function hipnn.abortBackward() check(C.vf_wgrad_group_abort(hipnn.ctx)) end   -- after an error inside a backward walk

fptr = function(t) return ffi.cast('float*', t:data()) end

----------------------------------------------------------------------------------------------------------------
-- Data parallel (one th process per GPU; the reference itself is single-device, train.lua:42).  Rank 0 writes the
-- 128-byte id to a file every rank can read; then, in the training loop:
--    optim.adam(function(x) local f, g = fDx(x); hipnn.allreduceAvg(gradParametersD); return f, g end, ...)
-- i.e. the flat gradient of train.lua:240-241 is averaged over ranks between the closure and the update.
----------------------------------------------------------------------------------------------------------------
function hipnn.initComm(world, rank, idfile)
   local id = ffi.new('uint8_t[128]')
   if rank == 0 then
      check(C.vf_comm_unique_id(id))
      local f = assert(io.open(idfile .. '.tmp', 'wb')); f:write(ffi.string(id, 128)); f:close()
      os.rename(idfile .. '.tmp', idfile)
   else
      local f = io.open(idfile, 'rb')
      while not f do os.execute('sleep 0.1'); f = io.open(idfile, 'rb') end
      ffi.copy(id, f:read(128), 128); f:close()
   end
   local out = ffi.new('vf_comm*[1]')
   check(C.vf_comm_init(out, id, world, rank))
   hipnn.comm, hipnn.world, hipnn.rank = out[0], world, rank
end
-- synchronous form: the context's stream waits for the average (the host does not)
function hipnn.allreduceAvg(flat)
   if not hipnn.comm then return end
   local t = ffi.new('int[1]')
   check(C.vf_comm_allreduce_avg_async(hipnn.comm, hipnn.ctx, fptr(flat), flat:nElement(), t))
   check(C.vf_comm_wait(hipnn.comm, hipnn.ctx, t[0]))
end
-- asynchronous form: returns the ticket; kernels launched before hipnn.waitComm(ticket) overlap the exchange
function hipnn.allreduceAvgAsync(flat)
   local t = ffi.new('int[1]')
   check(C.vf_comm_allreduce_avg_async(hipnn.comm, hipnn.ctx, fptr(flat), flat:nElement(), t))
   return t[0]
end
function hipnn.waitComm(ticket) check(C.vf_comm_wait(hipnn.comm, hipnn.ctx, ticket)) end
function hipnn.broadcastParameters(flat, root)
   check(C.vf_comm_broadcast(hipnn.comm, hipnn.ctx, fptr(flat), flat:nElement(), 0, root or 0))
end
function hipnn.barrier() check(C.vf_comm_barrier(hipnn.comm, hipnn.ctx)) end
function hipnn.destroyComm() if hipnn.comm then check(C.vf_comm_destroy(hipnn.comm)); hipnn.comm = nil end end

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

---------------------------------------------------------------------------------------------------------------
-- nn.SpatialFullConvolution replacement (train.lua:134-146).  weight logical nIn x nOut x kH x kW, physical
-- [nIn][kH][kW][nOut]; the three passes read the same buffer (vf_hip.h).
---------------------------------------------------------------------------------------------------------------
local Full, fparent = torch.class('hipnn.SpatialFullConvolution', 'nn.Module')
function Full:__init(nIn, nOut, kW, kH, dW, dH, padW, padH)
   fparent.__init(self)
   self.nInputPlane, self.nOutputPlane = nIn, nOut
   self.kW, self.kH, self.dW, self.dH = kW, kH, dW or 1, dH or 1
   self.padW, self.padH = padW or 0, padH or 0
   self.weight = hipnn.Tensor(nIn, kH, kW, nOut):permute(1, 4, 2, 3)
   self.gradWeight = hipnn.Tensor(nIn, kH, kW, nOut):permute(1, 4, 2, 3)
   self.bias, self.gradBias = hipnn.Tensor(nOut), hipnn.Tensor(nOut)
   self.fusedAct, self.fusedSlope = 'none', 0
end
function Full:updateOutput(input)
   local B, H, W = input:size(1), input:size(3), input:size(4)
   local Ho = (H - 1) * self.dH - 2 * self.padH + self.kH
   local Wo = (W - 1) * self.dW - 2 * self.padW + self.kW
   self.output = hipnn.resizeNHWC(self.output, B, self.nOutputPlane, Ho, Wo)
   check(C.vf_deconv2d_fwd(hipnn.ctx, fptr(input), fptr(self.weight), fptr(self.bias), fptr(self.output), B, H, W,
                           self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH, ACT[self.fusedAct], self.fusedSlope))
   return self.output
end
function Full:updateGradInput(input, gradOutput)
   local B, H, W = input:size(1), input:size(3), input:size(4)
   self.gradInput = hipnn.resizeNHWC(self.gradInput, B, self.nInputPlane, H, W)
   check(C.vf_deconv2d_bwd_data(hipnn.ctx, fptr(gradOutput), fptr(self.weight), fptr(self.gradInput), B, H, W,
                                self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH))
   return self.gradInput
end
function Full:accGradParameters(input, gradOutput, scale)
   assert((scale or 1) == 1)
   local B, H, W = input:size(1), input:size(3), input:size(4)
   check(C.vf_deconv2d_bwd_weight(hipnn.ctx, fptr(input), fptr(gradOutput), fptr(self.gradWeight), fptr(self.gradBias), B, H, W,
                                  self.nInputPlane, self.nOutputPlane, self.kH, self.dH, self.padH, 1.0))
end

---------------------------------------------------------------------------------------------------------------
-- nn.SpatialBatchNormalization replacement (train.lua:92).  Training: batch statistics, running_mean/var updated
-- with `momentum`, unbiased running_var (SURVEY A.3); evaluate(): running statistics (test_vid.lua:48).
---------------------------------------------------------------------------------------------------------------
local BN, bparent = torch.class('hipnn.SpatialBatchNormalization', 'nn.Module')
function BN:__init(nOutput, eps, momentum, affine)
   bparent.__init(self)
   assert(affine == nil or affine == true, 'the reference only builds affine BatchNorm')
   self.eps, self.momentum, self.affine, self.train = eps or 1e-5, momentum or 0.1, true, true
   self.weight, self.bias = hipnn.Tensor(nOutput), hipnn.Tensor(nOutput)
   self.gradWeight, self.gradBias = hipnn.Tensor(nOutput):zero(), hipnn.Tensor(nOutput):zero()
   self.running_mean, self.running_var = hipnn.Tensor(nOutput):zero(), hipnn.Tensor(nOutput):fill(1)
   self.save_mean, self.save_std = hipnn.Tensor(nOutput), hipnn.Tensor(nOutput)       -- save_std holds 1/sqrt(var+eps)
   self.sums = hipnn.DoubleTensor(2 * nOutput)
   self.fusedAct, self.fusedSlope = 'none', 0
end
function BN:updateOutput(input)
   local B, Cc, H, W = input:size(1), input:size(2), input:size(3), input:size(4)
   self.output = hipnn.resizeNHWC(self.output, B, Cc, H, W)
   if self.train then
      check(C.vf_bn_train_fwd(hipnn.ctx, fptr(input), fptr(self.output), fptr(self.weight), fptr(self.bias),
                              fptr(self.running_mean), fptr(self.running_var), fptr(self.save_mean), fptr(self.save_std),
                              ffi.cast('double*', self.sums:data()), B * H * W, Cc, self.momentum, self.eps,
                              ACT[self.fusedAct], self.fusedSlope))
   else
      check(C.vf_bn_eval_fwd(hipnn.ctx, fptr(input), fptr(self.output), fptr(self.weight), fptr(self.bias),
                             fptr(self.running_mean), fptr(self.running_var), B * H * W, Cc, self.eps,
                             ACT[self.fusedAct], self.fusedSlope))
   end
   return self.output
end
-- Torch7's BN computes gradInput and the parameter gradients from the same two reductions; `backward` does both in
-- one pass (the drivers only ever call net:backward or net:updateGradInput on the container).
function BN:backward(input, gradOutput, scale)
   assert(self.train, 'the reference never back-propagates through BatchNorm in evaluate mode')
   local B, Cc, H, W = input:size(1), input:size(2), input:size(3), input:size(4)
   self.gradInput = hipnn.resizeNHWC(self.gradInput, B, Cc, H, W)
   check(C.vf_bn_bwd(hipnn.ctx, fptr(input), fptr(self.output), fptr(gradOutput), fptr(self.gradInput),
                     self._noParamGrad and nil or fptr(self.gradWeight), self._noParamGrad and nil or fptr(self.gradBias),
                     fptr(self.weight), fptr(self.save_mean), fptr(self.save_std), ffi.cast('double*', self.sums:data()),
                     B * H * W, Cc, ACT[self.fusedAct], self.fusedSlope, 1.0))
   return self.gradInput
end
function BN:updateGradInput(input, gradOutput)
   self._noParamGrad = true
   local g = self:backward(input, gradOutput, 1)
   self._noParamGrad = nil
   return g
end
function BN:accGradParameters(input, gradOutput, scale) end      -- done by backward (see above)

---------------------------------------------------------------------------------------------------------------
-- Activations.  In-place LeakyReLU / ReLU directly after a conv or BatchNorm are folded into the producer by
-- hipnn.fuse(net) (its epilogue applies them; the module then only forwards tensors), exactly what nn.py's
-- Sequential does; unfused they are one vf_act_fwd / vf_act_bwd pass.
---------------------------------------------------------------------------------------------------------------
local function defAct(name, act, ctor)
   local A, aparent = torch.class('hipnn.' .. name, 'nn.Module')
   function A:__init(a, b)
      aparent.__init(self)
      self.act, self.slope, self.inplace, self.fusedInto = act, 0, false, nil
      if ctor then ctor(self, a, b) end
   end
   function A:updateOutput(input)
      if self.fusedInto then self.output = input; return input end
      self.output = self.inplace and input or hipnn.resizeLike(self.output, input)
      check(C.vf_act_fwd(hipnn.ctx, fptr(input), fptr(self.output), input:nElement(), ACT[act], self.slope))
      return self.output
   end
   function A:updateGradInput(input, gradOutput)
      -- f'(x) is taken from the OUTPUT (y > 0 <=> x > 0 for the leaky/plain ReLU; tanh, sigmoid are functions of y)
      self.gradInput = self.inplace and gradOutput or hipnn.resizeLike(self.gradInput, gradOutput)
      if not (self.fusedInto and self.fusedInto.bwdFused) then
         check(C.vf_act_bwd(hipnn.ctx, fptr(self.output), fptr(gradOutput), fptr(self.gradInput), gradOutput:nElement(), ACT[act], self.slope))
      else
         self.gradInput = gradOutput     -- BatchNorm's backward applies the activation derivative itself
      end
      return self.gradInput
   end
end
defAct('LeakyReLU', 'lrelu', function(self, negval, inplace) self.slope, self.negval, self.inplace = negval or 0.01, negval or 0.01, inplace or false end)
defAct('ReLU', 'relu', function(self, inplace) self.inplace = inplace or false end)
defAct('Tanh', 'tanh')
defAct('Sigmoid', 'sigmoid')

-- hipnn.fuse(net): mark (conv|BN, in-place activation) and (conv, Tanh|Sigmoid) pairs of a flat nn.Sequential.
function hipnn.fuse(net)
   local mods = net.modules
   for i = 1, #mods - 1 do
      local m, a = mods[i], mods[i + 1]
      local tm, ta = torch.type(m), torch.type(a)
      local prod = tm == 'hipnn.SpatialConvolution' or tm == 'hipnn.SpatialFullConvolution' or tm == 'hipnn.SpatialBatchNormalization'
      local isAct = ta == 'hipnn.LeakyReLU' or ta == 'hipnn.ReLU' or ta == 'hipnn.Tanh' or ta == 'hipnn.Sigmoid'
      local ok = prod and isAct and (a.inplace or ((ta == 'hipnn.Tanh' or ta == 'hipnn.Sigmoid') and tm ~= 'hipnn.SpatialBatchNormalization'))
      if ok then
         m.fusedAct, m.fusedSlope = a.act, a.slope
         m.bwdFused = (tm == 'hipnn.SpatialBatchNormalization')
         a.fusedInto = m
      end
   end
   return net
end

---------------------------------------------------------------------------------------------------------------
-- Criteria (train.lua:204-207, gdl_criterion.lua, MaskedMSECriterion.lua).  Losses are device doubles; :forward
-- returns a Lua number, i.e. it synchronises like Torch7's criteria do (use hipnn.lossSlot to keep them on the device).
---------------------------------------------------------------------------------------------------------------
local function readLoss(slot)
   local host = ffi.new('double[1]')
   check(C.vf_memcpy_d2h(hipnn.ctx, host, slot:data(), 8))
   return host[0]
end
local BCE = torch.class('hipnn.BCECriterion', 'nn.Criterion')
function BCE:__init() self.gradInput = hipnn.Tensor(); self.slot = hipnn.DoubleTensor(1) end
function BCE:updateOutput(input, target)        -- target: a label tensor filled with one value (label:fill(real_label))
   local label = type(target) == 'number' and target or target:hostScalar()
   check(C.vf_bce_fwd(hipnn.ctx, fptr(input), label, input:nElement(), ffi.cast('double*', self.slot:data())))
   self.output = readLoss(self.slot)
   return self.output
end
function BCE:updateGradInput(input, target)
   local label = type(target) == 'number' and target or target:hostScalar()
   self.gradInput = hipnn.resizeLike(self.gradInput, input)
   check(C.vf_bce_bwd(hipnn.ctx, fptr(input), label, fptr(self.gradInput), input:nElement()))
   return self.gradInput
end
local MSE = torch.class('hipnn.MSECriterion', 'nn.Criterion')
function MSE:__init() self.gradInput = hipnn.Tensor(); self.slot = hipnn.DoubleTensor(1) end
function MSE:updateOutput(input, target)
   check(C.vf_mse_fwd(hipnn.ctx, fptr(input), fptr(target), input:nElement(), ffi.cast('double*', self.slot:data())))
   self.output = readLoss(self.slot)
   return self.output
end
function MSE:updateGradInput(input, target)
   self.gradInput = hipnn.resizeLike(self.gradInput, input)
   check(C.vf_mse_bwd(hipnn.ctx, fptr(input), fptr(target), fptr(self.gradInput), input:nElement()))
   return self.gradInput
end
local GDL = torch.class('hipnn.GDLCriterion', 'nn.Criterion')      -- the drivers use the forward value only
function GDL:__init(alpha) assert((alpha or 1) == 1); self.gradInput = hipnn.Tensor(); self.slot = hipnn.DoubleTensor(1) end
function GDL:updateOutput(input, target)
   check(C.vf_gdl_fwd(hipnn.ctx, fptr(input), fptr(target), input:size(1), input:size(3), input:size(4), input:size(2),
                      ffi.cast('double*', self.slot:data())))
   self.output = readLoss(self.slot)
   return self.output
end
function GDL:updateGradInput(input, target)      -- gdl_criterion.lua:47-53
   self.gradInput = hipnn.resizeLike(self.gradInput, input)
   check(C.vf_gdl_bwd(hipnn.ctx, fptr(input), fptr(target), fptr(self.gradInput), input:size(1), input:size(3), input:size(4),
                      input:size(2)))
   return self.gradInput
end
local MMSE = torch.class('hipnn.MaskedMSECriterion', 'nn.Criterion')
function MMSE:__init(mWeight) self.mWeight = mWeight or 1; self.gradInput = hipnn.Tensor(); self.slot = hipnn.DoubleTensor(1) end
function MMSE:setMask(m) self.mask = m end      -- device uint8 0/1, same element order as the input
function MMSE:updateOutput(input, target)
   check(C.vf_masked_mse_fwd(hipnn.ctx, fptr(input), fptr(target), ffi.cast('const uint8_t*', self.mask:data()), self.mWeight,
                             input:nElement(), ffi.cast('double*', self.slot:data())))
   self.output = readLoss(self.slot)
   return self.output
end
function MMSE:updateGradInput(input, target)
   self.gradInput = hipnn.resizeLike(self.gradInput, input)
   check(C.vf_masked_mse_bwd(hipnn.ctx, fptr(input), fptr(target), ffi.cast('const uint8_t*', self.mask:data()), self.mWeight,
                             fptr(self.gradInput), input:nElement()))
   return self.gradInput
end

---------------------------------------------------------------------------------------------------------------
-- hipnn.Tensor: the device tensor the classes above hold — a device pointer plus sizes/strides, the few methods the
-- binding and the drivers touch.  (In a Torch7 tree built for ROCm, cutorch's CudaTensor plays this role and the
-- classes work on it unchanged: they only call :size, :nElement, :data.)
---------------------------------------------------------------------------------------------------------------
local DT = {}
DT.__index = DT
local function newTensor(elemSize, ...)
   local sizes = { ... }
   local n = #sizes > 0 and 1 or 0
   for _, s in ipairs(sizes) do n = n * s end
   local self = setmetatable({ sizes = sizes, elemSize = elemSize, n = n }, DT)
   self.strides = {}
   local st = 1
   for i = #sizes, 1, -1 do self.strides[i] = st; st = st * sizes[i] end
   if n > 0 then
      local p = ffi.new('void*[1]')
      check(C.vf_malloc(p, n * elemSize))
      self.ptr = ffi.gc(p[0], C.vf_free)
   end
   return self
end
function hipnn.Tensor(...) return newTensor(4, ...) end
function hipnn.DoubleTensor(...) return newTensor(8, ...) end
function hipnn.IntTensor(...) return newTensor(4, ...) end
function hipnn.ByteTensor(...) return newTensor(1, ...) end
function DT:data() return self.ptr end
function DT:nElement() return self.n end
function DT:dim() return #self.sizes end
function DT:size(i) if i then return self.sizes[i] end; return torch.LongStorage(self.sizes) end
function DT:permute(...)      -- a view with permuted logical axes over the same storage
   local o = setmetatable({ ptr = self.ptr, elemSize = self.elemSize, n = self.n, sizes = {}, strides = {}, base = self }, DT)
   for i, d in ipairs({ ... }) do o.sizes[i], o.strides[i] = self.sizes[d], self.strides[d] end
   return o
end
function DT:zero() if self.n > 0 then check(C.vf_zero(hipnn.ctx, self.ptr, self.n * self.elemSize)) end; return self end
function DT:fill(v)
   assert(self.elemSize == 4)
   self:zero()
   check(C.vf_scale_shift(hipnn.ctx, ffi.cast('float*', self.ptr), 0, v, self.n))
   self.hostValue = v
   return self
end
function DT:hostScalar() return assert(self.hostValue, 'label tensors are filled on the host side first (label:fill)') end
function DT:copy(src)          -- from a host torch.FloatTensor (contiguous, same element order) or another device tensor
   if getmetatable(src) == DT then
      check(C.vf_axpby(hipnn.ctx, 1, ffi.cast('const float*', src.ptr), 0, ffi.cast('float*', self.ptr), self.n))
   else
      local h = src:float():contiguous()
      check(C.vf_memcpy_h2d(hipnn.ctx, self.ptr, h:data(), self.n * self.elemSize))
   end
   return self
end
function DT:float()            -- to a host torch.FloatTensor in this tensor's PHYSICAL order
   local h = torch.FloatTensor(self.n)
   check(C.vf_memcpy_d2h(hipnn.ctx, h:data(), self.ptr, self.n * 4))
   return h
end
function hipnn.resizeNHWC(t, B, Cc, H, W)     -- logical B x C x H x W over [B][H][W][C]
   if t and getmetatable(t) == DT and t.n == B * Cc * H * W and t.sizes[2] == Cc and t.sizes[3] == H then return t end
   return hipnn.Tensor(B, H, W, Cc):permute(1, 4, 2, 3)
end
-- non-owning tensors over memory that belongs to someone else (the library's net object, a flat parameter storage)
local function aliasTensor(ptr, n, sizes, strides, keep)
   return setmetatable({ ptr = ptr, elemSize = 4, n = n, sizes = sizes, strides = strides, base = keep }, DT)
end
function hipnn.wrapNHWC(ptr, B, Cc, H, W)     -- logical B x C x H x W over the library's [B][H][W][C] buffer
   return aliasTensor(ffi.cast('void*', ptr), B * Cc * H * W, { B, Cc, H, W }, { H * W * Cc, 1, W * Cc, Cc })
end
function hipnn.viewOf(flat, off, like)        -- `like`'s logical sizes over flat[off ...]: channels-last strides for a 4-D weight
   local sizes, n = {}, 1
   for i = 1, like:dim() do sizes[i] = like:size(i); n = n * sizes[i] end
   local strides
   if #sizes == 4 then strides = { sizes[3] * sizes[4] * sizes[2], 1, sizes[4] * sizes[2], sizes[2] }
   else strides = { 1 } end
   return aliasTensor(ffi.cast('void*', ffi.cast('float*', flat.ptr) + off), n, sizes, strides, flat)
end
function DT:view(a, b)            -- (B, -1): the rows of a [B][...] buffer (nn.View(1):setNumInputDims(3) on B x 1 x 1 x 1)
   local rest = self.n / a
   assert(b == -1 or b == rest)
   return aliasTensor(self.ptr, self.n, { a, rest }, { rest, 1 }, self)
end
function DT:narrow(dim, first, len)   -- leading-dimension slices only (batch groups)
   assert(dim == 1)
   local per = self.n / self.sizes[1]
   local sizes = { len }
   for i = 2, #self.sizes do sizes[i] = self.sizes[i] end
   return aliasTensor(ffi.cast('void*', ffi.cast('float*', self.ptr) + (first - 1) * per), len * per, sizes, self.strides, self)
end
function hipnn.resizeLike(t, like)
   if t and getmetatable(t) == DT and t.n == like.n then t.sizes, t.strides = like.sizes, like.strides; return t end
   local o = newTensor(4, like.n)
   o.sizes, o.strides = like.sizes, like.strides
   return o
end
-- NCHW host tensor (B x C x H x W, contiguous) -> device channels-last, and back
function hipnn.toNHWC(host)
   local B, Cc, H, W = host:size(1), host:size(2), host:size(3), host:size(4)
   local stage = hipnn.Tensor(B * Cc * H * W):copy(host)
   local out = hipnn.resizeNHWC(nil, B, Cc, H, W)
   check(C.vf_nchw_to_nhwc(hipnn.ctx, fptr(stage), fptr(out), B, Cc, H, W))
   return out
end
function hipnn.toNCHW(dev)
   local B, Cc, H, W = dev:size(1), dev:size(2), dev:size(3), dev:size(4)
   local stage = hipnn.Tensor(B * Cc * H * W)
   check(C.vf_nhwc_to_nchw(hipnn.ctx, fptr(dev), fptr(stage), B, Cc, H, W))
   return stage:float():view(B, Cc, H, W)
end
-- copy an nn (NCHW-contiguous) 4-D weight into a channels-last parameter: dim 1 plays the batch role
function hipnn.copyParameterNCHW(dst, src)
   local h = src:float():contiguous()
   local stage = hipnn.Tensor(h:nElement()):copy(h)
   check(C.vf_nchw_to_nhwc(hipnn.ctx, fptr(stage), fptr(dst), h:size(1), h:size(2), h:size(3), h:size(4)))
end

---------------------------------------------------------------------------------------------------------------
-- util.hip(net): the backend swap.  Same walk as util.cudnn (util.lua:108-125): recurse into containers, replace by
-- exact type string, construct with the same 8 arguments, copy weight/bias (through vf_nchw_to_nhwc, since
-- nn weights are NCHW-contiguous).
---------------------------------------------------------------------------------------------------------------
local REPLACE = {
   ['nn.SpatialConvolution'] = function(l) return hipnn.SpatialConvolution(l.nInputPlane, l.nOutputPlane, l.kW, l.kH, l.dW, l.dH, l.padW, l.padH) end,
   ['nn.SpatialFullConvolution'] = function(l) return hipnn.SpatialFullConvolution(l.nInputPlane, l.nOutputPlane, l.kW, l.kH, l.dW, l.dH, l.padW, l.padH) end,
   ['nn.SpatialBatchNormalization'] = function(l) return hipnn.SpatialBatchNormalization(l.running_mean:size(1), l.eps, l.momentum, l.affine) end,
   ['nn.LeakyReLU'] = function(l) return hipnn.LeakyReLU(l.negval, l.inplace) end,
   ['nn.ReLU'] = function(l) return hipnn.ReLU(l.inplace) end,
   ['nn.Tanh'] = function(l) return hipnn.Tanh() end,
   ['nn.Sigmoid'] = function(l) return hipnn.Sigmoid() end,
}
local function recursiveHip(net)
   for k, l in ipairs(net.modules) do
      if net.modules[k].modules ~= nil then recursiveHip(net.modules[k]) end
      local make = REPLACE[torch.type(l)]
      if make then
         local new = make(l)
         if l.weight then
            if l.weight:dim() == 4 then hipnn.copyParameterNCHW(new.weight, l.weight) else new.weight:copy(l.weight) end
            new.bias:copy(l.bias)
         end
         if l.running_mean then new.running_mean:copy(l.running_mean); new.running_var:copy(l.running_var) end
         new.train = l.train
         net.modules[k] = new
      end
   end
end
function hipnn.convert(net) recursiveHip(net); return hipnn.fuse(net) end

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

-- optim.adam(opfunc, x, state) for a hipnn.Net whose bottleneck pair (train.lua:104,134: 92 % of the generator's weights) is
-- updated INSIDE the kernel that forms its weight gradient (vf_net_adam_fused): those two slices of dfdx are not written unless
-- keepGrad; everything else gets the plain one-pass update.  Single device (a data-parallel step needs the gradient on the wire).
--    hipnn.adamFused(fGx, parametersG, optimStateG, netG)
function hipnn.adamFused(opfunc, x, state, net, keepGrad)
   local cnt = ffi.new('int[1]')
   check(C.vf_net_set_fused_adam(net.h, 1, cnt))
   if cnt[0] == 0 then return hipnn.adam(opfunc, x, state) end
   local ok, fx, dfdx = pcall(opfunc, x)
   if not ok then C.vf_net_set_fused_adam(net.h, 0, cnt); error(fx, 0) end
   state.m = state.m or x.new(dfdx:size()):zero()
   state.v = state.v or x.new(dfdx:size()):zero()
   state.t_dev = state.t_dev or hipnn.IntTensor(2):zero()
   local b1, b2, eps = state.beta1 or 0.9, state.beta2 or 0.999, state.epsilon or 1e-8
   local t = ffi.cast('int32_t*', state.t_dev:data())
   check(C.vf_adam_prep(hipnn.ctx, state.learningRate or 0.001, b1, b2, t))
   local ranges, off, len = {}, ffi.new('int64_t[1]'), ffi.new('int64_t[1]')
   for i = 0, cnt[0] - 1 do
      check(C.vf_net_fused_adam_range(net.h, i, off, len))
      ranges[#ranges + 1] = { tonumber(off[0]), tonumber(off[0] + len[0]) }
   end
   table.sort(ranges, function(a, b) return a[1] < b[1] end)
   ranges[#ranges + 1] = { x:nElement(), x:nElement() }
   -- everything outside the fused slices: ONE launch over the ranges in between (vf_adam_apply_ranges)
   local offs, lens, n, pos = ffi.new('int64_t[8]'), ffi.new('int64_t[8]'), 0, 0
   for _, r in ipairs(ranges) do
      if r[1] > pos then offs[n], lens[n] = pos, r[1] - pos; n = n + 1 end
      pos = r[2]
   end
   if n > 0 then check(C.vf_adam_apply_ranges(hipnn.ctx, fptr(x), fptr(dfdx), fptr(state.m), fptr(state.v), offs, lens, n, b1, b2, eps, t)) end
   check(C.vf_net_adam_fused(net.h, fptr(state.m), fptr(state.v), b1, b2, eps, t, keepGrad and 1 or 0))
   check(C.vf_net_set_fused_adam(net.h, 0, cnt))
   return x, { fx }
end

return hipnn
