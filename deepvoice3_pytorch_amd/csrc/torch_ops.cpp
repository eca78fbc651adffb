// TORCH_LIBRARY(dv3hip): the C ABI of include/dv3hip.h as PyTorch-ROCm custom operators
// (torch.ops.dv3hip.*), the form BASELINE.json's north_star and SURVEY.md 8(b) name for the boundary.
// A thin shim: at::Tensor in / out at the operator boundary, raw device pointers + sizes inside, every call
// forwarded to libdv3hip.so on torch's CURRENT HIP stream; outputs come from torch's caching allocator.  No
// arithmetic lives here.  Host-only C++ (built by __graft_entry__.build() with g++ against the torch headers).
//
// Operators (reference lines they stand in for):
//   weight_norm_split_pack   nn.utils.weight_norm pre-hook                       modules.py:85,100
//   conv1d_glu               Conv1dGLU._forward / HighwayConv1d._forward (eval)   modules.py:145-164, 205-226
//   conv1x1                  1x1 Conv1d / Linear (+ReLU / sigmoid)                deepvoice3.py:51-54,565-578
//   sincos_pos               SinusoidalEncoding.forward                           modules.py:45-64
//   clip_adam_               clip_grad_norm_ + Adam.step on flat arenas           train.py:755-759
//   grad_sqnorm              the norm clip_grad_norm_ computes
//   abi_version
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <tuple>

#include "../../include/dv3hip.h"

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void check(int rc, const char* what) {
  TORCH_CHECK(rc == DV3_OK, "dv3hip::", what, " failed (", rc, "): ", dv3_last_error());
}
const at::Tensor& gpu_f32(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "dv3hip op got a non-GPU tensor for ", name, ": the HIP path has no CPU fallback");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
  return t;
}
int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

int64_t abi_version() { return dv3_abi_version(); }

// v (O, I[, J]), g (O, 1[, 1]) or None -> (fwd_split int16, bwd_split int16, scale f32 [O])
std::tuple<at::Tensor, at::Tensor, at::Tensor> weight_norm_split_pack(const at::Tensor& v_, const c10::optional<at::Tensor>& g_,
                                                                      int64_t glu_cg, bool f16) {
  at::Tensor v = gpu_f32(v_, "weight_v").contiguous();
  if (v.dim() == 2) v = v.unsqueeze(-1);
  TORCH_CHECK(v.dim() == 3, "weight_v must be (O, I) or (O, I, J)");
  const int64_t O = v.size(0), I = v.size(1), J = v.size(2);
  at::Tensor g;
  if (g_.has_value()) g = gpu_f32(*g_, "weight_g").contiguous();
  dv3_wn_desc d = {};
  d.v = v.data_ptr<float>();
  d.g = g.defined() ? g.data_ptr<float>() : nullptr;
  at::Tensor scale = at::empty({O}, v.options());
  d.scale = scale.data_ptr<float>();
  d.a_half = glu_cg ? (int32_t)round_up(glu_cg, 4) : 0;
  d.lda = glu_cg ? 2 * d.a_half : (int32_t)round_up(O, 4);
  d.ldb = (int32_t)round_up(I, 4);
  d.O = (int32_t)O; d.I = (int32_t)I; d.J = (int32_t)J; d.transposed = 0; d.glu_cg = (int32_t)glu_cg;
  d.fwd_dtype = f16 ? DV3_SPLIT_DTYPE_F16 : DV3_SPLIT_DTYPE_BF16;
  auto i16 = v.options().dtype(at::kShort);
  at::Tensor fs = at::zeros({2 * J * round_up(I, 32) * d.lda}, i16);
  at::Tensor bs = at::zeros({2 * J * round_up(O, 32) * d.ldb}, i16);
  check(dv3_weight_norm_split_pack_bf16(&d, (uint16_t*)fs.data_ptr<int16_t>(), (uint16_t*)bs.data_ptr<int16_t>(), cur_stream()),
        "weight_norm_split_pack");
  return {fs, bs, scale};
}

// shared by conv1d_glu / conv1x1: one tap-GEMM launch on the split-operand kernels
at::Tensor conv_launch(const at::Tensor& x_, const at::Tensor& fwd_split, bool f16, const c10::optional<at::Tensor>& bias_,
                       int64_t M, int64_t k, int64_t dilation, bool causal, int64_t mode, bool residual) {
  at::Tensor x = gpu_f32(x_, "x").contiguous();
  TORCH_CHECK(x.dim() == 3, "x must be (B, C, T)");
  TORCH_CHECK(fwd_split.is_cuda() && fwd_split.scalar_type() == at::kShort, "fwd_split must be the int16 image of weight_norm_split_pack");
  const int64_t B = x.size(0), Cin = x.size(1), T = x.size(2);
  const bool gated = mode == DV3_EPI_GLU || mode == DV3_EPI_HIGHWAY;
  const int64_t Cg = gated ? M / 2 : 0;
  const int64_t Cout = gated ? Cg : M;
  at::Tensor y = at::empty({B, Cout, T}, x.options());
  at::Tensor bias;
  if (bias_.has_value()) bias = gpu_f32(*bias_, "bias").contiguous();
  dv3_conv_desc d = {};
  d.x = x.data_ptr<float>(); d.x_bs = Cin * T; d.x_rs = T;
  d.a = nullptr; d.a_bs = 0;
  d.a_half = gated ? (int32_t)round_up(Cg, 4) : 0;
  d.lda = gated ? 2 * d.a_half : (int32_t)round_up(M, 4);
  d.bias = bias.defined() ? bias.data_ptr<float>() : nullptr;
  if (gated && (mode == DV3_EPI_HIGHWAY || residual)) { d.r = x.data_ptr<float>(); d.r_bs = Cin * T; d.r_rs = T; }
  d.y = y.data_ptr<float>(); d.y_bs = Cout * T; d.y_rs = T;
  d.drop_scale = 1.0f;
  d.B = (int32_t)B; d.Cin = (int32_t)Cin; d.Tin = (int32_t)T; d.M = (int32_t)M; d.Cg = (int32_t)Cg; d.Tout = (int32_t)T;
  d.J = (int32_t)k; d.dil = (int32_t)dilation;
  d.padL = (int32_t)(causal ? (k - 1) * dilation : (k - 1) / 2 * dilation);
  d.mode = (int32_t)mode; d.residual = residual ? 1 : 0; d.store_mode = DV3_STORE_BCT;
  d.a_split = (const uint16_t*)fwd_split.data_ptr<int16_t>();
  d.split_terms = f16 ? DV3_SPLIT_F16X3 : 0;
  check(dv3_conv_gemm_f32(&d, cur_stream()), "conv_gemm");
  return y;
}

at::Tensor conv1d_glu(const at::Tensor& x, const at::Tensor& fwd_split, bool f16, const c10::optional<at::Tensor>& bias,
                      int64_t kernel_size, int64_t dilation, bool causal, bool residual, bool highway) {
  TORCH_CHECK(x.dim() == 3, "x must be (B, C, T)");
  return conv_launch(x, fwd_split, f16, bias, 2 * x.size(1), kernel_size, dilation, causal,
                     highway ? DV3_EPI_HIGHWAY : DV3_EPI_GLU, residual);
}
// act: 0 linear, 1 relu, 2 sigmoid
at::Tensor conv1x1(const at::Tensor& x, const at::Tensor& fwd_split, bool f16, const c10::optional<at::Tensor>& bias,
                   int64_t out_channels, int64_t act) {
  TORCH_CHECK(act >= 0 && act <= 2, "act: 0 linear, 1 relu, 2 sigmoid");
  return conv_launch(x, fwd_split, f16, bias, out_channels, 1, 1, false, act, false);
}

at::Tensor sincos_pos(const at::Tensor& pos_, const at::Tensor& table_, double w) {
  TORCH_CHECK(pos_.is_cuda() && pos_.scalar_type() == at::kLong && pos_.dim() == 2, "pos must be a (B, T) int64 GPU tensor");
  at::Tensor pos = pos_.contiguous(), table = gpu_f32(table_, "table").contiguous();
  const int64_t B = pos.size(0), T = pos.size(1), n_pos = table.size(0), C = table.size(1);
  at::Tensor wt = at::full({1}, w, table.options());
  at::Tensor out = at::empty({B, C, T}, table.options());
  check(dv3_sincos_pos_bct_f32(pos.data_ptr<int64_t>(), table.data_ptr<float>(), wt.data_ptr<float>(), 0, nullptr,
                               out.data_ptr<float>(), (int32_t)B, (int32_t)T, (int32_t)C, (int32_t)n_pos, 1, cur_stream()),
        "sincos_pos");
  return out.transpose(1, 2);     // (B, T, C), the reference's layout
}

at::Tensor grad_sqnorm(const at::Tensor& g) {
  gpu_f32(g, "grad");
  TORCH_CHECK(g.is_contiguous(), "grad must be contiguous (a flat arena)");
  at::Tensor partial = at::empty({1024}, g.options()), out2 = at::zeros({2}, g.options());
  check(dv3_grad_sqnorm_f32(g.data_ptr<float>(), g.numel(), partial.data_ptr<float>(), 1024, out2.data_ptr<float>(), cur_stream()),
        "grad_sqnorm");
  return out2;     // [0] = norm, [1] = squared norm
}

void clip_adam_(at::Tensor p, const at::Tensor& g, at::Tensor m, at::Tensor v, const c10::optional<at::Tensor>& grad_norm,
                double clip, const at::Tensor& hyper, double beta1, double beta2, double eps, double weight_decay,
                double grad_prescale) {
  gpu_f32(p, "p"); gpu_f32(g, "g"); gpu_f32(m, "exp_avg"); gpu_f32(v, "exp_avg_sq"); gpu_f32(hyper, "hyper");
  TORCH_CHECK(p.is_contiguous() && g.is_contiguous() && m.is_contiguous() && v.is_contiguous(), "flat contiguous arenas expected");
  TORCH_CHECK(g.numel() == p.numel() && m.numel() == p.numel() && v.numel() == p.numel() && hyper.numel() >= 3, "size mismatch");
  const float* gn = nullptr;
  if (grad_norm.has_value()) gn = gpu_f32(*grad_norm, "grad_norm").data_ptr<float>();
  check(dv3_clip_adam_f32(p.data_ptr<float>(), g.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), p.numel(), gn,
                          (float)clip, hyper.data_ptr<float>(), (float)beta1, (float)beta2, (float)eps, (float)weight_decay,
                          (float)grad_prescale, cur_stream()),
        "clip_adam");
}

}  // namespace

TORCH_LIBRARY(dv3hip, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("weight_norm_split_pack(Tensor weight_v, Tensor? weight_g, int glu_cg, bool f16) -> (Tensor, Tensor, Tensor)");
  m.def("conv1d_glu(Tensor x, Tensor fwd_split, bool f16, Tensor? bias, int kernel_size, int dilation, bool causal, "
        "bool residual, bool highway) -> Tensor");
  m.def("conv1x1(Tensor x, Tensor fwd_split, bool f16, Tensor? bias, int out_channels, int act) -> Tensor");
  m.def("sincos_pos(Tensor pos, Tensor table, float w) -> Tensor");
  m.def("grad_sqnorm(Tensor grad) -> Tensor");
  m.def("clip_adam_(Tensor(a!) p, Tensor g, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor? grad_norm, float clip, "
        "Tensor hyper, float beta1, float beta2, float eps, float weight_decay, float grad_prescale) -> ()");
}
TORCH_LIBRARY_IMPL(dv3hip, CUDA, m) {     // the "CUDA" dispatch key is the GPU key of PyTorch-ROCm
  m.impl("weight_norm_split_pack", &weight_norm_split_pack);
  m.impl("conv1d_glu", &conv1d_glu);
  m.impl("conv1x1", &conv1x1);
  m.impl("sincos_pos", &sincos_pos);
  m.impl("grad_sqnorm", &grad_sqnorm);
  m.impl("clip_adam_", &clip_adam_);
}
