// DilatedConvBlock called on its own (dilated_cnn.py:76-99): Conv1d (no bias, symmetric zero padding (k-1)d/2) ->
// BatchNorm1d -> ReLU -> Dropout, plus the residual: the input itself when the widths agree, else
// BatchNorm1d(Conv1d 1x1) of it.  Activations are channels-last [B*T, C] (the Python side transposes the reference's
// [B, C, T]); the convolutions are the library's implicit GEMMs, the BatchNorm passes are plain kernels - this is off
// the model's hot path (DilatedCNN runs its blocks through ocean_dcnn_*), written for correctness, not speed.
#include "../../include/ocean_hip.h"
#include "ops.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }
inline int egrid(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// w [Co][Ci][Kt] -> wp [Kt][Co][Ci]
__global__ void cb_repack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int Kt) {
  const long n = (long)Co * Ci * Kt;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Kt), ci = (int)((i / Kt) % Ci), co = (int)(i / ((long)Kt * Ci));
    wp[((long)t * Co + co) * Ci + ci] = w[i];
  }
}
// dw [Co][Ci][Kt] += dwp [Kt][Co][Ci]
__global__ void cb_unpack_add_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int Kt) {
  const long n = (long)Co * Ci * Kt;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Kt), ci = (int)((i / Kt) % Ci), co = (int)(i / ((long)Kt * Ci));
    dw[i] += dwp[((long)t * Co + co) * Ci + ci];
  }
}

// One block per 32 columns: 8 row lanes each; fp64 partial sums meet in shared memory.
// training: mean / biased variance of u's columns, running statistics updated (unbiased variance, momentum);
// eval: the running statistics.  Leaves mean, invstd and the folded scale = gamma*invstd, shift = beta - mean*scale.
__global__ __launch_bounds__(256) void cb_stats_kernel(const float* __restrict__ u, int M, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv,
                                                       float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                                       float* __restrict__ shift, int training, float eps, float momentum) {
  __shared__ double part[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (training && c < C)
    for (int r = rl; r < M; r += 8) { const double x = u[(long)r * C + c]; s1 += x; s2 += x * x; }
  part[0][rl][cl] = s1; part[1][rl][cl] = s2;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  float mu, var;
  if (training) {
    s1 = 0.0; s2 = 0.0;
    for (int q = 0; q < 8; ++q) { s1 += part[0][q][cl]; s2 += part[1][q][cl]; }
    const double n = (double)M, m = s1 / n;
    double v = s2 / n - m * m;
    if (v < 0.0) v = 0.0;
    mu = (float)m; var = (float)v;
    const double unbiased = n > 1.0 ? v * n / (n - 1.0) : v;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mu;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
  } else {
    mu = rm[c]; var = rv[c];
  }
  const float is = 1.0f / sqrtf(var + eps);
  mean[c] = mu; invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc; shift[c] = beta[c] - mu * sc;
}

// out (+)= act(u*scale + shift) [* keep]; relu: ReLU and dropout (the main path); accumulate: add into out
__global__ void cb_apply_kernel(const float* __restrict__ u, const float* __restrict__ scale, const float* __restrict__ shift,
                                float* __restrict__ out, long n, int C, int relu, int accumulate, DropoutP dp) {
  dp = dropout_resolve(dp);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float y = fmaf(u[i], scale[c], shift[c]);
    if (relu) {
      y = fmaxf(y, 0.f);
      if (dp.p > 0.f) y *= dropout_keep(dp, (uint64_t)i);
    }
    out[i] = accumulate ? out[i] + y : y;
  }
}

// g = dy [* keep * (u*scale+shift > 0)]: per column sum g and sum g*xhat (fp64 atomics into sums [2][C], zeroed by the host)
__global__ __launch_bounds__(256) void cb_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          double* __restrict__ sums, int M, int C, int relu, DropoutP dp) {
  dp = dropout_resolve(dp);
  __shared__ double part[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int r0 = blockIdx.y * 1024, r1 = min(M, r0 + 1024);
  double sg = 0.0, sx = 0.0;
  if (c < C) {
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    for (int r = r0 + rl; r < r1; r += 8) {
      const long i = (long)r * C + c;
      float g = dy[i];
      if (relu) {
        if (!(fmaf(u[i], sc, sh) > 0.f)) g = 0.f;
        if (dp.p > 0.f) g *= dropout_keep(dp, (uint64_t)i);
      }
      sg += g;
      sx += (double)g * ((u[i] - mu) * is);
    }
  }
  part[0][rl][cl] = sg; part[1][rl][cl] = sx;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  sg = 0.0; sx = 0.0;
  for (int q = 0; q < 8; ++q) { sg += part[0][q][cl]; sx += part[1][q][cl]; }
  unsafeAtomicAdd(sums + c, sg);
  unsafeAtomicAdd(sums + C + c, sx);
}

// dgamma += sum g*xhat, dbeta += sum g
__global__ void cb_bwd_param_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += (float)sums[c];
  if (dgamma) dgamma[c] += (float)sums[C + c];
}

// du = scale * (g - sum(g)/n - xhat * sum(g*xhat)/n)  (training)   |   du = scale * g  (eval)
__global__ void cb_bwd_du_kernel(const float* __restrict__ dy, const float* __restrict__ u, const float* __restrict__ scale,
                                 const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                 const double* __restrict__ sums, float* __restrict__ du, long n, int M, int C, int relu, int training,
                                 DropoutP dp) {
  dp = dropout_resolve(dp);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float sc = scale[c];
    float g = dy[i];
    if (relu) {
      if (!(fmaf(u[i], sc, shift[c]) > 0.f)) g = 0.f;
      if (dp.p > 0.f) g *= dropout_keep(dp, (uint64_t)i);
    }
    float o = g;
    if (training) {
      const float xh = (u[i] - mean[c]) * invstd[c];
      o = g - (float)(sums[c] / M) - xh * (float)(sums[C + c] / M);
    }
    du[i] = sc * o;
  }
}

struct CbPlan {
  size_t wp, U, st, Ur, str, fwd_total;            // st / str: mean | invstd | scale | shift of the main / residual BatchNorm
  size_t dU, dwp, sums, bwd_total;
};

int cb_plan(const ocean_convblock_shape* s, CbPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->T <= 0 || s->Cin <= 0 || s->Cout <= 0 || s->kernel <= 0 || s->dilation <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (s->kernel % 2 == 0 || s->kernel > 15) return OCEAN_ERR_UNSUPPORTED;       // symmetric padding keeps the length only for odd kernels
  if (s->dropout < 0.f || s->dropout >= 1.f) return OCEAN_ERR_BAD_SHAPE;
  if (!s->res_conv && s->Cin != s->Cout) return OCEAN_ERR_BAD_SHAPE;
  const size_t M = (size_t)s->B * s->T, Ci = s->Cin, Co = s->Cout, W = Co * Ci * s->kernel;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->wp = take(W); p->U = take(M * Co); p->st = take(4 * Co);
  p->Ur = take(s->res_conv ? M * Co : 0); p->str = take(s->res_conv ? 4 * Co : 0);
  p->fwd_total = o;
  o = 0;
  p->dU = take(M * Co); p->dwp = take(W); p->sums = take(2 * 2 * Co);       // fp64 [2][Co]
  p->bwd_total = o;
  return OCEAN_OK;
}

DropoutP cb_drop(const ocean_convblock_shape* s) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed;
  d.epoch = s->seed_epoch;
  return d;
}

}  // namespace

extern "C" {

size_t ocean_convblock_fwd_workspace(const ocean_convblock_shape* s) { CbPlan p; return cb_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_convblock_bwd_workspace(const ocean_convblock_shape* s) { CbPlan p; return cb_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_convblock_fwd(const ocean_convblock_shape* s, const float* x, float* const* P_, float* ws, size_t ws_floats, float* out,
                        void* stream) {
  CbPlan pl;
  OCEAN_TRY(cb_plan(s, &pl));
  if (!x || !P_ || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  if (s->training && (long)s->B * s->T < 2) return OCEAN_ERR_BAD_SHAPE;      // BatchNorm refuses one value per channel
  hipStream_t st = (hipStream_t)stream;
  const int M = s->B * s->T, Ci = s->Cin, Co = s->Cout, k = s->kernel;
  const long n = (long)M * Co;
  hipLaunchKernelGGL(cb_repack_kernel, dim3(egrid((long)Co * Ci * k)), dim3(256), 0, st, P_[0], ws + pl.wp, Co, Ci, k);
  OCEAN_LAUNCH_CHECK();
  GemmP g = gemm_defaults();
  g.A = x; g.lda = Ci; g.B = ws + pl.wp; g.ldb = Ci; g.C = ws + pl.U; g.ldc = Co; g.M = M; g.N = Co; g.K = Ci;
  if (k > 1) { g.taps = k; g.T = s->T; g.shift0 = -(k / 2) * s->dilation; g.shift_step = s->dilation; g.tap_stride_b = (long)Co * Ci; }
  OCEAN_TRY(ocean_launch_gemm(g, st));
  float* S = ws + pl.st;
  hipLaunchKernelGGL(cb_stats_kernel, dim3(ceil_div(Co, 32)), dim3(256), 0, st, ws + pl.U, M, Co, P_[1], P_[2], P_[3], P_[4], S, S + Co,
                     S + 2 * Co, S + 3 * Co, s->training, s->eps, s->momentum);
  OCEAN_LAUNCH_CHECK();
  if (s->res_conv) {
    OCEAN_TRY(linear_fwd(x, Ci, P_[5], nullptr, ws + pl.Ur, Co, M, Co, Ci, 0, st));
    float* Sr = ws + pl.str;
    hipLaunchKernelGGL(cb_stats_kernel, dim3(ceil_div(Co, 32)), dim3(256), 0, st, ws + pl.Ur, M, Co, P_[6], P_[7], P_[8], P_[9], Sr,
                       Sr + Co, Sr + 2 * Co, Sr + 3 * Co, s->training, s->eps, s->momentum);
    OCEAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(cb_apply_kernel, dim3(egrid(n)), dim3(256), 0, st, ws + pl.Ur, Sr + 2 * Co, Sr + 3 * Co, out, n, Co, 0, 0,
                       DropoutP{0.f, 0, nullptr});
  } else {
    OCEAN_TRY(copy2d(out, Co, x, Ci, M, Co, 0, st));
  }
  OCEAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(cb_apply_kernel, dim3(egrid(n)), dim3(256), 0, st, ws + pl.U, S + 2 * Co, S + 3 * Co, out, n, Co, 1, 1, cb_drop(s));
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_convblock_bwd(const ocean_convblock_shape* s, const float* x, float* const* P_, const float* wf, const float* dy, float* wb,
                        size_t wb_floats, float* const* G_, float* dx, void* stream) {
  CbPlan pl;
  OCEAN_TRY(cb_plan(s, &pl));
  if (!x || !P_ || !wf || !dy || !wb || !G_) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int M = s->B * s->T, Ci = s->Cin, Co = s->Cout, k = s->kernel;
  const long n = (long)M * Co;
  double* sums = reinterpret_cast<double*>(wb + pl.sums);
  const dim3 sgrid(ceil_div(Co, 32), ceil_div(M, 1024));
  bool dx_started = false;
  // the residual branch first: dx = dy (identity) or its 1x1 convolution's data gradient
  if (s->res_conv) {
    const float* Sr = wf + pl.str;
    OCEAN_TRY(zero_f64(sums, 2 * (size_t)Co, st));
    hipLaunchKernelGGL(cb_bwd_sums_kernel, sgrid, dim3(256), 0, st, dy, wf + pl.Ur, Sr + 2 * Co, Sr + 3 * Co, Sr, Sr + Co, sums, M, Co, 0,
                       DropoutP{0.f, 0, nullptr});
    OCEAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(cb_bwd_param_kernel, dim3(ceil_div(Co, 64)), dim3(64), 0, st, sums, G_[6], G_[7], Co);
    OCEAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(cb_bwd_du_kernel, dim3(egrid(n)), dim3(256), 0, st, dy, wf + pl.Ur, Sr + 2 * Co, Sr + 3 * Co, Sr, Sr + Co, sums,
                       wb + pl.dU, n, M, Co, 0, s->training, DropoutP{0.f, 0, nullptr});
    OCEAN_LAUNCH_CHECK();
    if (G_[5]) OCEAN_TRY(linear_bwd_weight(wb + pl.dU, Co, x, Ci, G_[5], M, Co, Ci, st));
    if (dx) { OCEAN_TRY(linear_bwd_data(wb + pl.dU, Co, P_[5], dx, Ci, M, Co, Ci, 0, st)); dx_started = true; }
  } else if (dx) {
    OCEAN_TRY(copy2d(dx, Ci, dy, Co, M, Co, 0, st));
    dx_started = true;
  }
  // the main path
  const float* S = wf + pl.st;
  const DropoutP dp = cb_drop(s);
  OCEAN_TRY(zero_f64(sums, 2 * (size_t)Co, st));
  hipLaunchKernelGGL(cb_bwd_sums_kernel, sgrid, dim3(256), 0, st, dy, wf + pl.U, S + 2 * Co, S + 3 * Co, S, S + Co, sums, M, Co, 1, dp);
  OCEAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(cb_bwd_param_kernel, dim3(ceil_div(Co, 64)), dim3(64), 0, st, sums, G_[1], G_[2], Co);
  OCEAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(cb_bwd_du_kernel, dim3(egrid(n)), dim3(256), 0, st, dy, wf + pl.U, S + 2 * Co, S + 3 * Co, S, S + Co, sums, wb + pl.dU,
                     n, M, Co, 1, s->training, dp);
  OCEAN_LAUNCH_CHECK();
  if (G_[0]) {
    OCEAN_TRY(zero_f32(wb + pl.dwp, (size_t)Co * Ci * k, st));
    GemmP w = gemm_defaults();
    w.A = wb + pl.dU; w.lda = Co; w.a_kmajor = 1; w.B = x; w.ldb = Ci; w.b_kn = 1; w.C = wb + pl.dwp; w.ldc = Ci;
    w.M = Co; w.N = Ci; w.K = M; w.split_k = 0;
    if (k > 1) { w.taps = k; w.T = s->T; w.shift0 = -(k / 2) * s->dilation; w.shift_step = s->dilation; w.tap_stride_c = (long)Co * Ci; }
    OCEAN_TRY(ocean_launch_gemm(w, st));
    hipLaunchKernelGGL(cb_unpack_add_kernel, dim3(egrid((long)Co * Ci * k)), dim3(256), 0, st, wb + pl.dwp, G_[0], Co, Ci, k);
    OCEAN_LAUNCH_CHECK();
  }
  if (dx) {
    GemmP d = gemm_defaults();
    d.A = wb + pl.dU; d.lda = Co; d.B = wf + pl.wp; d.ldb = Ci; d.b_kn = 1; d.C = dx; d.ldc = Ci; d.M = M; d.N = Ci; d.K = Co;
    if (k > 1) { d.taps = k; d.T = s->T; d.shift0 = (k / 2) * s->dilation; d.shift_step = -s->dilation; d.tap_stride_b = (long)Co * Ci; }
    if (dx_started) { d.R = dx; d.ldr = Ci; }
    OCEAN_TRY(ocean_launch_gemm(d, st));
  }
  return OCEAN_OK;
}

}  // extern "C"
