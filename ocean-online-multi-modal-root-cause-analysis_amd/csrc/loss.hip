// InfoNCE (memory-bank and in-batch branches), prediction head + sigmoid, and
// binary cross-entropy: forward and backward.
//
// InfoNCE (graph_fusion.py:112-162): both inputs are L2-normalised again, the
// similarity matrix a_hat . x_hat^T is one fp32-MFMA GEMM ([B,K] with K = 16
// bank rows, or [B, B_global] for in-batch negatives), and a wave-per-row
// kernel takes the temperature-scaled log-sum-exp.  Backward turns the saved
// similarities into (softmax - onehot) * dloss / (B * tau) in place and feeds
// two GEMMs (d a_hat = dL . x_hat, d x_hat = dL^T . a_hat).
#include "../../include/ocean_hip.h"
#include "ops.h"
#include "rowchain.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
constexpr int kWaves = 4;

// sim[b, 0] = a_hat[b] . p_hat[b]   (bank branch keeps the positive in column 0)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ p, float* __restrict__ out,
                                                     int ldo, int B, int P) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (b >= B) return;
  float d = 0.f;
  for (int c = lane; c < P; c += 64) d += a[(long)b * P + c] * p[(long)b * P + c];
  d = wave_sum(d);
  if (lane == 0) out[(long)b * ldo] = d;
}

// loss += sum_b (lse_b - sim[b,label_b]/tau) * inv_rows;  lse saved per row
// group > 0: the rows are `B / group` stacked problems of `group` rows each; problem q adds into loss[q] and its row
// r has label label0 + label_step * r
__global__ __launch_bounds__(256) void nce_rows_fwd_kernel(const float* __restrict__ sim, int ld, int B, int K, int label0, int label_step,
                                                           float inv_tau, float inv_rows, float* __restrict__ lse_out,
                                                           float* __restrict__ loss, int group, float* __restrict__ row_loss = nullptr) {
  __shared__ float red[kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x * kWaves + wave;
  float contrib = 0.f;
  if (b < B) {
    const float* r = sim + (long)b * ld;
    float m = -INFINITY;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, r[k] * inv_tau);
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += __expf(r[k] * inv_tau - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    const int label = label0 + label_step * (group > 0 ? b % group : b);
    contrib = (lse - r[label] * inv_tau) * inv_rows;
    if (lane == 0) lse_out[b] = lse;
    if (row_loss != nullptr) {                // reduction='none': the row's own loss, nothing summed (block-uniform branch)
      if (lane == 0) row_loss[b] = lse - r[label] * inv_tau;
      contrib = 0.f;
    }
  }
  if (row_loss != nullptr) return;
  if (lane == 0) red[wave] = contrib;
  __syncthreads();
  // one add per block (same-address atomics serialise: one per row cost 46 us at 3 x 1024 rows); a block whose rows
  // straddle two stacked problems adds per wave instead
  const int b0 = blockIdx.x * kWaves, b1 = min(b0 + kWaves, B) - 1;
  if (group <= 0 || b0 / group == b1 / group) {
    if (threadIdx.x == 0) unsafeAtomicAdd(loss + (group > 0 ? b0 / group : 0), red[0] + red[1] + red[2] + red[3]);
  } else if (lane == 0 && b < B) {
    unsafeAtomicAdd(loss + b / group, contrib);
  }
}

// sim[b,k] <- (softmax_k(sim/tau) - [k == label_b]) * dloss * inv_rows * inv_tau
__global__ __launch_bounds__(256) void nce_rows_bwd_kernel(float* __restrict__ sim, int ld, int B, int K, int label0, int label_step,
                                                           float inv_tau, float inv_rows, const float* __restrict__ lse,
                                                           const float* __restrict__ dloss, int group, int dloss_per_row = 0) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (b >= B) return;
  float* r = sim + (long)b * ld;
  const float g = (dloss ? dloss[dloss_per_row ? b : (group > 0 ? b / group : 0)] : 1.0f) * inv_rows * inv_tau;
  const float l = lse[b];
  const int label = label0 + label_step * (group > 0 ? b % group : b);
  for (int k = lane; k < K; k += 64) r[k] = (__expf(r[k] * inv_tau - l) - (k == label ? 1.f : 0.f)) * g;
}

// out[b,:] (+)= col0[b] * src[b,:]
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ col0, int ld0, const float* __restrict__ src,
                                                       float* __restrict__ out, int B, int P, int accumulate) {
  const long n = (long)B * P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / P;
    const float v = col0[b * ld0] * src[i];
    if (accumulate) out[i] += v; else out[i] = v;
  }
}

__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = 1.0f / (1.0f + __expf(-x[i]));
}
// dlogits = dlogits_in + dprobs * p * (1 - p)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dlogits_in, const float* __restrict__ dprobs, const float* __restrict__ p,
                                   float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = dlogits_in ? dlogits_in[i] : 0.f;
    if (dprobs) v += dprobs[i] * p[i] * (1.0f - p[i]);
    out[i] = v;
  }
}

// F.binary_cross_entropy(p, y, 'mean'): log terms clamped at -100
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float inv_n,
                                                      float* __restrict__ loss) {
  __shared__ float red[kWaves];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(log1pf(-p[i]), -100.f);
    s -= y[i] * lp + (1.0f - y[i]) * lq;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}
// dp = dloss * (p - y) / max(p (1-p), 1e-12) / n
__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, const float* __restrict__ dloss, long n,
                               float inv_n, float* __restrict__ dp) {
  const float g = (dloss ? dloss[0] : 1.0f) * inv_n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dp[i] = g * (p[i] - y[i]) / fmaxf(p[i] * (1.0f - p[i]), 1e-12f);
}

// reduction='none': out[i] = -(y log p + (1-y) log(1-p)), same clamps; dp[i] = dout[i] * (p - y) / max(p (1-p), 1e-12)
__global__ void bce_elem_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(log1pf(-p[i]), -100.f);
    out[i] = -(y[i] * lp + (1.0f - y[i]) * lq);
  }
}
__global__ void bce_elem_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, const float* __restrict__ dout, long n,
                                    float* __restrict__ dp) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dp[i] = dout[i] * (p[i] - y[i]) / fmaxf(p[i] * (1.0f - p[i]), 1e-12f);
}

// loss += mean((a-b)^2)
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float inv_n,
                                                      float* __restrict__ loss) {
  __shared__ float red[kWaves];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float d = a[i] - b[i]; s += d * d; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}
__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dloss, long n,
                               float inv_n, float* __restrict__ da) {
  const float g = 2.0f * (dloss ? dloss[0] : 1.0f) * inv_n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) da[i] = g * (a[i] - b[i]);
}
// loss += sum_{i != j} |S[i,j]| * inv_pairs   over a [B,B] similarity matrix
__global__ __launch_bounds__(256) void offdiag_abs_fwd_kernel(const float* __restrict__ S, int B, float inv_pairs, float* __restrict__ loss) {
  __shared__ float red[kWaves];
  const long n = (long)B * B;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (i / B != i % B) s += fabsf(S[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_pairs);
}
// S[i,j] <- sign(S[i,j]) * dloss * inv_pairs * 2 off the diagonal, 0 on it (S is symmetric: dF = (dS + dS^T) F = 2 dS F)
__global__ void offdiag_abs_bwd_kernel(float* __restrict__ S, int B, float inv_pairs, const float* __restrict__ dloss) {
  const long n = (long)B * B;
  const float g = 2.0f * (dloss ? dloss[0] : 1.0f) * inv_pairs;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = S[i];
    S[i] = (i / B != i % B) ? (v > 0.f ? g : (v < 0.f ? -g : 0.f)) : 0.f;
  }
}

inline int egrid(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

struct NcePlan { size_t ah, ph, nh, na, np, nn, sim, lse, total; int ldsim; };
int nce_plan(const ocean_infonce_shape* s, NcePlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->Bp <= 0 || s->P <= 0 || s->K < 0 || !(s->tau > 0.f)) return OCEAN_ERR_BAD_SHAPE;
  if (s->K == 0 && (s->label_offset < 0 || s->label_offset + s->B > s->Bp)) return OCEAN_ERR_BAD_SHAPE;
  if (s->K > 0 && s->Bp != s->B) return OCEAN_ERR_BAD_SHAPE;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->ah = take((size_t)s->B * s->P); p->ph = take((size_t)s->Bp * s->P); p->nh = take((size_t)s->K * s->P);
  p->na = take(s->B); p->np = take(s->Bp); p->nn = take(s->K);
  p->ldsim = s->K > 0 ? al4(1 + s->K) + 4 : s->Bp;       // bank: column 0 = positive, negatives start at column 4 (16-byte aligned)
  p->sim = take((size_t)s->B * p->ldsim); p->lse = take(s->B);
  p->total = o;
  return OCEAN_OK;
}

}  // namespace

extern "C" {

size_t ocean_infonce_workspace(const ocean_infonce_shape* s) { NcePlan p; return nce_plan(s, &p) == OCEAN_OK ? p.total : 0; }
size_t ocean_infonce_bwd_workspace(const ocean_infonce_shape* s) {
  return s ? al4((size_t)s->B * s->P) + al4((size_t)s->Bp * s->P) : 0;
}

static int infonce_fwd_impl(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                            float* ws, size_t ws_floats, float* loss, float* row_loss, void* stream) {
  NcePlan pl;
  OCEAN_TRY(nce_plan(s, &pl));
  if (!anchor || !positive || !ws || (!loss && !row_loss) || (s->K > 0 && !negatives)) return OCEAN_ERR_NULL;
  if (ws_floats < pl.total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, P = s->P, K = s->K;
  const float inv_tau = 1.0f / s->tau, inv_rows = 1.0f / (float)(s->mean_rows > 0 ? s->mean_rows : B);
  OCEAN_TRY(l2norm_fwd(anchor, P, ws + pl.ah, P, ws + pl.na, B, P, 1e-12f, st));
  OCEAN_TRY(l2norm_fwd(positive, P, ws + pl.ph, P, ws + pl.np, s->Bp, P, 1e-12f, st));
  if (K > 0) {
    OCEAN_TRY(l2norm_fwd(negatives, P, ws + pl.nh, P, ws + pl.nn, K, P, 1e-12f, st));
    // similarity layout per row: [pos, (3 unused), neg_0 .. neg_{K-1}]; the row kernels see columns {0} U [4, 4+K)
    // by compacting: write pos at column 3 so that columns 3 .. 3+K are contiguous
    hipLaunchKernelGGL(rowdot_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, ws + pl.ah, ws + pl.ph, ws + pl.sim + 3, pl.ldsim, B, P);
    OCEAN_LAUNCH_CHECK();
    GemmP g = gemm_defaults();
    g.A = ws + pl.ah; g.lda = P; g.B = ws + pl.nh; g.ldb = P; g.C = ws + pl.sim + 4; g.ldc = pl.ldsim; g.M = B; g.N = K; g.K = P;
    OCEAN_TRY(ocean_launch_gemm(g, st));
    hipLaunchKernelGGL(nce_rows_fwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, ws + pl.sim + 3, pl.ldsim, B, K + 1, 0, 0, inv_tau,
                       inv_rows, ws + pl.lse, loss, 0, row_loss);
  } else {
    GemmP g = gemm_defaults();
    g.A = ws + pl.ah; g.lda = P; g.B = ws + pl.ph; g.ldb = P; g.C = ws + pl.sim; g.ldc = pl.ldsim; g.M = B; g.N = s->Bp; g.K = P;
    OCEAN_TRY(ocean_launch_gemm(g, st));
    hipLaunchKernelGGL(nce_rows_fwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, ws + pl.sim, pl.ldsim, B, s->Bp, s->label_offset, 1,
                       inv_tau, inv_rows, ws + pl.lse, loss, 0, row_loss);
  }
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_infonce_fwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                      float* ws, size_t ws_floats, float* loss, void* stream) {
  if (!loss) return OCEAN_ERR_NULL;
  return infonce_fwd_impl(s, anchor, positive, negatives, ws, ws_floats, loss, nullptr, stream);
}
int ocean_infonce_rows_fwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                           float* ws, size_t ws_floats, float* row_loss, void* stream) {
  if (!row_loss) return OCEAN_ERR_NULL;
  return infonce_fwd_impl(s, anchor, positive, negatives, ws, ws_floats, nullptr, row_loss, stream);
}

static int infonce_bwd_impl(const ocean_infonce_shape* s, const float* anchor, const float* positive, float* ws, float* wb,
                            size_t wb_floats, const float* dloss, int per_row, float* d_anchor, float* d_positive, void* stream) {
  NcePlan pl;
  OCEAN_TRY(nce_plan(s, &pl));
  if (!ws || !wb || !d_anchor || !d_positive) return OCEAN_ERR_NULL;
  if (wb_floats < ocean_infonce_bwd_workspace(s)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, P = s->P, K = s->K, Bp = s->Bp;
  const float inv_tau = 1.0f / s->tau, inv_rows = per_row ? 1.0f : 1.0f / (float)(s->mean_rows > 0 ? s->mean_rows : B);
  float* dah = wb; float* dph = wb + al4((size_t)B * P);
  if (K > 0) {
    float* sim = ws + pl.sim + 3;
    hipLaunchKernelGGL(nce_rows_bwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, sim, pl.ldsim, B, K + 1, 0, 0, inv_tau, inv_rows,
                       ws + pl.lse, dloss, 0, per_row);
    OCEAN_LAUNCH_CHECK();
    // d a_hat = dL[:,0] * p_hat + dL[:,1:] . n_hat ;  d p_hat = dL[:,0] * a_hat
    OCEAN_TRY(linear_bwd_data(sim + 1, pl.ldsim, ws + pl.nh, dah, P, B, K, P, 0, st));
    hipLaunchKernelGGL(rowscale_kernel, dim3(egrid((long)B * P)), dim3(256), 0, st, sim, pl.ldsim, ws + pl.ph, dah, B, P, 1);
    OCEAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(rowscale_kernel, dim3(egrid((long)B * P)), dim3(256), 0, st, sim, pl.ldsim, ws + pl.ah, dph, B, P, 0);
    OCEAN_LAUNCH_CHECK();
  } else {
    float* sim = ws + pl.sim;
    hipLaunchKernelGGL(nce_rows_bwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, sim, pl.ldsim, B, Bp, s->label_offset, 1, inv_tau,
                       inv_rows, ws + pl.lse, dloss, 0, per_row);
    OCEAN_LAUNCH_CHECK();
    OCEAN_TRY(linear_bwd_data(sim, pl.ldsim, ws + pl.ph, dah, P, B, Bp, P, 0, st));      // [B,Bp] . [Bp,P]
    OCEAN_TRY(zero_f32(dph, (size_t)Bp * P, st));
    OCEAN_TRY(linear_bwd_weight(sim, pl.ldsim, ws + pl.ah, P, dph, B, Bp, P, st));       // [B,Bp]^T . [B,P]
  }
  OCEAN_TRY(l2norm_bwd(dah, P, ws + pl.ah, P, ws + pl.na, d_anchor, P, B, P, 1e-12f, st));
  OCEAN_TRY(l2norm_bwd(dph, P, ws + pl.ph, P, ws + pl.np, d_positive, P, Bp, P, 1e-12f, st));
  return OCEAN_OK;
}

int ocean_infonce_bwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, float* ws, float* wb,
                      size_t wb_floats, const float* dloss, float* d_anchor, float* d_positive, void* stream) {
  return infonce_bwd_impl(s, anchor, positive, ws, wb, wb_floats, dloss, 0, d_anchor, d_positive, stream);
}
int ocean_infonce_rows_bwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, float* ws, float* wb,
                           size_t wb_floats, const float* d_row_loss, float* d_anchor, float* d_positive, void* stream) {
  if (!d_row_loss) return OCEAN_ERR_NULL;
  return infonce_bwd_impl(s, anchor, positive, ws, wb, wb_floats, d_row_loss, 1, d_anchor, d_positive, stream);
}

// ---------------------------------------------------------------- memory-bank ring
// ContrastiveLearningModule._update_memory_bank (graph_fusion.py:251-273) with the write pointer kept on the device:
// rows [R, P] are written at bank[(ptr + r) % K] (when R > K only the last K rows survive a full lap), then the
// pointer advances by the reference's formula, which leaves the out-of-range value K when ptr + R == 2K.  Two
// launches: every block of the first reads the pointer, so it may only change afterwards.
namespace {
__global__ void bank_ring_write_kernel(float* __restrict__ bank, const long long* __restrict__ ptr, const float* __restrict__ rows,
                                       int R, int K, int P) {
  const long long p0 = ptr[0];
  const int skip = R > K ? R - K : 0;                  // rows a later row of this same call would overwrite
  const long n = (long)(R - skip) * P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / P) + skip, c = (int)(i % P);
    const int dst = (int)(((p0 + r) % K + K) % K);
    bank[(long)dst * P + c] = rows[(long)r * P + c];
  }
}
__global__ void bank_ring_advance_kernel(long long* __restrict__ ptr, int R, int K) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long p0 = ptr[0], e = p0 + R;
    ptr[0] = (e <= K) ? e % K : (e <= 2LL * K ? e - K : e % K);
  }
}
}  // namespace

int ocean_bank_ring_write(float* bank, long long* ptr, const float* rows, int R, int K, int P, void* stream) {
  if (!bank || !ptr || !rows) return OCEAN_ERR_NULL;
  if (R <= 0 || K <= 0 || P <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bank_ring_write_kernel, dim3(egrid((long)(R > K ? K : R) * P)), dim3(256), 0, st, bank, ptr, rows, R, K, P);
  OCEAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bank_ring_advance_kernel, dim3(1), dim3(64), 0, st, ptr, R, K);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

// ---------------------------------------------------------------- InfoNCE forward for callers that cannot allocate
// The workspace of ocean_infonce_fwd (normalised copies, the similarity matrix, row norms and log-sum-exps) lives in
// one array that is part of the code object, so no allocation happens at any time: not here, not in the caller.
// Every GPU the library runs on holds its own copy of the array (OCEAN_STATIC_SCRATCH_FLOATS * 4 bytes = 6 MB) until
// the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: the device address is
// looked up once per device, with that device current, and every launch gets that address as an ordinary argument.
static __device__ __align__(256) float scratch[OCEAN_STATIC_SCRATCH_FLOATS];

static constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices];          // per device: address of that device's copy of `scratch`
static bool g_scratch_known[kMaxDevices];

// the current device's copy of `scratch`; 0 on success
static int static_scratch(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return OCEAN_ERR_HIP + (int)e;
  if (dev < 0 || dev >= kMaxDevices) return OCEAN_ERR_UNSUPPORTED;
  // two threads of one process may both look the address up the first time; they store the same value
  if (!__atomic_load_n(&g_scratch_known[dev], __ATOMIC_ACQUIRE)) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return OCEAN_ERR_HIP + (int)e;
    if (p == nullptr) return OCEAN_ERR_NULL;
    g_scratch_ptr[dev] = static_cast<float*>(p);
    __atomic_store_n(&g_scratch_known[dev], true, __ATOMIC_RELEASE);
  }
  *out = g_scratch_ptr[dev];
  return OCEAN_OK;
}

// Looks the array's address up for the current device.  The lookup may load the code object, which a stream capture in
// progress can refuse: a caller that records ocean_infonce_fwd_static into a HIP graph calls this (or the entry point
// itself) once per device beforehand, outside the capture.
int ocean_static_scratch_warmup(void) {
  float* ws = nullptr;
  return static_scratch(&ws);
}

int ocean_infonce_fwd_static(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                             float* loss, void* stream) {
  NcePlan pl;
  OCEAN_TRY(nce_plan(s, &pl));
  if (pl.total > (size_t)OCEAN_STATIC_SCRATCH_FLOATS) return OCEAN_ERR_SCRATCH;     // beyond the built-in array: use ocean_infonce_fwd
  float* ws = nullptr;
  OCEAN_TRY(static_scratch(&ws));
  // Nothing in the workspace is accumulated into: the normalised rows, the norms, the similarity matrix and the
  // log-sum-exps are all overwritten before they are read, so the previous call's contents need no clearing.
  return infonce_fwd_impl(s, anchor, positive, negatives, ws, pl.total, loss, nullptr, stream);
}

// ---------------------------------------------------------------- the three in-batch pairs at once
// Z [3, B, P]: the three modality projections.  Pair q of {(0,1), (0,2), (1,2)} takes rows of Z[a] as anchors and rows
// of the positives' [b] as positives; the positives are Z itself, or (data parallel) G [3, Bp, P] = Z gathered from
// every rank, in which local row r is matched with gathered row label_offset + r.  Every projection is re-normalised
// once (InfoNCELoss does it per call, with the same result), the three similarity matrices come from one batched GEMM
// launch, and one row kernel finishes all three losses.
namespace {
struct Nce3Plan { int Bp; size_t zh, nrm, gh, gnrm, sim, lse, total; };
int nce3_plan(const ocean_infonce3_shape* s, Nce3Plan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->P <= 0 || !(s->tau > 0.f) || s->Bp < 0 || s->label_offset < 0) return OCEAN_ERR_BAD_SHAPE;
  p->Bp = s->Bp > 0 ? s->Bp : s->B;
  if (s->label_offset + s->B > p->Bp) return OCEAN_ERR_BAD_SHAPE;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->zh = take((size_t)3 * s->B * s->P); p->nrm = take((size_t)3 * s->B);
  p->gh = take(s->Bp > 0 ? (size_t)3 * p->Bp * s->P : 0); p->gnrm = take(s->Bp > 0 ? (size_t)3 * p->Bp : 0);
  p->sim = take((size_t)3 * s->B * p->Bp); p->lse = take((size_t)3 * s->B);
  p->total = o;
  return OCEAN_OK;
}
constexpr int kPairA[3] = {0, 0, 1}, kPairB[3] = {1, 2, 2};
}  // namespace

size_t ocean_infonce3_workspace(const ocean_infonce3_shape* s) { Nce3Plan p; return nce3_plan(s, &p) == OCEAN_OK ? p.total : 0; }
size_t ocean_infonce3_bwd_workspace(const ocean_infonce3_shape* s) {
  return s ? (size_t)3 * s->B * s->P + (s->Bp > 0 ? (size_t)3 * s->Bp * s->P : 0) + 4 : 0;
}

int ocean_infonce3_fwd(const ocean_infonce3_shape* s, const float* Z, const float* G, float* ws, size_t ws_floats, float* losses,
                       void* stream) {
  Nce3Plan pl;
  OCEAN_TRY(nce3_plan(s, &pl));
  if (!Z || !ws || !losses || (s->Bp > 0 && !G)) return OCEAN_ERR_NULL;
  if (ws_floats < pl.total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, P = s->P, Bp = pl.Bp;
  const size_t BP = (size_t)B * P, GP = (size_t)Bp * P, BB = (size_t)B * Bp;
  OCEAN_TRY(l2norm_fwd(Z, P, ws + pl.zh, P, ws + pl.nrm, 3 * B, P, 1e-12f, st));
  const float* pos = ws + pl.zh;
  if (s->Bp > 0) {
    OCEAN_TRY(l2norm_fwd(G, P, ws + pl.gh, P, ws + pl.gnrm, 3 * Bp, P, 1e-12f, st));
    pos = ws + pl.gh;
  }
  GemmP g[3];
  for (int q = 0; q < 3; ++q) {
    g[q] = gemm_defaults();
    g[q].A = ws + pl.zh + kPairA[q] * BP; g[q].lda = P; g[q].B = pos + kPairB[q] * GP; g[q].ldb = P;
    g[q].C = ws + pl.sim + q * BB; g[q].ldc = Bp; g[q].M = B; g[q].N = Bp; g[q].K = P;
  }
  OCEAN_TRY(ocean_launch_gemm_batch(g, 3, st));
  hipLaunchKernelGGL(nce_rows_fwd_kernel, dim3(ceil_div(3 * B, kWaves)), dim3(256), 0, st, ws + pl.sim, Bp, 3 * B, Bp, s->label_offset, 1,
                     1.0f / s->tau, 1.0f / (float)B, ws + pl.lse, losses, B);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_infonce3_bwd(const ocean_infonce3_shape* s, float* ws, float* wb, size_t wb_floats, const float* dlosses, float* dZ,
                       float* dG, void* stream) {
  Nce3Plan pl;
  OCEAN_TRY(nce3_plan(s, &pl));
  if (!ws || !wb || !dZ || (s->Bp > 0 && !dG)) return OCEAN_ERR_NULL;
  if (wb_floats < ocean_infonce3_bwd_workspace(s)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, P = s->P, Bp = pl.Bp;
  const bool gathered = s->Bp > 0;
  const size_t BP = (size_t)B * P, GP = (size_t)Bp * P, BB = (size_t)B * Bp;
  float* sim = ws + pl.sim; float* zh = ws + pl.zh;
  const float* pos = gathered ? ws + pl.gh : zh;
  float* dzh = wb;                                  // [3, B, P]
  float* dph = gathered ? wb + 3 * BP : dzh;        // gradient of the normalised positives: its own [3, Bp, P], or the anchors' buffer
  hipLaunchKernelGGL(nce_rows_bwd_kernel, dim3(ceil_div(3 * B, kWaves)), dim3(256), 0, st, sim, Bp, 3 * B, Bp, s->label_offset, 1, 1.0f / s->tau,
                     1.0f / (float)B, ws + pl.lse, dlosses, B);
  OCEAN_LAUNCH_CHECK();
  // anchors: d zh[a] (+)= G_q . pos[b]   (G_q = the rewritten similarity matrix of pair q)
  GemmP g[3];
  for (int q = 0; q < 3; ++q) {
    g[q] = gemm_defaults();
    g[q].A = sim + q * BB; g[q].lda = Bp; g[q].B = pos + kPairB[q] * GP; g[q].ldb = P; g[q].b_kn = 1;
    g[q].C = dzh + kPairA[q] * BP; g[q].ldc = P; g[q].M = B; g[q].N = P; g[q].K = Bp;
  }
  GemmP first[2] = {g[0], g[2]};                       // pairs (0,1) and (1,2) write d zh[0] and d zh[1]
  OCEAN_TRY(ocean_launch_gemm_batch(first, 2, st));
  g[1].R = g[1].C; g[1].ldr = P;                       // pair (0,2) adds to d zh[0]
  OCEAN_TRY(ocean_launch_gemm(g[1], st));
  // positives: d pos[b] += G_q^T . zh[a]  (row reductions: they accumulate with atomics, so two may share an output).
  // d zh[2] has no anchor term; with gathered positives none of d pos has been written yet: one fill covers both.
  OCEAN_TRY(zero_f32(dzh + 2 * BP, BP + (gathered ? 3 * GP : 0), st));
  GemmP t[3];
  for (int q = 0; q < 3; ++q) {
    t[q] = gemm_defaults();
    t[q].A = sim + q * BB; t[q].lda = Bp; t[q].a_kmajor = 1; t[q].B = zh + kPairA[q] * BP; t[q].ldb = P; t[q].b_kn = 1;
    t[q].C = dph + kPairB[q] * GP; t[q].ldc = P; t[q].M = Bp; t[q].N = P; t[q].K = B; t[q].split_k = 0;
  }
  OCEAN_TRY(ocean_launch_gemm_batch(t, 3, st));
  OCEAN_TRY(l2norm_bwd(dzh, P, zh, P, ws + pl.nrm, dZ, P, 3 * B, P, 1e-12f, st));
  if (gathered) OCEAN_TRY(l2norm_bwd(dph, P, ws + pl.gh, P, ws + pl.gnrm, dG, P, 3 * Bp, P, 1e-12f, st));
  return OCEAN_OK;
}

// ---------------------------------------------------------------- mutual-information estimates
// f(cat(fused, x)) = w1 . relu(W0a fused + W0b x + b0) + b1 with W0 = [W0a | W0b]: the fused half U = fused W0a^T is
// shared by all six score sets (3 modalities x joint / shuffled), the modality half V_m = X_m W0b^T by a modality's
// joint and shuffled rows, so two small GEMM launches replace the reference's [6B, 2E] x [2E, E] product over
// materialised concatenations, and one row kernel forms both scores of a (modality, row).
namespace {
__global__ __launch_bounds__(256) void mi_rows_kernel(const float* __restrict__ U, const float* __restrict__ V, const long long* __restrict__ perms,
                                                      const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                                      float* __restrict__ acc, float* __restrict__ mi, int B, int E) {
  __shared__ float red[2][kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, idx = blockIdx.x * kWaves + wave;
  float joint = 0.f, marg = 0.f;
  if (idx < 3 * B) {
    const int m = idx / B, b = idx % B;
    const long pb = perms[idx];
    const float* u = U + (long)b * E;
    const float* vj = V + ((long)m * B + b) * E;
    const float* vm = V + ((long)m * B + pb) * E;
    float hj = 0.f, hm = 0.f;
    for (int c = lane; c < E; c += 64) {
      const float base = u[c] + b0[c], w = w1[c];
      hj = fmaf(fmaxf(base + vj[c], 0.f), w, hj);
      hm = fmaf(fmaxf(base + vm[c], 0.f), w, hm);
    }
    joint = wave_sum(hj) + b1[0];
    marg = __expf(wave_sum(hm) + b1[0]);
  }
  if (lane == 0) { red[0][wave] = joint; red[1][wave] = marg; }
  __syncthreads();
  const int i0 = blockIdx.x * kWaves, i1 = min(i0 + kWaves, 3 * B) - 1;
  if (i0 / B == i1 / B) {           // the block's rows belong to one modality: one pair of adds
    if (threadIdx.x < 2) atomicAdd(acc + 2 * (i0 / B) + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
  } else if (lane == 0 && idx < 3 * B) {
    atomicAdd(acc + 2 * (idx / B), joint);
    atomicAdd(acc + 2 * (idx / B) + 1, marg);
  }
  __threadfence();      // every wave that added (a straddling block adds from all four) has its adds in L2 before the sign-off
  // The block that finishes last turns the sums into the estimates:
  // mi[m] = mean(joint scores) - log(mean(exp(shuffled scores)) + 1e-8)      (graph_fusion.py:497-499)
  // (acc[6] counts finished blocks; the sums are read with atomics, i.e. from L2, where every block's adds have landed)
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(reinterpret_cast<unsigned*>(acc + 6), 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x < 3) {
    const int m = threadIdx.x;
    const float sj = atomicAdd(acc + 2 * m, 0.f), sm = atomicAdd(acc + 2 * m + 1, 0.f);
    mi[m] = sj / (float)B - __logf(sm / (float)B + 1e-8f);
  }
}

// Uniform random row permutations (torch.randperm's role, graph_fusion.py:494): block p sorts B (random key, index)
// pairs in LDS with a bitonic network and writes the indices.  Keys are the counter-based hash of (seed + epoch, p, i);
// equal keys are ordered by index, so the result is a function of the seed alone.
constexpr int kPermMax = 4096;
__global__ __launch_bounds__(1024) void perm_kernel(long long* __restrict__ out, int B, int n2, uint64_t seed, const uint64_t* __restrict__ epoch) {
  __shared__ unsigned long long key[kPermMax];
  if (epoch != nullptr) seed += epoch[0] * 0xD1B54A32D192ED03ull;
  const uint64_t sd = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x + 1);
  for (int i = threadIdx.x; i < n2; i += blockDim.x)
    key[i] = i < B ? ((unsigned long long)ocean_hash32_lo(sd, (uint32_t)i) << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
        const int i = 2 * t - (t & (j - 1)), l = i + j;         // the pair (i, i ^ j) with bit j of i clear
        const bool up = (i & k) == 0;
        const unsigned long long a = key[i], b = key[l];
        if ((a > b) == up) { key[i] = b; key[l] = a; }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < B; i += blockDim.x) out[(size_t)blockIdx.x * B + i] = (long long)(key[i] & 0xFFFFFFFFull);
}

// total = wp * pred + wc * (0 + c[0] + c[1] + ...), in the reference's order of operations and without contraction into
// fused multiply-adds (ocean_model.py:257-266); out = [total, pred, sum of c]
__global__ void loss_combine_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ c, int n_c, float wp, float wc, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float sc = 0.f;
  for (int q = 0; q < n_c; ++q) sc = __fadd_rn(sc, c[q]);
  out[0] = __fadd_rn(__fmul_rn(wp, pred[0]), __fmul_rn(wc, sc));
  out[1] = pred[0];
  out[2] = sc;
}
// d pred = wp * d total;  d c[q] = wc * d total + d csum     (either upstream gradient may be absent)
__global__ void loss_combine_bwd_kernel(const float* __restrict__ d_total, const float* __restrict__ d_csum, int n_c, float wp, float wc,
                                        float* __restrict__ d_pred, float* __restrict__ d_c) {
  const int q = threadIdx.x;
  const float dt = d_total ? d_total[0] : 0.f, ds = d_csum ? d_csum[0] : 0.f;
  if (q == 0) d_pred[0] = wp * dt;
  if (q < n_c) d_c[q] = __fadd_rn(__fmul_rn(wc, dt), ds);
}

// The prediction loss and compute_loss's arithmetic in one launch: every block leaves the sum of its elements' BCE terms
// in state->partial[block]; the block that signs off last adds the partial sums in block order (the same value on every
// run, where atomic adds would vary in the last bits), forms the mean and then [total, pred, sum of c] as
// loss_combine_fwd_kernel does, and puts the counter back to zero for the next call.
__global__ __launch_bounds__(256) void pred_total_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float inv_n,
                                                             const float* __restrict__ c, int n_c, float wp, float wc,
                                                             ocean_loss_state* __restrict__ state, float* __restrict__ out) {
  __shared__ float red[kWaves];
  __shared__ bool last;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(log1pf(-p[i]), -100.f);
    s -= y[i] * lp + (1.0f - y[i]) * lq;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&state->partial[blockIdx.x], red[0] + red[1] + red[2] + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(&state->done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) t += __hip_atomic_load(&state->partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float pred = (red[0] + red[1] + red[2] + red[3]) * inv_n;
  float sc = 0.f;
  for (int q = 0; q < n_c; ++q) sc = __fadd_rn(sc, c[q]);
  out[0] = __fadd_rn(__fmul_rn(wp, pred), __fmul_rn(wc, sc));
  out[1] = pred;
  out[2] = sc;
  state->done = 0u;
}
// dp = (wp * d total + d pred) * (p - y) / max(p (1-p), 1e-12) / n;   d c[q] = wc * d total + d csum   (block 0)
__global__ void pred_total_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, const float* __restrict__ d_total,
                                      const float* __restrict__ d_csum, const float* __restrict__ d_pred, long n, float inv_n, int n_c,
                                      float wp, float wc, float* __restrict__ dp, float* __restrict__ d_c) {
  const float dt = d_total ? d_total[0] : 0.f;
  if (blockIdx.x == 0 && (int)threadIdx.x < n_c) d_c[threadIdx.x] = __fadd_rn(__fmul_rn(wc, dt), d_csum ? d_csum[0] : 0.f);
  const float g = (wp * dt + (d_pred ? d_pred[0] : 0.f)) * inv_n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dp[i] = g * (p[i] - y[i]) / fmaxf(p[i] * (1.0f - p[i]), 1e-12f);
}
}  // namespace

int ocean_pred_total_fwd(const float* probs, const float* targets, long n, long mean_n, const float* c, int n_c, float wp, float wc,
                         ocean_loss_state* state, float* out3, void* stream) {
  if (!probs || !targets || !state || !out3 || (n_c > 0 && !c)) return OCEAN_ERR_NULL;
  if (n <= 0 || n_c < 0 || n_c > 64) return OCEAN_ERR_BAD_SHAPE;
  const int blocks = egrid(n) > OCEAN_LOSS_STATE_BLOCKS ? OCEAN_LOSS_STATE_BLOCKS : egrid(n);
  hipLaunchKernelGGL(pred_total_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, probs, targets, n,
                     1.0f / (float)(mean_n > 0 ? mean_n : n), c, n_c, wp, wc, state, out3);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_pred_total_bwd(const float* probs, const float* targets, const float* d_total, const float* d_csum, const float* d_pred, long n,
                         long mean_n, int n_c, float wp, float wc, float* dprobs, float* d_c, void* stream) {
  if (!probs || !targets || !dprobs || (n_c > 0 && !d_c)) return OCEAN_ERR_NULL;
  if (n <= 0 || n_c < 0 || n_c > 64) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pred_total_bwd_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, probs, targets, d_total, d_csum, d_pred, n,
                     1.0f / (float)(mean_n > 0 ? mean_n : n), n_c, wp, wc, dprobs, d_c);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_draw_permutations(int n_perm, int B, uint64_t seed, const uint64_t* seed_epoch, long long* out, void* stream) {
  if (!out) return OCEAN_ERR_NULL;
  if (n_perm <= 0 || B <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (B > kPermMax) return OCEAN_ERR_UNSUPPORTED;
  int n2 = 2;
  while (n2 < B) n2 <<= 1;
  hipLaunchKernelGGL(perm_kernel, dim3(n_perm), dim3(n2 / 2 < 1024 ? (n2 / 2 < 64 ? 64 : n2 / 2) : 1024), 0, (hipStream_t)stream, out, B, n2, seed, seed_epoch);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_loss_combine_fwd(const float* pred, const float* c, int n_c, float wp, float wc, float* out3, void* stream) {
  if (!pred || !out3 || (n_c > 0 && !c)) return OCEAN_ERR_NULL;
  if (n_c < 0 || n_c > 64) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pred, c, n_c, wp, wc, out3);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_loss_combine_bwd(const float* d_total, const float* d_csum, int n_c, float wp, float wc, float* d_pred, float* d_c, void* stream) {
  if (!d_pred || (n_c > 0 && !d_c)) return OCEAN_ERR_NULL;
  if (n_c < 0 || n_c > 64) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_total, d_csum, n_c, wp, wc, d_pred, d_c);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

size_t ocean_mi_workspace(const ocean_mi_shape* s) {
  return s && s->B > 0 && s->E > 0 ? al4((size_t)s->B * s->E) + al4((size_t)3 * s->B * s->E) + 8 : 0;
}

int ocean_mi_estimates(const ocean_mi_shape* s, const float* fused, const float* X, const long long* perms, const float* w0,
                       const float* b0, const float* w1, const float* b1, float* ws, size_t ws_floats, float* mi, void* stream) {
  if (!s || !fused || !X || !perms || !w0 || !b0 || !w1 || !b1 || !ws || !mi) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->E <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (ws_floats < ocean_mi_workspace(s)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, E = s->E;
  float* U = ws; float* V = U + al4((size_t)B * E); float* acc = V + al4((size_t)3 * B * E);
  OCEAN_TRY(zero_f32(acc, 8, st));
  GemmP u = gemm_defaults();
  u.A = fused; u.lda = E; u.B = w0; u.ldb = 2 * E; u.C = U; u.ldc = E; u.M = B; u.N = E; u.K = E;
  OCEAN_TRY(ocean_launch_gemm(u, st));
  GemmP v[3];
  for (int m = 0; m < 3; ++m) {
    v[m] = gemm_defaults();
    v[m].A = X + (size_t)m * E; v[m].lda = 3 * E; v[m].B = w0 + E; v[m].ldb = 2 * E;
    v[m].C = V + (size_t)m * B * E; v[m].ldc = E; v[m].M = B; v[m].N = E; v[m].K = E;
  }
  OCEAN_TRY(ocean_launch_gemm_batch(v, 3, st));
  hipLaunchKernelGGL(mi_rows_kernel, dim3(ceil_div(3 * B, kWaves)), dim3(256), 0, st, U, V, perms, b0, w1, b1, acc, mi, B, E);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

// ---------------------------------------------------------------- prediction head
size_t ocean_head_workspace(const ocean_head_shape* s) {
  return s ? al4((size_t)s->B * s->h1) + al4((size_t)s->B * s->h2) : 0;
}
size_t ocean_head_bwd_workspace(const ocean_head_shape* s) {
  return s ? al4((size_t)s->B * s->S) + al4((size_t)s->B * s->h1) + al4((size_t)s->B * s->h2) : 0;
}

static DropoutP head_site(const ocean_head_shape* s, int id) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0xC2B2AE3D27D4EB4Full * (uint64_t)(id + 1);
  d.epoch = s->seed_epoch;
  return d;
}

int ocean_head_fwd(const ocean_head_shape* s, const float* x, float* const* P_, float* ws, size_t ws_floats, float* logits,
                   float* probs, void* stream) {
  if (!s || !x || !P_ || !ws || !logits || !probs) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->E <= 0 || s->h1 <= 0 || s->h2 <= 0 || s->S <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (ws_floats < ocean_head_workspace(s)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* a1 = ws; float* a2 = ws + al4((size_t)s->B * s->h1);
  const DropoutP dr0 = head_site(s, 0), dr1 = head_site(s, 1);   // dropout rides in the epilogue of the producing GEMM
  Epi e0, e1; e0.drop = &dr0; e1.drop = &dr1;
  RcScope rc(s->B, st);       // small batches: the three layers and the sigmoid are one launch (rowchain.h)
  OCEAN_TRY(linear_fwd(x, s->E, P_[0], P_[1], a1, s->h1, s->B, s->h1, s->E, 1, st, &e0));
  OCEAN_TRY(linear_fwd(a1, s->h1, P_[2], P_[3], a2, s->h2, s->B, s->h2, s->h1, 1, st, &e1));
  OCEAN_TRY(linear_fwd(a2, s->h2, P_[4], P_[5], logits, s->S, s->B, s->S, s->h2, 0, st));
  if (!rc_sigmoid(logits, probs, s->B, s->S)) {
    hipLaunchKernelGGL(sigmoid_kernel, dim3(egrid((long)s->B * s->S)), dim3(256), 0, st, logits, probs, (long)s->B * s->S);
    OCEAN_LAUNCH_CHECK();
  }
  return rc.finish();
}

int ocean_head_bwd(const ocean_head_shape* s, const float* x, float* const* P_, float* ws, float* wb, size_t wb_floats,
                   const float* probs, const float* dlogits, const float* dprobs, float* const* G_, float* dx, void* stream) {
  if (!s || !x || !P_ || !ws || !wb || !probs || !G_) return OCEAN_ERR_NULL;
  if (wb_floats < ocean_head_bwd_workspace(s)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B;
  const float* a1 = ws; const float* a2 = ws + al4((size_t)B * s->h1);
  float* dl = wb; float* d1 = wb + al4((size_t)B * s->S); float* d2 = d1 + al4((size_t)B * s->h1);
  RcScope rc(B, st);
  if (!rc_sigmoid_bwd(dlogits, dprobs, probs, dl, B, s->S)) {
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(egrid((long)B * s->S)), dim3(256), 0, st, dlogits, dprobs, probs, dl, (long)B * s->S);
    OCEAN_LAUNCH_CHECK();
  }
  const DropoutP dr0 = head_site(s, 0), dr1 = head_site(s, 1);
  Epi e0, e1;      // dropout and ReLU backward on the way out of the data gradients
  e0.drop = &dr0; e0.relu_mask = a1; e0.ld_mask = s->h1;
  e1.drop = &dr1; e1.relu_mask = a2; e1.ld_mask = s->h2;
  // per layer: weight gradient, bias gradient (column sums of the same operand) and data gradient
  OCEAN_TRY(linear_bwd_pair(a2, s->h2, dl, s->S, P_[4], G_[4], G_[5], d2, s->h2, B, s->S, s->h2, 0, st, &e1));
  OCEAN_TRY(linear_bwd_pair(a1, s->h1, d2, s->h2, P_[2], G_[2], G_[3], d1, s->h1, B, s->h2, s->h1, 0, st, &e0));
  if (dx) OCEAN_TRY(linear_bwd_pair(x, s->E, d1, s->h1, P_[0], G_[0], G_[1], dx, s->E, B, s->h1, s->E, 0, st));
  else OCEAN_TRY(linear_bwd_weight(d1, s->h1, x, s->E, G_[0], B, s->h1, s->E, st, G_[1]));
  return rc.finish();
}

// ---------------------------------------------------------------- Trainer's extra loss terms
int ocean_mse_fwd(const float* a, const float* b, long n, float* loss, void* stream) {
  if (!a || !b || !loss) return OCEAN_ERR_NULL;
  if (n <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(egrid(n) > 512 ? 512 : egrid(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, 1.0f / (float)n, loss);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
int ocean_mse_bwd(const float* a, const float* b, const float* dloss, long n, float* da, void* stream) {
  if (!a || !b || !da) return OCEAN_ERR_NULL;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, a, b, dloss, n, 1.0f / (float)n, da);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
size_t ocean_diversity_workspace(int B, int E) { return al4((size_t)B * E) + al4(B) + al4((size_t)B * B); }
int ocean_diversity_fwd(const float* x, int B, int E, float* ws, size_t ws_floats, float* loss, void* stream) {
  if (!x || !ws || !loss) return OCEAN_ERR_NULL;
  if (B <= 0 || E <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (ws_floats < ocean_diversity_workspace(B, E)) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* fh = ws; float* nrm = ws + al4((size_t)B * E); float* S = nrm + al4(B);
  OCEAN_TRY(l2norm_fwd(x, E, fh, E, nrm, B, E, 1e-12f, st));
  GemmP g = gemm_defaults();
  g.A = fh; g.lda = E; g.B = fh; g.ldb = E; g.C = S; g.ldc = B; g.M = B; g.N = B; g.K = E;
  OCEAN_TRY(ocean_launch_gemm(g, st));
  // torch: mean over an empty selection is NaN when B == 1
  const float inv_pairs = 1.0f / ((float)B * (float)(B - 1));
  hipLaunchKernelGGL(offdiag_abs_fwd_kernel, dim3(egrid((long)B * B) > 512 ? 512 : egrid((long)B * B)), dim3(256), 0, st, S, B,
                     B > 1 ? inv_pairs : NAN, loss);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
int ocean_diversity_bwd(int B, int E, float* ws, const float* dloss, float* dx, void* stream) {
  if (!ws || !dx) return OCEAN_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  float* fh = ws; float* nrm = ws + al4((size_t)B * E); float* S = nrm + al4(B);
  hipLaunchKernelGGL(offdiag_abs_bwd_kernel, dim3(egrid((long)B * B)), dim3(256), 0, st, S, B, 1.0f / ((float)B * (float)(B - 1)), dloss);
  OCEAN_LAUNCH_CHECK();
  OCEAN_TRY(linear_bwd_data(S, B, fh, dx, E, B, B, E, 0, st));        // dF = dS . F   ([B,B] . [B,E])
  OCEAN_TRY(l2norm_bwd(dx, E, fh, E, nrm, dx, E, B, E, 1e-12f, st));
  return OCEAN_OK;
}

// ---------------------------------------------------------------- BCE
int ocean_bce_fwd(const float* probs, const float* targets, long n, long mean_n, float* loss, void* stream) {
  if (!probs || !targets || !loss) return OCEAN_ERR_NULL;
  if (n <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(egrid(n) > 512 ? 512 : egrid(n)), dim3(256), 0, (hipStream_t)stream, probs, targets, n,
                     1.0f / (float)(mean_n > 0 ? mean_n : n), loss);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
int ocean_bce_bwd(const float* probs, const float* targets, const float* dloss, long n, long mean_n, float* dprobs, void* stream) {
  if (!probs || !targets || !dprobs) return OCEAN_ERR_NULL;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, probs, targets, dloss, n,
                     1.0f / (float)(mean_n > 0 ? mean_n : n), dprobs);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_bce_elem_fwd(const float* probs, const float* targets, long n, float* out, void* stream) {
  if (!probs || !targets || !out) return OCEAN_ERR_NULL;
  if (n <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(bce_elem_fwd_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, probs, targets, n, out);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
int ocean_bce_elem_bwd(const float* probs, const float* targets, const float* dout, long n, float* dprobs, void* stream) {
  if (!probs || !targets || !dout || !dprobs) return OCEAN_ERR_NULL;
  if (n <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(bce_elem_bwd_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, probs, targets, dout, n, dprobs);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

}  // extern "C"
