// Row-wise kernels shared by the attention / fusion / head drivers: LayerNorm,
// L2 normalisation, dropout, token mean.  One 64-lane wave per row, lanes
// stride the row (coalesced), reductions by wavefront shuffles; rows are
// re-read from L1 rather than cached in registers so any width works.
#include "ops.h"
#include "rowbodies.h"
#include "rowchain.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int kWaves = 4;
constexpr int kLnSets = 3;
struct LnFwdBatch { LnFwdSet s[kLnSets]; };
struct LnBwdBatch { LnBwdSet s[kLnSets]; };

__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdBatch batch, int ldx, int ldy, int rows, int W, float eps, DropoutP dp) {
  dp = dropout_resolve(dp);
  const LnFwdSet& st = batch.s[blockIdx.y];
  const float* __restrict__ x = st.x;
  const float* __restrict__ gamma = st.gamma;
  const float* __restrict__ beta = st.beta;
  float* __restrict__ y = st.y;
  float* __restrict__ stats = st.stats;
  const int lane = threadIdx.x & 63, row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  rowbody::ln_fwd_row(x, ldx, gamma, beta, y, ldy, stats, row, (uint64_t)blockIdx.y * rows + row, W, eps, dp, lane);
}

// Each wave walks rows row0 + w, row0 + w + kWaves, ... of its block's chunk so the
// per-column gamma/beta partial sums stay in registers (up to 16 columns per lane).
constexpr int kLnRowsPerBlock = 8;    // two rows per wave: a wave works through its rows one dependent round trip after the other
constexpr int kMaxColsPerLane = rowbody::kMaxColsPerLane;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdBatch batch, int lddy, int ldx, int lddx, int rows, int W, DropoutP dp) {
  dp = dropout_resolve(dp);
  const LnBwdSet& st = batch.s[blockIdx.y];
  const float* __restrict__ dy = st.dy;
  const float* __restrict__ x = st.x;
  const float* __restrict__ stats = st.stats;
  const float* __restrict__ gamma = st.gamma;
  float* __restrict__ dx = st.dx;
  float* __restrict__ dgamma = st.dgamma;
  float* __restrict__ dbeta = st.dbeta;
  const uint64_t drow0 = (uint64_t)blockIdx.y * rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[kMaxColsPerLane], pb[kMaxColsPerLane];
#pragma unroll
  for (int k = 0; k < kMaxColsPerLane; ++k) { pg[k] = 0.f; pb[k] = 0.f; }
  const int r0 = blockIdx.x * kLnRowsPerBlock, r1 = min(rows, r0 + kLnRowsPerBlock);
  for (int row = r0 + wave; row < r1; row += kWaves)
    rowbody::ln_bwd_row(dy, lddy, x, ldx, stats, gamma, dx, lddx, row, drow0 + row, W, dp, lane, pg, pb);
  __shared__ float red[2][kWaves][64 * kMaxColsPerLane];
#pragma unroll
  for (int k = 0; k < kMaxColsPerLane; ++k) {
    const int c = lane + 64 * k;
    if (c < W) { red[0][wave][c] = pg[k]; red[1][wave][c] = pb[k]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < W; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
    if (dgamma) unsafeAtomicAdd(dgamma + c, a);
    if (dbeta) unsafeAtomicAdd(dbeta + c, b);
  }
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                         float* __restrict__ norms, int rows, int W, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  rowbody::l2norm_fwd_row(x, ldx, y, ldy, norms, row, W, eps, lane);
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                         const float* __restrict__ norms, float* __restrict__ dx, int lddx,
                                                         int rows, int W, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  rowbody::l2norm_bwd_row(dy, lddy, y, ldy, norms, dx, lddx, row, W, eps, lane);
}

__global__ void dropout_kernel(float* __restrict__ x, int ld, int M, int N, int group, DropoutP d) {
  d = dropout_resolve(d);
  const long n = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), c = (int)(i % N);
    const uint64_t idx = group > 1 ? (uint64_t)r * (N / group) + c / group : (uint64_t)i;
    x[(long)r * ld + c] *= dropout_keep(d, idx);
  }
}

// out[g, c] = mean over t of a[g, t, c].  Block = W column groups (VEC columns each) x 256/W row lanes; the row
// lanes split the middle axis and are combined through LDS.
template <int VEC>
__global__ __launch_bounds__(256) void mean_mid_kernel2(const float* __restrict__ a, float* __restrict__ out, int n, int C, int wlog) {
  __shared__ float red[256 * VEC];
  const int W = 1 << wlog, RL = 256 >> wlog;
  const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> wlog;
  const int c0 = (blockIdx.y * W + tx) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (c0 < C) {
    const float* p = a + (long)blockIdx.x * n * C + c0;
#pragma unroll 4
    for (int t = ty; t < n; t += RL) {
      if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4*>(p + (long)t * C);
        acc[0] += x.x; acc[1 % VEC] += x.y; acc[2 % VEC] += x.z; acc[3 % VEC] += x.w;
      } else {
        acc[0] += p[(long)t * C];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = acc[v];
  __syncthreads();
  if (ty == 0 && c0 < C) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float s = 0.f;
      for (int q = 0; q < RL; ++q) s += red[(q * W + tx) * VEC + v];
      out[(long)blockIdx.x * C + c0 + v] = s / (float)n;
    }
  }
}

}  // namespace

int layernorm_fwd_batch(const LnFwdSet* sets, int n, int ldx, int ldy, int rows, int W, float eps, DropoutP in_drop, hipStream_t s) {
  if (rows <= 0 || W <= 0 || n < 1 || n > kLnSets) return OCEAN_ERR_BAD_SHAPE;
  if (rc_active()) {
    // a stage kernel is being assembled (rowchain.h): one operation per set, numbered in the dropout site as here
    int taken = 0;
    while (taken < n && rc_ln_fwd(sets[taken].x, ldx, sets[taken].gamma, sets[taken].beta, sets[taken].y, ldy, sets[taken].stats, rows, W, eps,
                                  in_drop, taken * rows, taken > 0)) ++taken;
    if (taken == n) return OCEAN_OK;
    if (taken > 0) return OCEAN_ERR_UNSUPPORTED;     // (every set of a batch has the same shape: the first one decides)
  }
  LnFwdBatch b;
  for (int q = 0; q < kLnSets; ++q) b.s[q] = sets[q < n ? q : 0];
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, kWaves), n), dim3(256), 0, s, b, ldx, ldy, rows, W, eps, in_drop);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats,
                  int rows, int W, float eps, DropoutP in_drop, hipStream_t s) {
  LnFwdSet st;
  st.x = x; st.gamma = gamma; st.beta = beta; st.y = y; st.stats = stats;
  return layernorm_fwd_batch(&st, 1, ldx, ldy, rows, W, eps, in_drop, s);
}

int layernorm_bwd_batch(const LnBwdSet* sets, int n, int lddy, int ldx, int lddx, int rows, int W, DropoutP in_drop, hipStream_t s) {
  if (rows <= 0 || W <= 0 || n < 1 || n > kLnSets) return OCEAN_ERR_BAD_SHAPE;
  if (W > 64 * kMaxColsPerLane) return OCEAN_ERR_UNSUPPORTED;
  if (rc_active()) {
    int taken = 0;
    while (taken < n && rc_ln_bwd(sets[taken].dy, lddy, sets[taken].x, ldx, sets[taken].stats, sets[taken].gamma, sets[taken].dx, lddx,
                                  sets[taken].dgamma, sets[taken].dbeta, rows, W, in_drop, taken * rows, taken > 0)) ++taken;
    if (taken == n) return OCEAN_OK;
    if (taken > 0) return OCEAN_ERR_UNSUPPORTED;
  }
  LnBwdBatch b;
  for (int q = 0; q < kLnSets; ++q) b.s[q] = sets[q < n ? q : 0];
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(ceil_div(rows, kLnRowsPerBlock), n), dim3(256), 0, s, b, lddy, ldx, lddx, rows, W, in_drop);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma,
                  float* dx, int lddx, float* dgamma, float* dbeta, int rows, int W, DropoutP in_drop, hipStream_t s) {
  LnBwdSet st;
  st.dy = dy; st.x = x; st.stats = stats; st.gamma = gamma; st.dx = dx; st.dgamma = dgamma; st.dbeta = dbeta;
  return layernorm_bwd_batch(&st, 1, lddy, ldx, lddx, rows, W, in_drop, s);
}

int l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* norms, int rows, int W, float eps, hipStream_t s) {
  if (rc_active() && rc_l2norm_fwd(x, ldx, y, ldy, norms, rows, W, eps, false)) return OCEAN_OK;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(rows, kWaves)), dim3(256), 0, s, x, ldx, y, ldy, norms, rows, W, eps);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int l2norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* norms, float* dx, int lddx,
               int rows, int W, float eps, hipStream_t s) {
  if (rc_active() && rc_l2norm_bwd(dy, lddy, y, ldy, norms, dx, lddx, rows, W, eps, false)) return OCEAN_OK;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(rows, kWaves)), dim3(256), 0, s, dy, lddy, y, ldy, norms, dx, lddx, rows, W, eps);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int dropout_inplace(float* x, int ld, int M, int N, int group, DropoutP d, hipStream_t s) {
  if (d.p <= 0.f) return OCEAN_OK;
  rc_flush();
  long g = ((long)M * N + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((int)g), dim3(256), 0, s, x, ld, M, N, group, d);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int mean_mid(const float* a, float* out, int G, int n, int C, hipStream_t s) {
  if (rc_active() && rc_mean_mid(a, out, G, n, C)) return OCEAN_OK;
  const bool v4 = C % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
  const int cg = v4 ? C / 4 : C;
  int wlog = 0;
  while ((1 << wlog) < cg && wlog < 8) ++wlog;
  const dim3 grid(G, ceil_div(cg, 1 << wlog));
  if (v4) hipLaunchKernelGGL(mean_mid_kernel2<4>, grid, dim3(256), 0, s, a, out, n, C, wlog);
  else hipLaunchKernelGGL(mean_mid_kernel2<1>, grid, dim3(256), 0, s, a, out, n, C, wlog);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
