// DilatedCNN forward / backward driver and its BatchNorm / pooling kernels.
//
// Data layout: activations are channels-last [B*T, C] (row = one (sample,
// timestep)); the reference's [B,C,T] (dilated_cnn.py:211) is never formed.
// Per BatchNorm layer j (blocks and transitions interleaved, forward order) the
// forward workspace keeps u_j (pre-BN convolution output) and a_j (the layer's
// activation) for the backward pass, plus fp64 column statistics.
//
// Training-mode BatchNorm needs full-batch column statistics between the
// convolution and its activation, so the driver is cut into phases at those
// points: the convolution GEMM accumulates (sum, sum of squares) per channel in
// its epilogue; the next phase finalises them (a data-parallel caller
// all-reduces the fp64 block in between) and applies scale/shift + ReLU
// (+ residual) in one elementwise pass.
#include "../../include/ocean_hip.h"
#include "ops.h"
#include <stdlib.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int kMaxBN = 2 * OCEAN_MAX_CONV_LAYERS - 1;
// Column statistics are accumulated with fp64 atomics; atomics on one address serialise in L2, so blocks spread
// their partial sums over kStatSlots copies of the statistics block, which a small kernel then adds up.
constexpr int kStatSlots = 32;

struct DcnnPlan {
  int M, L, nbn, cmax;
  int width[kMaxBN], in_width[kMaxBN], is_block[kMaxBN], li[kMaxBN];
  size_t h0, u[kMaxBN], a[kMaxBN], pooled;
  size_t last_cnt, last_sx;      // [B][C] per-sample sums of the last layer's mask and mask * xhat (see bn_apply_pool_kernel)
  size_t stats[kMaxBN], fslots[kMaxBN], mean[kMaxBN], invstd[kMaxBN], scale[kMaxBN], shift[kMaxBN];
  size_t wp[OCEAN_MAX_CONV_LAYERS];
  size_t wpT[OCEAN_MAX_CONV_LAYERS], wtT[OCEAN_MAX_CONV_LAYERS];   // transposed copies: block taps [Kt][Ci][Co], transitions [Ci][Co]
  size_t fwd_total;
  size_t gA, gB, dU, dpool, bstats[kMaxBN], bslots[kMaxBN], dwp[OCEAN_MAX_CONV_LAYERS], bwd_total;
};

inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

int make_plan(const ocean_dcnn_shape* s, DcnnPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->T <= 0 || s->F <= 0 || s->L < 1 || s->L > OCEAN_MAX_CONV_LAYERS ||
      s->out_dim <= 0 || s->kernel_size < 1 || (s->kernel_size & 1) == 0)
    return OCEAN_ERR_BAD_SHAPE;
  for (int i = 0; i < s->L; ++i)
    if (s->channels[i] <= 0 || s->dilations[i] <= 0) return OCEAN_ERR_BAD_SHAPE;
  p->M = s->B * s->T;
  p->L = s->L;
  p->nbn = 2 * s->L - 1;
  p->cmax = 0;
  const size_t M = (size_t)p->M;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->h0 = take(M * s->channels[0]);
  for (int j = 0; j < p->nbn; ++j) {
    const int i = j / 2;
    p->is_block[j] = (j % 2 == 0);
    p->li[j] = i;
    p->in_width[j] = s->channels[i];
    p->width[j] = p->is_block[j] ? s->channels[i] : s->channels[i + 1];
    if (p->width[j] > p->cmax) p->cmax = p->width[j];
    p->u[j] = take(M * p->width[j]);
    p->a[j] = take(M * p->width[j]);
  }
  p->pooled = take((size_t)s->B * s->channels[s->L - 1]);
  p->last_cnt = take((size_t)s->B * s->channels[s->L - 1]);
  p->last_sx = take((size_t)s->B * s->channels[s->L - 1]);
  for (int j = 0; j < p->nbn; ++j) {
    const size_t C = p->width[j];
    p->stats[j] = take(4 * C);   // 2*C doubles
    p->mean[j] = take(C); p->invstd[j] = take(C); p->scale[j] = take(C); p->shift[j] = take(C);
  }
  for (int j = 0; j < p->nbn; ++j) p->fslots[j] = take(4 * (size_t)p->width[j] * kStatSlots);
  for (int i = 0; i < s->L; ++i) p->wp[i] = take((size_t)s->kernel_size * s->channels[i] * s->channels[i]);
  for (int i = 0; i < s->L; ++i) p->wpT[i] = take((size_t)s->kernel_size * s->channels[i] * s->channels[i]);
  for (int i = 0; i + 1 < s->L; ++i) p->wtT[i] = take((size_t)s->channels[i] * s->channels[i + 1]);
  p->fwd_total = o;
  o = 0;
  p->gA = take(M * p->cmax); p->gB = take(M * p->cmax); p->dU = take(M * p->cmax);
  p->dpool = take((size_t)s->B * s->channels[s->L - 1]);
  for (int j = 0; j < p->nbn; ++j) p->bstats[j] = take(4 * (size_t)p->width[j]);
  for (int j = 0; j < p->nbn; ++j) p->bslots[j] = take(4 * (size_t)p->width[j] * kStatSlots);
  for (int i = 0; i < s->L; ++i) p->dwp[i] = take((size_t)s->kernel_size * s->channels[i] * s->channels[i]);
  p->bwd_total = o;
  return OCEAN_OK;
}

// parameter table indices
inline int pi_block(int i, int k) { return 2 + 5 * i + k; }                 // k: 0 conv_w 1 bn_w 2 bn_b 3 rm 4 rv
inline int pi_trans(int L, int i, int k) { return 2 + 5 * L + 5 * i + k; }
inline int pi_out(int L, int k) { return 2 + 5 * L + 5 * (L - 1) + k; }
inline int pi_bn(const DcnnPlan& p, int j, int k) { return p.is_block[j] ? pi_block(p.li[j], k) : pi_trans(p.L, p.li[j], k); }

// ------------------------------------------------------------------ kernels
// stats: [2][C] sums, or - with nslots > 0 - nslots copies of them that are added up here (the convolution epilogue's
// slot copies, when no other rank's sums have to be added in between)
constexpr int kFinCh = 32, kFinParts = 8;
__global__ __launch_bounds__(kFinCh * kFinParts) void bn_finalize_kernel(const double* __restrict__ stats, int nslots, double n, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int C, int training, float eps, float momentum) {
  // 32 channels per block, 8 threads per channel: each takes every 8th slot copy (loads that do not depend on each
  // other), then the 8 partial sums meet in shared memory; one thread per channel walking all copies in turn spent
  // ~10 us on a chain of dependent L2 round trips
  __shared__ double part[2][kFinParts][kFinCh];
  const int cl = threadIdx.x % kFinCh, pt = threadIdx.x / kFinCh;
  const int c = blockIdx.x * kFinCh + cl;
  if (training) {
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
      if (nslots <= 0) { if (pt == 0) { a1 = stats[c]; a2 = stats[C + c]; } }
      else for (int q = pt; q < nslots; q += kFinParts) { a1 += stats[(size_t)q * 2 * C + c]; a2 += stats[(size_t)q * 2 * C + C + c]; }
    }
    part[0][pt][cl] = a1; part[1][pt][cl] = a2;
    __syncthreads();
  }
  if (pt != 0 || c >= C) return;
  float mu, var;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < kFinParts; ++q) { s1 += part[0][q][cl]; s2 += part[1][q][cl]; }
    const double m = s1 / n;
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
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
}

// Elementwise BatchNorm kernels share one thread layout: a block is W column groups (VEC consecutive channels
// each, W a power of two) by 256/W row lanes, so a thread keeps its channels' coefficients in registers, walks
// down rows, and no index needs a division.  Consecutive lanes read consecutive addresses of a row; with W < 64
// a wavefront covers several whole rows, which are contiguous too.
struct ColTile {
  int wlog;            // log2(W)
  int rows;            // rows per block
};
inline ColTile col_tile(int C, int vec, int M, int rows_per_thread, int min_rows) {
  const int cg = (C + vec - 1) / vec;
  int wlog = 0;
  while ((1 << wlog) < cg && wlog < 8) ++wlog;
  ColTile t;
  t.wlog = wlog;
  t.rows = (256 >> wlog) * rows_per_thread;
  if (t.rows < min_rows) t.rows = min_rows;
  while ((M + t.rows - 1) / t.rows > 65535) t.rows *= 2;   // grid.y limit
  return t;
}
inline dim3 col_grid(const ColTile& t, int C, int vec, int M) {
  const int cg = (C + vec - 1) / vec;
  return dim3(ceil_div(cg, 1 << t.wlog), ceil_div(M, t.rows));
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if (VEC == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1 % VEC] = a.y; v[2 % VEC] = a.z; v[3 % VEC] = a.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
  else *p = v[0];
}

// out = dropout(relu(u*scale + shift)) (+ res)
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ u, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res,
                                                           float* __restrict__ out, int M, int C, ColTile ct, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int W = 1 << ct.wlog;
  const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> ct.wlog, RL = 256 >> ct.wlog;
  const int c0 = (blockIdx.x * W + tx) * VEC;
  if (c0 >= C) return;
  const int r0 = blockIdx.y * ct.rows, r1 = min(M, r0 + ct.rows);
  float sc[VEC], sh[VEC];
  load_vec<VEC>(scale + c0, sc);
  load_vec<VEC>(shift + c0, sh);
#pragma unroll 2
  for (int r = r0 + ty; r < r1; r += RL) {
    const long i = (long)r * C + c0;
    float x[VEC], y[VEC];
    load_vec<VEC>(u + i, x);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      y[v] = fmaxf(fmaf(x[v], sc[v], sh[v]), 0.f);
      if (dp.p > 0.f) y[v] *= dropout_keep(dp, (uint64_t)(i + v));
    }
    if (res) {
      float q[VEC];
      load_vec<VEC>(res + i, q);
#pragma unroll
      for (int v = 0; v < VEC; ++v) y[v] += q[v];
    }
    store_vec<VEC>(out + i, y);
  }
}

// The last layer's apply pass with what follows it folded in.  Its output is only ever averaged over time (the
// pooling), and in the backward pass its incoming gradient is that pooled gradient broadcast over time: constant per
// (sample, channel).  So one block takes one sample and 256 channels, walks the sample's T rows once and leaves per
// (sample, channel) - the activation itself is not written unless `out` is given -: the time mean (the pooled output - no second pass over the
// activation), the sum of m_t = [u*scale+shift > 0] * keep_t and the sum of m_t * xhat_t.  The BatchNorm-backward
// column sums of this layer are then sum_b dpool[b][c]/T * cnt[b][c] and sum_b dpool[b][c]/T * sx[b][c]: a pass over
// [B, C] instead of one over [B*T, C] (bn_last_reduce_kernel).
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float* __restrict__ u, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ res,
                                                            float* __restrict__ out, float* __restrict__ pooled,
                                                            float* __restrict__ cnt, float* __restrict__ sxo, int T, int C, DropoutP dp) {
  dp = dropout_resolve(dp);
  __shared__ float red[3][4][256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + tx) * 4, b = blockIdx.y;
  float ps[4] = {0.f, 0.f, 0.f, 0.f}, pc[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    float sc[4], sh[4], mu[4], is[4];
    load_vec<4>(scale + c0, sc); load_vec<4>(shift + c0, sh); load_vec<4>(mean + c0, mu); load_vec<4>(invstd + c0, is);
#pragma unroll 2
    for (int t = ty; t < T; t += 4) {
      const long i = ((long)b * T + t) * C + c0;
      float x[4], y[4];
      load_vec<4>(u + i, x);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float z = fmaf(x[v], sc[v], sh[v]);
        float m = z > 0.f ? 1.f : 0.f;
        if (dp.p > 0.f) m *= dropout_keep(dp, (uint64_t)(i + v));
        y[v] = fmaxf(z, 0.f) * (dp.p > 0.f ? m : 1.f);
        pc[v] += m;
        px[v] = fmaf(m, (x[v] - mu[v]) * is[v], px[v]);
      }
      if (res) {
        float q[4];
        load_vec<4>(res + i, q);
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] += q[v];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) ps[v] += y[v];
      if (out != nullptr) store_vec<4>(out + i, y);
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) { red[0][ty][tx * 4 + v] = ps[v]; red[1][ty][tx * 4 + v] = pc[v]; red[2][ty][tx * 4 + v] = px[v]; }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const int l = threadIdx.x;
    const long o = (long)b * C + c;
    pooled[o] = (red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]) * (1.0f / (float)T);
    cnt[o] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    sxo[o] = red[2][0][l] + red[2][1][l] + red[2][2][l] + red[2][3][l];
  }
}

// bslots += (sum_b dpool*cnt, sum_b dpool*sx) / T for the last layer: a block takes 64 channels of kLastRows samples
// and adds its partial sums into slot copy blockIdx.y % kStatSlots
constexpr int kLastRows = 32;
__global__ __launch_bounds__(256) void bn_last_reduce_kernel(const float* __restrict__ dpool, const float* __restrict__ cnt,
                                                             const float* __restrict__ sxi, double* __restrict__ bslots, int B, int C, float inv_T) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int b0 = blockIdx.y * kLastRows, b1 = min(B, b0 + kLastRows);
  float sg = 0.f, sx = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int b = b0 + ty; b < b1; b += 4) {
      const long o = (long)b * C + c;
      const float g = dpool[o] * inv_T;
      sg = fmaf(g, cnt[o], sg);
      sx = fmaf(g, sxi[o], sx);
    }
  }
  red[0][ty][tx] = sg; red[1][ty][tx] = sx;
  __syncthreads();
  if (ty == 0 && c < C) {
    double* slot = bslots + (size_t)(blockIdx.y % kStatSlots) * 2 * C;
    unsafeAtomicAdd(slot + c, (double)(red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx]));
    unsafeAtomicAdd(slot + C + c, (double)(red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx]));
  }
}

// Column sums of g and g*xhat, g = dy * (u*scale+shift > 0), xhat = (u-mean)*invstd; fp64 atomics per column.
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, int bcast_T, float dy_scale,
                                                            const float* __restrict__ u, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ bslots,
                                                            int M, int C, ColTile ct, DropoutP dp) {
  dp = dropout_resolve(dp);
  __shared__ float red[2][256 * VEC];
  const int W = 1 << ct.wlog;
  const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> ct.wlog, RL = 256 >> ct.wlog;
  const int c0 = (blockIdx.x * W + tx) * VEC;
  const int r0 = blockIdx.y * ct.rows, r1 = min(M, r0 + ct.rows);
  float sg[VEC], sx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { sg[v] = 0.f; sx[v] = 0.f; }
  if (c0 < C) {
    float sc[VEC], sh[VEC], mu[VEC], is[VEC];
    load_vec<VEC>(scale + c0, sc); load_vec<VEC>(shift + c0, sh);
    load_vec<VEC>(mean + c0, mu); load_vec<VEC>(invstd + c0, is);
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += RL) {
      float uu[VEC], gg[VEC];
      const long ui = (long)r * C + c0;
      const long gi = bcast_T > 0 ? (long)(r / bcast_T) * C + c0 : ui;
      load_vec<VEC>(u + ui, uu);
      load_vec<VEC>(dy + gi, gg);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float g = (fmaf(uu[v], sc[v], sh[v]) > 0.f) ? gg[v] * dy_scale : 0.f;
        if (dp.p > 0.f) g *= dropout_keep(dp, (uint64_t)(ui + v));
        sg[v] += g;
        sx[v] += g * (uu[v] - mu[v]) * is[v];
      }
    }
  }
  // red[.][(ty*W + tx)*VEC + v]
#pragma unroll
  for (int v = 0; v < VEC; ++v) { red[0][threadIdx.x * VEC + v] = sg[v]; red[1][threadIdx.x * VEC + v] = sx[v]; }
  __syncthreads();
  if (ty == 0 && c0 < C) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < RL; ++q) { a += red[0][(q * W + tx) * VEC + v]; b += red[1][(q * W + tx) * VEC + v]; }
      double* slot = bslots + (size_t)(blockIdx.y % kStatSlots) * 2 * C;
      unsafeAtomicAdd(slot + c0 + v, (double)a);
      unsafeAtomicAdd(slot + C + c0 + v, (double)b);
    }
  }
}

// du = gamma*invstd * (g - sum(g)/n - xhat*sum(g*xhat)/n)   (training)
// du = g * scale                                            (eval: fixed statistics)
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int bcast_T, float dy_scale,
                                                           const float* __restrict__ u, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const double* __restrict__ bstats,
                                                           double n, float* __restrict__ du, int M, int C, int training,
                                                           ColTile ct, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int W = 1 << ct.wlog;
  const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> ct.wlog, RL = 256 >> ct.wlog;
  const int c0 = (blockIdx.x * W + tx) * VEC;
  if (c0 >= C) return;
  const int r0 = blockIdx.y * ct.rows, r1 = min(M, r0 + ct.rows);
  // per channel: du = k1*g - k0 - k2*u  with the batch terms folded (training), or k1*g (eval)
  float sc[VEC], sh[VEC], k0[VEC], k2[VEC];
  load_vec<VEC>(scale + c0, sc); load_vec<VEC>(shift + c0, sh);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (training) {
      const float mg = (float)(bstats[c0 + v] / n), mgx = (float)(bstats[C + c0 + v] / n);
      const float mu = mean[c0 + v], is = invstd[c0 + v];
      // sc*(g - mg - (u-mu)*is*mgx)
      k2[v] = sc[v] * is * mgx;
      k0[v] = sc[v] * mg - k2[v] * mu;
    } else {
      k2[v] = 0.f; k0[v] = 0.f;
    }
  }
#pragma unroll 2
  for (int r = r0 + ty; r < r1; r += RL) {
    float uu[VEC], gg[VEC], o[VEC];
    const long ui = (long)r * C + c0;
    const long gi = bcast_T > 0 ? (long)(r / bcast_T) * C + c0 : ui;
    load_vec<VEC>(u + ui, uu);
    load_vec<VEC>(dy + gi, gg);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float g = (fmaf(uu[v], sc[v], sh[v]) > 0.f) ? gg[v] * dy_scale : 0.f;
      if (dp.p > 0.f) g *= dropout_keep(dp, (uint64_t)(ui + v));
      o[v] = fmaf(sc[v], g, -fmaf(k2[v], uu[v], k0[v]));
    }
    store_vec<VEC>(du + ui, o);
  }
}

// stats = sum of the slot copies the convolution epilogue accumulated into
__global__ void bn_stats_collapse_kernel(const double* __restrict__ slots, double* __restrict__ stats, int n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0;
  for (int q = 0; q < kStatSlots; ++q) s += slots[(size_t)q * n + c];
  stats[c] = s;
}

// bstats = sum of the slots; dbeta += sum(g), dgamma += sum(g*xhat)
__global__ void bn_bwd_param_kernel(const double* __restrict__ bslots, double* __restrict__ bstats, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sg = 0.0, sx = 0.0;
  for (int q = 0; q < kStatSlots; ++q) { sg += bslots[(size_t)q * 2 * C + c]; sx += bslots[(size_t)q * 2 * C + C + c]; }
  bstats[c] = sg; bstats[C + c] = sx;
  if (dbeta) dbeta[c] += (float)sg;
  if (dgamma) dgamma[c] += (float)sx;
}

// Per-tap weight layouts of all convolution layers in one launch (blockIdx.y = set):
// w [Co][Ci][Kt] -> wp [Kt][Co][Ci] (the forward product's [out][in] operand, and the data gradient's [k][n] operand) and
// wpT [Kt][Ci][Co] (the data gradient's operand with the reduction index contiguous) before the forward pass;
// dw [Co][Ci][Kt] += dwp [Kt][Co][Ci] after the backward pass.  A 1x1 transition is a set with Kt = 1 and no wp.
struct RepackSet { const float* src; float* dst; float* dstT; int Co, Ci, Kt; };
struct RepackBatch { RepackSet s[2 * OCEAN_MAX_CONV_LAYERS]; };
__global__ void repack_w_kernel(const RepackBatch b) {
  const RepackSet r = b.s[blockIdx.y];
  const int Kt = r.Kt;
  const long n = (long)r.Co * r.Ci * Kt;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Kt);
    const long oc = i / Kt;          // co*Ci + ci
    const float v = r.src[i];
    if (r.dst) r.dst[(long)t * r.Co * r.Ci + oc] = v;
    const long co = oc / r.Ci, ci = oc % r.Ci;
    r.dstT[((long)t * r.Ci + ci) * r.Co + co] = v;
  }
}
__global__ void unpack_dw_kernel(const RepackBatch b) {
  const RepackSet r = b.s[blockIdx.y];
  const int Kt = r.Kt;
  const long n = (long)r.Co * r.Ci * Kt;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Kt);
    const long oc = i / Kt;
    r.dst[i] += r.src[(long)t * r.Co * r.Ci + oc];
  }
}

inline int egrid(long n, int per = 256) {
  long g = (n + per - 1) / per;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

int bn_finalize(const DcnnPlan& pl, const ocean_dcnn_shape* s, int j, float* const* params, float* ws, bool from_slots,
                hipStream_t st) {
  const int C = pl.width[j];
  const double n = (double)(s->bn_count > 0 ? s->bn_count : pl.M);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, kFinCh)), dim3(kFinCh * kFinParts), 0, st,
                     reinterpret_cast<const double*>(ws + (from_slots ? pl.fslots[j] : pl.stats[j])), from_slots ? kStatSlots : 0, n, params[pi_bn(pl, j, 1)], params[pi_bn(pl, j, 2)],
                     params[pi_bn(pl, j, 3)], params[pi_bn(pl, j, 4)], ws + pl.mean[j], ws + pl.invstd[j],
                     ws + pl.scale[j], ws + pl.shift[j], C, s->training, s->eps, s->momentum);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

inline DropoutP layer_dropout(const ocean_dcnn_shape* s, int j) {
  DropoutP dp;
  dp.p = s->training ? s->dropout : 0.f;
  dp.seed = s->seed + 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1);
  dp.epoch = s->seed_epoch;
  return dp;
}

int bn_apply_fwd(const DcnnPlan& pl, const ocean_dcnn_shape* s, int j, float* ws, const float* res, hipStream_t st) {
  const int C = pl.width[j];
  const DropoutP dp = layer_dropout(s, j);
  if (C % 4 == 0) {
    const ColTile ct = col_tile(C, 4, pl.M, 8, 1);
    hipLaunchKernelGGL(bn_apply_fwd_kernel<4>, col_grid(ct, C, 4, pl.M), dim3(256), 0, st, ws + pl.u[j], ws + pl.scale[j],
                       ws + pl.shift[j], res, ws + pl.a[j], pl.M, C, ct, dp);
  } else {
    const ColTile ct = col_tile(C, 1, pl.M, 8, 1);
    hipLaunchKernelGGL(bn_apply_fwd_kernel<1>, col_grid(ct, C, 1, pl.M), dim3(256), 0, st, ws + pl.u[j], ws + pl.scale[j],
                       ws + pl.shift[j], res, ws + pl.a[j], pl.M, C, ct, dp);
  }
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

// the convolution of BN layer j: input activation -> u_j, column statistics in the epilogue
// collapse: add the epilogue's slot copies into stats[j] (needed when the caller all-reduces that block over ranks
// before the next phase; otherwise bn_finalize adds them itself)
int conv_fwd(const DcnnPlan& pl, const ocean_dcnn_shape* s, int j, const float* in, float* const* params, float* ws, bool collapse,
             hipStream_t st) {
  GemmP g = gemm_defaults();
  g.A = in; g.lda = pl.in_width[j]; g.C = ws + pl.u[j]; g.ldc = pl.width[j];
  g.M = pl.M; g.N = pl.width[j]; g.K = pl.in_width[j];
  g.colstats = s->training ? reinterpret_cast<double*>(ws + pl.fslots[j]) : nullptr;
  g.colstat_slots = kStatSlots;
  if (pl.is_block[j]) {
    const int i = pl.li[j], k = s->kernel_size;
    g.B = ws + pl.wp[i]; g.ldb = pl.in_width[j];
    if (k > 1) {
      g.taps = k; g.T = s->T; g.shift0 = -(k / 2) * s->dilations[i]; g.shift_step = s->dilations[i];
      g.tap_stride_b = (long)pl.width[j] * pl.in_width[j];
    }
  } else {
    g.B = params[pi_trans(pl.L, pl.li[j], 0)]; g.ldb = pl.in_width[j];
  }
  OCEAN_TRY(ocean_launch_gemm(g, st));
  if (s->training && collapse) {
    const int n2 = 2 * pl.width[j];
    hipLaunchKernelGGL(bn_stats_collapse_kernel, dim3(ceil_div(n2, 64)), dim3(64), 0, st,
                       reinterpret_cast<const double*>(ws + pl.fslots[j]), reinterpret_cast<double*>(ws + pl.stats[j]), n2);
    OCEAN_LAUNCH_CHECK();
  }
  return OCEAN_OK;
}

// whether the last layer's apply pass also pools and leaves the per-sample sums its backward reduction needs
inline bool last_pool_fused(const DcnnPlan& pl) {
  static const bool off = getenv("OCEAN_DCNN_SEPARATE_POOL") != nullptr;     // measurement switch
  return !off && pl.width[pl.nbn - 1] % 4 == 0;
}

const float* layer_input(const DcnnPlan& pl, int j, const float* ws) { return j == 0 ? ws + pl.h0 : ws + pl.a[j - 1]; }

}  // namespace

extern "C" {

size_t ocean_dcnn_param_count(const ocean_dcnn_shape* s) { return s ? (size_t)(2 + 5 * s->L + 5 * (s->L - 1) + 2) : 0; }

size_t ocean_dcnn_fwd_workspace(const ocean_dcnn_shape* s) {
  DcnnPlan p;
  return make_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0;
}
size_t ocean_dcnn_bwd_workspace(const ocean_dcnn_shape* s) {
  DcnnPlan p;
  return make_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0;
}
int ocean_dcnn_num_phases(const ocean_dcnn_shape* s) { return s ? 2 * s->L : 0; }

int ocean_dcnn_stats_block(const ocean_dcnn_shape* s, int phase, int backward, size_t* offset_floats, int* n_doubles) {
  DcnnPlan p;
  OCEAN_TRY(make_plan(s, &p));
  if (phase < 0 || phase >= p.nbn) return OCEAN_ERR_BAD_SHAPE;
  // forward phase j completes stats of BN layer j; backward phase j completes layer nbn-1-j
  const int j = backward ? p.nbn - 1 - phase : phase;
  if (offset_floats) *offset_floats = backward ? p.bstats[j] : p.stats[j];
  if (n_doubles) *n_doubles = 2 * p.width[j];
  return OCEAN_OK;
}

int ocean_dcnn_block_output(const ocean_dcnn_shape* s, int block, size_t* offset_floats, int* width) {
  DcnnPlan p;
  OCEAN_TRY(make_plan(s, &p));
  if (block < 0 || block >= s->L) return OCEAN_ERR_BAD_SHAPE;
  if (block == s->L - 1 && last_pool_fused(p)) return OCEAN_ERR_UNSUPPORTED;   // only its time mean is kept: ocean_dcnn_pooled_output
  if (offset_floats) *offset_floats = p.a[2 * block];      // BN layers alternate block, transition, block, ...
  if (width) *width = p.width[2 * block];
  return OCEAN_OK;
}

int ocean_dcnn_pooled_output(const ocean_dcnn_shape* s, size_t* offset_floats, int* width) {
  DcnnPlan p;
  OCEAN_TRY(make_plan(s, &p));
  if (offset_floats) *offset_floats = p.pooled;
  if (width) *width = s->channels[s->L - 1];
  return OCEAN_OK;
}

int ocean_dcnn_fwd(const ocean_dcnn_shape* s, const float* x, float* const* params, float* ws, size_t ws_floats,
                   float* out, int phase_lo, int phase_hi, void* stream) {
  DcnnPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!x || !params || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int L = s->L, nph = 2 * L;
  if (phase_lo < 0) phase_lo = 0;
  if (phase_hi > nph) phase_hi = nph;
  for (int ph = phase_lo; ph < phase_hi; ++ph) {
    if (ph == 0) {
      if (s->training) OCEAN_TRY(zero_f32(ws + pl.stats[0], pl.wp[0] - pl.stats[0], st));
      RepackBatch rb;
      long biggest = 0;
      for (int i = 0; i < L; ++i) {
        const int C = s->channels[i];
        rb.s[i] = RepackSet{params[pi_block(i, 0)], ws + pl.wp[i], ws + pl.wpT[i], C, C, s->kernel_size};
        if ((long)C * C * s->kernel_size > biggest) biggest = (long)C * C * s->kernel_size;
      }
      for (int i = 0; i + 1 < L; ++i) {
        rb.s[L + i] = RepackSet{params[pi_trans(L, i, 0)], nullptr, ws + pl.wtT[i], s->channels[i + 1], s->channels[i], 1};
        if ((long)s->channels[i] * s->channels[i + 1] > biggest) biggest = (long)s->channels[i] * s->channels[i + 1];
      }
      hipLaunchKernelGGL(repack_w_kernel, dim3(egrid(biggest), 2 * L - 1), dim3(256), 0, st, rb);
      OCEAN_LAUNCH_CHECK();
      OCEAN_TRY(linear_fwd(x, s->F, params[0], params[1], ws + pl.h0, s->channels[0], pl.M, s->channels[0], s->F, 0, st));
      OCEAN_TRY(conv_fwd(pl, s, 0, ws + pl.h0, params, ws, ph + 1 >= phase_hi, st));
    } else {
      const int j = ph - 1;   // finish BN layer j
      // the statistics were collapsed (and possibly all-reduced by the caller) iff this call starts at this phase
      OCEAN_TRY(bn_finalize(pl, s, j, params, ws, s->training && ph > phase_lo, st));
      const bool pool_fused = ph == nph - 1 && last_pool_fused(pl);
      if (pool_fused) {
        const int C = pl.width[j];
        hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(ceil_div(C, 256), s->B), dim3(256), 0, st, ws + pl.u[j], ws + pl.scale[j],
                           ws + pl.shift[j], ws + pl.mean[j], ws + pl.invstd[j], pl.is_block[j] ? layer_input(pl, j, ws) : nullptr,
                           nullptr, ws + pl.pooled, ws + pl.last_cnt, ws + pl.last_sx, s->T, C, layer_dropout(s, j));
        OCEAN_LAUNCH_CHECK();
      } else {
        OCEAN_TRY(bn_apply_fwd(pl, s, j, ws, pl.is_block[j] ? layer_input(pl, j, ws) : nullptr, st));
      }
      if (ph < nph - 1) {
        OCEAN_TRY(conv_fwd(pl, s, j + 1, ws + pl.a[j], params, ws, ph + 1 >= phase_hi, st));
      } else {
        const int C = s->channels[L - 1];
        if (!pool_fused) OCEAN_TRY(mean_mid(ws + pl.a[j], ws + pl.pooled, s->B, s->T, C, st));
        OCEAN_TRY(linear_fwd(ws + pl.pooled, C, params[pi_out(L, 0)], params[pi_out(L, 1)], out, s->out_dim, s->B, s->out_dim, C, 0, st));
      }
    }
  }
  return OCEAN_OK;
}

}  // extern "C"

namespace {

// parts: 1 = what the NEXT phase's statistics wait for (this layer's BatchNorm-backward apply pass and data gradient, the
// column sums of the layer below), 2 = the rest of the phase (the layer's weight gradient; in the last phase also the
// input projection's gradients and the unpacking of the per-tap weight gradients).  A data-parallel caller runs part 1,
// starts the all-reduce of the statistics block, runs part 2 under it, and waits before the next phase.
int dcnn_bwd_impl(const ocean_dcnn_shape* s, const float* x, float* const* params, float* wf, float* wb, size_t wb_floats,
                  const float* dout, float* const* grads, float* dx, int phase_lo, int phase_hi, int parts, void* stream) {
  DcnnPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!x || !params || !wf || !wb || !dout || !grads) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int L = s->L, nph = 2 * L, M = pl.M;
  const double n = (double)(s->bn_count > 0 ? s->bn_count : pl.M);
  if (phase_lo < 0) phase_lo = 0;
  if (phase_hi > nph) phase_hi = nph;
  const int Cl = s->channels[L - 1];

  // which gradient buffer holds d(a_j) when BN layer j is processed: layers are
  // walked from the last to the first; a block keeps the buffer (in-place
  // residual), a transition flips it.
  auto grad_buf = [&](int j) {
    int flips = 0;
    for (int q = pl.nbn - 1; q > j; --q) if (!pl.is_block[q]) ++flips;
    return (flips & 1) ? wb + pl.gB : wb + pl.gA;
  };

  auto reduce = [&](int j) -> int {
    const int C = pl.width[j];
    const bool last = (j == pl.nbn - 1);
    const float* dy = last ? wb + pl.dpool : grad_buf(j);
    const int bT = last ? s->T : 0;
    const float sc = last ? 1.0f / (float)s->T : 1.0f;
    double* slots = reinterpret_cast<double*>(wb + pl.bslots[j]);
    if (last && last_pool_fused(pl)) {
      // the forward pass left this layer's per-sample mask sums: the reduction runs over [B, C]
      hipLaunchKernelGGL(bn_last_reduce_kernel, dim3(ceil_div(C, 64), ceil_div(s->B, kLastRows)), dim3(256), 0, st, wb + pl.dpool, wf + pl.last_cnt, wf + pl.last_sx,
                         slots, s->B, C, sc);
      OCEAN_LAUNCH_CHECK();
      return OCEAN_OK;
    }
    // about 1024 row blocks: every block ends with 2*C fp64 atomics, and measured over 256..8192 blocks the pass is
    // fastest here (fewer blocks leave CUs idle at the tail, more pile up on the atomics)
    const int min_rows = ceil_div(M, 1024);
    if (C % 4 == 0) {
      const ColTile ct = col_tile(C, 4, M, 4, min_rows);
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, col_grid(ct, C, 4, M), dim3(256), 0, st, dy, bT, sc, wf + pl.u[j], wf + pl.scale[j],
                         wf + pl.shift[j], wf + pl.mean[j], wf + pl.invstd[j], slots, M, C, ct, layer_dropout(s, j));
    } else {
      const ColTile ct = col_tile(C, 1, M, 4, min_rows);
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, col_grid(ct, C, 1, M), dim3(256), 0, st, dy, bT, sc, wf + pl.u[j], wf + pl.scale[j],
                         wf + pl.shift[j], wf + pl.mean[j], wf + pl.invstd[j], slots, M, C, ct, layer_dropout(s, j));
    }
    OCEAN_LAUNCH_CHECK();
    return OCEAN_OK;
  };
  // the column sums of layer j are complete in their slots (from `reduce`, or from the epilogue of the data-gradient
  // GEMM that produced the layer's dy): fold the slots, and take the gamma / beta gradients from THIS rank's sums
  // (before any cross-rank all-reduce of the block), so that averaging parameter gradients over ranks yields the
  // global-batch gradient like every other parameter
  auto reduce_finish = [&](int j) -> int {
    const int C = pl.width[j];
    hipLaunchKernelGGL(bn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, reinterpret_cast<double*>(wb + pl.bslots[j]),
                       reinterpret_cast<double*>(wb + pl.bstats[j]), grads[pi_bn(pl, j, 1)], grads[pi_bn(pl, j, 2)], C);
    OCEAN_LAUNCH_CHECK();
    return OCEAN_OK;
  };
  // BatchNorm-backward sums of layer j-1 ride in the epilogue of layer j's data-gradient GEMM, whose output is that
  // layer's dy (it is in registers there; a separate pass would read it and u again from memory)
  static const bool fuse_reduce = getenv("OCEAN_DCNN_SEPARATE_BN_REDUCE") == nullptr;     // measurement switch
  auto ride_along = [&](GemmP& g, int jprev) -> bool {
    if (!fuse_reduce || !ocean_gemm_bn_rides(g)) return false;
    g.bn_u = wf + pl.u[jprev]; g.ld_bn_u = pl.width[jprev];
    g.bn_scale = wf + pl.scale[jprev]; g.bn_shift = wf + pl.shift[jprev];
    g.bn_mean = wf + pl.mean[jprev]; g.bn_invstd = wf + pl.invstd[jprev];
    g.bn_slots = reinterpret_cast<double*>(wb + pl.bslots[jprev]); g.bn_nslots = kStatSlots;
    g.bn_drop = layer_dropout(s, jprev);
    return true;
  };

  auto nt_form = [](const GemmP& g) { GemmBatch b; b.n = 1; b.g[0] = g; return ocean_gemm_nt_applies(b); };

  const bool part_a = (parts & 1) != 0, part_b = (parts & 2) != 0;
  for (int ph = phase_lo; ph < phase_hi; ++ph) {
    if (ph == 0) {
      if (!part_a) continue;
      OCEAN_TRY(zero_f32(wb + pl.bstats[0], pl.bwd_total - pl.bstats[0], st));   // bstats + packed weight grads
      // output projection
      // weight, bias (column sums of the same operand) and data gradients of the output projection
      OCEAN_TRY(linear_bwd_pair(wf + pl.pooled, Cl, dout, s->out_dim, params[pi_out(L, 0)], grads[pi_out(L, 0)], grads[pi_out(L, 1)], wb + pl.dpool, Cl,
                                s->B, s->out_dim, Cl, 0, st));
      OCEAN_TRY(reduce(pl.nbn - 1));
      OCEAN_TRY(reduce_finish(pl.nbn - 1));
      continue;
    }
    const int j = pl.nbn - ph;          // BN layer finished in this phase
    const int C = pl.width[j], Cin = pl.in_width[j];
    const bool last = (j == pl.nbn - 1);
    const float* dy = last ? wb + pl.dpool : grad_buf(j);
    const int bT = last ? s->T : 0;
    const float dsc = last ? 1.0f / (float)s->T : 1.0f;
    const double* bs = reinterpret_cast<const double*>(wb + pl.bstats[j]);
    float* dU = wb + pl.dU;
    if (!part_a) {}
    else if (C % 4 == 0) {
      const ColTile ct = col_tile(C, 4, M, 8, 1);
      hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, col_grid(ct, C, 4, M), dim3(256), 0, st, dy, bT, dsc, wf + pl.u[j], wf + pl.scale[j],
                         wf + pl.shift[j], wf + pl.mean[j], wf + pl.invstd[j], bs, n, dU, M, C, s->training, ct, layer_dropout(s, j));
    } else {
      const ColTile ct = col_tile(C, 1, M, 8, 1);
      hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, col_grid(ct, C, 1, M), dim3(256), 0, st, dy, bT, dsc, wf + pl.u[j], wf + pl.scale[j],
                         wf + pl.shift[j], wf + pl.mean[j], wf + pl.invstd[j], bs, n, dU, M, C, s->training, ct, layer_dropout(s, j));
    }
    OCEAN_LAUNCH_CHECK();
    const float* in = layer_input(pl, j, wf);
    bool rode = false;
    if (pl.is_block[j]) {
      const int i = pl.li[j], k = s->kernel_size, d = s->dilations[i];
      // data gradient: taps with the opposite shift, plus the identity-residual gradient.  Preferred form: both operands
      // with the reduction index (the output channels) contiguous, i.e. against the transposed weight copy; the
      // [k][n] form serves the shapes that kernel does not take.
      GemmP h = gemm_defaults();
      h.A = dU; h.lda = C; h.M = M; h.N = Cin; h.K = C;
      if (k > 1) { h.taps = k; h.T = s->T; h.shift0 = (k / 2) * d; h.shift_step = -d; h.tap_stride_b = (long)C * Cin; }
      float* gout = grad_buf(j);          // same buffer: d(input) = conv^T(dU) + d(output)
      h.C = gout; h.ldc = Cin;
      if (last) { h.R = wb + pl.dpool; h.ldr = Cl; h.r_bcast_T = s->T; h.r_scale = 1.0f / (float)s->T; }
      else { h.R = gout; h.ldr = Cin; }
      h.B = wf + pl.wpT[i]; h.ldb = C; h.b_kn = 0;
      bool nt = false;
      if (j > 0) { GemmP t = h; if (ride_along(t, j - 1)) { h = t; rode = nt = true; } }
      if (!nt) nt = nt_form(h);
      if (!nt) {
        h.B = wf + pl.wp[i]; h.ldb = Cin; h.b_kn = 1;
        if (j > 0) rode = ride_along(h, j - 1);
      }
      if (part_a) OCEAN_TRY(ocean_launch_gemm(h, st));
      if (part_b) {
        // weight gradient per tap into the packed buffer (the packed gradients of all blocks are unpacked together at the end)
        GemmP g = gemm_defaults();
        g.A = dU; g.lda = C; g.a_kmajor = 1; g.B = in; g.ldb = Cin; g.b_kn = 1;
        g.C = wb + pl.dwp[i]; g.ldc = Cin; g.M = C; g.N = Cin; g.K = M; g.split_k = 0;
        g.taps = k; g.tap_stride_c = (long)C * Cin;
        if (k > 1) { g.T = s->T; g.shift0 = -(k / 2) * d; g.shift_step = d; }
        OCEAN_TRY(ocean_launch_gemm(g, st));
      }
    } else {
      const int i = pl.li[j];
      if (part_b) OCEAN_TRY(linear_bwd_weight(dU, C, in, Cin, grads[pi_trans(L, i, 0)], M, C, Cin, st));
      GemmP h = gemm_defaults();        // dx = dU . W: against the transposed copy [Cin][C] when that kernel takes the shape, else W [C, Cin] read as [k][n]
      h.A = dU; h.lda = C; h.C = grad_buf(j - 1); h.ldc = Cin;
      h.M = M; h.N = Cin; h.K = C;
      h.B = wf + pl.wtT[i]; h.ldb = C; h.b_kn = 0;
      bool nt = false;
      { GemmP t = h; if (ride_along(t, j - 1)) { h = t; rode = nt = true; } }
      if (!nt) nt = nt_form(h);
      if (!nt) {
        h.B = params[pi_trans(L, i, 0)]; h.ldb = Cin; h.b_kn = 1;
        rode = ride_along(h, j - 1);
      }
      if (part_a) OCEAN_TRY(ocean_launch_gemm(h, st));
    }
    if (j > 0) {
      if (part_a) {
        if (!rode) OCEAN_TRY(reduce(j - 1));
        OCEAN_TRY(reduce_finish(j - 1));
      }
    } else if (part_b) {
      // input projection: grad wrt h0 sits in grad_buf(0)
      const float* dh0 = grad_buf(0);
      const int C0 = s->channels[0];
      OCEAN_TRY(linear_bwd_weight(dh0, C0, x, s->F, grads[0], M, C0, s->F, st, grads[1]));   // + bias gradient (column sums of the same operand)
      if (dx) OCEAN_TRY(linear_bwd_data(dh0, C0, params[0], dx, s->F, M, C0, s->F, 0, st));
      // last phase: every block's per-tap weight gradient is complete
      RepackBatch rb;
      long biggest = 0;
      for (int i = 0; i < L; ++i) {
        const int Cb = s->channels[i];
        rb.s[i] = RepackSet{wb + pl.dwp[i], grads[pi_block(i, 0)], nullptr, Cb, Cb, s->kernel_size};
        if ((long)Cb * Cb * s->kernel_size > biggest) biggest = (long)Cb * Cb * s->kernel_size;
      }
      hipLaunchKernelGGL(unpack_dw_kernel, dim3(egrid(biggest), L), dim3(256), 0, st, rb);
      OCEAN_LAUNCH_CHECK();
    }
  }
  return OCEAN_OK;
}

}  // namespace

extern "C" {

int ocean_dcnn_bwd(const ocean_dcnn_shape* s, const float* x, float* const* params, float* wf, float* wb, size_t wb_floats,
                   const float* dout, float* const* grads, float* dx, int phase_lo, int phase_hi, void* stream) {
  const OceanBackwardScope scope;
  return dcnn_bwd_impl(s, x, params, wf, wb, wb_floats, dout, grads, dx, phase_lo, phase_hi, 3, stream);
}

int ocean_dcnn_bwd_part(const ocean_dcnn_shape* s, const float* x, float* const* params, float* wf, float* wb, size_t wb_floats,
                        const float* dout, float* const* grads, float* dx, int phase, int part, void* stream) {
  if (part != 0 && part != 1) return OCEAN_ERR_BAD_SHAPE;
  const OceanBackwardScope scope;
  return dcnn_bwd_impl(s, x, params, wf, wb, wb_floats, dout, grads, dx, phase, phase + 1, part == 0 ? 1 : 2, stream);
}

}  // extern "C"
