// MultiHeadAttention called on its own (multi_factor_attention.py:68-148, ScaledDotProductAttention :17-65): linear
// projections of query / key / value, per head softmax(Q K^T / sqrt(d_k)) with an optional mask (0 = masked, filled
// with -1e9 as the reference does), dropout on the attention weights, context, output projection, dropout.
//
// Off the model's hot path (inside OCEANModel the sequences have length 1 and MultiFactorAttention's fused driver
// handles them): one wave per (sample, head, query row), lane j takes key j for the scores and lane c takes feature c
// for the context; the projections go through the library's GEMMs.
#include "../../include/ocean_hip.h"
#include "ops.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cmath>

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxL = 1024;          // scores of one query row live in LDS

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Q, K, V: [B*L, D] rows (sample, position), head h = columns h*dk .. (h+1)*dk.  P [B,H,L,L]: softmax before dropout.
// A [B,H,L,L]: attention weights as returned (after dropout).  ctx [B*L, D].
__global__ __launch_bounds__(64 * kWavesPerBlock) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                        const float* __restrict__ V, const float* __restrict__ mask,
                                                                        float* __restrict__ P, float* __restrict__ A,
                                                                        float* __restrict__ ctx, int B, int H, int L, int D,
                                                                        float inv_sqrt_dk, DropoutP dp) {
  __shared__ float sc[kWavesPerBlock][kMaxL];
  dp = dropout_resolve(dp);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * kWavesPerBlock + w;          // (b, h, i)
  if (row >= (long)B * H * L) return;
  const int i = (int)(row % L), h = (int)((row / L) % H), b = (int)(row / ((long)L * H));
  const int dk = D / H;
  const float* q = Q + ((long)b * L + i) * D + h * dk;
  float* s = sc[w];
  float mx = -INFINITY;
  for (int j = lane; j < L; j += 64) {
    const float* k = K + ((long)b * L + j) * D + h * dk;
    float d = 0.f;
    for (int c = 0; c < dk; ++c) d = fmaf(q[c], k[c], d);
    d *= inv_sqrt_dk;
    if (mask != nullptr && mask[((long)b * L + i) * L + j] == 0.f) d = -1e9f;
    s[j] = d;
    mx = fmaxf(mx, d);
  }
  mx = wmax(mx);
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float e = __expf(s[j] - mx);
    s[j] = e;
    sum += e;
  }
  sum = wsum(sum);
  const float inv = 1.0f / sum;
  float* Pr = P + row * L;
  float* Ar = A + row * L;
  for (int j = lane; j < L; j += 64) {
    const float p = s[j] * inv;
    const float a = dp.p > 0.f ? p * dropout_keep(dp, (uint64_t)row * L + j) : p;
    Pr[j] = p;
    Ar[j] = a;
    s[j] = a;
  }
  __builtin_amdgcn_wave_barrier();
  float* o = ctx + ((long)b * L + i) * D + h * dk;
  for (int c = lane; c < dk; c += 64) {
    float acc = 0.f;
    for (int j = 0; j < L; ++j) acc = fmaf(s[j], V[((long)b * L + j) * D + h * dk + c], acc);
    o[c] = acc;
  }
}

// dctx [B*L, D]; dA_in [B,H,L,L] or null (gradient arriving through the returned weights).  dQ is written, dK and dV are
// accumulated atomically (zeroed by the host).
__global__ __launch_bounds__(64 * kWavesPerBlock) void attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                        const float* __restrict__ V, const float* __restrict__ P,
                                                                        const float* __restrict__ A, const float* __restrict__ dctx,
                                                                        const float* __restrict__ dA_in, float* __restrict__ dQ,
                                                                        float* __restrict__ dK, float* __restrict__ dV, int B, int H,
                                                                        int L, int D, float inv_sqrt_dk, float keep_scale) {
  __shared__ float ds[kWavesPerBlock][kMaxL];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * kWavesPerBlock + w;
  if (row >= (long)B * H * L) return;
  const int i = (int)(row % L), h = (int)((row / L) % H), b = (int)(row / ((long)L * H));
  const int dk = D / H;
  const float* go = dctx + ((long)b * L + i) * D + h * dk;
  const float* Pr = P + row * L;
  const float* Ar = A + row * L;
  float* s = ds[w];
  // dA_ij = dctx_i . V_j (+ dA_in); dP_ij = dA_ij * keep_ij / (1 - p) where keep_ij = [A_ij != 0 or P_ij == 0]
  float dot = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float* v = V + ((long)b * L + j) * D + h * dk;
    float d = 0.f;
    for (int c = 0; c < dk; ++c) d = fmaf(go[c], v[c], d);
    if (dA_in != nullptr) d += dA_in[row * L + j];
    const float p = Pr[j], a = Ar[j];
    // accumulate dV_j += A_ij * dctx_i below needs a; the softmax backward needs dP
    const float dP = (a != 0.f || p == 0.f) ? d * keep_scale : 0.f;
    s[j] = dP;
    dot += p * dP;
  }
  dot = wsum(dot);
  for (int j = lane; j < L; j += 64) s[j] = Pr[j] * (s[j] - dot) * inv_sqrt_dk;      // dS_ij / sqrt(dk)
  __builtin_amdgcn_wave_barrier();
  const float* q = Q + ((long)b * L + i) * D + h * dk;
  float* gq = dQ + ((long)b * L + i) * D + h * dk;
  for (int c = lane; c < dk; c += 64) {
    float acc = 0.f;
    const float qc = q[c], gc = go[c];
    for (int j = 0; j < L; ++j) {
      const long kr = ((long)b * L + j) * D + h * dk + c;
      acc = fmaf(s[j], K[kr], acc);
      unsafeAtomicAdd(dK + kr, s[j] * qc);
      unsafeAtomicAdd(dV + kr, Ar[j] * gc);
    }
    gq[c] = acc;
  }
}

struct MhaPlan { size_t Q, K, V, P, ctx, fwd_total, dctx, dQ, dK, dV, dout, bwd_total; };
inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

int mha_plan(const ocean_mha_shape* s, MhaPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->L <= 0 || s->D <= 0 || s->heads <= 0 || s->D % s->heads != 0) return OCEAN_ERR_BAD_SHAPE;
  if (s->dropout < 0.f || s->dropout >= 1.f) return OCEAN_ERR_BAD_SHAPE;
  if (s->L > kMaxL) return OCEAN_ERR_UNSUPPORTED;
  const size_t R = (size_t)s->B * s->L, D = s->D, LL = (size_t)s->B * s->heads * s->L * s->L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->Q = take(R * D); p->K = take(R * D); p->V = take(R * D); p->P = take(LL); p->ctx = take(R * D);
  p->fwd_total = o;
  o = 0;
  p->dctx = take(R * D); p->dQ = take(R * D); p->dK = take(R * D); p->dV = take(R * D); p->dout = take(R * D);
  p->bwd_total = o;
  return OCEAN_OK;
}

DropoutP mha_site(const ocean_mha_shape* s, int id) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0xD6E8FEB86659FD93ull * (uint64_t)(id + 1);
  d.epoch = s->seed_epoch;
  return d;
}

}  // namespace

extern "C" {

size_t ocean_mha_fwd_workspace(const ocean_mha_shape* s) { MhaPlan p; return mha_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_mha_bwd_workspace(const ocean_mha_shape* s) { MhaPlan p; return mha_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_mha_fwd(const ocean_mha_shape* s, const float* query, const float* key, const float* value, const float* mask,
                  float* const* P_, float* ws, size_t ws_floats, float* out, float* attn, void* stream) {
  MhaPlan pl;
  OCEAN_TRY(mha_plan(s, &pl));
  if (!query || !key || !value || !ws || !out || !attn || (s->project && !P_)) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = s->B * s->L, D = s->D;
  const float scale = s->scale > 0.f ? s->scale : 1.0f / sqrtf((float)(D / s->heads));
  const long rows = (long)s->B * s->heads * s->L;
  if (!s->project) {
    // the attention alone (ScaledDotProductAttention.forward): the inputs are the heads' queries / keys / values
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(64 * kWavesPerBlock), 0, st, query, key,
                       value, mask, ws + pl.P, attn, out, s->B, s->heads, s->L, D, scale, mha_site(s, 0));
    OCEAN_LAUNCH_CHECK();
    return OCEAN_OK;
  }
  OCEAN_TRY(linear_fwd(query, D, P_[0], P_[1], ws + pl.Q, D, R, D, D, 0, st));
  OCEAN_TRY(linear_fwd(key, D, P_[2], P_[3], ws + pl.K, D, R, D, D, 0, st));
  OCEAN_TRY(linear_fwd(value, D, P_[4], P_[5], ws + pl.V, D, R, D, D, 0, st));
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(64 * kWavesPerBlock), 0, st, ws + pl.Q, ws + pl.K,
                     ws + pl.V, mask, ws + pl.P, attn, ws + pl.ctx, s->B, s->heads, s->L, D, scale, mha_site(s, 0));
  OCEAN_LAUNCH_CHECK();
  const DropoutP d1 = mha_site(s, 1);
  Epi e; e.drop = &d1;
  return linear_fwd(ws + pl.ctx, D, P_[6], P_[7], out, D, R, D, D, 0, st, d1.p > 0.f ? &e : nullptr);
}

int ocean_mha_bwd(const ocean_mha_shape* s, const float* query, const float* key, const float* value, float* const* P_,
                  const float* wf, const float* attn, const float* d_out, const float* d_attn, float* wb, size_t wb_floats,
                  float* const* G_, float* dq, float* dk, float* dv, void* stream) {
  MhaPlan pl;
  OCEAN_TRY(mha_plan(s, &pl));
  if (!query || !key || !value || !wf || !attn || !wb || (s->project && (!P_ || !G_))) return OCEAN_ERR_NULL;
  if (!d_out && !d_attn) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = s->B * s->L, D = s->D;
  const size_t RD = (size_t)R * D;
  const float scale = s->scale > 0.f ? s->scale : 1.0f / sqrtf((float)(D / s->heads));
  const float keep = (s->training && s->dropout > 0.f) ? 1.0f / (1.0f - s->dropout) : 1.0f;
  const long nrows = (long)s->B * s->heads * s->L;
  if (!s->project) {
    if (!dq || !dk || !dv) return OCEAN_ERR_NULL;
    const float* g = d_out;
    if (!g) { OCEAN_TRY(zero_f32(wb + pl.dctx, RD, st)); g = wb + pl.dctx; }
    OCEAN_TRY(zero_f32(dk, RD, st));
    OCEAN_TRY(zero_f32(dv, RD, st));
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((nrows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(64 * kWavesPerBlock), 0, st, query, key,
                       value, wf + pl.P, attn, g, d_attn, dq, dk, dv, s->B, s->heads, s->L, D, scale, keep);
    OCEAN_LAUNCH_CHECK();
    return OCEAN_OK;
  }
  if (d_out) {
    // out = dropout(ctx Wo^T + bo): the mask goes onto a copy of the incoming gradient
    const float* g = d_out;
    const DropoutP d1 = mha_site(s, 1);
    if (d1.p > 0.f) {
      OCEAN_TRY(copy2d(wb + pl.dout, D, d_out, D, R, D, 0, st));
      OCEAN_TRY(dropout_inplace(wb + pl.dout, D, R, D, 1, d1, st));
      g = wb + pl.dout;
    }
    if (G_[6]) OCEAN_TRY(linear_bwd_weight(g, D, wf + pl.ctx, D, G_[6], R, D, D, st, G_[7]));
    else if (G_[7]) OCEAN_TRY(colsum_add(g, D, R, D, G_[7], st));
    OCEAN_TRY(linear_bwd_data(g, D, P_[6], wb + pl.dctx, D, R, D, D, 0, st));
  } else {
    OCEAN_TRY(zero_f32(wb + pl.dctx, RD, st));
  }
  OCEAN_TRY(zero_f32(wb + pl.dK, 2 * al4(RD), st));        // dK and dV are neighbours in the plan
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((nrows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(64 * kWavesPerBlock), 0, st, wf + pl.Q, wf + pl.K,
                     wf + pl.V, wf + pl.P, attn, wb + pl.dctx, d_attn, wb + pl.dQ, wb + pl.dK, wb + pl.dV, s->B, s->heads, s->L, D,
                     scale, keep);
  OCEAN_LAUNCH_CHECK();
  const float* x[3] = {query, key, value};
  const float* g3[3] = {wb + pl.dQ, wb + pl.dK, wb + pl.dV};
  float* dx[3] = {dq, dk, dv};
  for (int m = 0; m < 3; ++m) {
    if (G_[2 * m]) OCEAN_TRY(linear_bwd_weight(g3[m], D, x[m], D, G_[2 * m], R, D, D, st, G_[2 * m + 1]));
    else if (G_[2 * m + 1]) OCEAN_TRY(colsum_add(g3[m], D, R, D, G_[2 * m + 1], st));
    if (dx[m]) OCEAN_TRY(linear_bwd_data(g3[m], D, P_[2 * m], dx[m], D, R, D, D, 0, st));
  }
  return OCEAN_OK;
}

// ---- the rest of CrossModalAttention.forward (multi_factor_attention.py:205-215) after its MultiHeadAttention:
//   y1 = LayerNorm(query + attended);  out = LayerNorm(y1 + Dropout(Linear(Dropout(ReLU(Linear(y1))))))
// params / grads: {layer_norm.weight, .bias, feed_forward.0.weight [F,D], .0.bias, feed_forward.3.weight [D,F], .3.bias,
// ff_layer_norm.weight, .bias}.
namespace {
struct TailPlan { size_t S1, st1, Y1, H, S2, st2, fwd_total, dS2, dF, dH, dY1, bwd_total; };
int tail_plan(const ocean_xattn_tail_shape* s, TailPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->rows <= 0 || s->D <= 0 || s->F <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (s->dropout < 0.f || s->dropout >= 1.f) return OCEAN_ERR_BAD_SHAPE;
  if (s->D > 1024) return OCEAN_ERR_UNSUPPORTED;                 // row-wise LayerNorm kernels: at most 1024 columns
  const size_t R = s->rows, D = s->D, Fd = s->F;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->S1 = take(R * D); p->st1 = take(2 * R); p->Y1 = take(R * D); p->H = take(R * Fd); p->S2 = take(R * D); p->st2 = take(2 * R);
  p->fwd_total = o;
  o = 0;
  p->dS2 = take(R * D); p->dF = take(R * D); p->dH = take(R * Fd); p->dY1 = take(R * D);
  p->bwd_total = o;
  return OCEAN_OK;
}
DropoutP tail_site(const ocean_xattn_tail_shape* s, int id) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0xD6E8FEB86659FD93ull * (uint64_t)(id + 1);
  d.epoch = s->seed_epoch;
  return d;
}
}  // namespace

size_t ocean_xattn_tail_fwd_workspace(const ocean_xattn_tail_shape* s) { TailPlan p; return tail_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_xattn_tail_bwd_workspace(const ocean_xattn_tail_shape* s) { TailPlan p; return tail_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_xattn_tail_fwd(const ocean_xattn_tail_shape* s, const float* query, const float* attended, float* const* P_, float* ws,
                         size_t ws_floats, float* out, void* stream) {
  TailPlan pl;
  OCEAN_TRY(tail_plan(s, &pl));
  if (!query || !attended || !P_ || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = s->rows, D = s->D, Fd = s->F;
  const DropoutP none = {0.f, 0, nullptr};
  OCEAN_TRY(copy2d(ws + pl.S1, D, query, D, R, D, 0, st));
  OCEAN_TRY(copy2d(ws + pl.S1, D, attended, D, R, D, 1, st));
  OCEAN_TRY(layernorm_fwd(ws + pl.S1, D, P_[0], P_[1], ws + pl.Y1, D, ws + pl.st1, R, D, s->eps, none, st));
  const DropoutP d2 = tail_site(s, 2), d3 = tail_site(s, 3);
  Epi e2; e2.drop = &d2;
  OCEAN_TRY(linear_fwd(ws + pl.Y1, D, P_[2], P_[3], ws + pl.H, Fd, R, Fd, D, 1, st, d2.p > 0.f ? &e2 : nullptr));
  Epi e3; e3.drop = &d3;
  OCEAN_TRY(linear_fwd(ws + pl.H, Fd, P_[4], P_[5], ws + pl.S2, D, R, D, Fd, 0, st, d3.p > 0.f ? &e3 : nullptr));
  OCEAN_TRY(copy2d(ws + pl.S2, D, ws + pl.Y1, D, R, D, 1, st));
  return layernorm_fwd(ws + pl.S2, D, P_[6], P_[7], out, D, ws + pl.st2, R, D, s->eps, none, st);
}

int ocean_xattn_tail_bwd(const ocean_xattn_tail_shape* s, float* const* P_, const float* wf, const float* d_out, float* wb,
                         size_t wb_floats, float* const* G_, float* d_sum /* [rows, D]: gradient of query + attended */, void* stream) {
  TailPlan pl;
  OCEAN_TRY(tail_plan(s, &pl));
  if (!P_ || !wf || !d_out || !wb || !G_ || !d_sum) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int R = s->rows, D = s->D, Fd = s->F;
  const DropoutP none = {0.f, 0, nullptr};
  OCEAN_TRY(layernorm_bwd(d_out, D, wf + pl.S2, D, wf + pl.st2, P_[6], wb + pl.dS2, D, G_[6], G_[7], R, D, none, st));
  const DropoutP d2 = tail_site(s, 2), d3 = tail_site(s, 3);
  const float* dF = wb + pl.dS2;
  if (d3.p > 0.f) {
    OCEAN_TRY(copy2d(wb + pl.dF, D, wb + pl.dS2, D, R, D, 0, st));
    OCEAN_TRY(dropout_inplace(wb + pl.dF, D, R, D, 1, d3, st));
    dF = wb + pl.dF;
  }
  Epi e2; e2.drop = d2.p > 0.f ? &d2 : nullptr; e2.relu_mask = wf + pl.H; e2.ld_mask = Fd;
  if (G_[4]) OCEAN_TRY(linear_bwd_weight(dF, D, wf + pl.H, Fd, G_[4], R, D, Fd, st, G_[5]));
  else if (G_[5]) OCEAN_TRY(colsum_add(dF, D, R, D, G_[5], st));
  OCEAN_TRY(linear_bwd_data(dF, D, P_[4], wb + pl.dH, Fd, R, D, Fd, 0, st, &e2));
  if (G_[2]) OCEAN_TRY(linear_bwd_weight(wb + pl.dH, Fd, wf + pl.Y1, D, G_[2], R, Fd, D, st, G_[3]));
  else if (G_[3]) OCEAN_TRY(colsum_add(wb + pl.dH, Fd, R, Fd, G_[3], st));
  OCEAN_TRY(copy2d(wb + pl.dY1, D, wb + pl.dS2, D, R, D, 0, st));                        // the residual branch
  OCEAN_TRY(linear_bwd_data(wb + pl.dH, Fd, P_[2], wb + pl.dY1, D, R, Fd, D, 1, st));
  return layernorm_bwd(wb + pl.dY1, D, wf + pl.S1, D, wf + pl.st1, P_[0], d_sum, D, G_[0], G_[1], R, D, none, st);
}

}  // extern "C"
