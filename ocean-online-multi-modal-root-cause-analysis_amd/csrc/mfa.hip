// MultiFactorAttention forward / backward driver (multi_factor_attention.py:312-440).
//
// All projections are fp32-MFMA GEMMs on [B, .] (or stacked [3B, .]) matrices;
// the glue between them is three small row kernels: the 2-key cross-modal
// softmax over the three LayerNormed modality vectors, the sigmoid feature
// gate, and LayerNorm (rowops).  The per-modality self-attention runs on a
// length-1 sequence, so its softmax is identically 1 and the block reduces to
// W_o(W_v x + b_v) + b_o with the reference's two dropouts (per-head on the
// attention weight, then on the output); W_q / W_k never influence the result
// and receive no gradient here (the reference gives them exact zeros).
#include "../../include/ocean_hip.h"
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

// Nn: [3][B][W] LayerNormed modality vectors (t, s, l) -> G, cw (rowbody::cross_fwd_row), one wave per sample
__global__ __launch_bounds__(256) void cross_fwd_kernel(const float* __restrict__ Nn, const float* __restrict__ mw,
                                                        float* __restrict__ G, float* __restrict__ cw, int B, int W) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (b >= B) return;
  rowbody::cross_fwd_row(Nn, W, (long)B * W, mw, G, W, (long)B * W, cw, W, b, lane);
}

__global__ __launch_bounds__(256) void cross_bwd_kernel(const float* __restrict__ dG, const float* __restrict__ Nn,
                                                        const float* __restrict__ cw, const float* __restrict__ mw,
                                                        float* __restrict__ dN, float* __restrict__ dpi, int B, int W) {
  __shared__ float red[kWaves][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x * kWaves + wave;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
  if (b < B) rowbody::cross_bwd_row(dG, W, (long)B * W, Nn, W, (long)B * W, cw, mw, dN, W, (long)B * W, W, b, lane, dp0, dp1, dp2);
  if (lane == 0) { red[wave][0] = dp0; red[wave][1] = dp1; red[wave][2] = dp2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w2 = 0; w2 < kWaves; ++w2) s += red[w2][threadIdx.x];
    unsafeAtomicAdd(dpi + threadIdx.x, s);
  }
}

// d(modality_weights) through pi = softmax(mw)
__global__ void modality_bwd_kernel(const float* __restrict__ dpi, const float* __restrict__ mw, float* __restrict__ dmw) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float m0 = mw[0], m1 = mw[1], m2 = mw[2], mm = fmaxf(m0, fmaxf(m1, m2));
  const float e0 = __expf(m0 - mm), e1 = __expf(m1 - mm), e2 = __expf(m2 - mm), es = e0 + e1 + e2;
  const float p[3] = {e0 / es, e1 / es, e2 / es};
  const float dot = p[0] * dpi[0] + p[1] * dpi[1] + p[2] * dpi[2];
  for (int i = 0; i < 3; ++i) dmw[i] += p[i] * (dpi[i] - dot);
}

// rows r in [0,3B): the sigmoid feature gate (rowbody::gate_fwd_row / gate_bwd_row), one wave per row
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ hid, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, const float* __restrict__ G,
                                                       float* __restrict__ ig, float* __restrict__ cat, int B, int W, int Wh) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (r >= 3 * B) return;
  rowbody::gate_fwd_row(hid, Wh, (long)B * Wh, w2, b2, G, W, (long)B * W, ig, cat, 3 * W, B, W, Wh, r / B, r % B, lane);
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dcat, const float* __restrict__ G,
                                                       const float* __restrict__ ig, const float* __restrict__ hid,
                                                       const float* __restrict__ w2, float* __restrict__ dG,
                                                       float* __restrict__ dz, float* __restrict__ dhid, int B, int W, int Wh) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (r >= 3 * B) return;
  rowbody::gate_bwd_row(dcat, 3 * W, G, W, (long)B * W, ig, hid, Wh, (long)B * Wh, w2, dG, W, (long)B * W, dz, dhid, Wh, (long)B * Wh, B, W, Wh, r / B, r % B, lane);
}

struct MfaPlan {
  int B, W, Wh;
  size_t P, V, A, lnst, Nn, cw, G, Hd, ig, cat, F, O1, O2, lnso, fwd_total;
  size_t d1, d2, d3, d4, dHd, dz, dcat, e1, e2, e3, dpi, bwd_total;
};
inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

int make_plan(const ocean_mfa_shape* s, MfaPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->td <= 0 || s->sd <= 0 || s->ld <= 0 || s->W <= 1 || s->heads <= 0 || s->W % s->heads != 0)
    return OCEAN_ERR_BAD_SHAPE;
  if (s->W > 1024) return OCEAN_ERR_UNSUPPORTED;
  p->B = s->B; p->W = s->W; p->Wh = s->W / 2;
  const size_t B = s->B, W = s->W, Wh = p->Wh;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->P = take(3 * B * W); p->V = take(3 * B * W); p->A = take(3 * B * W); p->lnst = take(6 * B); p->Nn = take(3 * B * W);
  p->cw = take(6 * B); p->G = take(3 * B * W); p->Hd = take(3 * B * Wh); p->ig = take(3 * B); p->cat = take(3 * B * W);
  p->F = take(B * W); p->O1 = take(B * W); p->O2 = take(B * W); p->lnso = take(2 * B);
  p->fwd_total = o;
  o = 0;
  p->d1 = take(3 * B * W); p->d2 = take(3 * B * W); p->dHd = take(3 * B * Wh); p->dz = take(3 * B); p->dcat = take(3 * B * W);
  p->e1 = take(B * W); p->e2 = take(B * W); p->dpi = take(4);
  // a layer's output gradient stays untouched until the end of the pass (the weight gradients of a small batch are all
  // formed there, in one launch: rowchain.h), so no two layers share a gradient buffer
  p->e3 = take(B * W); p->d3 = take(3 * B * W); p->d4 = take(3 * B * W);
  p->bwd_total = o;
  return OCEAN_OK;
}

DropoutP site(const ocean_mfa_shape* s, int id) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0xA24BAED4963EE407ull * (uint64_t)(id + 1);
  d.epoch = s->seed_epoch;
  return d;
}

// weight + bias gradients and (optionally) the data gradient of y = x W^T + b
int lin_bwd(const float* x, int ldx, const float* dy, int lddy, int M, int N, int K, const float* W, float* dW, float* db,
            float* dx, int lddx, int accumulate, hipStream_t st, const Epi* dx_epi = nullptr) {
  if (dW && dx) return linear_bwd_pair(x, ldx, dy, lddy, W, dW, db, dx, lddx, M, N, K, accumulate, st, dx_epi);   // db rides along as the column sums of dy
  if (dW) OCEAN_TRY(linear_bwd_weight(dy, lddy, x, ldx, dW, M, N, K, st, db));
  else if (db) OCEAN_TRY(colsum_add(dy, lddy, M, N, db, st));
  if (dx) OCEAN_TRY(linear_bwd_data(dy, lddy, W, dx, lddx, M, N, K, accumulate, st, dx_epi));
  return OCEAN_OK;
}

}  // namespace

extern "C" {

size_t ocean_mfa_fwd_workspace(const ocean_mfa_shape* s) { MfaPlan p; return make_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_mfa_bwd_workspace(const ocean_mfa_shape* s) { MfaPlan p; return make_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_mfa_fwd(const ocean_mfa_shape* s, const float* t, const float* sp, const float* lg, float* const* P_, float* ws,
                  size_t ws_floats, float* out, void* stream) {
  MfaPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!t || !sp || !lg || !P_ || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, W = s->W, Wh = pl.Wh;
  const size_t BW = (size_t)B * W;
  const float* xin[3] = {t, sp, lg};
  const int xd[3] = {s->td, s->sd, s->ld};
  const DropoutP none = {0.f, 0, nullptr};
  RcScope rc(B, st);       // small batches: the whole chain below becomes one launch (rowchain.h)
  // the three modality branches run as one launch per layer; dropout sites cover all three (contiguous [3, B, W])
  LinFwd l[3];
  for (int m = 0; m < 3; ++m) l[m] = LinFwd{xin[m], xd[m], P_[2 * m], P_[2 * m + 1], ws + pl.P + m * BW, W, B, W, xd[m]};
  OCEAN_TRY(linear_fwd_batch(l, 3, 0, st));
  for (int m = 0; m < 3; ++m) l[m] = LinFwd{ws + pl.P + m * BW, W, P_[6 + 4 * m], P_[7 + 4 * m], ws + pl.V + m * BW, W, B, W, W};
  const DropoutP d0 = site(s, 0), d6 = site(s, 6);   // dropout rides in the epilogue of the GEMM that produces the tensor
  OCEAN_TRY(linear_fwd_batch(l, 3, 0, st, &d0, W / s->heads));
  for (int m = 0; m < 3; ++m) l[m] = LinFwd{ws + pl.V + m * BW, W, P_[8 + 4 * m], P_[9 + 4 * m], ws + pl.A + m * BW, W, B, W, W};
  OCEAN_TRY(linear_fwd_batch(l, 3, 0, st));
  LnFwdSet ln[3];
  for (int m = 0; m < 3; ++m)
    ln[m] = LnFwdSet{ws + pl.A + m * BW, P_[18 + 2 * m], P_[19 + 2 * m], ws + pl.Nn + m * BW, ws + pl.lnst + (size_t)m * 2 * B};
  OCEAN_TRY(layernorm_fwd_batch(ln, 3, W, W, B, W, s->eps, site(s, 3), st));   // A stays as computed; the LayerNorm drops its input
  if (!rc_cross_fwd(ws + pl.Nn, P_[26], ws + pl.G, ws + pl.cw, B, W)) {
    hipLaunchKernelGGL(cross_fwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, ws + pl.Nn, P_[26], ws + pl.G, ws + pl.cw, B, W);
    OCEAN_LAUNCH_CHECK();
  }
  if (rc_active()) {
    // the stage kernel walks samples: the [3][B][.] stack is three matrices of one row per sample
    for (int m = 0; m < 3; ++m) l[m] = LinFwd{ws + pl.G + m * BW, W, P_[27], P_[28], ws + pl.Hd + (size_t)m * B * Wh, Wh, B, Wh, W};
    OCEAN_TRY(linear_fwd_batch(l, 3, 1, st));
  } else {
    OCEAN_TRY(linear_fwd(ws + pl.G, W, P_[27], P_[28], ws + pl.Hd, Wh, 3 * B, Wh, W, 1, st));
  }
  if (!rc_gate_fwd(ws + pl.Hd, P_[29], P_[30], ws + pl.G, ws + pl.ig, ws + pl.cat, B, W, Wh)) {
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(ceil_div(3 * B, kWaves)), dim3(256), 0, st, ws + pl.Hd, P_[29], P_[30], ws + pl.G, ws + pl.ig,
                       ws + pl.cat, B, W, Wh);
    OCEAN_LAUNCH_CHECK();
  }
  OCEAN_TRY(linear_fwd(ws + pl.cat, 3 * W, P_[31], P_[32], ws + pl.F, W, B, W, 3 * W, 0, st));
  Epi e6; e6.drop = &d6;
  OCEAN_TRY(linear_fwd(ws + pl.F, W, P_[33], P_[34], ws + pl.O1, W, B, W, W, 1, st, &e6));
  OCEAN_TRY(linear_fwd(ws + pl.O1, W, P_[35], P_[36], ws + pl.O2, W, B, W, W, 0, st));
  OCEAN_TRY(layernorm_fwd(ws + pl.O2, W, P_[24], P_[25], out, W, ws + pl.lnso, B, W, s->eps, none, st));
  return rc.finish();
}

int ocean_mfa_bwd(const ocean_mfa_shape* s, const float* t, const float* sp, const float* lg, float* const* P_, float* wf,
                  float* wb, size_t wb_floats, const float* dout, float* const* G_, float* dt, float* dsp, float* dlg, void* stream) {
  MfaPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!t || !sp || !lg || !P_ || !wf || !wb || !dout || !G_) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, W = s->W, Wh = pl.Wh;
  const size_t BW = (size_t)B * W;
  const DropoutP none = {0.f, 0, nullptr};
  float* e1 = wb + pl.e1; float* e2 = wb + pl.e2; float* e3 = wb + pl.e3;
  float* d1 = wb + pl.d1; float* d2 = wb + pl.d2; float* d3 = wb + pl.d3; float* d4 = wb + pl.d4;
  // the accumulator of d(pi) is cleared first: with a small batch everything below up to the weight gradients is ONE
  // launch (rowchain.h), and nothing of it may sit in front of this fill
  OCEAN_TRY(zero_f32(wb + pl.dpi, 4, st));
  RcScope rc(B, st);
  // output LayerNorm and MLP
  OCEAN_TRY(layernorm_bwd(dout, W, wf + pl.O2, W, wf + pl.lnso, P_[24], e1, W, G_[24], G_[25], B, W, none, st));
  const DropoutP d0 = site(s, 0), d6 = site(s, 6);
  Epi e6; e6.drop = &d6; e6.relu_mask = wf + pl.O1; e6.ld_mask = W;     // dropout and ReLU backward on the way out of the dgrad GEMM
  OCEAN_TRY(lin_bwd(wf + pl.O1, W, e1, W, B, W, W, P_[35], G_[35], G_[36], e2, W, 0, st, &e6));
  OCEAN_TRY(lin_bwd(wf + pl.F, W, e2, W, B, W, W, P_[33], G_[33], G_[34], e3, W, 0, st));
  OCEAN_TRY(lin_bwd(wf + pl.cat, 3 * W, e3, W, B, W, 3 * W, P_[31], G_[31], G_[32], wb + pl.dcat, 3 * W, 0, st));
  // feature gate
  if (!rc_gate_bwd(wb + pl.dcat, wf + pl.G, wf + pl.ig, wf + pl.Hd, P_[29], d1, wb + pl.dz, wb + pl.dHd, B, W, Wh)) {
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(3 * B, kWaves)), dim3(256), 0, st, wb + pl.dcat, wf + pl.G, wf + pl.ig, wf + pl.Hd,
                       P_[29], d1, wb + pl.dz, wb + pl.dHd, B, W, Wh);
    OCEAN_LAUNCH_CHECK();
  }
  OCEAN_TRY(lin_bwd(wf + pl.Hd, Wh, wb + pl.dz, 1, 3 * B, 1, Wh, P_[29], G_[29], G_[30], nullptr, 0, 0, st));
  if (rc_active()) {
    // weight gradient over all 3 B rows; the data gradient per modality (one row per sample each)
    OCEAN_TRY(linear_bwd_weight(wb + pl.dHd, Wh, wf + pl.G, W, G_[27], 3 * B, Wh, W, st, G_[28]));
    LinBwd g3[3];
    for (int m = 0; m < 3; ++m)
      g3[m] = LinBwd{wf + pl.G + m * BW, W, wb + pl.dHd + (size_t)m * B * Wh, Wh, B, Wh, W, P_[27], nullptr, nullptr, d1 + m * BW, W};
    OCEAN_TRY(linear_bwd_batch(g3, 3, 1, st));
  } else {
    OCEAN_TRY(lin_bwd(wf + pl.G, W, wb + pl.dHd, Wh, 3 * B, Wh, W, P_[27], G_[27], G_[28], d1, W, 1, st));
  }
  // cross-modal softmax and modality weights
  if (!rc_cross_bwd(d1, wf + pl.Nn, wf + pl.cw, P_[26], d2, wb + pl.dpi, B, W)) {
    hipLaunchKernelGGL(cross_bwd_kernel, dim3(ceil_div(B, kWaves)), dim3(256), 0, st, d1, wf + pl.Nn, wf + pl.cw, P_[26], d2, wb + pl.dpi, B, W);
    OCEAN_LAUNCH_CHECK();
  }
  const float* xin[3] = {t, sp, lg};
  const int xd[3] = {s->td, s->sd, s->ld};
  float* dxin[3] = {dt, dsp, dlg};
  LnBwdSet ln[3];
  for (int m = 0; m < 3; ++m)
    ln[m] = LnBwdSet{d2 + m * BW, wf + pl.A + m * BW, wf + pl.lnst + (size_t)m * 2 * B, P_[18 + 2 * m], d2 + m * BW, G_[18 + 2 * m],
                     G_[19 + 2 * m]};
  OCEAN_TRY(layernorm_bwd_batch(ln, 3, W, W, W, B, W, site(s, 3), st));
  LinBwd l[3];
  for (int m = 0; m < 3; ++m)
    l[m] = LinBwd{wf + pl.V + m * BW, W, d2 + m * BW, W, B, W, W, P_[8 + 4 * m], G_[8 + 4 * m], G_[9 + 4 * m], d3 + m * BW, W};
  OCEAN_TRY(linear_bwd_batch(l, 3, 0, st, &d0, W / s->heads));
  for (int m = 0; m < 3; ++m)
    l[m] = LinBwd{wf + pl.P + m * BW, W, d3 + m * BW, W, B, W, W, P_[6 + 4 * m], G_[6 + 4 * m], G_[7 + 4 * m], d4 + m * BW, W};
  OCEAN_TRY(linear_bwd_batch(l, 3, 0, st));
  for (int m = 0; m < 3; ++m)
    l[m] = LinBwd{xin[m], xd[m], d4 + m * BW, W, B, W, xd[m], P_[2 * m], G_[2 * m], G_[2 * m + 1], dxin[m], xd[m]};
  OCEAN_TRY(linear_bwd_batch(l, 3, 0, st));
  OCEAN_TRY(rc.finish());
  // d(modality_weights) needs every sample's term of d(pi): after the chain
  hipLaunchKernelGGL(modality_bwd_kernel, dim3(1), dim3(64), 0, st, wb + pl.dpi, P_[26], G_[26]);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

}  // extern "C"
