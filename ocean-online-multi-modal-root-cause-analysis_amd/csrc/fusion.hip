// GraphFusionModule forward / backward driver (graph_fusion.py:388-473,
// fusion_strategy='attention') with the contrastive projection heads
// (graph_fusion.py:77-92, 217-221).
//
// The three processed modality vectors are written side by side as one
// [B, 3, E] matrix so nn.MultiheadAttention's in_proj is a single [3B,E] GEMM.
// The 3-token, H-head attention itself is one thread per (sample, head): nine
// dot products of width E/H.  mean over tokens commutes with out_proj
// (mean_l(W o_l + b) = W mean_l(o_l) + b), so out_proj runs on [B,E] rows.
#include "../../include/ocean_hip.h"
#include "ops.h"
#include "rowbodies.h"
#include "rowchain.h"

namespace {

constexpr int kTok = 3;

// qkv [B*3, 3E] rows (b,l): [q | k | v].  probs [B,H,3,3] post-softmax (pre-dropout).  o [B*3, E].
__global__ void mha3_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ probs, float* __restrict__ o,
                                int B, int E, int H, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, h = idx % H, dh = E / H;
  const float sc = 1.0f / sqrtf((float)dh);
  const float* base = qkv + (long)b * kTok * 3 * E + h * dh;
  float S[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i)
#pragma unroll
    for (int j = 0; j < kTok; ++j) S[i][j] = 0.f;
  for (int d = 0; d < dh; ++d) {
    float q[kTok], k[kTok];
#pragma unroll
    for (int i = 0; i < kTok; ++i) { q[i] = base[(long)i * 3 * E + d] * sc; k[i] = base[(long)i * 3 * E + E + d]; }
#pragma unroll
    for (int i = 0; i < kTok; ++i)
#pragma unroll
      for (int j = 0; j < kTok; ++j) S[i][j] = fmaf(q[i], k[j], S[i][j]);
  }
  float A[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i) {
    const float m = fmaxf(S[i][0], fmaxf(S[i][1], S[i][2]));
    const float e0 = __expf(S[i][0] - m), e1 = __expf(S[i][1] - m), e2 = __expf(S[i][2] - m), es = e0 + e1 + e2;
    A[i][0] = e0 / es; A[i][1] = e1 / es; A[i][2] = e2 / es;
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      probs[((long)idx * kTok + i) * kTok + j] = A[i][j];
      if (dp.p > 0.f) A[i][j] *= dropout_keep(dp, ((uint64_t)idx * kTok + i) * kTok + j);
    }
  }
  for (int d = 0; d < dh; ++d) {
    float v[kTok];
#pragma unroll
    for (int j = 0; j < kTok; ++j) v[j] = base[(long)j * 3 * E + 2 * E + d];
#pragma unroll
    for (int i = 0; i < kTok; ++i)
      o[((long)b * kTok + i) * E + h * dh + d] = A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2];
  }
}

// dOm [B,E]: gradient of the token mean (each token receives dOm/3).  dqkv [B*3, 3E].
__global__ void mha3_bwd_kernel(const float* __restrict__ dOm, const float* __restrict__ qkv, const float* __restrict__ probs,
                                float* __restrict__ dqkv, int B, int E, int H, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, h = idx % H, dh = E / H;
  const float sc = 1.0f / sqrtf((float)dh);
  const float* base = qkv + (long)b * kTok * 3 * E + h * dh;
  float* dbase = dqkv + (long)b * kTok * 3 * E + h * dh;
  const float* go = dOm + (long)b * E + h * dh;
  float A[kTok][kTok], Ad[kTok][kTok], keep[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i)
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      A[i][j] = probs[((long)idx * kTok + i) * kTok + j];
      keep[i][j] = dp.p > 0.f ? dropout_keep(dp, ((uint64_t)idx * kTok + i) * kTok + j) : 1.f;
      Ad[i][j] = A[i][j] * keep[i][j];
    }
  // dA'[i][j] = dO_i . v_j with dO_i = dOm/3 for every i;  dV_j = sum_i A'[i][j] dO_i
  float dv_dot[kTok] = {0.f, 0.f, 0.f};
  for (int d = 0; d < dh; ++d) {
    const float g = go[d] * (1.0f / kTok);
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      dv_dot[j] = fmaf(g, base[(long)j * 3 * E + 2 * E + d], dv_dot[j]);
      dbase[(long)j * 3 * E + 2 * E + d] = g * (Ad[0][j] + Ad[1][j] + Ad[2][j]);
    }
  }
  float dS[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i) {
    float dA[kTok], dot = 0.f;
#pragma unroll
    for (int j = 0; j < kTok; ++j) { dA[j] = dv_dot[j] * keep[i][j]; dot += A[i][j] * dA[j]; }
#pragma unroll
    for (int j = 0; j < kTok; ++j) dS[i][j] = A[i][j] * (dA[j] - dot);
  }
  for (int d = 0; d < dh; ++d) {
    float q[kTok], k[kTok];
#pragma unroll
    for (int i = 0; i < kTok; ++i) { q[i] = base[(long)i * 3 * E + d]; k[i] = base[(long)i * 3 * E + E + d]; }
#pragma unroll
    for (int i = 0; i < kTok; ++i) {
      dbase[(long)i * 3 * E + d] = sc * (dS[i][0] * k[0] + dS[i][1] * k[1] + dS[i][2] * k[2]);
      dbase[(long)i * 3 * E + E + d] = sc * (dS[0][i] * q[0] + dS[1][i] * q[1] + dS[2][i] * q[2]);
    }
  }
}

// One wave per sample for the usual shapes (E = 64 * VEC columns, a power-of-two head count): rowbody::mha3_*_sample.
// (One thread per (sample, head) leaves 8192 threads on the whole chip walking 128-byte strides: 33 + 53 us for a few MB.)
using rowbody::mha3_hlog;
using rowbody::mha3_vec;
template <int VEC>
__global__ __launch_bounds__(256) void mha3_fwd_wave_kernel(const float* __restrict__ qkv, float* __restrict__ probs, float* __restrict__ o,
                                                            int B, int H, int hlog, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;                                   // wave-uniform
  rowbody::mha3_fwd_sample<VEC>(qkv, probs, o, H, hlog, dp, b, lane);
}

template <int VEC>
__global__ __launch_bounds__(256) void mha3_bwd_wave_kernel(const float* __restrict__ dOm, const float* __restrict__ qkv,
                                                            const float* __restrict__ probs, float* __restrict__ dqkv, int B, int H,
                                                            int hlog, DropoutP dp) {
  dp = dropout_resolve(dp);
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;                                   // wave-uniform
  rowbody::mha3_bwd_sample<VEC>(dOm, qkv, probs, dqkv, H, hlog, dp, b, lane);
}

struct FusPlan {
  int B, E, Pd;
  size_t R, lnsp, QKV, probs, O, Om, Ao, lnsa, Fn, H1, Y1, Y2, nrm, fwd_total;
  size_t e1, e2, dOm, dQKV, dX, dY2, dY1, dR, bwd_total;
};
inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

int make_plan(const ocean_fusion_shape* s, FusPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->td <= 0 || s->sd <= 0 || s->ld <= 0 || s->E <= 0 || s->heads <= 0 || s->E % s->heads != 0 || s->Pd <= 0)
    return OCEAN_ERR_BAD_SHAPE;
  if (s->E > 1024) return OCEAN_ERR_UNSUPPORTED;
  p->B = s->B; p->E = s->E; p->Pd = s->Pd;
  const size_t B = s->B, E = s->E, Pd = s->Pd;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->R = take(3 * B * E); p->lnsp = take(6 * B); p->QKV = take(9 * B * E); p->probs = take(B * s->heads * 9);
  p->O = take(3 * B * E); p->Om = take(B * E); p->Ao = take(B * E); p->lnsa = take(2 * B); p->Fn = take(B * E); p->H1 = take(B * E);
  p->Y1 = take(3 * B * E); p->Y2 = take(3 * B * Pd); p->nrm = take(3 * B);
  p->fwd_total = o;
  o = 0;
  p->e1 = take(B * E); p->e2 = take(B * E); p->dOm = take(B * E); p->dQKV = take(9 * B * E); p->dX = take(3 * B * E);
  p->dY2 = take(3 * B * Pd); p->dY1 = take(3 * B * E); p->dR = take(3 * B * E);   // one slice per modality branch
  p->bwd_total = o;
  return OCEAN_OK;
}

DropoutP site(const ocean_fusion_shape* s, int id) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0xD6E8FEB86659FD93ull * (uint64_t)(id + 1);
  d.epoch = s->seed_epoch;
  return d;
}

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

size_t ocean_fusion_fwd_workspace(const ocean_fusion_shape* s) { FusPlan p; return make_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_fusion_bwd_workspace(const ocean_fusion_shape* s) { FusPlan p; return make_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_fusion_fwd(const ocean_fusion_shape* s, const float* t, const float* sp, const float* lg, float* const* P_, float* ws,
                     size_t ws_floats, float* fused, float* X, float* Zp, void* stream) {
  FusPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!t || !sp || !lg || !P_ || !ws || !fused || !X) return OCEAN_ERR_NULL;
  if (s->contrastive && !Zp) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, E = s->E, Pd = s->Pd;
  const size_t BE = (size_t)B * E;
  const float* xin[3] = {t, sp, lg};
  const int xd[3] = {s->td, s->sd, s->ld};
  const DropoutP none = {0.f, 0, nullptr};
  RcScope rc(B, st);       // small batches: the whole chain below becomes one launch (rowchain.h)
  // the three modality branches run as one launch per layer
  LinFwd l[3];
  for (int m = 0; m < 3; ++m) l[m] = LinFwd{xin[m], xd[m], P_[4 * m], P_[4 * m + 1], ws + pl.R + m * BE, E, B, E, xd[m]};
  OCEAN_TRY(linear_fwd_batch(l, 3, 1, st));
  LnFwdSet ln[3];
  for (int m = 0; m < 3; ++m)
    ln[m] = LnFwdSet{ws + pl.R + m * BE, P_[4 * m + 2], P_[4 * m + 3], X + m * E, ws + pl.lnsp + (size_t)m * 2 * B};
  OCEAN_TRY(layernorm_fwd_batch(ln, 3, E, 3 * E, B, E, s->eps, site(s, 0), st));
  if (s->contrastive) {
    for (int m = 0; m < 3; ++m) l[m] = LinFwd{X + m * E, 3 * E, P_[22 + 4 * m], P_[23 + 4 * m], ws + pl.Y1 + m * BE, E, B, E, E};
    const DropoutP d5 = site(s, 5);
    OCEAN_TRY(linear_fwd_batch(l, 3, 1, st, &d5, 1));
    for (int m = 0; m < 3; ++m)
      l[m] = LinFwd{ws + pl.Y1 + m * BE, E, P_[24 + 4 * m], P_[25 + 4 * m], ws + pl.Y2 + (size_t)m * B * Pd, Pd, B, Pd, E};
    OCEAN_TRY(linear_fwd_batch(l, 3, 0, st));
    if (rc_active()) {      // the stage kernel walks samples: the [3][B][.] stack is three matrices of one row per sample
      for (int m = 0; m < 3; ++m)
        OCEAN_TRY(l2norm_fwd(ws + pl.Y2 + (size_t)m * B * Pd, Pd, Zp + (size_t)m * B * Pd, Pd, ws + pl.nrm + (size_t)m * B, B, Pd, 1e-12f, st));
    } else {
      OCEAN_TRY(l2norm_fwd(ws + pl.Y2, Pd, Zp, Pd, ws + pl.nrm, 3 * B, Pd, 1e-12f, st));
    }
  }
  {
    const RcTokenMajor tokens;      // X is [B, 3, E]: three consecutive rows per sample
    OCEAN_TRY(linear_fwd(X, E, P_[12], P_[13], ws + pl.QKV, 3 * E, 3 * B, 3 * E, E, 0, st));
  }
  if (rc_mha3_fwd(ws + pl.QKV, ws + pl.probs, ws + pl.O, B, E, s->heads, site(s, 3))) {}
  else switch (mha3_vec(E, s->heads)) {
    case 4: hipLaunchKernelGGL(mha3_fwd_wave_kernel<4>, dim3(ceil_div(B, 4)), dim3(256), 0, st, ws + pl.QKV, ws + pl.probs, ws + pl.O, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
    case 2: hipLaunchKernelGGL(mha3_fwd_wave_kernel<2>, dim3(ceil_div(B, 4)), dim3(256), 0, st, ws + pl.QKV, ws + pl.probs, ws + pl.O, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
    case 1: hipLaunchKernelGGL(mha3_fwd_wave_kernel<1>, dim3(ceil_div(B, 4)), dim3(256), 0, st, ws + pl.QKV, ws + pl.probs, ws + pl.O, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
    default:
      hipLaunchKernelGGL(mha3_fwd_kernel, dim3(ceil_div((long)B * s->heads, 128)), dim3(128), 0, st, ws + pl.QKV, ws + pl.probs, ws + pl.O, B, E,
                         s->heads, site(s, 3));
  }
  OCEAN_LAUNCH_CHECK();
  OCEAN_TRY(mean_mid(ws + pl.O, ws + pl.Om, B, kTok, E, st));
  OCEAN_TRY(linear_fwd(ws + pl.Om, E, P_[14], P_[15], ws + pl.Ao, E, B, E, E, 0, st));
  OCEAN_TRY(layernorm_fwd(ws + pl.Ao, E, P_[16], P_[17], ws + pl.Fn, E, ws + pl.lnsa, B, E, s->eps, none, st));
  const DropoutP d4 = site(s, 4);
  Epi e4; e4.drop = &d4;
  OCEAN_TRY(linear_fwd(ws + pl.Fn, E, P_[18], P_[19], ws + pl.H1, E, B, E, E, 1, st, &e4));
  OCEAN_TRY(linear_fwd(ws + pl.H1, E, P_[20], P_[21], fused, E, B, E, E, 0, st));
  return rc.finish();
}

int ocean_fusion_bwd(const ocean_fusion_shape* s, const float* t, const float* sp, const float* lg, float* const* P_, float* wf,
                     const float* X, const float* Zp, float* wb, size_t wb_floats, const float* dfused, const float* dX_in,
                     const float* dZp, float* const* G_, float* dt, float* dsp, float* dlg, void* stream) {
  FusPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!t || !sp || !lg || !P_ || !wf || !X || !wb || !G_) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, E = s->E, Pd = s->Pd;
  const size_t BE = (size_t)B * E;
  const DropoutP none = {0.f, 0, nullptr};
  float* e1 = wb + pl.e1; float* e2 = wb + pl.e2; float* dX = wb + pl.dX;
  RcScope rc(B, st);       // small batches: the data-gradient chain is one launch, the weight gradients another (rowchain.h)
  if (dfused) {
    const DropoutP d4 = site(s, 4);
    Epi e4; e4.drop = &d4; e4.relu_mask = wf + pl.H1; e4.ld_mask = E;   // dropout and ReLU backward on the way out of the dgrad GEMM
    OCEAN_TRY(lin_bwd(wf + pl.H1, E, dfused, E, B, E, E, P_[20], G_[20], G_[21], e2, E, 0, st, &e4));
    OCEAN_TRY(lin_bwd(wf + pl.Fn, E, e2, E, B, E, E, P_[18], G_[18], G_[19], e1, E, 0, st));
    OCEAN_TRY(layernorm_bwd(e1, E, wf + pl.Ao, E, wf + pl.lnsa, P_[16], e1, E, G_[16], G_[17], B, E, none, st));
    OCEAN_TRY(lin_bwd(wf + pl.Om, E, e1, E, B, E, E, P_[14], G_[14], G_[15], wb + pl.dOm, E, 0, st));
    if (rc_mha3_bwd(wb + pl.dOm, wf + pl.QKV, wf + pl.probs, wb + pl.dQKV, B, E, s->heads, site(s, 3))) {}
    else switch (mha3_vec(E, s->heads)) {
      case 4: hipLaunchKernelGGL(mha3_bwd_wave_kernel<4>, dim3(ceil_div(B, 4)), dim3(256), 0, st, wb + pl.dOm, wf + pl.QKV, wf + pl.probs, wb + pl.dQKV, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
      case 2: hipLaunchKernelGGL(mha3_bwd_wave_kernel<2>, dim3(ceil_div(B, 4)), dim3(256), 0, st, wb + pl.dOm, wf + pl.QKV, wf + pl.probs, wb + pl.dQKV, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
      case 1: hipLaunchKernelGGL(mha3_bwd_wave_kernel<1>, dim3(ceil_div(B, 4)), dim3(256), 0, st, wb + pl.dOm, wf + pl.QKV, wf + pl.probs, wb + pl.dQKV, B, s->heads, mha3_hlog(s->heads), site(s, 3)); break;
      default:
        hipLaunchKernelGGL(mha3_bwd_kernel, dim3(ceil_div((long)B * s->heads, 128)), dim3(128), 0, st, wb + pl.dOm, wf + pl.QKV, wf + pl.probs,
                           wb + pl.dQKV, B, E, s->heads, site(s, 3));
    }
    OCEAN_LAUNCH_CHECK();
    {
      const RcTokenMajor tokens;      // dQKV and dX are [B, 3, .]: three consecutive rows per sample
      OCEAN_TRY(lin_bwd(X, E, wb + pl.dQKV, 3 * E, 3 * B, 3 * E, E, P_[12], G_[12], G_[13], dX, E, 0, st));
    }
  } else {
    OCEAN_TRY(zero_f32(dX, 3 * BE, st));
  }
  if (dX_in) OCEAN_TRY(copy2d(dX, 3 * E, dX_in, 3 * E, B, 3 * E, 1, st));
  if (s->contrastive && dZp) {
    if (!Zp) return OCEAN_ERR_NULL;
    LinBwd l[3];
    if (rc_active()) {
      for (int m = 0; m < 3; ++m)
        OCEAN_TRY(l2norm_bwd(dZp + (size_t)m * B * Pd, Pd, Zp + (size_t)m * B * Pd, Pd, wf + pl.nrm + (size_t)m * B, wb + pl.dY2 + (size_t)m * B * Pd, Pd,
                             B, Pd, 1e-12f, st));
    } else {
      OCEAN_TRY(l2norm_bwd(dZp, Pd, Zp, Pd, wf + pl.nrm, wb + pl.dY2, Pd, 3 * B, Pd, 1e-12f, st));
    }
    for (int m = 0; m < 3; ++m)
      l[m] = LinBwd{wf + pl.Y1 + m * BE, E, wb + pl.dY2 + (size_t)m * B * Pd, Pd, B, Pd, E, P_[24 + 4 * m], G_[24 + 4 * m], G_[25 + 4 * m],
                    wb + pl.dY1 + m * BE, E};
    const DropoutP d5 = site(s, 5);
    const float* y1[3] = {wf + pl.Y1, wf + pl.Y1 + BE, wf + pl.Y1 + 2 * BE};
    OCEAN_TRY(linear_bwd_batch(l, 3, 0, st, &d5, 1, y1, E));
    for (int m = 0; m < 3; ++m)
      l[m] = LinBwd{X + m * E, 3 * E, wb + pl.dY1 + m * BE, E, B, E, E, P_[22 + 4 * m], G_[22 + 4 * m], G_[23 + 4 * m], dX + m * E, 3 * E};
    OCEAN_TRY(linear_bwd_batch(l, 3, 1, st));
  }
  const float* xin[3] = {t, sp, lg};
  const int xd[3] = {s->td, s->sd, s->ld};
  float* dxin[3] = {dt, dsp, dlg};
  LnBwdSet ln[3];
  for (int m = 0; m < 3; ++m)
    ln[m] = LnBwdSet{dX + m * E, wf + pl.R + m * BE, wf + pl.lnsp + (size_t)m * 2 * B, P_[4 * m + 2], wb + pl.dR + m * BE, G_[4 * m + 2],
                     G_[4 * m + 3]};
  OCEAN_TRY(layernorm_bwd_batch(ln, 3, 3 * E, E, E, B, E, site(s, 0), st));
  if (rc_active()) {
    for (int m = 0; m < 3; ++m) OCEAN_TRY(relu_bwd_inplace(wb + pl.dR + m * BE, E, wf + pl.R + m * BE, E, B, E, st));
  } else {
    OCEAN_TRY(relu_bwd_inplace(wb + pl.dR, E, wf + pl.R, E, 3 * B, E, st));
  }
  LinBwd lb[3];
  for (int m = 0; m < 3; ++m)
    lb[m] = LinBwd{xin[m], xd[m], wb + pl.dR + m * BE, E, B, E, xd[m], P_[4 * m], G_[4 * m], G_[4 * m + 1], dxin[m], xd[m]};
  OCEAN_TRY(linear_bwd_batch(lb, 3, 0, st));
  return rc.finish();
}

// ---- one ProjectionHead called on its own: Linear -> ReLU -> Dropout -> Linear -> L2 normalise.
// Forward workspace: h [B, hidden] (after ReLU and dropout) | z [B, out] (before normalising) | norms [B].
namespace {
struct ProjPlan { size_t H, Z, nrm, fwd_total, dZ, dH, bwd_total; };
int proj_plan(const ocean_proj_shape* s, ProjPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->B <= 0 || s->in_dim <= 0 || s->hidden <= 0 || s->out_dim <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (s->dropout < 0.f || s->dropout >= 1.f) return OCEAN_ERR_BAD_SHAPE;
  const size_t B = s->B;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->H = take(B * s->hidden); p->Z = take(B * s->out_dim); p->nrm = take(B);
  p->fwd_total = o;
  o = 0;
  p->dZ = take(B * s->out_dim); p->dH = take(B * s->hidden);
  p->bwd_total = o;
  return OCEAN_OK;
}
DropoutP proj_site(const ocean_proj_shape* s) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed;
  d.epoch = s->seed_epoch;
  return d;
}
}  // namespace

size_t ocean_projection_head_fwd_workspace(const ocean_proj_shape* s) { ProjPlan p; return proj_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_projection_head_bwd_workspace(const ocean_proj_shape* s) { ProjPlan p; return proj_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_projection_head_fwd(const ocean_proj_shape* s, const float* x, float* const* P_, float* ws, size_t ws_floats, float* y,
                              void* stream) {
  ProjPlan pl;
  OCEAN_TRY(proj_plan(s, &pl));
  if (!x || !P_ || !ws || !y) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const DropoutP d = proj_site(s);
  Epi e; e.drop = &d;
  OCEAN_TRY(linear_fwd(x, s->in_dim, P_[0], P_[1], ws + pl.H, s->hidden, s->B, s->hidden, s->in_dim, 1, st, d.p > 0.f ? &e : nullptr));
  if (!s->normalize) return linear_fwd(ws + pl.H, s->hidden, P_[2], P_[3], y, s->out_dim, s->B, s->out_dim, s->hidden, 0, st);
  OCEAN_TRY(linear_fwd(ws + pl.H, s->hidden, P_[2], P_[3], ws + pl.Z, s->out_dim, s->B, s->out_dim, s->hidden, 0, st));
  return l2norm_fwd(ws + pl.Z, s->out_dim, y, s->out_dim, ws + pl.nrm, s->B, s->out_dim, 1e-12f, st);
}

int ocean_projection_head_bwd(const ocean_proj_shape* s, const float* x, float* const* P_, const float* wf, const float* y,
                              const float* dy, float* wb, size_t wb_floats, float* const* G_, float* dx, void* stream) {
  ProjPlan pl;
  OCEAN_TRY(proj_plan(s, &pl));
  if (!x || !P_ || !wf || !y || !dy || !wb || !G_) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float* dz = dy;
  if (s->normalize) {
    OCEAN_TRY(l2norm_bwd(dy, s->out_dim, y, s->out_dim, wf + pl.nrm, wb + pl.dZ, s->out_dim, s->B, s->out_dim, 1e-12f, st));
    dz = wb + pl.dZ;
  }
  const DropoutP d = proj_site(s);
  Epi e; e.drop = d.p > 0.f ? &d : nullptr; e.relu_mask = wf + pl.H; e.ld_mask = s->hidden;   // dropout and ReLU backward leaving the data gradient
  OCEAN_TRY(lin_bwd(wf + pl.H, s->hidden, dz, s->out_dim, s->B, s->out_dim, s->hidden, P_[2], G_[2], G_[3], wb + pl.dH, s->hidden, 0, st, &e));
  return lin_bwd(x, s->in_dim, wb + pl.dH, s->hidden, s->B, s->hidden, s->in_dim, P_[0], G_[0], G_[1], dx, s->in_dim, 0, st);
}

}  // extern "C"
