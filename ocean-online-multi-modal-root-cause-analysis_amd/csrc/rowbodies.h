// Per-row (one 64-lane wave per row or per sample) device routines of the attention / fusion / head stages.
// Each is the body of a stand-alone kernel (rowops.hip, mfa.hip, fusion.hip, loss.hip) and, unchanged, of the
// corresponding operation of the stage kernel (rowchain.hip), which carries a tile of samples through a whole chain of
// layers in one launch.  Everything here is a function of one row (or of one sample's rows) only.
#pragma once
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace rowbody {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int kMaxColsPerLane = 16;          // LayerNorm backward keeps a row in registers: widths up to 1024

// ---------------------------------------------------------------- LayerNorm
// y = LayerNorm(dropout(x)) * gamma + beta; stats[row] = {mean, rstd}.  drow: the row's number in the dropout site.
__device__ __forceinline__ void ln_fwd_row(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* __restrict__ y, int ldy, float* __restrict__ stats,
                                           int row, uint64_t drow, int W, float eps, const DropoutP& dp, int lane) {
  const float* xr = x + (long)row * ldx;
  auto val = [&](int c) { float v = xr[c]; if (dp.p > 0.f) v *= dropout_keep(dp, drow * W + c); return v; };
  float s = 0.f;
  for (int c = lane; c < W; c += 64) s += val(c);
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
  for (int c = lane; c < W; c += 64) { const float d = val(c) - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + eps);
  for (int c = lane; c < W; c += 64) y[(long)row * ldy + c] = (val(c) - mean) * rstd * gamma[c] + beta[c];
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx of one row (dropout mask re-applied); the row's terms of dgamma / dbeta are added into pg / pb (lane l owns
// columns l + 64 k).  dx may alias dy.
__device__ __forceinline__ void ln_bwd_row(const float* dy, int lddy, const float* __restrict__ x, int ldx,
                                           const float* __restrict__ stats, const float* __restrict__ gamma, float* dx, int lddx,
                                           int row, uint64_t drow, int W, const DropoutP& dp, int lane,
                                           float (&pg)[kMaxColsPerLane], float (&pb)[kMaxColsPerLane]) {
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const float* xr = x + (long)row * ldx;
  const float* gr = dy + (long)row * lddy;
  float s1 = 0.f, s2 = 0.f;
  float gv[kMaxColsPerLane];
#pragma unroll
  for (int k = 0; k < kMaxColsPerLane; ++k) {
    const int c = lane + 64 * k;
    gv[k] = 0.f;
    if (c < W) {
      float xv = xr[c];
      if (dp.p > 0.f) xv *= dropout_keep(dp, drow * W + c);
      const float xh = (xv - mean) * rstd, g = gr[c];
      gv[k] = g;
      pg[k] += g * xh; pb[k] += g;
      const float dxh = g * gamma[c];
      s1 += dxh; s2 += dxh * xh;
    }
  }
  s1 = wave_sum(s1) / (float)W; s2 = wave_sum(s2) / (float)W;
#pragma unroll
  for (int k = 0; k < kMaxColsPerLane; ++k) {
    const int c = lane + 64 * k;
    if (c < W) {
      float keep = 1.f, xv = xr[c];
      if (dp.p > 0.f) { keep = dropout_keep(dp, drow * W + c); xv *= keep; }
      const float xh = (xv - mean) * rstd;
      dx[(long)row * lddx + c] = rstd * (gv[k] * gamma[c] - s1 - xh * s2) * keep;
    }
  }
}

// ---------------------------------------------------------------- L2 normalisation
__device__ __forceinline__ void l2norm_fwd_row(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                               float* __restrict__ norms, int row, int W, float eps, int lane) {
  const float* xr = x + (long)row * ldx;
  float q = 0.f;
  for (int c = lane; c < W; c += 64) q += xr[c] * xr[c];
  const float nrm = sqrtf(wave_sum(q));
  const float inv = 1.0f / fmaxf(nrm, eps);
  for (int c = lane; c < W; c += 64) y[(long)row * ldy + c] = xr[c] * inv;
  if (lane == 0) norms[row] = nrm;
}

__device__ __forceinline__ void l2norm_bwd_row(const float* dy, int lddy, const float* __restrict__ y, int ldy,
                                               const float* __restrict__ norms, float* dx, int lddx, int row, int W, float eps, int lane) {
  const float* gr = dy + (long)row * lddy;
  const float* yr = y + (long)row * ldy;
  float d = 0.f;
  for (int c = lane; c < W; c += 64) d += gr[c] * yr[c];
  d = wave_sum(d);
  const float nrm = norms[row];
  // below eps the forward is x/eps (linear): no projection term
  const float inv = 1.0f / fmaxf(nrm, eps);
  const float proj = nrm > eps ? d : 0.f;
  for (int c = lane; c < W; c += 64) {
    const float g = gr[c];
    dx[(long)row * lddx + c] = (g - yr[c] * proj) * inv;
  }
}

// ---------------------------------------------------------------- MultiFactorAttention: cross-modal softmax
// Nn: [3][B][W] LayerNormed modality vectors (t, s, l).  G[m] = pi_m * (n_m + w1 k1 + w2 k2) with
// (k1,k2) = (s,l) for t, (t,l) for s, (t,s) for l and (w1,w2) = softmax([n_m.k1, n_m.k2]/sqrt(W)).
struct Softmax3 { float p0, p1, p2; };
__device__ __forceinline__ Softmax3 softmax3(const float* __restrict__ mw) {
  const float m0 = mw[0], m1 = mw[1], m2 = mw[2], mm = fmaxf(m0, fmaxf(m1, m2));
  const float e0 = __expf(m0 - mm), e1 = __expf(m1 - mm), e2 = __expf(m2 - mm), es = e0 + e1 + e2;
  return Softmax3{e0 / es, e1 / es, e2 / es};
}

// The three modality matrices of a stacked operand are addressed as base + m * bs + b * ld (stand-alone kernels:
// ld = W, bs = B * W; the stage kernel passes tiles that live in LDS with their own strides).
__device__ __forceinline__ void cross_fwd_row(const float* __restrict__ Nn, int ldn, long bsn, const float* __restrict__ mw,
                                              float* __restrict__ G, int ldg, long bsg, float* __restrict__ cw, int W, int b, int lane) {
  const float* nt = Nn + (long)b * ldn;
  const float* ns = nt + bsn;
  const float* nl = ns + bsn;
  float dts = 0.f, dtl = 0.f, dsl = 0.f;
  for (int c = lane; c < W; c += 64) { const float t = nt[c], s = ns[c], l = nl[c]; dts += t * s; dtl += t * l; dsl += s * l; }
  const float sc = 1.0f / sqrtf((float)W);
  dts = wave_sum(dts) * sc; dtl = wave_sum(dtl) * sc; dsl = wave_sum(dsl) * sc;
  auto sm2 = [](float a, float bb, float& w1, float& w2) { const float m = fmaxf(a, bb); const float ea = __expf(a - m), eb = __expf(bb - m); w1 = ea / (ea + eb); w2 = eb / (ea + eb); };
  float wt1, wt2, ws1, ws2, wl1, wl2;
  sm2(dts, dtl, wt1, wt2); sm2(dts, dsl, ws1, ws2); sm2(dtl, dsl, wl1, wl2);
  const Softmax3 pi = softmax3(mw);
  float* g0 = G + (long)b * ldg;
  float* g1 = g0 + bsg;
  float* g2 = g1 + bsg;
  for (int c = lane; c < W; c += 64) {
    const float t = nt[c], s = ns[c], l = nl[c];
    g0[c] = pi.p0 * (t + wt1 * s + wt2 * l);
    g1[c] = pi.p1 * (s + ws1 * t + ws2 * l);
    g2[c] = pi.p2 * (l + wl1 * t + wl2 * s);
  }
  if (lane == 0) { float* o = cw + (long)b * 6; o[0] = wt1; o[1] = wt2; o[2] = ws1; o[3] = ws2; o[4] = wl1; o[5] = wl2; }
}

// dN of one sample; returns the sample's terms of d(pi) in dp0..dp2 (wave-reduced, the same value in every lane)
__device__ __forceinline__ void cross_bwd_row(const float* __restrict__ dG, int ldg, long bsg, const float* __restrict__ Nn, int ldn, long bsn,
                                              const float* __restrict__ cw, const float* __restrict__ mw, float* __restrict__ dN, int lddn,
                                              long bsdn, int W, int b, int lane, float& dp0, float& dp1, float& dp2) {
  const float* nt = Nn + (long)b * ldn;
  const float* ns = nt + bsn;
  const float* nl = ns + bsn;
  const float* gt = dG + (long)b * ldg;
  const float* gs = gt + bsg;
  const float* gl = gs + bsg;
  const float* w = cw + (long)b * 6;
  const float wt1 = w[0], wt2 = w[1], ws1 = w[2], ws2 = w[3], wl1 = w[4], wl2 = w[5];
  const Softmax3 pi = softmax3(mw);
  const float p0 = pi.p0, p1 = pi.p1, p2 = pi.p2;
  // dots of de_m = pi_m * dg_m with each key, and dpi_m = dg_m . e_m
  float at_s = 0.f, at_l = 0.f, as_t = 0.f, as_l = 0.f, al_t = 0.f, al_s = 0.f;
  dp0 = dp1 = dp2 = 0.f;
  for (int c = lane; c < W; c += 64) {
    const float t = nt[c], s = ns[c], l = nl[c], a = gt[c], bb = gs[c], cc = gl[c];
    at_s += a * s; at_l += a * l; as_t += bb * t; as_l += bb * l; al_t += cc * t; al_s += cc * s;
    dp0 += a * (t + wt1 * s + wt2 * l); dp1 += bb * (s + ws1 * t + ws2 * l); dp2 += cc * (l + wl1 * t + wl2 * s);
  }
  at_s = wave_sum(at_s) * p0; at_l = wave_sum(at_l) * p0; as_t = wave_sum(as_t) * p1; as_l = wave_sum(as_l) * p1;
  al_t = wave_sum(al_t) * p2; al_s = wave_sum(al_s) * p2;
  dp0 = wave_sum(dp0); dp1 = wave_sum(dp1); dp2 = wave_sum(dp2);
  const float sc = 1.0f / sqrtf((float)W);
  // softmax backward per modality: ds1 = w1*(dw1 - (w1 dw1 + w2 dw2)), scaled by 1/sqrt(W)
  const float tt = wt1 * at_s + wt2 * at_l, dst1 = wt1 * (at_s - tt) * sc, dst2 = wt2 * (at_l - tt) * sc;
  const float ss = ws1 * as_t + ws2 * as_l, dss1 = ws1 * (as_t - ss) * sc, dss2 = ws2 * (as_l - ss) * sc;
  const float ll = wl1 * al_t + wl2 * al_s, dsl1 = wl1 * (al_t - ll) * sc, dsl2 = wl2 * (al_s - ll) * sc;
  float* d0 = dN + (long)b * lddn;
  float* d1 = d0 + bsdn;
  float* d2 = d1 + bsdn;
  for (int c = lane; c < W; c += 64) {
    const float t = nt[c], s = ns[c], l = nl[c];
    const float a = gt[c] * p0, bb = gs[c] * p1, cc = gl[c] * p2;       // de_t, de_s, de_l
    // t: query of its own block (keys s,l); key 1 of s's and l's blocks
    const float dt = a + dst1 * s + dst2 * l + ws1 * bb + dss1 * s + wl1 * cc + dsl1 * l;
    // s: query (keys t,l); key 1 of t's block, key 2 of l's block
    const float ds = bb + dss1 * t + dss2 * l + wt1 * a + dst1 * t + wl2 * cc + dsl2 * l;
    // l: query (keys t,s); key 2 of t's and s's blocks
    const float dl = cc + dsl1 * t + dsl2 * s + wt2 * a + dst2 * t + ws2 * bb + dss2 * s;
    d0[c] = dt; d1[c] = ds; d2[c] = dl;
  }
}

// ---------------------------------------------------------------- MultiFactorAttention: sigmoid feature gate
// modality m of sample b (row r = m*B + b of the stacked matrices): i = sigmoid(hid[r].w2 + b2);  cat[b, m*W + c] = G[r,c] * i
__device__ __forceinline__ void gate_fwd_row(const float* __restrict__ hid, int ldh, long bsh, const float* __restrict__ w2,
                                             const float* __restrict__ b2, const float* __restrict__ G, int ldg, long bsg,
                                             float* __restrict__ ig, float* __restrict__ cat, int ldcat, int B, int W, int Wh, int m, int b, int lane) {
  const float* hr = hid + m * bsh + (long)b * ldh;
  const float* gr = G + m * bsg + (long)b * ldg;
  float z = 0.f;
  for (int k = lane; k < Wh; k += 64) z += hr[k] * w2[k];
  z = wave_sum(z) + b2[0];
  const float i = 1.0f / (1.0f + __expf(-z));
  float* cr = cat + (long)b * ldcat + m * W;
  for (int c = lane; c < W; c += 64) cr[c] = gr[c] * i;
  if (lane == 0) ig[m * B + b] = i;
}

// dG = dcat * i;  dz = (dcat . G) * i(1-i);  dhid = dz * w2 where hid > 0
__device__ __forceinline__ void gate_bwd_row(const float* __restrict__ dcat, int ldcat, const float* __restrict__ G, int ldg, long bsg,
                                             const float* __restrict__ ig, const float* __restrict__ hid, int ldh, long bsh,
                                             const float* __restrict__ w2, float* __restrict__ dG, int lddg, long bsdg, float* __restrict__ dz,
                                             float* __restrict__ dhid, int lddh, long bsdh, int B, int W, int Wh, int m, int b, int lane) {
  const float i = ig[m * B + b];
  const float* dc = dcat + (long)b * ldcat + m * W;
  const float* gr = G + m * bsg + (long)b * ldg;
  float* dg = dG + m * bsdg + (long)b * lddg;
  float d = 0.f;
  for (int c = lane; c < W; c += 64) {
    const float g = dc[c];
    d += g * gr[c];
    dg[c] = g * i;
  }
  d = wave_sum(d) * i * (1.0f - i);
  const float* hr = hid + m * bsh + (long)b * ldh;
  float* dh = dhid + m * bsdh + (long)b * lddh;
  for (int k = lane; k < Wh; k += 64) dh[k] = hr[k] > 0.f ? d * w2[k] : 0.f;
  if (lane == 0) dz[m * B + b] = d;
}

// ---------------------------------------------------------------- GraphFusionModule: three-token attention
// One wave per sample (E = 64 * VEC columns, a power-of-two head count): lane l owns columns l*VEC .. l*VEC+VEC-1 of
// every row, the 64/H lanes of a head reduce their partial dot products with xor shuffles.
constexpr int kTok = 3;
template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else { v[0] = p[0]; }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  else p[0] = v[0];
}

// qkv [B*3, 3E] rows (b,l): [q | k | v].  probs [B,H,3,3] post-softmax (pre-dropout).  o [B*3, E].
template <int VEC>
__device__ __forceinline__ void mha3_fwd_sample(const float* __restrict__ qkv, float* __restrict__ probs, float* __restrict__ o,
                                                int H, int hlog, const DropoutP& dp, int b, int lane) {
  const int E = 64 * VEC, lph = 64 >> hlog, h = lane >> (6 - hlog), c = lane * VEC;
  const float sc = 1.0f / sqrtf((float)(lph * VEC));
  const float* base = qkv + (long)b * kTok * 3 * E + c;
  float q[kTok][VEC], k[kTok][VEC], v[kTok][VEC];
#pragma unroll
  for (int i = 0; i < kTok; ++i) { ldv<VEC>(base + (long)i * 3 * E, q[i]); ldv<VEC>(base + (long)i * 3 * E + E, k[i]); ldv<VEC>(base + (long)i * 3 * E + 2 * E, v[i]); }
  float S[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i)
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < VEC; ++u) t = fmaf(q[i][u] * sc, k[j][u], t);
      for (int sh = 1; sh < lph; sh <<= 1) t += __shfl_xor(t, sh);
      S[i][j] = t;
    }
  const int idx = b * H + h;
  float A[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i) {
    const float m = fmaxf(S[i][0], fmaxf(S[i][1], S[i][2]));
    const float e0 = __expf(S[i][0] - m), e1 = __expf(S[i][1] - m), e2 = __expf(S[i][2] - m), es = e0 + e1 + e2;
    A[i][0] = e0 / es; A[i][1] = e1 / es; A[i][2] = e2 / es;
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      if ((lane & (lph - 1)) == 0) probs[((long)idx * kTok + i) * kTok + j] = A[i][j];
      if (dp.p > 0.f) A[i][j] *= dropout_keep(dp, ((uint64_t)idx * kTok + i) * kTok + j);
    }
  }
#pragma unroll
  for (int i = 0; i < kTok; ++i) {
    float r[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) r[u] = A[i][0] * v[0][u] + A[i][1] * v[1][u] + A[i][2] * v[2][u];
    stv<VEC>(o + ((long)b * kTok + i) * E + c, r);
  }
}

// dOm [B,E]: gradient of the token mean (each token receives dOm/3).  dqkv [B*3, 3E].
template <int VEC>
__device__ __forceinline__ void mha3_bwd_sample(const float* __restrict__ dOm, const float* __restrict__ qkv, const float* __restrict__ probs,
                                                float* __restrict__ dqkv, int H, int hlog, const DropoutP& dp, int b, int lane) {
  const int E = 64 * VEC, lph = 64 >> hlog, h = lane >> (6 - hlog), c = lane * VEC;
  const float sc = 1.0f / sqrtf((float)(lph * VEC));
  const float* base = qkv + (long)b * kTok * 3 * E + c;
  float* dbase = dqkv + (long)b * kTok * 3 * E + c;
  const int idx = b * H + h;
  float q[kTok][VEC], k[kTok][VEC], v[kTok][VEC], g[VEC];
#pragma unroll
  for (int i = 0; i < kTok; ++i) { ldv<VEC>(base + (long)i * 3 * E, q[i]); ldv<VEC>(base + (long)i * 3 * E + E, k[i]); ldv<VEC>(base + (long)i * 3 * E + 2 * E, v[i]); }
  ldv<VEC>(dOm + (long)b * E + c, g);
#pragma unroll
  for (int u = 0; u < VEC; ++u) g[u] *= (1.0f / kTok);
  float A[kTok][kTok], Ad[kTok][kTok], keep[kTok][kTok];
#pragma unroll
  for (int i = 0; i < kTok; ++i)
#pragma unroll
    for (int j = 0; j < kTok; ++j) {
      A[i][j] = probs[((long)idx * kTok + i) * kTok + j];
      keep[i][j] = dp.p > 0.f ? dropout_keep(dp, ((uint64_t)idx * kTok + i) * kTok + j) : 1.f;
      Ad[i][j] = A[i][j] * keep[i][j];
    }
  float dv_dot[kTok];
#pragma unroll
  for (int j = 0; j < kTok; ++j) {
    float t = 0.f, dv[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) { t = fmaf(g[u], v[j][u], t); dv[u] = g[u] * (Ad[0][j] + Ad[1][j] + Ad[2][j]); }
    for (int sh = 1; sh < lph; sh <<= 1) t += __shfl_xor(t, sh);
    dv_dot[j] = t;
    stv<VEC>(dbase + (long)j * 3 * E + 2 * E, dv);
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
#pragma unroll
  for (int i = 0; i < kTok; ++i) {
    float dq[VEC], dk[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      dq[u] = sc * (dS[i][0] * k[0][u] + dS[i][1] * k[1][u] + dS[i][2] * k[2][u]);
      dk[u] = sc * (dS[0][i] * q[0][u] + dS[1][i] * q[1][u] + dS[2][i] * q[2][u]);
    }
    stv<VEC>(dbase + (long)i * 3 * E, dq);
    stv<VEC>(dbase + (long)i * 3 * E + E, dk);
  }
}

// wave-per-sample variants apply: E = 64, 128 or 256 columns and a power-of-two head count of at most 64
inline int mha3_vec(int E, int H) {
  if (H <= 0 || H > 64 || (H & (H - 1)) != 0) return 0;
  return E == 256 ? 4 : E == 128 ? 2 : E == 64 ? 1 : 0;
}
inline int mha3_hlog(int H) { int l = 0; while ((1 << l) < H) ++l; return l; }

}  // namespace rowbody
