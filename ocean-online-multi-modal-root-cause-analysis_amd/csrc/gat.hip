// GraphNeuralNetwork (GAT) forward / backward on gfx950.
//
// The reference batches B disjoint copies of one service graph
// (ocean_model.py:155-171) and runs torch_geometric's GATConv on the union.
// Here the graph is held once as two CSRs (incoming edges per target for the
// forward gather, outgoing edges per source for the backward gather, self loops
// already normalised the way GATConv does) and the copies are a replica axis:
// node tensors are [R*N, .].  With dropout off every replica is identical and
// the caller passes R = 1.
//
// Node kernels use one 64-lane wave per (replica, node): lane l owns CPL
// consecutive channels of the H*C-wide row, so a row gather is one coalesced
// load; per-head reductions (edge scores' dot products) are wavefront shuffles
// when 64/H lanes hold exactly one head, and go through wave-private LDS
// otherwise (odd sizes).  Edge softmax is two passes over the target's incoming
// edges (max, then sum and weighted accumulation); scores are never stored -
// backward recomputes alpha from the saved per-(node,head) max and denominator.
#include "../../include/ocean_hip.h"
#include "ops.h"
#include <stdlib.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct GatK {
  int N, R, H, C, HC;                 // nodes, replicas, heads, channels per head, H*C
  int E;                              // edges in the CSRs (self loops included)
  const int* rp_in; const int* src_in;                       // by target
  const int* rp_out; const int* dst_out; const int* eid_out; // by source; eid = position in the by-target order
  float slope;
  DropoutP att_drop;                  // attention-coefficient dropout (GATConv's own)
  int z_shared;                       // Z, a_s, a_d hold ONE replica ([N, .]) that every replica reads (the first layer's
                                      // input is the same for all replicas; only the dropout masks differ)
};

// The layer's output stage y = dropout(relu(headmix(agg) + bias)) (+ xin), for the kernels that apply it to the
// aggregated row while it is still in registers (y == nullptr: left to gat_epilogue_kernel).
struct GatEpi {
  const float* bias; const float* xin; float* y;
  int xin_rows;                       // > 0: xin holds one replica of that many rows, read by all
  int concat;
  DropoutP fd;
};

constexpr int kWavesPerBlock = 4;

template <int CPL>
struct Lane {
  int c0, hk[CPL];
  bool ok[CPL];
  __device__ Lane(int lane, const GatK& g) {
    c0 = lane * CPL;
#pragma unroll
    for (int k = 0; k < CPL; ++k) { ok[k] = c0 + k < g.HC; hk[k] = ok[k] ? (c0 + k) / g.C : 0; }
  }
};

// Orders one wave's LDS traffic: lanes exchange values through wave-private LDS, and the LDS unit serves one wave's
// instructions in program order, so only the compiler has to be kept from moving accesses across this point.  (A
// workgroup-scope fence here also waits for every outstanding global load and store of the wave, which made each
// node wait for the previous node's output stores to be acknowledged.)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// hs[k] = sum over the head of channel (c0+k) of prod[.]
template <int CPL, bool FAST>
__device__ __forceinline__ void head_sums(const float (&prod)[CPL], float (&hs)[CPL], const Lane<CPL>& L, const GatK& g,
                                          int lane, float* wlds) {
  if (FAST) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) v += prod[k];
    for (int o = (64 / g.H) >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
#pragma unroll
    for (int k = 0; k < CPL; ++k) hs[k] = v;
  } else {
    // wave-private LDS: [HC] products then [H] sums
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < CPL; ++k) if (L.ok[k]) wlds[L.c0 + k] = prod[k];
    wave_lds_sync();
    for (int h = lane; h < g.H; h += 64) {
      float s = 0.f;
      for (int c = 0; c < g.C; ++c) s += wlds[h * g.C + c];
      wlds[g.HC + h] = s;
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < CPL; ++k) hs[k] = L.ok[k] ? wlds[g.HC + L.hk[k]] : 0.f;
  }
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : slope * x; }

// Cross-lane reductions on the VALU data-parallel-primitive path.  __shfl_xor goes through the LDS crossbar
// (ds_bpermute), and at ~50 shuffles per node the gather kernels were bound by that pipe; within a row of 16 lanes
// the same sums are available as DPP modifiers on ordinary VALU instructions.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;   // quad_perm / row mirrors
// sum over aligned groups of `width` consecutive lanes (power of two); every lane of the group gets the total
__device__ __forceinline__ float group_sum(float v, int width) {
  if (width >= 2) v += dpp_f<kDppXor1>(v);
  if (width >= 4) v += dpp_f<kDppXor2>(v);
  if (width >= 8) v += dpp_f<kDppHalfMirror>(v);     // lane i <- lane 7-i of its half row: the other quad's total
  if (width >= 16) v += dpp_f<kDppMirror>(v);        // lane i <- lane 15-i: the other half row's total
  for (int o = 16; o < width; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
// reductions over the lanes that share (lane % stride), stride a power of two: row rotations inside a 16-lane
// row, LDS-crossbar shuffles across rows
template <bool MAX>
__device__ __forceinline__ float strided_reduce(float v, int stride) {
  auto comb = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  if (stride <= 1) v = comb(v, dpp_f<0x121>(v));     // row_ror:1
  if (stride <= 2) v = comb(v, dpp_f<0x122>(v));     // row_ror:2
  if (stride <= 4) v = comb(v, dpp_f<0x124>(v));     // row_ror:4
  if (stride <= 8) v = comb(v, dpp_f<0x128>(v));     // row_ror:8
  for (int o = stride > 16 ? stride : 16; o < 64; o <<= 1) v = comb(v, __shfl_xor(v, o));
  return v;
}

// a_s[row,h] = sum_c Z[row,h,c]*att_src[h,c]; same for a_d
template <int CPL, bool FAST>
__global__ __launch_bounds__(256) void gat_scores_kernel(GatK g, const float* __restrict__ Z, const float* __restrict__ att_src,
                                                         const float* __restrict__ att_dst, float* __restrict__ a_s,
                                                         float* __restrict__ a_d) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wl = lds + wave * (g.HC + g.H);
  const long node = (long)blockIdx.x * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  const Lane<CPL> L(lane, g);
  float ps[CPL], pd[CPL], hs[CPL], hd[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const float z = L.ok[k] ? Z[node * g.HC + L.c0 + k] : 0.f;
    ps[k] = L.ok[k] ? z * att_src[L.c0 + k] : 0.f;
    pd[k] = L.ok[k] ? z * att_dst[L.c0 + k] : 0.f;
  }
  head_sums<CPL, FAST>(ps, hs, L, g, lane, wl);
  head_sums<CPL, FAST>(pd, hd, L, g, lane, wl);
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (L.ok[k] && (L.c0 + k) % g.C == 0) { a_s[node * g.H + L.hk[k]] = hs[k]; a_d[node * g.H + L.hk[k]] = hd[k]; }
}

// agg[node,h,c] = sum_{j->i} dropout(alpha_ji) * Z[j,h,c];  saves softmax max / denominator per (node, head)
template <int CPL>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(GatK g, const float* __restrict__ Z, const float* __restrict__ a_s,
                                                            const float* __restrict__ a_d, float* __restrict__ agg,
                                                            float* __restrict__ smax, float* __restrict__ sden) {
  g.att_drop = dropout_resolve(g.att_drop);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long node = (long)blockIdx.x * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  const int r = (int)(node / g.N), i = (int)(node % g.N);
  const long rbase = (long)r * g.N;
  const Lane<CPL> L(lane, g);
  float adi[CPL], mx[CPL], den[CPL], acc[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) { adi[k] = L.ok[k] ? a_d[node * g.H + L.hk[k]] : 0.f; mx[k] = -INFINITY; den[k] = 0.f; acc[k] = 0.f; }
  const int e0 = g.rp_in[i], e1 = g.rp_in[i + 1];
  for (int e = e0; e < e1; ++e) {
    const long nj = rbase + g.src_in[e];
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (L.ok[k]) mx[k] = fmaxf(mx[k], lrelu(a_s[nj * g.H + L.hk[k]] + adi[k], g.slope));
  }
  for (int e = e0; e < e1; ++e) {
    const long nj = rbase + g.src_in[e];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (!L.ok[k]) continue;
      const float ex = __expf(lrelu(a_s[nj * g.H + L.hk[k]] + adi[k], g.slope) - mx[k]);
      den[k] += ex;
      float w = ex;
      if (g.att_drop.p > 0.f) w *= dropout_keep(g.att_drop, ((uint64_t)r * g.E + e) * g.H + L.hk[k]);
      acc[k] = fmaf(w, Z[nj * g.HC + L.c0 + k], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    if (!L.ok[k]) continue;
    agg[node * g.HC + L.c0 + k] = acc[k] / (den[k] + 1e-16f);
    if ((L.c0 + k) % g.C == 0) { smax[node * g.H + L.hk[k]] = mx[k]; sden[node * g.H + L.hk[k]] = den[k]; }
  }
}

// ---- fast variants (H a power of two <= 16, H*C == 64*CPL) -------------------------------------------------
// The per-edge scalar work (score, softmax weight, dropout) is done with the wave laid out as
// (edge slot, head) = (lane / H, lane % H), 64/H edges at a time, and parked in wave-private LDS; the row
// gathers then run as an unrolled loop whose only memory dependency is the row address read back from LDS,
// so four 64-lane row loads are in flight at once instead of one load chain per edge.
// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Remapping the block id so that every
// XCD walks a contiguous range of rows keeps one graph replica's rows (which its edges gather from) in one L2
// instead of fetching them into all eight.
__device__ __forceinline__ long xcd_contiguous_block(unsigned orig, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7, xcd = orig & 7;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return (long)(base + (orig >> 3));
}

template <int CPL>
__device__ __forceinline__ void load_row(const float* p, float (&v)[CPL]) {
  if (CPL == 1) { v[0] = p[0]; }
  else if (CPL == 2) { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[1 % CPL] = a.y; }
  else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const float4 a = reinterpret_cast<const float4*>(p)[q];
      v[(4 * q) % CPL] = a.x; v[(4 * q + 1) % CPL] = a.y; v[(4 * q + 2) % CPL] = a.z; v[(4 * q + 3) % CPL] = a.w;
    }
  }
}
template <int CPL>
__device__ __forceinline__ void store_row(float* p, const float (&v)[CPL]) {
  if (CPL == 1) { p[0] = v[0]; }
  else if (CPL == 2) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1 % CPL]); }
  else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q)
      reinterpret_cast<float4*>(p)[q] = make_float4(v[(4 * q) % CPL], v[(4 * q + 1) % CPL], v[(4 * q + 2) % CPL], v[(4 * q + 3) % CPL]);
  }
}

template <int CPL>
__global__ __launch_bounds__(256) void gat_aggregate_fast_kernel(GatK g, int hlog, const float* __restrict__ Z,
                                                                 const float* __restrict__ a_s, const float* __restrict__ a_d,
                                                                 float* __restrict__ agg, float* __restrict__ smax,
                                                                 float* __restrict__ sden, GatEpi epi) {
  g.att_drop = dropout_resolve(g.att_drop);
  epi.fd = dropout_resolve(epi.fd);
  __shared__ float wl_all[kWavesPerBlock][64];
  __shared__ int jl_all[kWavesPerBlock][64];
  __shared__ float dl_all[kWavesPerBlock][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long node = xcd_contiguous_block(blockIdx.x, gridDim.x) * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  float* wl = wl_all[wave]; int* jl = jl_all[wave]; float* dl = dl_all[wave];
  const int r = (int)(node / g.N), i = (int)(node % g.N);
  const long rbase = g.z_shared ? 0 : (long)r * g.N;   // base row of this replica in Z / a_s / a_d
  const int H = g.H, EPW = 64 >> hlog;
  const int le = lane >> hlog, lh = lane & (H - 1);   // edge-role coordinates
  const int hk = lane >> (6 - hlog), c0 = lane * CPL;  // channel-role coordinates
  const float adl = a_d[(rbase + i) * H + lh];
  const int e0 = g.rp_in[i], e1 = g.rp_in[i + 1];
  // what the output stage at the end needs from memory is requested now, ahead of the gathers
  const int Cw = g.HC >> hlog, od = epi.concat ? g.HC : Cw;
  const int oc0 = epi.concat ? c0 : c0 - hk * Cw;          // this lane's first output channel
  const bool writes_y = epi.y != nullptr && (epi.concat || hk == 0);
  float bias_r[CPL], xin_r[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) { bias_r[k] = 0.f; xin_r[k] = 0.f; }
  if (epi.y != nullptr) load_row<CPL>(epi.bias + oc0, bias_r);
  if (writes_y && epi.xin != nullptr) load_row<CPL>(epi.xin + (epi.xin_rows > 0 ? (long)i : node) * od + oc0, xin_r);
  const bool single = (e1 - e0) <= EPW;
  float mx = -INFINITY, s_last = -INFINITY;
  int j_last = 0;
  for (int base = e0; base < e1; base += EPW) {
    const int e = base + le;
    const bool valid = e < e1;
    const int j = valid ? g.src_in[e] : 0;
    const float sv = valid ? lrelu(a_s[(rbase + j) * H + lh] + adl, g.slope) : -INFINITY;
    mx = fmaxf(mx, sv);
    s_last = sv; j_last = j;
  }
  mx = strided_reduce<true>(mx, H);
  float den = 0.f, acc[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
  for (int base = e0; base < e1; base += EPW) {
    const int e = base + le;
    const bool valid = e < e1;
    int j = j_last;
    float sv = s_last;
    if (!single) {
      j = valid ? g.src_in[e] : 0;
      sv = valid ? lrelu(a_s[(rbase + j) * H + lh] + adl, g.slope) : -INFINITY;
    }
    const float ex = valid ? __expf(sv - mx) : 0.f;
    den += ex;
    float w = ex;
    if (g.att_drop.p > 0.f && valid) w *= dropout_keep(g.att_drop, ((uint64_t)r * g.E + e) * H + lh);
    wave_lds_sync();
    wl[lane] = w;
    if (lh == 0) jl[le] = j;
    wave_lds_sync();
    const int n = min(EPW, e1 - base);
    for (int q = 0; q < n; q += 4) {     // slots n..n+3 hold weight 0 and a valid row index
      float zz[4][CPL], ww[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load_row<CPL>(Z + (rbase + jl[q + u]) * g.HC + c0, zz[u]);
        ww[u] = wl[((q + u) << hlog) + hk];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = fmaf(ww[u], zz[u][k], acc[k]);
    }
  }
  den = strided_reduce<false>(den, H);
  wave_lds_sync();
  if (le == 0) { smax[node * H + lh] = mx; sden[node * H + lh] = den; dl[lh] = den; }
  wave_lds_sync();
  const float inv = 1.0f / (dl[hk] + 1e-16f);
#pragma unroll
  for (int k = 0; k < CPL; ++k) acc[k] *= inv;
  store_row<CPL>(agg + node * g.HC + c0, acc);     // kept: the backward pass takes the ReLU mask from it
  if (epi.y != nullptr) {
    // the layer's output stage on the row while it is in registers (what gat_epilogue_kernel does in a pass of its own):
    // mean over the heads (or their concatenation), bias, ReLU, feature dropout, residual
    float o[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      float v = acc[k];
      if (!epi.concat) {
        for (int sh = 64 >> hlog; sh < 64; sh <<= 1) v += __shfl_xor(v, sh);     // sum over the heads
        v *= 1.0f / (float)H;
      }
      v = fmaxf(v + bias_r[k], 0.f);
      if (epi.fd.p > 0.f && writes_y) v *= dropout_keep(epi.fd, (uint64_t)(node * od + oc0 + k));
      o[k] = v + xin_r[k];
    }
    if (writes_y) store_row<CPL>(epi.y + node * od + oc0, o);
  }
}

// da_d[row, h] = sum over the incoming edges e of row's node of ds_edge[(r*E + e), h]   (edges of one target are
// consecutive in by-target order, so each thread reads a short contiguous run)
__global__ void gat_bwd_target_sum_kernel(GatK g, const float* __restrict__ ds_edge, float* __restrict__ da_d) {
  const long n = (long)g.R * g.N * g.H;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int h = (int)(t % g.H);
    const long row = t / g.H;
    const int i = (int)(row % g.N);
    const long r = row / g.N;
    float s = 0.f;
    for (int e = g.rp_in[i]; e < g.rp_in[i + 1]; ++e) s += ds_edge[(r * g.E + e) * g.H + h];
    da_d[t] = s;
  }
}

template <int CPL>
__global__ __launch_bounds__(256) void gat_bwd_source_fast_kernel(GatK g, int hlog, const float* __restrict__ Z,
                                                                  const float* __restrict__ a_s, const float* __restrict__ a_d,
                                                                  const float* __restrict__ smax, const float* __restrict__ sden,
                                                                  const float* __restrict__ dagg, const float* __restrict__ Dn,
                                                                  float* __restrict__ dZ, float* __restrict__ da_s,
                                                                  float* __restrict__ ds_edge) {
  g.att_drop = dropout_resolve(g.att_drop);
  __shared__ float wl_all[kWavesPerBlock][64];
  __shared__ float hl_all[kWavesPerBlock][64];
  __shared__ int il_all[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long node = xcd_contiguous_block(blockIdx.x, gridDim.x) * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  float* wl = wl_all[wave]; float* hl = hl_all[wave]; int* il = il_all[wave];
  const int r = (int)(node / g.N), j = (int)(node % g.N);
  const long rbase = (long)r * g.N;
  const long zbase = g.z_shared ? 0 : rbase;               // base row of this replica in Z / a_s / a_d
  const int H = g.H, EPW = 64 >> hlog, LPH = 64 >> hlog;   // edges per pass; lanes per head in the channel role
  const int le = lane >> hlog, lh = lane & (H - 1);
  const int hk = lane >> (6 - hlog), c0 = lane * CPL;
  float z[CPL], dz[CPL];
  load_row<CPL>(Z + (zbase + j) * g.HC + c0, z);
#pragma unroll
  for (int k = 0; k < CPL; ++k) dz[k] = 0.f;
  const float asl = a_s[(zbase + j) * H + lh];
  float das = 0.f;
  const int e0 = g.rp_out[j], e1 = g.rp_out[j + 1];
  for (int base = e0; base < e1; base += EPW) {
    const int e = base + le;
    const bool valid = e < e1;
    const int i = valid ? g.dst_out[e] : j;
    const long nh = (rbase + i) * H + lh;
    const float raw = asl + a_d[(zbase + i) * H + lh];
    const float alpha = valid ? __expf(lrelu(raw, g.slope) - smax[nh]) / (sden[nh] + 1e-16f) : 0.f;
    const long eslot = valid ? ((long)r * g.E + g.eid_out[e]) * H + lh : 0;   // this edge's slot in by-target order
    float keep = 1.f;
    if (g.att_drop.p > 0.f && valid) keep = dropout_keep(g.att_drop, (uint64_t)eslot);
    const float dnv = Dn[nh];
    wave_lds_sync();
    wl[lane] = alpha * keep;
    if (lh == 0) il[le] = i;
    wave_lds_sync();
    const int n = min(EPW, e1 - base);
    for (int q = 0; q < n; q += 4) {
      float dg[4][CPL], ww[4], pr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load_row<CPL>(dagg + (rbase + il[q + u]) * g.HC + c0, dg[u]);
        ww[u] = wl[((q + u) << hlog) + hk];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { dz[k] = fmaf(ww[u], dg[u][k], dz[k]); p = fmaf(dg[u][k], z[k], p); }
        pr[u] = p;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) pr[u] = group_sum(pr[u], LPH);
      if ((lane & (LPH - 1)) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) hl[((q + u) << hlog) + hk] = pr[u];
      }
    }
    wave_lds_sync();
    if (valid) {
      const float de = alpha * (keep * hl[lane] - dnv);
      const float ds = de * (raw > 0.f ? 1.f : g.slope);
      das += ds;
      ds_edge[eslot] = ds;     // summed per target by gat_bwd_target_sum_kernel: a plain store instead of an atomic
    }
  }
  das = strided_reduce<false>(das, H);
  if (le == 0) da_s[node * H + lh] = das;
  store_row<CPL>(dZ + node * g.HC + c0, dz);
}

// ---- replica-block variants for small graphs ------------------------------------------------------------------
// Counters on the one-node-per-wave kernels above show the vector ALU as their limit (about 250 vector instructions
// per node, most of them address arithmetic and table lookups that are the same for all 64 lanes).  Here a block
// takes a whole replica (or a node range of it): the per-(node, head) tables the edge lanes index are staged in LDS
// once per block, the node index is wave-uniform so the CSR walk and the row addresses of the gathers are scalar
// loads and scalar arithmetic, and the head count is a template parameter so LDS offsets fold into the instructions.
constexpr int kTileWaves = 16;
constexpr int kWaitVm0 = 0x0F70;   // s_waitcnt vmcnt(0) only (expcnt, lgkmcnt fields at their maxima)

// a * b + c on the full-rate 24-bit multiplier (a, b < 2^24); the 32-bit integer multiply runs at a quarter of that
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}

// CPL consecutive floats at byte offset `off` of a bounds-checked buffer (reads past the end return 0)
template <int CPL>
__device__ __forceinline__ void load_row_buf(__amdgpu_buffer_rsrc_t rs, unsigned off, float (&v)[CPL]) {
  // (a bit_cast applied to a vector element reads element 0: copy the elements out first)
  if constexpr (CPL == 1) {
    v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
  } else if constexpr (CPL == 2) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    const unsigned x = r[0], y = r[1];
    v[0] = __uint_as_float(x); v[1] = __uint_as_float(y);
  } else {
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 16 * q, 0, 0);
      const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
      v[4 * q] = __uint_as_float(x); v[4 * q + 1] = __uint_as_float(y); v[4 * q + 2] = __uint_as_float(z); v[4 * q + 3] = __uint_as_float(w);
    }
  }
}

// Block id -> (replica, node range).  The node ranges of one replica must share an L2 and run at the same time: a
// replica's rows are gathered ~9 times over, and an XCD's 64 resident blocks on 64 different replicas of 128 KB
// overflow its 4 MB L2 (measured: 5x the compulsory fetch).  Blocks are dealt to the 8 XCDs round-robin by id, so
// ids that differ by 8 are neighbours on one XCD: those become the ranges of one replica.
__device__ __forceinline__ void tile_block(unsigned id, int R, int chunks, int& r, int& chunk) {
  if ((R & 7) == 0) {
    const unsigned per = 8u * (unsigned)chunks;
    r = (int)((id / per) * 8u + (id & 7u));
    chunk = (int)((id >> 3) % (unsigned)chunks);
  } else {
    r = (int)(id / (unsigned)chunks);
    chunk = (int)(id % (unsigned)chunks);
  }
}

template <int CPL, int HLOG>
__global__ __launch_bounds__(kTileWaves * 64) void gat_aggregate_tile_kernel(GatK g, int npb, int chunks, const float* __restrict__ Z,
                                                                             const float* __restrict__ a_s,
                                                                             const float* __restrict__ a_d, float* __restrict__ agg,
                                                                             float* __restrict__ smax, float* __restrict__ sden,
                                                                             GatEpi epi) {
  constexpr int H = 1 << HLOG, EPW = 64 >> HLOG;
  constexpr int RB = CPL <= 2 ? 8 : 4, GB = EPW < RB ? EPW : RB;   // rows in flight per wave
  g.att_drop = dropout_resolve(g.att_drop);
  epi.fd = dropout_resolve(epi.fd);
  extern __shared__ float lds[];
  const int N = g.N, E = g.E;
  float* as_l = lds;                                         // [N*H]   source logits of this replica
  float* ad_l = as_l + N * H;                                // [N*H]   target logits
  int* src_l = reinterpret_cast<int*>(ad_l + N * H);         // [E+GB]  padded: the early row requests read GB indices
  float* wsc = reinterpret_cast<float*>(src_l + E + GB);      // per wave: weights [64], row indices [64], denominators [16]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int r, chunk;
  tile_block(blockIdx.x, g.R, chunks, r, chunk);
  const long rbase = (long)r * N;
  const long zbase = g.z_shared ? 0 : rbase;
  for (int t = threadIdx.x; t < N * H; t += kTileWaves * 64) { as_l[t] = a_s[zbase * H + t]; ad_l[t] = a_d[zbase * H + t]; }
  for (int t = threadIdx.x; t < E + GB; t += kTileWaves * 64) src_l[t] = t < E ? g.src_in[t] : 0;
  __syncthreads();
  float* wl = wsc + wave * 144;
  int* jl = reinterpret_cast<int*>(wl + 64);
  float* dl = wl + 128;
  const int le = lane >> HLOG, lh = lane & (H - 1);    // edge-role coordinates
  const int hk = lane >> (6 - HLOG), c0 = lane * CPL;  // channel-role coordinates
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)g.HC * 4u), lane_off = (unsigned)c0 * 4u;
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z + zbase * g.HC), 0, N * (int)row_bytes, 0x00020000);
  const int i0 = chunk * npb, i1 = min(N, i0 + npb);
  // lane k holds the edge range of the k-th node this wave will visit (npb <= 64 * kTileWaves)
  const int imine = min(i0 + wave + lane * kTileWaves, N - 1);
  const int rp0v = g.rp_in[imine], rp1v = g.rp_in[imine + 1];
  // output stage: a lane's channels c0 .. c0+CPL-1 belong to head hk; with head mixing = mean the lanes of head 0 end
  // up with the sums and write channel (c0 + k) of the C-wide output
  const int Cw = g.HC >> HLOG;
  float bias_r[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) bias_r[k] = epi.y != nullptr ? epi.bias[epi.concat ? c0 + k : (c0 + k) % Cw] : 0.f;
  // wait for them here: left to the compiler, the wait lands at the top of the node loop, where it also waits for the
  // previous node's output stores to be acknowledged
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  int trip = 0;
  for (int i = i0 + wave; i < i1; i += kTileWaves, ++trip) {       // i is wave-uniform
    const long node = rbase + i;
    const float adl = ad_l[i * H + lh];
    const int e0 = __builtin_amdgcn_readlane(rp0v, trip), e1 = __builtin_amdgcn_readlane(rp1v, trip);
    const bool single = (e1 - e0) <= EPW;
    // The first GB source rows are requested before the softmax over the edges is worked out (their indices come
    // straight from the CSR), so that work runs under the rows' L2 round trip instead of in front of it.
    const int n0 = min(EPW, e1 - e0);
    float z0[GB][CPL];
#pragma unroll
    for (int q = 0; q < GB; q += 4) {
      if (q < n0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jr = src_l[e0 + q + u];
          const int jj = q + u < n0 ? jr : N;       // row N is past the end of the buffer: nothing is fetched
          load_row_buf<CPL>(zrs, mad_u24((unsigned)jj, row_bytes, lane_off), z0[q + u]);
        }
      }
    }
    float mx = -INFINITY, s_last = -INFINITY;
    int j_last = N;
    for (int base = e0; base < e1; base += EPW) {
      const int e = base + le;
      const bool valid = e < e1;
      const int j = valid ? src_l[e] : N;
      const float sv = valid ? lrelu(as_l[j * H + lh] + adl, g.slope) : -INFINITY;
      mx = fmaxf(mx, sv);
      s_last = sv; j_last = j;
    }
    mx = strided_reduce<true>(mx, H);
    float den = 0.f, acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
    for (int base = e0; base < e1; base += EPW) {
      const int e = base + le;
      const bool valid = e < e1;
      float sv = s_last;
      int j = j_last;
      if (!single) {
        j = valid ? src_l[e] : N;
        sv = valid ? lrelu(as_l[j * H + lh] + adl, g.slope) : -INFINITY;
      }
      const float ex = valid ? __expf(sv - mx) : 0.f;
      den += ex;
      float w = ex;
      if (g.att_drop.p > 0.f && valid) w *= dropout_keep(g.att_drop, ((uint64_t)r * E + e) * H + lh);
      wave_lds_sync();
      wl[lane] = w;
      if (lh == 0) jl[le] = j;
      wave_lds_sync();
      const int n = min(EPW, e1 - base);
      const bool first = base == e0;
#pragma unroll
      for (int qb = 0; qb < EPW; qb += GB) {
        if (qb < n) {
          // rows of one batch are all requested before the first is used: one round trip per batch
          float zz[GB][CPL];
          if (!(qb == 0 && first)) {
#pragma unroll
            for (int q = 0; q < GB; q += 4) {
              if (qb + q < n) {            // slots n.. of the last group hold weight 0 and the out-of-range row
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  load_row_buf<CPL>(zrs, mad_u24((unsigned)jl[qb + q + u], row_bytes, lane_off), qb == 0 ? z0[q + u] : zz[q + u]);
              }
            }
          }
          asm volatile("" ::: "memory");   // keep the loads above the first use
#pragma unroll
          for (int q = 0; q < GB; q += 4) {
            if (qb + q < n) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float w4 = wl[(qb + q + u) * H + hk];
#pragma unroll
                for (int k = 0; k < CPL; ++k) acc[k] = fmaf(w4, qb == 0 ? z0[q + u][k] : zz[q + u][k], acc[k]);
              }
            }
          }
          // every load issued so far has been consumed; saying so keeps the compiler from guarding the next
          // batch's registers with waits on loads it cannot prove finished
          __builtin_amdgcn_s_waitcnt(kWaitVm0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);   // (no load is in flight here on any path; stated for the compiler as above)
    den = strided_reduce<false>(den, H);
    wave_lds_sync();
    if (le == 0) { smax[node * H + lh] = mx; sden[node * H + lh] = den; dl[lh] = den; }
    wave_lds_sync();
    const float inv = __builtin_amdgcn_rcpf(dl[hk] + 1e-16f);
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] *= inv;
    store_row<CPL>(agg + node * g.HC + c0, acc);     // kept: the backward pass takes the ReLU mask from it
    if (epi.y != nullptr) {
      const int od = epi.concat ? g.HC : Cw;
      float o[CPL];
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        float v = acc[k];
        if (!epi.concat) {
#pragma unroll
          for (int sh = 64 >> HLOG; sh < 64; sh <<= 1) v += __shfl_xor(v, sh);     // sum over the heads
          v *= 1.0f / (float)H;
        }
        v = fmaxf(v + bias_r[k], 0.f);
        const long idx = node * od + c0 + k;
        if (epi.fd.p > 0.f) v *= dropout_keep(epi.fd, (uint64_t)idx);
        o[k] = v;
      }
      if (epi.concat || hk == 0) {
        if (epi.xin != nullptr) {
          const float* xr = epi.xin + (epi.xin_rows > 0 ? (long)i : node) * od + c0;
#pragma unroll
          for (int k = 0; k < CPL; ++k) o[k] += xr[k];
        }
        store_row<CPL>(epi.y + node * od + c0, o);
      }
    }
  }
}

template <int CPL, int HLOG>
__global__ __launch_bounds__(kTileWaves * 64) void gat_bwd_source_tile_kernel(GatK g, int npb, int chunks, const float* __restrict__ Z,
                                                                              const float* __restrict__ a_s,
                                                                              const float* __restrict__ a_d,
                                                                              const float* __restrict__ smax,
                                                                              const float* __restrict__ sden,
                                                                              const float* __restrict__ dagg, const float* __restrict__ Dn,
                                                                              float* __restrict__ dZ, float* __restrict__ da_s,
                                                                              float* __restrict__ ds_edge) {
  constexpr int H = 1 << HLOG, EPW = 64 >> HLOG, LPH = 64 >> HLOG;
  constexpr int RB = CPL <= 2 ? 8 : 4, GB = EPW < RB ? EPW : RB;   // rows in flight per wave
  g.att_drop = dropout_resolve(g.att_drop);
  extern __shared__ float lds[];
  const int N = g.N, E = g.E, NH = N * H;
  float* as_l = lds;                                   // [N*H] source logits
  float* ad_l = as_l + NH;                             // [N*H] target logits
  float* mx_l = ad_l + NH;                             // [N*H] softmax maxima of the targets
  float* id_l = mx_l + NH;                             // [N*H] 1 / (softmax denominators + 1e-16)
  float* dn_l = id_l + NH;                             // [N*H] sum_c dAgg*agg of the targets
  int* dst_l = reinterpret_cast<int*>(dn_l + NH);      // [E]
  int* eid_l = dst_l + E;                              // [E]
  float* wsc = reinterpret_cast<float*>(eid_l + E);    // per wave: weights [64], edge dot products [64], row indices [64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int r, chunk;
  tile_block(blockIdx.x, g.R, chunks, r, chunk);
  const long rbase = (long)r * N;
  const long zbase = g.z_shared ? 0 : rbase;
  for (int t = threadIdx.x; t < NH; t += kTileWaves * 64) {
    as_l[t] = a_s[zbase * H + t];
    ad_l[t] = a_d[zbase * H + t];
    mx_l[t] = smax[rbase * H + t];
    id_l[t] = 1.0f / (sden[rbase * H + t] + 1e-16f);
    dn_l[t] = Dn[rbase * H + t];
  }
  for (int t = threadIdx.x; t < E; t += kTileWaves * 64) { dst_l[t] = g.dst_out[t]; eid_l[t] = g.eid_out[t]; }
  __syncthreads();
  float* wl = wsc + wave * 192;
  float* hl = wl + 64;
  int* il = reinterpret_cast<int*>(wl + 128);
  const int le = lane >> HLOG, lh = lane & (H - 1);
  const int hk = lane >> (6 - HLOG), c0 = lane * CPL;
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)g.HC * 4u), lane_off = (unsigned)c0 * 4u;
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dagg + rbase * g.HC), 0, N * (int)row_bytes, 0x00020000);
  const long eslot0 = (long)r * E * H + lh;
  const int j0 = chunk * npb, j1 = min(N, j0 + npb);
  const int jmine = min(j0 + wave + lane * kTileWaves, N - 1);
  const int rp0v = g.rp_out[jmine], rp1v = g.rp_out[jmine + 1];
  __builtin_amdgcn_s_waitcnt(kWaitVm0);   // as in the forward kernel
  int trip = 0;
  for (int j = j0 + wave; j < j1; j += kTileWaves, ++trip) {       // j is wave-uniform
    const long node = rbase + j;
    float z[CPL], dz[CPL];
    load_row<CPL>(Z + (zbase + j) * g.HC + c0, z);
#pragma unroll
    for (int k = 0; k < CPL; ++k) dz[k] = 0.f;
    const float asl = as_l[j * H + lh];
    float das = 0.f;
    const int e0 = __builtin_amdgcn_readlane(rp0v, trip), e1 = __builtin_amdgcn_readlane(rp1v, trip);
    for (int base = e0; base < e1; base += EPW) {
      const int e = base + le;
      const bool valid = e < e1;
      const int it = valid ? dst_l[e] : N;           // row N is past the end of the buffer: its load returns 0 untouched
      const int ih = (valid ? it : j) * H + lh;
      const float raw = asl + ad_l[ih];
      const float alpha = valid ? __expf(lrelu(raw, g.slope) - mx_l[ih]) * id_l[ih] : 0.f;
      const long eslot = valid ? eslot0 + (long)eid_l[e] * H : 0;   // this edge's slot in by-target order
      float keep = 1.f;
      if (g.att_drop.p > 0.f && valid) keep = dropout_keep(g.att_drop, (uint64_t)eslot);
      const float dnv = dn_l[ih];
      wave_lds_sync();
      wl[lane] = alpha * keep;
      if (lh == 0) il[le] = it;
      wave_lds_sync();
      const int n = min(EPW, e1 - base);
#pragma unroll
      for (int qb = 0; qb < EPW; qb += GB) {
        if (qb < n) {
          float dg[GB][CPL];
#pragma unroll
          for (int q = 0; q < GB; q += 4) {
            if (qb + q < n) {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                load_row_buf<CPL>(drs, mad_u24((unsigned)il[qb + q + u], row_bytes, lane_off), dg[q + u]);
            }
          }
          asm volatile("" ::: "memory");   // keep the loads above the first use
#pragma unroll
          for (int q = 0; q < GB; q += 4) {
            if (qb + q < n) {
              float pr[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float w4 = wl[(qb + q + u) * H + hk];
                float p = 0.f;
#pragma unroll
                for (int k = 0; k < CPL; ++k) { dz[k] = fmaf(w4, dg[q + u][k], dz[k]); p = fmaf(dg[q + u][k], z[k], p); }
                pr[u] = p;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) pr[u] = group_sum(pr[u], LPH);
              if ((lane & (LPH - 1)) == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) hl[(qb + q + u) * H + hk] = pr[u];
              }
            }
          }
          __builtin_amdgcn_s_waitcnt(kWaitVm0);   // as in the forward kernel: all of this batch's loads are consumed
        }
      }
      wave_lds_sync();
      if (valid) {
        const float de = alpha * (keep * hl[lane] - dnv);
        const float ds = de * (raw > 0.f ? 1.f : g.slope);
        das += ds;
        ds_edge[eslot] = ds;
      }
    }
    das = strided_reduce<false>(das, H);
    if (le == 0) da_s[node * H + lh] = das;
    store_row<CPL>(dZ + node * g.HC + c0, dz);
  }
}

// ---- node-streaming passes of the fast path ----------------------------------------------------------------
// A launch of kStreamBlocks blocks walks all (replica, node) rows, one row per wave per trip, lanes on channels
// as above.  Per-channel sums over rows (bias and attention-vector gradients) stay in registers for the whole
// walk and leave the block as one atomic per channel into one of kSumSlots copies (same-address atomics
// serialise at the memory side), which fold_slots_kernel adds into the gradient.
constexpr int kStreamBlocks = 2048;
constexpr int kSumSlots = 32;

template <int CPL>
__device__ __forceinline__ void block_channel_sums(const float (&v)[CPL], float* slots, int width, int c0, bool active, float* red) {
  // red: [kWavesPerBlock][64*CPL]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CPL; ++k) red[(wave * 64 + lane) * CPL + k] = v[k];
  __syncthreads();
  if (wave == 0 && active) {
    float* dst = slots + (size_t)(blockIdx.x % kSumSlots) * width;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) t += red[(w * 64 + lane) * CPL + k];
      unsafeAtomicAdd(dst + c0 + k, t);
    }
  }
  __syncthreads();
}

// up to three slot sets per launch (blockIdx.y): a layer's bias and two attention-vector gradients fold together
struct FoldSet { const float* slots; float* dst; int n; };
struct FoldBatch { FoldSet s[3]; };
__global__ void fold_slots_kernel(const FoldBatch b) {
  const FoldSet f = b.s[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= f.n) return;
  float t = 0.f;
#pragma unroll 8
  for (int q = 0; q < kSumSlots; ++q) t += f.slots[(size_t)q * f.n + c];
  f.dst[c] += t;
}

// a_s / a_d of every row (same result as gat_scores_kernel)
template <int CPL>
__global__ __launch_bounds__(256) void gat_scores_fast_kernel(GatK g, int hlog, const float* __restrict__ Z,
                                                              const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                              float* __restrict__ a_s, float* __restrict__ a_d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int LPH = 64 >> hlog, hk = lane >> (6 - hlog), c0 = lane * CPL;
  float as_[CPL], ad_[CPL];
  load_row<CPL>(att_src + c0, as_);
  load_row<CPL>(att_dst + c0, ad_);
  const long rows = g.z_shared ? (long)g.N : (long)g.R * g.N, stride = (long)gridDim.x * kWavesPerBlock;
  for (long node = (long)blockIdx.x * kWavesPerBlock + wave; node < rows; node += 2 * stride) {
    const long n2 = node + stride;
    const bool two = n2 < rows;
    float z0[CPL], z1[CPL];
    load_row<CPL>(Z + node * g.HC + c0, z0);
    load_row<CPL>(Z + (two ? n2 : node) * g.HC + c0, z1);
    float s0 = 0.f, d0 = 0.f, s1 = 0.f, d1 = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      s0 = fmaf(z0[k], as_[k], s0); d0 = fmaf(z0[k], ad_[k], d0);
      s1 = fmaf(z1[k], as_[k], s1); d1 = fmaf(z1[k], ad_[k], d1);
    }
    s0 = group_sum(s0, LPH); d0 = group_sum(d0, LPH); s1 = group_sum(s1, LPH); d1 = group_sum(d1, LPH);
    if ((lane & (LPH - 1)) == 0) {
      a_s[node * g.H + hk] = s0; a_d[node * g.H + hk] = d0;
      if (two) { a_s[n2 * g.H + hk] = s1; a_d[n2 * g.H + hk] = d1; }
    }
  }
}

// dAgg, Dn and the bias-gradient partial sums in one pass (gat_bwd_epilogue_kernel + gat_bwd_dot_kernel + colsum)
template <int CPL>
__global__ __launch_bounds__(256) void gat_bwd_prep_fast_kernel(GatK g, int hlog, const float* __restrict__ dy, int dy_bcast_N,
                                                                float dy_scale, const float* __restrict__ agg,
                                                                const float* __restrict__ bias, float* __restrict__ dagg,
                                                                float* __restrict__ Dn, float* __restrict__ bias_slots,
                                                                int concat, DropoutP fd) {
  fd = dropout_resolve(fd);
  fd = dropout_resolve(fd);
  __shared__ float red[kWavesPerBlock * 64 * CPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = g.H, C = g.C, LPH = 64 >> hlog, hk = lane >> (6 - hlog), c0 = lane * CPL;
  const int od = concat ? g.HC : C;
  const int oc0 = concat ? c0 : c0 - hk * C;          // this lane's first output channel
  const float invH = 1.0f / (float)H;
  float bs[CPL], gb[CPL];
  load_row<CPL>(bias + oc0, bs);
#pragma unroll
  for (int k = 0; k < CPL; ++k) gb[k] = 0.f;
  const long rows = (long)g.R * g.N, stride = (long)gridDim.x * kWavesPerBlock;
  for (long node = (long)blockIdx.x * kWavesPerBlock + wave; node < rows; node += stride) {
    float a[CPL], gr[CPL], v[CPL], da[CPL];
    load_row<CPL>(agg + node * g.HC + c0, a);
    const long drow = dy_bcast_N > 0 ? node / dy_bcast_N : node;
    load_row<CPL>(dy + drow * od + oc0, gr);
#pragma unroll
    for (int k = 0; k < CPL; ++k) v[k] = a[k];
    if (!concat) {
      for (int o = LPH; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) v[k] += __shfl_xor(v[k], o);
      }
#pragma unroll
      for (int k = 0; k < CPL; ++k) v[k] *= invH;
    }
    float dn = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      float gq = dy_bcast_N > 0 ? gr[k] * dy_scale : gr[k];
      if (!(v[k] + bs[k] > 0.f)) gq = 0.f;
      if (fd.p > 0.f) gq *= dropout_keep(fd, (uint64_t)(node * od + oc0 + k));
      if (concat || hk == 0) gb[k] += gq;
      da[k] = concat ? gq : gq * invH;
      dn = fmaf(da[k], a[k], dn);
    }
    store_row<CPL>(dagg + node * g.HC + c0, da);
    dn = group_sum(dn, LPH);
    if ((lane & (LPH - 1)) == 0) Dn[node * H + hk] = dn;
  }
  block_channel_sums<CPL>(gb, bias_slots, od, oc0, concat || hk == 0, red);
}

// dZ += da_s*att_src + da_d*att_dst, and the attention-vector gradient partial sums
// (gat_bwd_att_apply_kernel + gat_bwd_att_param_kernel)
template <int CPL>
__global__ __launch_bounds__(256) void gat_bwd_att_fast_kernel(GatK g, int hlog, const float* __restrict__ Z, float* __restrict__ dZ,
                                                               const float* __restrict__ da_s, const float* __restrict__ da_d,
                                                               const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                               float* __restrict__ src_slots, float* __restrict__ dst_slots) {
  __shared__ float red[kWavesPerBlock * 64 * CPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hk = lane >> (6 - hlog), c0 = lane * CPL;
  float as_[CPL], ad_[CPL], gs[CPL], gd[CPL];
  load_row<CPL>(att_src + c0, as_);
  load_row<CPL>(att_dst + c0, ad_);
#pragma unroll
  for (int k = 0; k < CPL; ++k) { gs[k] = 0.f; gd[k] = 0.f; }
  const long rows = (long)g.R * g.N, stride = (long)gridDim.x * kWavesPerBlock;
  for (long node = (long)blockIdx.x * kWavesPerBlock + wave; node < rows; node += stride) {
    float z[CPL], dz[CPL];
    load_row<CPL>(Z + (g.z_shared ? node % g.N : node) * g.HC + c0, z);
    load_row<CPL>(dZ + node * g.HC + c0, dz);
    const float s = da_s[node * g.H + hk], d = da_d[node * g.H + hk];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      dz[k] += s * as_[k] + d * ad_[k];
      gs[k] = fmaf(s, z[k], gs[k]);
      gd[k] = fmaf(d, z[k], gd[k]);
    }
    store_row<CPL>(dZ + node * g.HC + c0, dz);
  }
  block_channel_sums<CPL>(gs, src_slots, g.HC, c0, true, red);
  block_channel_sums<CPL>(gd, dst_slots, g.HC, c0, true, red);
}

// y = dropout(relu(headmix(agg) + bias)) (+ xin), headmix = identity (concat) or mean over heads
__global__ void gat_epilogue_kernel(const float* __restrict__ agg, const float* __restrict__ bias, const float* __restrict__ xin,
                                    int xin_rows, float* __restrict__ y, long rows, int H, int C, int concat, DropoutP fd) {
  fd = dropout_resolve(fd);
  const int od = concat ? H * C : C;
  const long n = rows * od;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / od;
    const int c = (int)(idx % od);
    float v;
    if (concat) v = agg[row * H * C + c];
    else {
      v = 0.f;
      for (int h = 0; h < H; ++h) v += agg[row * H * C + h * C + c];
      v /= (float)H;
    }
    v = fmaxf(v + bias[c], 0.f);
    if (fd.p > 0.f) v *= dropout_keep(fd, (uint64_t)idx);
    if (xin) v += xin_rows > 0 ? xin[(row % xin_rows) * od + c] : xin[idx];   // xin_rows > 0: one replica, read by all
    y[idx] = v;
  }
}

// dAgg = d(loss)/d(agg) from the gradient of the layer output (undo dropout, ReLU, head mixing)
// dy_bcast_N > 0: dy has one row per replica (mean-pool backward), scaled by dy_scale.
__global__ void gat_bwd_epilogue_kernel(const float* __restrict__ dy, int dy_bcast_N, float dy_scale, const float* __restrict__ agg,
                                        const float* __restrict__ bias, float* __restrict__ dagg, long rows, int H, int C,
                                        int concat, DropoutP fd) {
  fd = dropout_resolve(fd);
  const int od = concat ? H * C : C;
  const long n = rows * od;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / od;
    const int c = (int)(idx % od);
    float v;
    if (concat) v = agg[row * H * C + c];
    else {
      v = 0.f;
      for (int h = 0; h < H; ++h) v += agg[row * H * C + h * C + c];
      v /= (float)H;
    }
    float gr = dy_bcast_N > 0 ? dy[(row / dy_bcast_N) * od + c] * dy_scale : dy[idx];
    if (!(v + bias[c] > 0.f)) gr = 0.f;
    if (fd.p > 0.f) gr *= dropout_keep(fd, (uint64_t)idx);
    if (concat) dagg[row * H * C + c] = gr;
    else for (int h = 0; h < H; ++h) dagg[row * H * C + h * C + c] = gr / (float)H;
  }
}

// Dn[node,h] = sum_c dAgg[node,h,c] * agg[node,h,c]
template <int CPL, bool FAST>
__global__ __launch_bounds__(256) void gat_bwd_dot_kernel(GatK g, const float* __restrict__ dagg, const float* __restrict__ agg,
                                                          float* __restrict__ Dn) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wl = lds + wave * (g.HC + g.H);
  const long node = (long)blockIdx.x * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  const Lane<CPL> L(lane, g);
  float pr[CPL], hs[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) pr[k] = L.ok[k] ? dagg[node * g.HC + L.c0 + k] * agg[node * g.HC + L.c0 + k] : 0.f;
  head_sums<CPL, FAST>(pr, hs, L, g, lane, wl);
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (L.ok[k] && (L.c0 + k) % g.C == 0) Dn[node * g.H + L.hk[k]] = hs[k];
}

// By-source gather: dZ[j] = sum_{j->i} dropout(alpha_ji) dAgg[i];  da_s[j] = sum_i ds_ji;  da_d[i] += ds_ji (atomic)
template <int CPL, bool FAST>
__global__ __launch_bounds__(256) void gat_bwd_source_kernel(GatK g, const float* __restrict__ Z, const float* __restrict__ a_s,
                                                             const float* __restrict__ a_d, const float* __restrict__ smax,
                                                             const float* __restrict__ sden, const float* __restrict__ dagg,
                                                             const float* __restrict__ Dn, float* __restrict__ dZ,
                                                             float* __restrict__ da_s, float* __restrict__ da_d) {
  g.att_drop = dropout_resolve(g.att_drop);
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wl = lds + wave * (g.HC + g.H);
  const long node = (long)blockIdx.x * kWavesPerBlock + wave;
  if (node >= (long)g.R * g.N) return;
  const int r = (int)(node / g.N), j = (int)(node % g.N);
  const long rbase = (long)r * g.N;
  const Lane<CPL> L(lane, g);
  float z[CPL], asj[CPL], dz[CPL], das[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    z[k] = L.ok[k] ? Z[node * g.HC + L.c0 + k] : 0.f;
    asj[k] = L.ok[k] ? a_s[node * g.H + L.hk[k]] : 0.f;
    dz[k] = 0.f; das[k] = 0.f;
  }
  const int e0 = g.rp_out[j], e1 = g.rp_out[j + 1];
  for (int e = e0; e < e1; ++e) {
    const long ni = rbase + g.dst_out[e];
    const int eid = g.eid_out[e];
    float dg[CPL], pr[CPL], hs[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) { dg[k] = L.ok[k] ? dagg[ni * g.HC + L.c0 + k] : 0.f; pr[k] = dg[k] * z[k]; }
    head_sums<CPL, FAST>(pr, hs, L, g, lane, wl);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (!L.ok[k]) continue;
      const long nh = ni * g.H + L.hk[k];
      const float raw = asj[k] + a_d[nh];
      const float alpha = __expf(lrelu(raw, g.slope) - smax[nh]) / (sden[nh] + 1e-16f);
      float keep = 1.f;
      if (g.att_drop.p > 0.f) keep = dropout_keep(g.att_drop, ((uint64_t)r * g.E + eid) * g.H + L.hk[k]);
      dz[k] = fmaf(alpha * keep, dg[k], dz[k]);
      const float de = alpha * (keep * hs[k] - Dn[nh]);
      const float ds = de * (raw > 0.f ? 1.f : g.slope);
      das[k] += ds;
      if ((L.c0 + k) % g.C == 0) unsafeAtomicAdd(da_d + nh, ds);
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    if (!L.ok[k]) continue;
    dZ[node * g.HC + L.c0 + k] = dz[k];
    if ((L.c0 + k) % g.C == 0) da_s[node * g.H + L.hk[k]] = das[k];
  }
}

// dZ[row,c] += da_s[row,h]*att_src[c] + da_d[row,h]*att_dst[c]
__global__ void gat_bwd_att_apply_kernel(float* __restrict__ dZ, const float* __restrict__ da_s, const float* __restrict__ da_d,
                                         const float* __restrict__ att_src, const float* __restrict__ att_dst, long rows,
                                         int H, int C) {
  const int HC = H * C;
  const long n = rows * HC;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / HC;
    const int c = (int)(idx % HC), h = c / C;
    dZ[idx] += da_s[row * H + h] * att_src[c] + da_d[row * H + h] * att_dst[c];
  }
}

// datt_src[c] += sum_rows da_s[row,h(c)] * Z[row,c]; same for dst.   Threads along c, blocks over row chunks.
constexpr int kAttRows = 256;
__global__ void gat_bwd_att_param_kernel(const float* __restrict__ Z, const float* __restrict__ da_s, const float* __restrict__ da_d,
                                         float* __restrict__ datt_src, float* __restrict__ datt_dst, long rows, int H, int C) {
  const int HC = H * C;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= HC) return;
  const int h = c / C;
  const long r0 = (long)blockIdx.y * kAttRows, r1 = min(rows, r0 + kAttRows);
  float s = 0.f, d = 0.f;
  for (long r = r0; r < r1; ++r) {
    const float z = Z[r * HC + c];
    s = fmaf(da_s[r * H + h], z, s);
    d = fmaf(da_d[r * H + h], z, d);
  }
  unsafeAtomicAdd(datt_src + c, s);
  unsafeAtomicAdd(datt_dst + c, d);
}

// out[r*?] helpers -----------------------------------------------------------
__global__ void bcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, int R) {
  const long n = per * R;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i % per];
}
// dst[row, c] = src[row / N, c] * scale
__global__ void bcast_pool_grad_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int N, int od, float scale) {
  const long n = rows * od;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[(i / od / N) * od + (i % od)] * scale;
}
// dst[i] += sum over the replicas of chunk blockIdx.y of src[r*per + i]   (dst zeroed by the caller)
constexpr int kReplicaChunk = 32;
__global__ void sum_replicas_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, int R) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= per) return;
  const int r0 = blockIdx.y * kReplicaChunk, r1 = min(R, r0 + kReplicaChunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += src[(long)r * per + i]; s1 += src[(long)(r + 1) * per + i];
    s2 += src[(long)(r + 2) * per + i]; s3 += src[(long)(r + 3) * per + i];
  }
  for (; r < r1; ++r) s0 += src[(long)r * per + i];
  unsafeAtomicAdd(dst + i, (s0 + s1) + (s2 + s3));
}
inline int egrid(long n) {
  long g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ plan
struct GnnPlan {
  int N, R, G, D, H, C, layers;
  int pool_n, P;                   // nodes per pooled group, number of pooled rows (rows / pool_n)
  long rows;                       // R*N
  int in_dim[8], out_dim[8], concat[8], residual[8];
  size_t x0, xb, Z[8], as_[8], ad[8], smax[8], sden[8], agg[8], y[8], pooled, fwd_total;
  size_t dY, dZ, dagg, Dn, das, dad, dse, dpool, dx0, slots, bwd_total;
};

inline size_t al4(size_t x) { return (x + 3) & ~size_t(3); }

int make_plan(const ocean_gnn_shape* s, GnnPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->N <= 0 || s->R <= 0 || s->G <= 0 || s->D <= 0 || s->H <= 0 || s->layers < 1 || s->layers > 8 || s->E < s->N ||
      s->D % s->H != 0)
    return OCEAN_ERR_BAD_SHAPE;
  p->N = s->N; p->R = s->R; p->G = s->G; p->D = s->D; p->H = s->H; p->C = s->D / s->H; p->layers = s->layers;
  p->rows = (long)s->R * s->N;
  p->pool_n = s->pool_n > 0 ? s->pool_n : s->N;
  if (p->rows % p->pool_n != 0) return OCEAN_ERR_BAD_SHAPE;
  p->P = (int)(p->rows / p->pool_n);
  if (p->D > 64 * 8) return OCEAN_ERR_UNSUPPORTED;   // at most 8 channels per lane
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  const size_t rows = (size_t)p->rows;
  p->x0 = take((size_t)s->N * s->D);
  p->xb = take(s->R > 1 ? rows * s->D : 0);
  for (int l = 0; l < s->layers; ++l) {
    p->in_dim[l] = s->D;
    p->concat[l] = (l < s->layers - 1);
    p->out_dim[l] = p->concat[l] ? s->D : p->C;
    p->residual[l] = (p->out_dim[l] == p->in_dim[l]);
    p->Z[l] = take(rows * s->D);
    p->as_[l] = take(rows * s->H); p->ad[l] = take(rows * s->H);
    p->smax[l] = take(rows * s->H); p->sden[l] = take(rows * s->H);
    p->agg[l] = take(rows * s->D);
    p->y[l] = take(rows * p->out_dim[l]);
  }
  p->pooled = take((size_t)p->P * p->out_dim[s->layers - 1]);
  p->fwd_total = o;
  o = 0;
  p->dY = take(rows * s->D); p->dZ = take(rows * s->D); p->dagg = take(rows * s->D);
  p->Dn = take(rows * s->H); p->das = take(rows * s->H); p->dad = take(rows * s->H);
  p->dse = take((size_t)s->R * s->E * s->H);    // per-edge score gradients (fast path)
  p->dpool = take((size_t)p->P * p->out_dim[s->layers - 1]);
  p->dx0 = take((size_t)s->N * s->D);
  p->slots = take((size_t)3 * kSumSlots * s->D);   // bias, att_src, att_dst partial sums
  p->bwd_total = o;
  return OCEAN_OK;
}

GatK make_k(const ocean_gnn_shape* s, const GnnPlan& p, const int* const* graph, int layer) {
  GatK g;
  g.N = p.N; g.R = p.R; g.H = p.H; g.C = p.C; g.HC = p.D; g.E = s->E;
  g.rp_in = graph[0]; g.src_in = graph[1]; g.rp_out = graph[2]; g.dst_out = graph[3]; g.eid_out = graph[4];
  g.slope = s->negative_slope;
  g.att_drop.p = s->training ? s->dropout : 0.f;
  g.att_drop.seed = s->seed + 0x51ED270B7F4A7C15ull * (uint64_t)(2 * layer + 1);
  g.att_drop.epoch = s->seed_epoch;
  g.z_shared = 0;
  return g;
}
DropoutP feat_drop(const ocean_gnn_shape* s, int layer) {
  DropoutP d;
  d.p = s->training ? s->dropout : 0.f;
  d.seed = s->seed + 0x51ED270B7F4A7C15ull * (uint64_t)(2 * layer + 2);
  d.epoch = s->seed_epoch;
  return d;
}

inline bool fast_heads(const GnnPlan& p) {
  const int H = p.H;
  return p.D % 64 == 0 && H <= 64 && (H & (H - 1)) == 0;
}
inline int cpl_of(const GnnPlan& p) {
  const int c = (p.D + 63) / 64;
  return c <= 1 ? 1 : (c <= 2 ? 2 : (c <= 4 ? 4 : 8));
}

#define GAT_DISPATCH(KERNEL, ...)                                                                         \
  do {                                                                                                    \
    const int cpl__ = cpl_of(pl);                                                                         \
    const bool fast__ = fast_heads(pl) && cpl__ * 64 == pl.D;                                             \
    const dim3 grid__(ceil_div(pl.rows, kWavesPerBlock)), block__(256);                                   \
    const size_t sh__ = kWavesPerBlock * (pl.D + pl.H) * sizeof(float);                                   \
    if (fast__) {                                                                                         \
      if (cpl__ == 1) hipLaunchKernelGGL((KERNEL<1, true>), grid__, block__, sh__, st, __VA_ARGS__);      \
      else if (cpl__ == 2) hipLaunchKernelGGL((KERNEL<2, true>), grid__, block__, sh__, st, __VA_ARGS__); \
      else if (cpl__ == 4) hipLaunchKernelGGL((KERNEL<4, true>), grid__, block__, sh__, st, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<8, true>), grid__, block__, sh__, st, __VA_ARGS__);                 \
    } else {                                                                                              \
      if (cpl__ == 1) hipLaunchKernelGGL((KERNEL<1, false>), grid__, block__, sh__, st, __VA_ARGS__);     \
      else if (cpl__ == 2) hipLaunchKernelGGL((KERNEL<2, false>), grid__, block__, sh__, st, __VA_ARGS__);\
      else if (cpl__ == 4) hipLaunchKernelGGL((KERNEL<4, false>), grid__, block__, sh__, st, __VA_ARGS__);\
      else hipLaunchKernelGGL((KERNEL<8, false>), grid__, block__, sh__, st, __VA_ARGS__);                \
    }                                                                                                     \
    OCEAN_LAUNCH_CHECK();                                                                                 \
  } while (0)

inline bool fast_edges(const GnnPlan& p) {
  const bool off = getenv("OCEAN_GAT_GENERIC") != nullptr;   // test / measurement aid: force the one-edge-at-a-time kernels
  return !off && fast_heads(p) && cpl_of(p) * 64 == p.D && p.H <= 16;
}
inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

#define GAT_DISPATCH_FAST(KERNEL, ...)                                                                   \
  do {                                                                                                    \
    const int cpl__ = cpl_of(pl), hlog__ = ilog2(pl.H);                                                   \
    const dim3 grid__(ceil_div(pl.rows, kWavesPerBlock)), block__(256);                                   \
    if (cpl__ == 1) hipLaunchKernelGGL((KERNEL<1>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__);      \
    else if (cpl__ == 2) hipLaunchKernelGGL((KERNEL<2>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__); \
    else if (cpl__ == 4) hipLaunchKernelGGL((KERNEL<4>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<8>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__);                 \
    OCEAN_LAUNCH_CHECK();                                                                                 \
  } while (0)

#define GAT_STREAM_FAST(KERNEL, ...)                                                                     \
  do {                                                                                                    \
    const int cpl__ = cpl_of(pl), hlog__ = ilog2(pl.H);                                                   \
    const long nb__ = ceil_div(pl.rows, kWavesPerBlock);                                                  \
    const dim3 grid__((unsigned)(nb__ < kStreamBlocks ? nb__ : kStreamBlocks)), block__(256);             \
    if (cpl__ == 1) hipLaunchKernelGGL((KERNEL<1>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__);      \
    else if (cpl__ == 2) hipLaunchKernelGGL((KERNEL<2>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__); \
    else if (cpl__ == 4) hipLaunchKernelGGL((KERNEL<4>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<8>), grid__, block__, 0, st, g, hlog__, __VA_ARGS__);                 \
    OCEAN_LAUNCH_CHECK();                                                                                 \
  } while (0)

// replica-block kernels: LDS bytes of the forward / backward table sets, and whether they apply
inline size_t tile_lds_fwd(const GnnPlan& p, int E) {
  return sizeof(float) * ((size_t)2 * p.N * p.H + E + 8 + (size_t)kTileWaves * 144);
}
inline size_t tile_lds_bwd(const GnnPlan& p, int E) {
  return sizeof(float) * ((size_t)5 * p.N * p.H + 2 * (size_t)E + (size_t)kTileWaves * 192);
}
inline bool tile_edges(const GnnPlan& p, int E) {
  const bool off = getenv("OCEAN_GAT_NO_TILE") != nullptr;   // test / measurement aid: keep the one-node-per-wave kernels
  return !off && fast_edges(p) && tile_lds_bwd(p, E) <= 64 * 1024 && tile_lds_fwd(p, E) <= 64 * 1024;
}
// blocks: one per (replica, node range); node ranges only when there are too few replicas to fill the chip
inline int tile_chunks(const GnnPlan& p) {
  int c = (int)ceil_div(1024, p.R);
  // keep the replicas in flight on one XCD (64 resident blocks) within half of its 4 MB L2
  const long l2c = ceil_div((long)64 * p.N * p.D * 4, (long)2 << 20);
  if (l2c > c) c = (int)l2c;
  const int cmin = (int)ceil_div(p.N, 64 * kTileWaves);   // a wave keeps the edge ranges of its nodes in its 64 lanes
  const int cmax = (int)ceil_div(p.N, kTileWaves);
  c = c > cmax ? cmax : c;
  return c < cmin ? cmin : c;
}

#define GAT_TILE_HEADS(KERNEL, CPLV, LDS, ...)                                                                               \
  switch (hlog__) {                                                                                                          \
    case 0: hipLaunchKernelGGL((KERNEL<CPLV, 0>), grid__, block__, LDS, st, g, npb__, ch__, __VA_ARGS__); break;                   \
    case 1: hipLaunchKernelGGL((KERNEL<CPLV, 1>), grid__, block__, LDS, st, g, npb__, ch__, __VA_ARGS__); break;                   \
    case 2: hipLaunchKernelGGL((KERNEL<CPLV, 2>), grid__, block__, LDS, st, g, npb__, ch__, __VA_ARGS__); break;                   \
    case 3: hipLaunchKernelGGL((KERNEL<CPLV, 3>), grid__, block__, LDS, st, g, npb__, ch__, __VA_ARGS__); break;                   \
    default: hipLaunchKernelGGL((KERNEL<CPLV, 4>), grid__, block__, LDS, st, g, npb__, ch__, __VA_ARGS__); break;                  \
  }
#define GAT_DISPATCH_TILE(KERNEL, LDS, ...)                                                                   \
  do {                                                                                                         \
    const int cpl__ = cpl_of(pl), hlog__ = ilog2(pl.H), ch__ = tile_chunks(pl);                                \
    const int npb__ = (int)ceil_div(pl.N, ch__);                                                               \
    const dim3 grid__((unsigned)pl.R * ch__), block__(kTileWaves * 64);                                                   \
    if (cpl__ == 1) { GAT_TILE_HEADS(KERNEL, 1, LDS, __VA_ARGS__) }                                            \
    else if (cpl__ == 2) { GAT_TILE_HEADS(KERNEL, 2, LDS, __VA_ARGS__) }                                       \
    else if (cpl__ == 4) { GAT_TILE_HEADS(KERNEL, 4, LDS, __VA_ARGS__) }                                       \
    else { GAT_TILE_HEADS(KERNEL, 8, LDS, __VA_ARGS__) }                                                       \
    OCEAN_LAUNCH_CHECK();                                                                                      \
  } while (0)

// parameter table: in_w, in_b, then per layer (lin_w, att_src, att_dst, bias), then out_w, out_b
// y = headmix(agg) + bias: a GATConv layer's own output stage (no activation, residual or feature dropout)
__global__ void gat_plain_epilogue_kernel(const float* __restrict__ agg, const float* __restrict__ bias, float* __restrict__ y, long rows,
                                          int H, int C, int concat) {
  const int od = concat ? H * C : C;
  const long n = rows * od;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / od;
    const int c = (int)(idx % od);
    float v;
    if (concat) v = agg[row * H * C + c];
    else {
      v = 0.f;
      for (int h = 0; h < H; ++h) v += agg[row * H * C + h * C + c];
      v /= (float)H;
    }
    y[idx] = v + (bias ? bias[c] : 0.f);
  }
}
// dAgg from dy through that stage: dy itself (concat) or dy / H for each head (mean)
__global__ void gat_plain_bwd_epilogue_kernel(const float* __restrict__ dy, float* __restrict__ dagg, long rows, int H, int C, int concat) {
  const long n = rows * H * C;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / (H * C);
    const int hc = (int)(idx % (H * C));
    dagg[idx] = concat ? dy[idx] : dy[row * C + hc % C] / (float)H;
  }
}

struct GatConvPlan { GnnPlan pl; ocean_gnn_shape gs; size_t Z, as_, ad, smax, sden, agg, fwd_total, dagg, Dn, dZ, das, dad, bwd_total; };
int gatconv_plan(const ocean_gatconv_shape* s, GatConvPlan* p) {
  if (!s) return OCEAN_ERR_NULL;
  if (s->N <= 0 || s->in_dim <= 0 || s->H <= 0 || s->C <= 0 || s->E < s->N) return OCEAN_ERR_BAD_SHAPE;
  if (s->dropout < 0.f || s->dropout >= 1.f) return OCEAN_ERR_BAD_SHAPE;
  const int D = s->H * s->C;
  if (D > 64 * 8) return OCEAN_ERR_UNSUPPORTED;               // at most 8 channels per lane
  GnnPlan& pl = p->pl;
  pl = GnnPlan{};
  pl.N = s->N; pl.R = 1; pl.G = s->in_dim; pl.D = D; pl.H = s->H; pl.C = s->C; pl.layers = 1; pl.rows = s->N; pl.pool_n = s->N; pl.P = 1;
  ocean_gnn_shape& gs = p->gs;
  gs = ocean_gnn_shape{};
  gs.N = s->N; gs.R = 1; gs.G = s->in_dim; gs.D = D; gs.H = s->H; gs.layers = 1; gs.E = s->E; gs.pool_n = 0; gs.training = s->training;
  gs.dropout = s->dropout; gs.negative_slope = s->negative_slope; gs.seed = s->seed; gs.seed_epoch = s->seed_epoch;
  const size_t rows = (size_t)s->N;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o = al4(o + n); return r; };
  p->Z = take(rows * D); p->as_ = take(rows * s->H); p->ad = take(rows * s->H); p->smax = take(rows * s->H); p->sden = take(rows * s->H);
  p->agg = take(rows * D);
  p->fwd_total = o;
  o = 0;
  p->dagg = take(rows * D); p->Dn = take(rows * s->H); p->dZ = take(rows * D); p->das = take(rows * s->H); p->dad = take(rows * s->H);
  p->bwd_total = o;
  return OCEAN_OK;
}

inline int pg_layer(int l, int k) { return 2 + 4 * l + k; }
inline int pg_out(int layers, int k) { return 2 + 4 * layers + k; }

}  // namespace

extern "C" {

size_t ocean_gnn_fwd_workspace(const ocean_gnn_shape* s) {
  GnnPlan p;
  return make_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0;
}
size_t ocean_gnn_bwd_workspace(const ocean_gnn_shape* s) {
  GnnPlan p;
  return make_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0;
}

int ocean_gnn_layer_output(const ocean_gnn_shape* s, int layer, size_t* offset_floats, int* width) {
  GnnPlan p;
  OCEAN_TRY(make_plan(s, &p));
  if (layer < 0 || layer >= p.layers) return OCEAN_ERR_BAD_SHAPE;
  if (offset_floats) *offset_floats = p.y[layer];
  if (width) *width = p.out_dim[layer];
  return OCEAN_OK;
}

int ocean_gnn_edge_kernel_family(const ocean_gnn_shape* s) {
  GnnPlan p;
  if (make_plan(s, &p) != OCEAN_OK) return -1;
  return tile_edges(p, s->E) ? 2 : (fast_edges(p) ? 1 : 0);
}

int ocean_gnn_fwd(const ocean_gnn_shape* s, const float* node_features, const int* const* graph, float* const* params,
                  float* ws, size_t ws_floats, float* out, void* stream) {
  GnnPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!node_features || !graph || !params || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < pl.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int D = pl.D, H = pl.H, C = pl.C;
  const long rows = pl.rows;
  OCEAN_TRY(linear_fwd(node_features, s->G, params[0], params[1], ws + pl.x0, D, s->N, D, s->G, 1, st));
  // Every replica starts from the same projected node features, so the first layer's linear transform and
  // attention logits are computed once and shared (fast path); otherwise the features are replicated first.
  const bool share0 = s->R > 1 && fast_edges(pl);
  const float* xin = ws + pl.x0;
  if (s->R > 1 && !share0) {
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(egrid(rows * D)), dim3(256), 0, st, ws + pl.x0, ws + pl.xb, (long)s->N * D, s->R);
    OCEAN_LAUNCH_CHECK();
    xin = ws + pl.xb;
  }
  for (int l = 0; l < pl.layers; ++l) {
    GatK g = make_k(s, pl, graph, l);
    const bool shared = share0 && l == 0;
    g.z_shared = shared;
    const int zrows = shared ? s->N : (int)rows;
    OCEAN_TRY(linear_fwd(xin, D, params[pg_layer(l, 0)], nullptr, ws + pl.Z[l], D, zrows, D, D, 0, st));
    if (fast_edges(pl))
      GAT_STREAM_FAST(gat_scores_fast_kernel, ws + pl.Z[l], params[pg_layer(l, 1)], params[pg_layer(l, 2)], ws + pl.as_[l], ws + pl.ad[l]);
    else
      GAT_DISPATCH(gat_scores_kernel, g, ws + pl.Z[l], params[pg_layer(l, 1)], params[pg_layer(l, 2)], ws + pl.as_[l], ws + pl.ad[l]);
    if (tile_edges(pl, s->E)) {
      // the output stage rides along: the aggregated row is still in registers when it is stored
      const bool fused = getenv("OCEAN_GAT_NO_FUSED_EPI") == nullptr;       // measurement switch, read per call
      const GatEpi epi{params[pg_layer(l, 3)], pl.residual[l] ? xin : nullptr, fused ? ws + pl.y[l] : nullptr, shared ? s->N : 0, pl.concat[l], feat_drop(s, l)};
      GAT_DISPATCH_TILE(gat_aggregate_tile_kernel, tile_lds_fwd(pl, s->E), ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l],
                        ws + pl.smax[l], ws + pl.sden[l], epi);
      if (fused) { xin = ws + pl.y[l]; continue; }
    } else if (fast_edges(pl)) {
      const bool fused = getenv("OCEAN_GAT_NO_FUSED_EPI") == nullptr;       // measurement switch, read per call
      const GatEpi epi{params[pg_layer(l, 3)], pl.residual[l] ? xin : nullptr, fused ? ws + pl.y[l] : nullptr, shared ? s->N : 0, pl.concat[l], feat_drop(s, l)};
      GAT_DISPATCH_FAST(gat_aggregate_fast_kernel, ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l], ws + pl.smax[l], ws + pl.sden[l], epi);
      if (fused) { xin = ws + pl.y[l]; continue; }
    } else {
      const int cpl = cpl_of(pl);
      const dim3 grid(ceil_div(rows, kWavesPerBlock)), block(256);
      if (cpl == 1) hipLaunchKernelGGL(gat_aggregate_kernel<1>, grid, block, 0, st, g, ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l], ws + pl.smax[l], ws + pl.sden[l]);
      else if (cpl == 2) hipLaunchKernelGGL(gat_aggregate_kernel<2>, grid, block, 0, st, g, ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l], ws + pl.smax[l], ws + pl.sden[l]);
      else if (cpl == 4) hipLaunchKernelGGL(gat_aggregate_kernel<4>, grid, block, 0, st, g, ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l], ws + pl.smax[l], ws + pl.sden[l]);
      else hipLaunchKernelGGL(gat_aggregate_kernel<8>, grid, block, 0, st, g, ws + pl.Z[l], ws + pl.as_[l], ws + pl.ad[l], ws + pl.agg[l], ws + pl.smax[l], ws + pl.sden[l]);
      OCEAN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gat_epilogue_kernel, dim3(egrid(rows * pl.out_dim[l])), dim3(256), 0, st, ws + pl.agg[l], params[pg_layer(l, 3)],
                       pl.residual[l] ? xin : nullptr, shared ? s->N : 0, ws + pl.y[l], rows, H, C, pl.concat[l], feat_drop(s, l));
    OCEAN_LAUNCH_CHECK();
    xin = ws + pl.y[l];
  }
  const int Cl = pl.out_dim[pl.layers - 1];
  OCEAN_TRY(mean_mid(xin, ws + pl.pooled, pl.P, pl.pool_n, Cl, st));
  OCEAN_TRY(linear_fwd(ws + pl.pooled, Cl, params[pg_out(pl.layers, 0)], params[pg_out(pl.layers, 1)], out, D, pl.P, D, Cl, 0, st));
  return OCEAN_OK;
}

int ocean_gnn_bwd(const ocean_gnn_shape* s, const float* node_features, const int* const* graph, float* const* params,
                  float* wf, float* wb, size_t wb_floats, const float* dout, float* const* grads, void* stream) {
  const OceanBackwardScope scope;
  GnnPlan pl;
  OCEAN_TRY(make_plan(s, &pl));
  if (!node_features || !graph || !params || !wf || !wb || !dout || !grads) return OCEAN_ERR_NULL;
  if (wb_floats < pl.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int D = pl.D, H = pl.H, C = pl.C, Lr = pl.layers;
  const long rows = pl.rows;
  const int Cl = pl.out_dim[Lr - 1];
  // output projection
  // weight, bias (column sums of the same operand) and data gradients of the output projection
  OCEAN_TRY(linear_bwd_pair(wf + pl.pooled, Cl, dout, D, params[pg_out(Lr, 0)], grads[pg_out(Lr, 0)], grads[pg_out(Lr, 1)], wb + pl.dpool, Cl, pl.P, D, Cl, 0, st));
  const bool share0 = s->R > 1 && fast_edges(pl);
  float* dx0 = wb + pl.dx0;
  for (int l = Lr - 1; l >= 0; --l) {
    GatK g = make_k(s, pl, graph, l);
    const bool shared = share0 && l == 0;
    g.z_shared = shared;
    const float* xin = l == 0 ? (s->R > 1 && !share0 ? wf + pl.xb : wf + pl.x0) : wf + pl.y[l - 1];
    const bool last = (l == Lr - 1);
    const int od = pl.out_dim[l];
    const bool fast = fast_edges(pl);
    float* slot_b = wb + pl.slots;
    float* slot_s = slot_b + (size_t)kSumSlots * D;
    float* slot_d = slot_s + (size_t)kSumSlots * D;
    if (fast) {
      OCEAN_TRY(zero_f32(slot_b, (size_t)3 * kSumSlots * D, st));
      GAT_STREAM_FAST(gat_bwd_prep_fast_kernel, last ? wb + pl.dpool : wb + pl.dY, last ? pl.pool_n : 0,
                      last ? 1.0f / (float)pl.pool_n : 1.0f, wf + pl.agg[l], params[pg_layer(l, 3)], wb + pl.dagg, wb + pl.Dn,
                      slot_b, pl.concat[l], feat_drop(s, l));
    } else {
      hipLaunchKernelGGL(gat_bwd_epilogue_kernel, dim3(egrid(rows * od)), dim3(256), 0, st, last ? wb + pl.dpool : wb + pl.dY,
                         last ? pl.pool_n : 0, last ? 1.0f / (float)pl.pool_n : 1.0f, wf + pl.agg[l], params[pg_layer(l, 3)], wb + pl.dagg, rows,
                         H, C, pl.concat[l], feat_drop(s, l));
      OCEAN_LAUNCH_CHECK();
      // bias gradient: column sums of g; for head-averaged output the H slices of dAgg each hold g/H
      if (pl.concat[l]) OCEAN_TRY(colsum_add(wb + pl.dagg, D, (int)rows, D, grads[pg_layer(l, 3)], st));
      else for (int h = 0; h < H; ++h) OCEAN_TRY(colsum_add(wb + pl.dagg + h * C, D, (int)rows, C, grads[pg_layer(l, 3)], st));
      GAT_DISPATCH(gat_bwd_dot_kernel, g, wb + pl.dagg, wf + pl.agg[l], wb + pl.Dn);
    }
    if (fast_edges(pl)) {
      if (tile_edges(pl, s->E))
        GAT_DISPATCH_TILE(gat_bwd_source_tile_kernel, tile_lds_bwd(pl, s->E), wf + pl.Z[l], wf + pl.as_[l], wf + pl.ad[l], wf + pl.smax[l],
                          wf + pl.sden[l], wb + pl.dagg, wb + pl.Dn, wb + pl.dZ, wb + pl.das, wb + pl.dse);
      else
        GAT_DISPATCH_FAST(gat_bwd_source_fast_kernel, wf + pl.Z[l], wf + pl.as_[l], wf + pl.ad[l], wf + pl.smax[l], wf + pl.sden[l],
                          wb + pl.dagg, wb + pl.Dn, wb + pl.dZ, wb + pl.das, wb + pl.dse);
      hipLaunchKernelGGL(gat_bwd_target_sum_kernel, dim3(egrid(rows * H)), dim3(256), 0, st, g, wb + pl.dse, wb + pl.dad);
      OCEAN_LAUNCH_CHECK();
    } else {
      OCEAN_TRY(zero_f32(wb + pl.dad, rows * H, st));
      GAT_DISPATCH(gat_bwd_source_kernel, g, wf + pl.Z[l], wf + pl.as_[l], wf + pl.ad[l], wf + pl.smax[l], wf + pl.sden[l],
                   wb + pl.dagg, wb + pl.Dn, wb + pl.dZ, wb + pl.das, wb + pl.dad);
    }
    if (fast) {
      GAT_STREAM_FAST(gat_bwd_att_fast_kernel, wf + pl.Z[l], wb + pl.dZ, wb + pl.das, wb + pl.dad, params[pg_layer(l, 1)],
                      params[pg_layer(l, 2)], slot_s, slot_d);
      const FoldBatch fb{{{slot_b, grads[pg_layer(l, 3)], od}, {slot_s, grads[pg_layer(l, 1)], D}, {slot_d, grads[pg_layer(l, 2)], D}}};
      hipLaunchKernelGGL(fold_slots_kernel, dim3(ceil_div(D, 64), 3), dim3(64), 0, st, fb);
      OCEAN_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(gat_bwd_att_param_kernel, dim3(ceil_div(D, 64), ceil_div(rows, kAttRows)), dim3(64), 0, st, wf + pl.Z[l],
                         wb + pl.das, wb + pl.dad, grads[pg_layer(l, 1)], grads[pg_layer(l, 2)], rows, H, C);
      OCEAN_LAUNCH_CHECK();
      hipLaunchKernelGGL(gat_bwd_att_apply_kernel, dim3(egrid(rows * D)), dim3(256), 0, st, wb + pl.dZ, wb + pl.das, wb + pl.dad,
                         params[pg_layer(l, 1)], params[pg_layer(l, 2)], rows, H, C);
      OCEAN_LAUNCH_CHECK();
    }
    if (shared) {
      // one transform for all replicas: its gradients come from the replica sum of dZ, and the layer's input
      // gradient is that sum times W plus (residual) the replica sum of the incoming output gradient
      float* dzs = wb + pl.dagg;               // [N, D] scratch: dAgg is no longer needed at this point
      const long per = (long)s->N * D;
      const dim3 rg(ceil_div(per, 256), ceil_div(s->R, kReplicaChunk));
      OCEAN_TRY(zero_f32(dzs, (size_t)per, st));
      hipLaunchKernelGGL(sum_replicas_kernel, rg, dim3(256), 0, st, wb + pl.dZ, dzs, per, s->R);
      OCEAN_LAUNCH_CHECK();
      OCEAN_TRY(linear_bwd_weight(dzs, D, xin, D, grads[pg_layer(l, 0)], s->N, D, D, st));
      OCEAN_TRY(zero_f32(dx0, (size_t)per, st));
      if (pl.residual[l]) {
        if (last) {   // a one-layer network: the incoming gradient is the mean-pool broadcast, materialised first
          hipLaunchKernelGGL(bcast_pool_grad_kernel, dim3(egrid(rows * od)), dim3(256), 0, st, wb + pl.dpool, wb + pl.dY, rows, pl.pool_n, od,
                             1.0f / (float)pl.pool_n);
          OCEAN_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(sum_replicas_kernel, rg, dim3(256), 0, st, wb + pl.dY, dx0, per, s->R);
        OCEAN_LAUNCH_CHECK();
      }
      OCEAN_TRY(linear_bwd_data(dzs, D, params[pg_layer(l, 0)], dx0, D, s->N, D, D, 1, st));
      continue;
    }
    OCEAN_TRY(linear_bwd_weight(wb + pl.dZ, D, xin, D, grads[pg_layer(l, 0)], (int)rows, D, D, st));
    // d(xin) = dZ . W (+ dY through the residual); written over dY for the next (earlier) layer
    if (pl.residual[l] && !last) {
      OCEAN_TRY(linear_bwd_data(wb + pl.dZ, D, params[pg_layer(l, 0)], wb + pl.dY, D, (int)rows, D, D, 1, st));
    } else if (pl.residual[l] && last) {
      // residual of the pooled layer (only when H == 1): materialise the broadcast mean-pool gradient, then accumulate
      hipLaunchKernelGGL(bcast_pool_grad_kernel, dim3(egrid(rows * od)), dim3(256), 0, st, wb + pl.dpool, wb + pl.dY, rows, pl.pool_n, od,
                         1.0f / (float)pl.pool_n);
      OCEAN_LAUNCH_CHECK();
      OCEAN_TRY(linear_bwd_data(wb + pl.dZ, D, params[pg_layer(l, 0)], wb + pl.dY, D, (int)rows, D, D, 1, st));
    } else {
      OCEAN_TRY(linear_bwd_data(wb + pl.dZ, D, params[pg_layer(l, 0)], wb + pl.dY, D, (int)rows, D, D, 0, st));
    }
  }
  // input projection: x0 = relu(W nf + b), shared by all replicas
  if (share0) {
    // dx0 already holds the replica-summed gradient
  } else if (s->R > 1) {
    OCEAN_TRY(zero_f32(dx0, (size_t)s->N * D, st));
    hipLaunchKernelGGL(sum_replicas_kernel, dim3(ceil_div((long)s->N * D, 256), ceil_div(s->R, kReplicaChunk)), dim3(256), 0, st,
                       wb + pl.dY, dx0, (long)s->N * D, s->R);
    OCEAN_LAUNCH_CHECK();
  } else {
    OCEAN_TRY(copy2d(dx0, D, wb + pl.dY, D, s->N, D, 0, st));
  }
  OCEAN_TRY(relu_bwd_inplace(dx0, D, wf + pl.x0, D, s->N, D, st));
  OCEAN_TRY(linear_bwd_weight(dx0, D, node_features, s->G, grads[0], s->N, D, s->G, st, grads[1]));   // + bias gradient (column sums of the same operand)
  return OCEAN_OK;
}

// ---- one GATConv layer called on its own (the one-node-per-wave kernel family; dropout acts on the attention
// coefficients only, as in torch_geometric's layer).  params / grads: {lin.weight [H*C, in], att_src [H*C], att_dst [H*C],
// bias [H*C or C] or NULL}.
size_t ocean_gatconv_fwd_workspace(const ocean_gatconv_shape* s) { GatConvPlan p; return gatconv_plan(s, &p) == OCEAN_OK ? p.fwd_total : 0; }
size_t ocean_gatconv_bwd_workspace(const ocean_gatconv_shape* s) { GatConvPlan p; return gatconv_plan(s, &p) == OCEAN_OK ? p.bwd_total : 0; }

int ocean_gatconv_fwd(const ocean_gatconv_shape* s, const float* x, const int* const* graph, float* const* params, float* ws,
                      size_t ws_floats, float* out, void* stream) {
  GatConvPlan cp;
  OCEAN_TRY(gatconv_plan(s, &cp));
  if (!x || !graph || !params || !ws || !out) return OCEAN_ERR_NULL;
  if (ws_floats < cp.fwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const GnnPlan& pl = cp.pl;
  const int D = pl.D, H = pl.H, C = pl.C;
  const long rows = pl.rows;
  GatK g = make_k(&cp.gs, pl, graph, 0);
  OCEAN_TRY(linear_fwd(x, s->in_dim, params[0], nullptr, ws + cp.Z, D, (int)rows, D, s->in_dim, 0, st));
  GAT_DISPATCH(gat_scores_kernel, g, ws + cp.Z, params[1], params[2], ws + cp.as_, ws + cp.ad);
  {
    const int cpl = cpl_of(pl);
    const dim3 grid(ceil_div(rows, kWavesPerBlock)), block(256);
    if (cpl == 1) hipLaunchKernelGGL(gat_aggregate_kernel<1>, grid, block, 0, st, g, ws + cp.Z, ws + cp.as_, ws + cp.ad, ws + cp.agg, ws + cp.smax, ws + cp.sden);
    else if (cpl == 2) hipLaunchKernelGGL(gat_aggregate_kernel<2>, grid, block, 0, st, g, ws + cp.Z, ws + cp.as_, ws + cp.ad, ws + cp.agg, ws + cp.smax, ws + cp.sden);
    else if (cpl == 4) hipLaunchKernelGGL(gat_aggregate_kernel<4>, grid, block, 0, st, g, ws + cp.Z, ws + cp.as_, ws + cp.ad, ws + cp.agg, ws + cp.smax, ws + cp.sden);
    else hipLaunchKernelGGL(gat_aggregate_kernel<8>, grid, block, 0, st, g, ws + cp.Z, ws + cp.as_, ws + cp.ad, ws + cp.agg, ws + cp.smax, ws + cp.sden);
    OCEAN_LAUNCH_CHECK();
  }
  const int od = s->concat ? D : C;
  hipLaunchKernelGGL(gat_plain_epilogue_kernel, dim3(egrid(rows * od)), dim3(256), 0, st, ws + cp.agg, params[3], out, rows, H, C, s->concat);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_gatconv_bwd(const ocean_gatconv_shape* s, const float* x, const int* const* graph, float* const* params, float* wf,
                      const float* dy, float* wb, size_t wb_floats, float* const* grads, float* dx, void* stream) {
  GatConvPlan cp;
  OCEAN_TRY(gatconv_plan(s, &cp));
  if (!x || !graph || !params || !wf || !dy || !wb || !grads) return OCEAN_ERR_NULL;
  if (!grads[0] || !grads[1] || !grads[2]) return OCEAN_ERR_NULL;
  if (wb_floats < cp.bwd_total) return OCEAN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const GnnPlan& pl = cp.pl;
  const int D = pl.D, H = pl.H, C = pl.C;
  const long rows = pl.rows;
  const int od = s->concat ? D : C;
  GatK g = make_k(&cp.gs, pl, graph, 0);
  hipLaunchKernelGGL(gat_plain_bwd_epilogue_kernel, dim3(egrid(rows * D)), dim3(256), 0, st, dy, wb + cp.dagg, rows, H, C, s->concat);
  OCEAN_LAUNCH_CHECK();
  if (grads[3]) OCEAN_TRY(colsum_add(dy, od, (int)rows, od, grads[3], st));
  GAT_DISPATCH(gat_bwd_dot_kernel, g, wb + cp.dagg, wf + cp.agg, wb + cp.Dn);
  OCEAN_TRY(zero_f32(wb + cp.dad, rows * H, st));
  GAT_DISPATCH(gat_bwd_source_kernel, g, wf + cp.Z, wf + cp.as_, wf + cp.ad, wf + cp.smax, wf + cp.sden, wb + cp.dagg, wb + cp.Dn, wb + cp.dZ,
               wb + cp.das, wb + cp.dad);
  hipLaunchKernelGGL(gat_bwd_att_param_kernel, dim3(ceil_div(D, 64), ceil_div(rows, kAttRows)), dim3(64), 0, st, wf + cp.Z, wb + cp.das,
                     wb + cp.dad, grads[1], grads[2], rows, H, C);
  OCEAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(gat_bwd_att_apply_kernel, dim3(egrid(rows * D)), dim3(256), 0, st, wb + cp.dZ, wb + cp.das, wb + cp.dad, params[1],
                     params[2], rows, H, C);
  OCEAN_LAUNCH_CHECK();
  OCEAN_TRY(linear_bwd_weight(wb + cp.dZ, D, x, s->in_dim, grads[0], (int)rows, D, s->in_dim, st));
  if (dx) OCEAN_TRY(linear_bwd_data(wb + cp.dZ, D, params[0], dx, s->in_dim, (int)rows, D, s->in_dim, 0, st));
  return OCEAN_OK;
}

}  // extern "C"
