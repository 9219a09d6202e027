// fp32 GEMM for outputs of a few dozen 32x32 tiles (the attention / fusion / head layers: a few hundred to a few
// thousand rows, 64-768 columns, reductions of 64-768).
//
// With so few output tiles the ring-pipelined kernel of gemm.hip runs on a handful of compute units, and each of its
// waves walks the whole reduction as one chain of dependent MFMAs (64 cycles apiece): ~10 us per launch whatever the
// size.  Here a block owns ONE 32x32 output tile and its four waves split the reduction four ways, so four times as
// many compute units work and each chain is a quarter as long; the partial tiles meet in LDS, and the epilogue (bias,
// residual, ReLU, dropout, ReLU-backward mask) runs on the sum.  Operands go from memory straight into MFMA fragment
// registers - lane (m, h) of a wave loads four consecutive k of row m (offset 4h inside each 8-deep step), which pairs
// k and k+4 in one 32x32x2 MFMA on both operands alike - so there is no staging through LDS and every load of a wave's
// slice can be in flight at once.
#include "common.h"
#include <cstdlib>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;          // 8-deep reduction steps loaded together (4 x 2 x float4 = 32 registers)

template <bool BKN>
__global__ __launch_bounds__(kThreads) void gemm_skinny_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[blockIdx.y];
  const int tiles_n = (p.N + 31) / 32, tiles_m = (p.M + 31) / 32;
  if ((int)blockIdx.x >= tiles_m * tiles_n) return;              // the grid is sized for the largest problem of the batch
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, h = lane >> 5;

  // this wave's share of the reduction, in whole 8-deep steps
  const int steps = (p.K + 7) / 8;
  const int per = (steps + 3) / 4;
  const int s_begin = wave * per, s_end = min(steps, s_begin + per);

  constexpr unsigned kOob = 0x80000000u;                         // out-of-range offset: a buffer load returns zeros
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7FFFFFF0, 0x00020000);
  const bool a_row_ok = m0 + lm < p.M, b_col_ok = n0 + lm < p.N;
  const unsigned a_base = (unsigned)((m0 + lm) * p.lda + 4 * h) * 4u;
  // B as [N][K] (nn.Linear weight): row n0 + lm, four consecutive k; B as [K][N]: column n0 + lm of four consecutive rows
  const unsigned b_base = BKN ? (unsigned)((4 * h) * p.ldb + n0 + lm) * 4u : (unsigned)((n0 + lm) * p.ldb + 4 * h) * 4u;
  const unsigned b_row = (unsigned)p.ldb * 4u;

  auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, float (&v)[4]) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
    v[0] = __uint_as_float(x); v[1] = __uint_as_float(y); v[2] = __uint_as_float(z); v[3] = __uint_as_float(w);
  };
  auto load_step = [&](int s, float (&a)[4], float (&b)[4]) {
    const int k = 8 * s + 4 * h;                                  // K % 4 == 0: a group of four is inside or outside
    const bool k_ok = s < s_end && k < p.K;
    ld4(rsrc_a, (a_row_ok && k_ok) ? a_base + (unsigned)(8 * s) * 4u : kOob, a);
    if (BKN) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        b[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            rsrc_b, (int)((b_col_ok && k_ok) ? b_base + (unsigned)(8 * s + t) * b_row : kOob), 0, 0));
    } else {
      ld4(rsrc_b, (b_col_ok && k_ok) ? b_base + (unsigned)(8 * s) * 4u : kOob, b);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a[2][kUnroll][4], b[2][kUnroll][4];
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) load_step(s_begin + u, a[0][u], b[0][u]);
  int cur = 0;
  for (int s = s_begin; s < s_end; s += kUnroll) {
    // the next group is requested before this one is multiplied
    if (cur == 0) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) load_step(s + kUnroll + u, a[1][u], b[1][u]);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][u][t], b[0][u][t], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) load_step(s + kUnroll + u, a[0][u], b[0][u]);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][u][t], b[1][u][t], acc, 0, 0, 0);
    }
    cur ^= 1;
  }

  // partial tiles -> LDS; acc register r of lane (lm, h) is C[(r&3) + 8*(r>>2) + 4*h][lm]
  __shared__ float part[4][32][33];
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][lm] = acc[r];
  __syncthreads();

  const DropoutP drop = dropout_resolve(p.drop);
  const int col = n0 + (tid & 31);
  const bool col_ok = col < p.N;
  const float bias = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = (tid >> 5) + 8 * i, row = m0 + lr;
    if (row >= p.M || !col_ok) continue;
    float v = part[0][lr][tid & 31] + part[1][lr][tid & 31] + part[2][lr][tid & 31] + part[3][lr][tid & 31] + bias;
    if (p.R != nullptr) v += p.R[(long)row * p.ldr + col];
    if (p.relu) v = fmaxf(v, 0.f);
    if (drop.p > 0.f) {
      const uint64_t di = p.drop_group > 1 ? (uint64_t)(p.drop_row0 + row) * (p.N / p.drop_group) + col / p.drop_group
                                           : (uint64_t)(p.drop_row0 + row) * p.N + col;
      v *= dropout_keep(drop, di);
    }
    if (p.relu_mask != nullptr && !(p.relu_mask[(long)row * p.ld_mask + col] > 0.f)) v = 0.f;
    p.C[(long)row * p.ldc + col] = v;
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Whether the batch should take this kernel: plain products (one tap, no row reduction, no column statistics) whose
// 64x64 tiling would occupy less than one block per compute unit, with operands the vector loads can address.
bool ocean_gemm_skinny_applies(const GemmBatch& b) {
  static const bool off = getenv("OCEAN_GEMM_NO_SKINNY") != nullptr;       // measurement switch
  if (off) return false;
  long blocks64 = 0, tiles32 = 0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    if (p.a_kmajor || p.taps != 1 || p.colstats != nullptr || p.bn_u != nullptr || p.a_colsum != nullptr || p.r_bcast_T > 0) return false;
    if (p.b_kn != b.g[0].b_kn) return false;
    if (p.K % 4 != 0 || p.K < 32 || !al16(p.A) || p.lda % 4 != 0) return false;
    if (!p.b_kn && (!al16(p.B) || p.ldb % 4 != 0)) return false;
    if ((double)p.M * p.lda >= 5.0e8 || (double)(p.b_kn ? p.K : p.N) * p.ldb >= 5.0e8) return false;   // 32-bit byte offsets
    blocks64 += (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
    tiles32 += (long)ceil_div(p.M, 32) * ceil_div(p.N, 32);
  }
  return blocks64 < 256 && tiles32 <= 2048;
}

int ocean_launch_gemm_skinny(const GemmBatch& b, hipStream_t stream) {
  int tiles = 0;
  double flops = 0.0, bytes = 0.0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    const int t = ceil_div(p.M, 32) * ceil_div(p.N, 32);
    if (t > tiles) tiles = t;
    flops += 2.0 * p.M * p.N * (double)p.K;
    bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
  }
  void* tok = nullptr;
  if (ocean_prof_on()) ocean_prof_begin(b.g[0].b_kn ? "gemm_skinny_kernel<true>" : "gemm_skinny_kernel<false>", flops, bytes, stream, &tok);
  if (b.g[0].b_kn) hipLaunchKernelGGL(gemm_skinny_kernel<true>, dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
  else hipLaunchKernelGGL(gemm_skinny_kernel<false>, dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
