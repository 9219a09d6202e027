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
#include "rowchain.h"
#include <cstdlib>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 256;
constexpr int kSkinnyUnroll = 4;    // 8-deep reduction steps loaded together (4 x 2 x float4 = 32 registers)
constexpr int kPairUnroll = 2;      // the same inside gemm_pair_kernel, whose 1024-thread blocks leave a thread 128 registers

// One 32x32 output tile by one group of 256 threads (tid = 0..255 inside the group; `part` is the group's own four
// partial tiles).  Every thread of the block must call it, active or not: it holds a block-wide barrier.  An inactive
// group (its tile index is past the problem's last tile) loads and stores nothing.
// CONV: the convolution-shaped launches of small batches as well - several taps (A's rows shifted per tap inside their
// sample, B advancing by tap_stride_b; all taps are one reduction of taps * K, split over the four waves like any
// other), the BatchNorm column statistics of the raw output, the BatchNorm-backward column sums of GemmP::bn_u and a
// residual broadcast from one row per sample.  Only from gemm_skinny_kernel (one tile per block: the extra block-wide
// barrier of the column sums needs every thread of the block in the same tile).
// NW: waves that split the reduction of the tile (4; 8 for the long reductions of the convolution-shaped launches, where
// one wave per SIMD with eight loads in flight left the tile waiting on L2 round trips); the epilogue is the first 256
// threads' either way.
template <bool BKN, int kUnroll, bool CONV = false, int NW = 4>
__device__ __forceinline__ void skinny_tile(const GemmP& p, const int tile, const int tid, float (*part)[32][33]) {
  const int tiles_n = (p.N + 31) / 32, tiles_m = (p.M + 31) / 32;
  const bool active = tile < tiles_m * tiles_n;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, h = lane >> 5;

  // this wave's share of the reduction, in whole 8-deep steps
  const int steps_tap = (p.K + 7) / 8;
  const int steps = CONV ? steps_tap * p.taps : steps_tap;
  const int per = (steps + NW - 1) / NW;
  const int s_begin = wave * per, s_end = active ? min(steps, s_begin + per) : s_begin;

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
  const int t_row = (CONV && p.T > 0) ? (m0 + lm) % p.T : 0;      // this lane's row inside its sample
  auto load_step = [&](int s, float (&a)[4], float (&b)[4]) {
    int tap = 0, ks = s;
    if (CONV && p.taps > 1) { tap = s / steps_tap; ks = s - tap * steps_tap; }
    const int k = 8 * ks + 4 * h;                                 // K % 4 == 0: a group of four is inside or outside
    const bool k_ok = s < s_end && k < p.K;
    bool a_ok = a_row_ok && k_ok;
    unsigned a_off = a_base + (unsigned)(8 * ks) * 4u, b_tap = 0u;
    if (CONV) {
      // (byte offsets are taken modulo 2^32: a negative shift wraps and comes back once the row part is added)
      const int shift = p.shift0 + tap * p.shift_step;
      if (p.T > 0) a_ok = a_ok && (unsigned)(t_row + shift) < (unsigned)p.T;
      a_off += (unsigned)(shift * p.lda) * 4u;
      b_tap = (unsigned)((long)tap * p.tap_stride_b) * 4u;
    }
    ld4(rsrc_a, a_ok ? a_off : kOob, a);
    if (BKN) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        b[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            rsrc_b, (int)((b_col_ok && k_ok) ? b_tap + b_base + (unsigned)(8 * ks + t) * b_row : kOob), 0, 0));
    } else {
      ld4(rsrc_b, (b_col_ok && k_ok) ? b_tap + b_base + (unsigned)(8 * ks) * 4u : kOob, b);
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
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][lm] = acc[r];
  __syncthreads();
  if (!active) return;

  const DropoutP drop = dropout_resolve(p.drop);
  const int col = n0 + (tid & 31);
  const bool col_ok = col < p.N;
  const float bias = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
  const bool bn_on = CONV && p.bn_u != nullptr, stats_on = CONV && p.colstats != nullptr;
  DropoutP bn_drop = DropoutP{0.f, 0, nullptr};
  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
  if (bn_on) {
    bn_drop = dropout_resolve(p.bn_drop);
    if (col_ok) { sc = p.bn_scale[col]; sh = p.bn_shift[col]; mu = p.bn_mean[col]; is = p.bn_invstd[col]; }
  }
  float cs_s = 0.f, cs_q = 0.f, bn_sg = 0.f, bn_sx = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = (tid >> 5) + 8 * i, row = m0 + lr;
    if (row >= p.M || !col_ok || (NW > 4 && tid >= 256)) continue;
    float raw = part[0][lr][tid & 31] + part[1][lr][tid & 31] + part[2][lr][tid & 31] + part[3][lr][tid & 31];
    if (NW > 4) {
#pragma unroll
      for (int w = 4; w < NW; ++w) raw += part[w][lr][tid & 31];
    }
    if (stats_on) { cs_s += raw; cs_q += raw * raw; }
    float v = raw + bias;
    if (p.R != nullptr) {
      if (CONV && p.r_bcast_T > 0) v += p.r_scale * p.R[(long)(row / p.r_bcast_T) * p.ldr + col];
      else v += p.R[(long)row * p.ldr + col];
    }
    if (p.relu) v = fmaxf(v, 0.f);
    if (drop.p > 0.f) {
      const uint64_t di = p.drop_group > 1 ? (uint64_t)(p.drop_row0 + row) * (p.N / p.drop_group) + col / p.drop_group
                                           : (uint64_t)(p.drop_row0 + row) * p.N + col;
      v *= dropout_keep(drop, di);
    }
    if (p.relu_mask != nullptr && !(p.relu_mask[(long)row * p.ld_mask + col] > 0.f)) v = 0.f;
    p.C[(long)row * p.ldc + col] = v;
    if (bn_on) {
      const float u = p.bn_u[(long)row * p.ld_bn_u + col];
      float g = (fmaf(u, sc, sh) > 0.f) ? v : 0.f;
      if (bn_drop.p > 0.f) g *= dropout_keep(bn_drop, (uint64_t)row * p.N + col);
      bn_sg += g;
      bn_sx += g * (u - mu) * is;
    }
  }
  if (CONV && (stats_on || bn_on)) {
    // column sums over the tile's 32 rows: the eight row groups meet in shared memory (the partial tiles are spent)
    __syncthreads();
    float* red = &part[0][0][0];                  // [4][8][32]
    const int rg = tid >> 5, cl = tid & 31;
    if (NW == 4 || tid < 256) {
      red[(0 * 8 + rg) * 32 + cl] = cs_s; red[(1 * 8 + rg) * 32 + cl] = cs_q;
      red[(2 * 8 + rg) * 32 + cl] = bn_sg; red[(3 * 8 + rg) * 32 + cl] = bn_sx;
    }
    __syncthreads();
    if (tid < 32 && col_ok) {
      float t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t[q] = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) t[q] += red[(q * 8 + g8) * 32 + tid];
      }
      if (stats_on) {
        double* cs = p.colstats + (size_t)(tile_m % p.colstat_slots) * 2 * p.N;
        unsafeAtomicAdd(cs + col, (double)t[0]);
        unsafeAtomicAdd(cs + p.N + col, (double)t[1]);
      }
      if (bn_on) {
        double* bs = p.bn_slots + (size_t)(tile_m % p.bn_nslots) * 2 * p.N;
        unsafeAtomicAdd(bs + col, (double)t[2]);
        unsafeAtomicAdd(bs + p.N + col, (double)t[3]);
      }
    }
  }
}

template <bool BKN, bool CONV = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[blockIdx.y];
  if ((int)blockIdx.x >= ((p.N + 31) / 32) * ((p.M + 31) / 32)) return;   // the grid is sized for the largest problem of the batch
  __shared__ float part[NW][32][33];
  skinny_tile<BKN, kSkinnyUnroll, CONV, NW>(p, blockIdx.x, threadIdx.x, part);
}

// Row reductions with a small output (a linear layer's weight gradient C[M][N] += sum_k A[k][m] B[k][n] over a few
// thousand rows, optionally the column sums of A on the side: its bias gradient).  Such a launch is bound by memory
// round trips and by the cost of atomic adds, not by arithmetic.  A block of SIXTEEN waves owns one 32x32 output tile
// and splits the rows sixteen ways, so a wave's share is a few dozen rows whose loads it can request all at once; both
// operands are row-major over the reduction index, so lane (c, h) loads element c of row k = 2s + h of either operand
// straight into its MFMA register (32 consecutive floats of a row per half-wave).  The sixteen partial tiles meet in
// LDS, one thread per output element.  With one block per tile (the usual case) and no other problem of the batch writing
// the same matrix, the block is the tile's only writer and adds it into C with plain loads and stores; only outputs of very few tiles split the rows over blocks as well
// (grid.z) and then add atomically (measured: 8 slices adding into 3 x 128 x 768 outputs took 48 us, the atomics alone).
constexpr int kRowsWaves = 16;
constexpr int kRedUnroll = 8;          // MFMA steps whose 2 x 8 loads are requested together (two such groups in flight; 12 would spill at 16 waves per block)

// One 32x32 output tile (index `tile`, which the caller has checked) by a block of sixteen waves, over row slice
// `slice` of `nslices`.
__device__ __forceinline__ void rows_tile(const GemmP& p, const int tile, const int slice, const int nslices, const int plain,
                                          float (*part)[32][33], float (*csum)[32]) {
  const int tiles_n = (p.N + 31) / 32;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 31, h = lane >> 5;

  // rows of this block's slice, then of this wave's share of it (whole MFMA steps of two rows)
  const int steps = (p.K + 1) / 2;
  const int per_block = (steps + nslices - 1) / nslices;
  const int b_begin = slice * per_block, b_end = min(steps, b_begin + per_block);
  const int per_wave = (per_block + kRowsWaves - 1) / kRowsWaves;
  const int s_begin = b_begin + wave * per_wave, s_end = min(b_end, s_begin + per_wave);

  constexpr unsigned kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7FFFFFF0, 0x00020000);
  const bool a_ok = m0 + lm < p.M, b_ok = n0 + lm < p.N;
  const unsigned a_base = (unsigned)(h * p.lda + m0 + lm) * 4u, b_base = (unsigned)(h * p.ldb + n0 + lm) * 4u;
  const unsigned a_step = (unsigned)p.lda * 8u, b_step = (unsigned)p.ldb * 8u;      // two rows
  auto load_step = [&](int s, float& a, float& b) {
    const bool k_ok = s < s_end && 2 * s + h < p.K;
    a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_a, (int)((a_ok && k_ok) ? a_base + (unsigned)s * a_step : kOob), 0, 0));
    b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_b, (int)((b_ok && k_ok) ? b_base + (unsigned)s * b_step : kOob), 0, 0));
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f;
  // two groups of kRedUnroll steps per trip, each group's loads requested while the other group is multiplied
  float a0[kRedUnroll], b0[kRedUnroll], a1[kRedUnroll], b1[kRedUnroll];
#pragma unroll
  for (int u = 0; u < kRedUnroll; ++u) load_step(s_begin + u, a0[u], b0[u]);
  for (int s = s_begin; s < s_end; s += 2 * kRedUnroll) {
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) load_step(s + kRedUnroll + u, a1[u], b1[u]);
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) { asum += a0[u]; acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) load_step(s + 2 * kRedUnroll + u, a0[u], b0[u]);
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) { asum += a1[u]; acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0); }
  }

#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][lm] = acc[r];
  asum += __shfl_xor(asum, 32);
  if (h == 0) csum[wave][lm] = asum;
  __syncthreads();
  const int lr = tid >> 5, lc = tid & 31, row = m0 + lr, col = n0 + lc;      // one output element per thread
  if (row < p.M && col < p.N) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kRowsWaves; ++w) v += part[w][lr][lc];
    float* dst = p.C + (long)row * p.ldc + col;
    if (plain) *dst += v;
    else unsafeAtomicAdd(dst, v);
  }
  if (p.a_colsum != nullptr && tile_n == 0 && tid < 32 && m0 + tid < p.M) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kRowsWaves; ++w) v += csum[w][tid];
    if (plain) p.a_colsum[m0 + tid] += v;
    else unsafeAtomicAdd(p.a_colsum + m0 + tid, v);
  }
}

__global__ __launch_bounds__(kRowsWaves * 64) void gemm_skinny_rows_kernel(const GemmBatch batch, const int plain) {
  const GemmP& p = batch.g[blockIdx.y];
  if ((int)blockIdx.x >= ((p.N + 31) / 32) * ((p.M + 31) / 32)) return;
  __shared__ float part[kRowsWaves][32][33];
  __shared__ float csum[kRowsWaves][32];
  rows_tile(p, blockIdx.x, blockIdx.z, gridDim.z, plain, part, csum);
}

// A linear layer's two gradients in ONE launch: they read the same dy and are independent of each other, and each alone
// is a launch of a few microseconds that leaves most of the chip idle.  Blocks with blockIdx.z < wsplit take the weight
// gradient's role (rows_tile over batch `wb`, row slice blockIdx.z), blocks with blockIdx.z == wsplit the data gradient's
// (batch `db`: four groups of 256 threads, a tile each).  The grid spans the larger of the two roles in x and y; a block
// outside its role's range leaves at once.
__global__ __launch_bounds__(kRowsWaves * 64) void gemm_pair_kernel(const GemmBatch wb, const GemmBatch db, const int wsplit, const int plain) {
  __shared__ float part[kRowsWaves][32][33];
  __shared__ float csum[kRowsWaves][32];
  if ((int)blockIdx.z < wsplit) {
    if ((int)blockIdx.y >= wb.n) return;
    const GemmP& p = wb.g[blockIdx.y];
    if ((int)blockIdx.x >= ((p.N + 31) / 32) * ((p.M + 31) / 32)) return;
    rows_tile(p, blockIdx.x, blockIdx.z, wsplit, plain, part, csum);
  } else {
    if ((int)blockIdx.y >= db.n) return;
    const GemmP& p = db.g[blockIdx.y];
    if ((int)blockIdx.x * 4 >= ((p.N + 31) / 32) * ((p.M + 31) / 32)) return;
    const int group = threadIdx.x >> 8;
    skinny_tile<true, kPairUnroll>(p, blockIdx.x * 4 + group, threadIdx.x & 255, part + 4 * group);
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Whether the batch should take this kernel: products without a row reduction whose 64x64 tiling would occupy less than
// one block per compute unit, with operands the vector loads can address - plain ones (one tap, no column statistics)
// in batches of up to three, convolution-shaped ones (skinny_conv below) one at a time.
// Row-reduction batches for gemm_skinny_rows_kernel: one tap, no row shifts, a reduction of at most 8192 rows (longer
// ones fill the chip through the ring kernel's split, unless the output is a handful of tiles), at most 128 output tiles (with more, the 32x32 tiles' operand
// re-reads from L2 cost more than the ring kernel's 64x64 tiles: 20-25 us measured at 384 tiles).
static bool skinny_rows_applies(const GemmBatch& b, const int min_rows = 64) {
  static const bool off = getenv("OCEAN_GEMM_NO_SKINNY_ROWS") != nullptr;  // measurement switch
  if (off) return false;
  long tiles32 = 0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    if (!p.a_kmajor || !p.b_kn || p.taps != 1 || (p.T > 0 && p.shift0 != 0)) return false;
    if (p.K != b.g[0].K || p.K < min_rows) return false;
    if ((double)p.K * p.lda >= 5.0e8 || (double)p.K * p.ldb >= 5.0e8) return false;     // 32-bit byte offsets
    tiles32 += (long)ceil_div(p.M, 32) * ceil_div(p.N, 32);
  }
  // long reductions only into one or two tiles (the temporal encoder's input projection: a 64 x 12 weight gradient over
  // 131 072 rows, 42 us through the ring kernel's 64x64 tiles, 19 us here as 256 row slices; at 8 and 16 tiles the
  // slices' atomic adds make this kernel the slower one: 44 and 129 us)
  static const int long_tiles = [] { const char* e = getenv("OCEAN_GEMM_ROWS_LONG_TILES"); return e ? atoi(e) : 2; }();
  if (b.g[0].K > 8192) return tiles32 <= long_tiles;
  return tiles32 <= 128;
}

// a problem that needs the convolution-shaped variant of the tile (taps, column statistics, BatchNorm-backward sums, a
// residual broadcast per sample)
static bool skinny_conv(const GemmP& p) {
  return p.taps != 1 || p.T > 0 || p.colstats != nullptr || p.bn_u != nullptr || p.r_bcast_T > 0;
}

bool ocean_gemm_skinny_applies(const GemmBatch& b) {
  static const bool off = getenv("OCEAN_GEMM_NO_SKINNY") != nullptr;       // measurement switch
  if (off) return false;
  if (b.g[0].a_kmajor) return skinny_rows_applies(b);
  static const bool conv_off = getenv("OCEAN_GEMM_NO_SKINNY_CONV") != nullptr;   // measurement switch
  long blocks64 = 0, tiles32 = 0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    if (p.a_kmajor || p.a_colsum != nullptr) return false;
    if (skinny_conv(p) && (conv_off || b.n != 1)) return false;
    if (p.b_kn != b.g[0].b_kn) return false;
    if (p.K % 4 != 0 || p.K < 32 || !al16(p.A) || p.lda % 4 != 0) return false;
    if (!p.b_kn && (!al16(p.B) || p.ldb % 4 != 0 || p.tap_stride_b % 4 != 0)) return false;
    // 32-bit byte offsets
    if ((double)p.M * p.lda >= 5.0e8 || (double)(p.b_kn ? p.K : p.N) * p.ldb + (double)(p.taps - 1) * (double)p.tap_stride_b >= 5.0e8) return false;
    blocks64 += (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
    tiles32 += (long)ceil_div(p.M, 32) * ceil_div(p.N, 32);
  }
  return blocks64 < 256 && tiles32 <= 2048;
}

// Grid of a row-reduction batch: the most tiles of any problem, the row slices, and whether the block of a tile may add
// into C with plain loads and stores.
struct RowsGrid { int tiles, split, plain; double flops, bytes; };
static RowsGrid rows_grid(const GemmBatch& b) {
  RowsGrid r{0, 1, 0, 0.0, 0.0};
  long all_tiles = 0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    const int t = ceil_div(p.M, 32) * ceil_div(p.N, 32);
    if (t > r.tiles) r.tiles = t;
    all_tiles += t;
    r.flops += 2.0 * p.M * p.N * (double)p.K;
    r.bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
  }
  // One block per tile unless that leaves most of the chip idle: then the rows are sliced over blocks too, down to
  // one request round per wave (2 x kRedUnroll rows), the slices adding atomically.
  const int K = b.g[0].K;
  int split = 1;
  if (all_tiles < 64) {
    split = (int)(128 / (all_tiles > 0 ? all_tiles : 1));
    // a long reduction: two request rounds per wave, as long as the slices' atomic adds stay below ~half a million
    if (K > 8192) {
      const long by_rows = ceil_div(K, kRowsWaves * 4 * kRedUnroll), by_adds = 512 / (all_tiles > 0 ? all_tiles : 1);
      split = (int)(by_rows < by_adds ? by_rows : by_adds);
    }
    const int most = ceil_div(K, kRowsWaves * 2 * kRedUnroll);
    if (split > most) split = most;
    if (split < 1) split = 1;
  }
  r.split = split;
  // plain (non-atomic) accumulation needs every output element to have one writer in this launch: one slice, and no
  // two problems of the batch whose outputs (or bias-gradient vectors) overlap
  bool plain = split == 1;
  for (int q = 0; q < b.n && plain; ++q)
    for (int r2 = q + 1; r2 < b.n && plain; ++r2) {
      const GemmP &x = b.g[q], &y = b.g[r2];
      const float *x0 = x.C, *x1 = x.C + (size_t)(x.M - 1) * x.ldc + x.N, *y0 = y.C, *y1 = y.C + (size_t)(y.M - 1) * y.ldc + y.N;
      if (x0 < y1 && y0 < x1) plain = false;
      if (x.a_colsum && y.a_colsum && x.a_colsum < y.a_colsum + y.M && y.a_colsum < x.a_colsum + x.M) plain = false;
    }
  r.plain = plain ? 1 : 0;
  return r;
}

static int launch_skinny_rows(const GemmBatch& b, hipStream_t stream) {
  const RowsGrid r = rows_grid(b);
  void* tok = nullptr;
  if (ocean_prof_on()) ocean_prof_begin("gemm_skinny_rows_kernel", r.flops, r.bytes, stream, &tok);
  hipLaunchKernelGGL(gemm_skinny_rows_kernel, dim3(r.tiles, b.n, r.split), dim3(kRowsWaves * 64), 0, stream, b, r.plain);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

// [lo, hi) of the floats a row-major operand of `rows` rows of `cols` floats, `ld` apart, occupies
static inline bool spans_overlap(const float* a, long rows_a, long ld_a, long cols_a, const float* b, long rows_b, long ld_b, long cols_b) {
  const float *a1 = a + (rows_a - 1) * ld_a + cols_a, *b1 = b + (rows_b - 1) * ld_b + cols_b;
  return a < b1 && b < a1;
}

// Weight-gradient problems `w` (row reductions, a_kmajor && b_kn) and data-gradient problems `d` (b_kn) of the same
// layers: one launch of gemm_pair_kernel when both are work for the few-tile kernels and nothing one role writes is read
// or written by the other; otherwise the two ordinary launches, weight gradients first.
int ocean_launch_gemm_pair(const GemmP* w, int nw, const GemmP* d, int nd, hipStream_t stream) {
  if (rc_collecting()) {
    // a scope of rowchain.h is alive: the weight gradients join its batched reduction (and the data gradients the stage
    // kernel, when one is being recorded) - through the ordinary batch launcher, which hands them over problem by problem
    if (nw > 0) OCEAN_TRY(ocean_launch_gemm_batch(w, nw, stream));
    if (nd > 0) OCEAN_TRY(ocean_launch_gemm_batch(d, nd, stream));
    return OCEAN_OK;
  }
  static const bool off = getenv("OCEAN_GEMM_NO_PAIR") != nullptr;         // measurement switch
  bool pair = !off && nw >= 1 && nd >= 1 && nw <= kGemmBatchMax && nd <= kGemmBatchMax;
  GemmBatch wb, db;
  if (pair) {
    wb.n = nw; db.n = nd;
    for (int q = 0; q < nw; ++q) wb.g[q] = w[q];
    for (int q = 0; q < nd; ++q) db.g[q] = d[q];
    for (int q = 0; q < nw && pair; ++q) pair = w[q].a_kmajor && w[q].b_kn && w[q].M > 0 && w[q].N > 0 && w[q].K > 0 && w[q].A && w[q].B && w[q].C;
    for (int q = 0; q < nd && pair; ++q) pair = !d[q].a_kmajor && d[q].b_kn && d[q].M > 0 && d[q].N > 0 && d[q].K > 0 && d[q].A && d[q].B && d[q].C;
    // a reduction of a few rows (small batches) is no work for sixteen waves, but here it rides in a launch that is made anyway
    static const bool skinny_off = getenv("OCEAN_GEMM_NO_SKINNY") != nullptr;
    pair = pair && !skinny_off && wb.g[0].K <= 8192 && skinny_rows_applies(wb, 1) && ocean_gemm_skinny_applies(db);
    for (int q = 0; q < nd && pair; ++q) pair = !skinny_conv(d[q]);      // the pair kernel's data-gradient role is the plain tile
    for (int q = 0; q < nd && pair; ++q)
      for (int r = 0; r < nw && pair; ++r) {
        const GemmP &x = d[q], &y = w[r];
        // the data gradient's output against the weight gradient's operands and outputs
        if (spans_overlap(x.C, x.M, x.ldc, x.N, y.A, y.K, y.lda, y.M) || spans_overlap(x.C, x.M, x.ldc, x.N, y.B, y.K, y.ldb, y.N) ||
            spans_overlap(x.C, x.M, x.ldc, x.N, y.C, y.M, y.ldc, y.N) || (y.a_colsum && spans_overlap(x.C, x.M, x.ldc, x.N, y.a_colsum, 1, 0, y.M)))
          pair = false;
        // the weight gradient's outputs against the data gradient's operands
        if (spans_overlap(y.C, y.M, y.ldc, y.N, x.A, x.M, x.lda, x.K) || spans_overlap(y.C, y.M, y.ldc, y.N, x.B, x.K, x.ldb, x.N)) pair = false;
      }
  }
  if (!pair) {
    if (nw > 0) OCEAN_TRY(ocean_launch_gemm_batch(w, nw, stream));
    if (nd > 0) OCEAN_TRY(ocean_launch_gemm_batch(d, nd, stream));
    return OCEAN_OK;
  }
  const RowsGrid r = rows_grid(wb);
  int dblocks = 0;
  double flops = r.flops, bytes = r.bytes;
  for (int q = 0; q < nd; ++q) {
    const GemmP& p = d[q];
    const int t = ceil_div(ceil_div(p.M, 32) * ceil_div(p.N, 32), 4);
    if (t > dblocks) dblocks = t;
    flops += 2.0 * p.M * p.N * (double)p.K;
    bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
  }
  void* tok = nullptr;
  if (ocean_prof_on()) ocean_prof_begin("gemm_pair_kernel", flops, bytes, stream, &tok);
  hipLaunchKernelGGL(gemm_pair_kernel, dim3(r.tiles > dblocks ? r.tiles : dblocks, nw > nd ? nw : nd, r.split + 1), dim3(kRowsWaves * 64), 0,
                     stream, wb, db, r.split, r.plain);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_launch_gemm_skinny(const GemmBatch& b, hipStream_t stream) {
  if (b.g[0].a_kmajor) return launch_skinny_rows(b, stream);
  int tiles = 0;
  double flops = 0.0, bytes = 0.0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    const int t = ceil_div(p.M, 32) * ceil_div(p.N, 32);
    if (t > tiles) tiles = t;
    flops += 2.0 * p.M * p.N * (double)p.K * p.taps;
    bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K * p.taps + (double)p.M * p.N);
  }
  void* tok = nullptr;
  const bool conv = skinny_conv(b.g[0]);
  if (ocean_prof_on()) ocean_prof_begin(conv ? (b.g[0].b_kn ? "gemm_skinny_kernel<true, true>" : "gemm_skinny_kernel<false, true>")
                                             : (b.g[0].b_kn ? "gemm_skinny_kernel<true>" : "gemm_skinny_kernel<false>"), flops, bytes, stream, &tok);
  if (conv) {
    // reductions of 384 and more (48 eight-deep steps) are split over eight waves
    const bool wide = (long)((b.g[0].K + 7) / 8) * b.g[0].taps >= 48;
    if (b.g[0].b_kn) {
      if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<true, true, 8>), dim3(tiles, b.n), dim3(512), 0, stream, b);
      else hipLaunchKernelGGL((gemm_skinny_kernel<true, true>), dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
    } else {
      if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<false, true, 8>), dim3(tiles, b.n), dim3(512), 0, stream, b);
      else hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
    }
  }
  else if (b.g[0].b_kn) hipLaunchKernelGGL(gemm_skinny_kernel<true>, dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
  else hipLaunchKernelGGL(gemm_skinny_kernel<false>, dim3(tiles, b.n), dim3(kThreads), 0, stream, b);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
