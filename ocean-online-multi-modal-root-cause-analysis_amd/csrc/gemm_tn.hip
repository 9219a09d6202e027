// fp32 row reductions with full 128x128 output tiles: the convolutions' weight gradients
//   C_t[m][n] += sum_k A[k][m] * B[k + shift_t][n]      (A = dU [rows][Cout], B = x [rows][Cin], k = (sample, time) rows)
// split over the rows (grid.z = taps x slices, atomic accumulation).
//
// Same arithmetic and LDS layout as the AKM / BKN instantiation of the general kernel (gemm.hip): k-major tiles
// [16][128 + 4], float4 staging writes, one ds_read_b32 per fragment.  What differs is everything around the MFMAs: the
// host sends only problems whose tiles are full and whose slices are whole 16-row tiles, so a staged piece needs no
// bounds arithmetic beyond the time window of a shifted row (two wraps and a compare: samples may be of any length);
// tile coordinates are two scalar adds; fragments are read half a tile (16 MFMAs) ahead.  Per 16-deep tile the general
// kernel's loop carries ~48 vector, ~49 scalar and ~34 wait instructions next to its 32 MFMAs, this one ~19, ~9 and ~6.
#include "gemm_epilogue.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LD = BM + 4;                    // floats per k-row of a tile image

__global__ __launch_bounds__(kThreads, 3) void gemm_tn_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[0];
  constexpr int TM = 2, TN = 2, KS = BK / 2, HS = KS / 2;
  constexpr int SLOT = BK * LD;               // floats per ring slot of one operand
  __shared__ __align__(16) float smem[3 * SLOT * 2];
  float* const As = smem;
  float* const Bs = smem + 3 * SLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, h = lane >> 5;

  // block -> (row slice, tap, output tile): every block of one slice (all output tiles, all taps) reads the same rows
  // of A and B, so slices - not tiles - are what an XCD's L2 should share (blocks are dealt to XCDs round-robin by id)
  const int tiles_n = p.N / BN, nwg = (p.M / BM) * tiles_n;
  const int split = p.split_k;
  const int total = nwg * (int)gridDim.z;
  const int q = xcd_swizzle((int)(blockIdx.x + gridDim.x * blockIdx.z), total);
  const int per_slice = nwg * p.taps;
  const int zs = q / per_slice, ztap = (q % per_slice) / nwg, bid = q % nwg;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int chunk = (p.K + split - 1) / split;
  chunk = (chunk + BK - 1) / BK * BK;
  const int k_begin = zs * chunk, k_end = min(p.K, k_begin + chunk);
  if (k_begin >= k_end) return;
  const int n_iter = (k_end - k_begin) / BK;
  const int shift = p.T > 0 ? p.shift0 + ztap * p.shift_step : 0;

  // B's descriptor starts `shift` rows into (or before) the matrix, so that the rows of a tile are addressed by the
  // tile's own row number; a row whose shifted time index leaves its sample is never fetched (out-of-range offset),
  // which also covers the rows before the matrix's first and after its last
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B) + (long)shift * p.ldb, 0, 0x7FFFFFF0, 0x00020000);
  constexpr unsigned kOob = 0x80000000u;
  auto buf_load = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
  };

  // staging: piece i of a tile is columns 4 (idx & 31) .. + 3 of k-row (idx >> 5), idx = tid + 256 i: rows r and r + 8
  const int krow = tid >> 5, kcol = (tid & 31) * 4;
  const unsigned a_goff = (unsigned)(krow * p.lda + m0 + kcol) * 4u, b_goff = (unsigned)(krow * p.ldb + n0 + kcol) * 4u;
  const unsigned a_half = (unsigned)p.lda * 32u, b_half = (unsigned)p.ldb * 32u;       // 8 rows further
  const unsigned st_off = (unsigned)(krow * LD + kcol);
  float4 ra[2], rb[2];
  // scalar state of the next tile to load: byte offsets of its first row, its time index plus the shift.  Past the end
  // of the slice the last tile is loaded again, with its own time index (it goes to a ring slot nobody reads): no
  // validity test per piece, and no row outside the slice is ever addressed.
  const int a_step = BK * p.lda * 4, b_step = BK * p.ldb * 4;
  int a_s = k_begin * p.lda * 4, b_s = k_begin * p.ldb * 4;
  const int a_last = (k_end - BK) * p.lda * 4;
  // ts: time index of the tile's first row.  A tile may run over sample boundaries (samples of any length >= 8 rows), so
  // each staged row takes its own time index: ts + r, wrapped (twice covers r < 16 <= 2 T).
  int ts = p.T > 0 ? k_begin % p.T : 0;
  const int Tw = p.T > 0 ? p.T : 0x3FFFFFFF;
  auto in_sample = [&](int r) {
    int t = ts + r;
    if (t >= Tw) t -= Tw;
    if (t >= Tw) t -= Tw;
    return (unsigned)(t + shift) < (unsigned)Tw;
  };
  auto load_tile = [&]() {
    ra[0] = buf_load(rsrc_a, a_goff, a_s);
    ra[1] = buf_load(rsrc_a, a_goff + a_half, a_s);
    rb[0] = buf_load(rsrc_b, in_sample(krow) ? b_goff : kOob, b_s);
    rb[1] = buf_load(rsrc_b, in_sample(krow + 8) ? b_goff + b_half : kOob, b_s);
    if (a_s < a_last) {
      a_s += a_step;
      b_s += b_step;
      ts += BK;
      if (ts >= Tw) ts -= Tw;
      if (ts >= Tw) ts -= Tw;
    }
  };
  auto store_tile = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(As + slot * SLOT + st_off + i * 8 * LD) = ra[i];
      *reinterpret_cast<float4*>(Bs + slot * SLOT + st_off + i * 8 * LD) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tile();
  store_tile(0);
  load_tile();
  __syncthreads();

  // fragments of half a tile (4 k-steps) at a time, two halves in registers
  const float* a_rd = As + h * LD + wm * 64 + lm;
  const float* b_rd = Bs + h * LD + wn * 64 + lm;
  float fa[2][HS][TM], fb[2][HS][TN];
  auto read_half = [&](int slot, int half, int hb) {
#pragma unroll
    for (int s = 0; s < HS; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[hb][s][i] = a_rd[slot * SLOT + 2 * (half * HS + s) * LD + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[hb][s][j] = b_rd[slot * SLOT + 2 * (half * HS + s) * LD + j * 32];
    }
  };
  auto mfma_step = [&](int hb, int s) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[hb][s][i], fb[hb][s][j], acc[i][j], 0, 0, 0);
  };
  read_half(0, 0, 0);

  // One tile: first half multiplied while the second half's fragments are read, the registers holding tile it+1 go to
  // the next ring slot and are reloaded with tile it+2; barrier; second half multiplied while tile it+1's first
  // fragments are read.  (Ring slot (it+1) % 3 was last read as tile it-2: before the barrier of tile it-1.)
  auto tile_step = [&](auto slot_c, const int it) {
    constexpr int S = decltype(slot_c)::value, NS = (S + 1) % 3;
    const bool more = it + 1 < n_iter;
    read_half(S, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0, 0);
    store_tile(NS);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0, 1);
    load_tile();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0, 2);
    mfma_step(0, 3);
    __syncthreads();
    if (more) read_half(NS, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(1, 0);
    mfma_step(1, 1);
    mfma_step(1, 2);
    mfma_step(1, 3);
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int it = 0;
  for (; it + 3 <= n_iter; it += 3) {
    tile_step(I0{}, it); tile_step(I1{}, it + 1); tile_step(I2{}, it + 2);
  }
  if (it < n_iter) {
    tile_step(I0{}, it);
    if (it + 1 < n_iter) tile_step(I1{}, it + 1);
  }

  const EpiArgs e = make_epi_args(p);
  float* Cz = e.C + (long)ztap * e.tap_stride_c;
  epilogue_full<BM, BN, true, false>(acc, e, Cz, DropoutP{0.f, 0, nullptr}, m0, n0, tile_m, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// The same reduction on split operands (x = hi + lo in bf16, three bf16 matrix instructions per k-step: see
// gemm_nt.hip).  A staging thread loads the same four columns of two consecutive rows k = 2j, 2j + 1 and packs each
// column's pair into one dword, so a tile image is [8 row pairs][128 + 4 columns] dwords per half (hi, lo); lane (m, h)
// reads the four dwords of row pairs 4h .. 4h + 3 of its column - rows k = 8h .. 8h + 7, the operand of
// v_mfma_f32_32x32x16_bf16 - with four ds_read_b32 per half.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// columns c .. c + 3 of rows k (a) and k + 1 (b) -> per column one dword of hi halves {k, k + 1} and one of lo halves
__device__ __forceinline__ void split_pair4(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  hi.x = pack_bf16(a.x, b.x); hi.y = pack_bf16(a.y, b.y); hi.z = pack_bf16(a.z, b.z); hi.w = pack_bf16(a.w, b.w);
  lo.x = pack_bf16(a.x - __uint_as_float(hi.x << 16), b.x - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pack_bf16(a.y - __uint_as_float(hi.y << 16), b.y - __uint_as_float(hi.y & 0xffff0000u));
  lo.z = pack_bf16(a.z - __uint_as_float(hi.z << 16), b.z - __uint_as_float(hi.z & 0xffff0000u));
  lo.w = pack_bf16(a.w - __uint_as_float(hi.w << 16), b.w - __uint_as_float(hi.w & 0xffff0000u));
}

__global__ __launch_bounds__(kThreads, 3) void gemm_tn_bf3_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[0];
  constexpr int TM = 2, TN = 2;
  constexpr int PAIRS = BK / 2;                 // row pairs per tile
  constexpr int HALF = PAIRS * LD;              // dwords per half (hi or lo) of one operand's tile image
  constexpr int SLOT = 2 * HALF;
  __shared__ __align__(16) unsigned smem[3 * SLOT * 2];
  unsigned* const As = smem;
  unsigned* const Bs = smem + 3 * SLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, h = lane >> 5;

  const int tiles_n = p.N / BN, nwg = (p.M / BM) * tiles_n;
  const int split = p.split_k;
  const int total = nwg * (int)gridDim.z;
  const int q = xcd_swizzle((int)(blockIdx.x + gridDim.x * blockIdx.z), total);
  const int per_slice = nwg * p.taps;
  const int zs = q / per_slice, ztap = (q % per_slice) / nwg, bid = q % nwg;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int chunk = (p.K + split - 1) / split;
  chunk = (chunk + BK - 1) / BK * BK;
  const int k_begin = zs * chunk, k_end = min(p.K, k_begin + chunk);
  if (k_begin >= k_end) return;
  const int n_iter = (k_end - k_begin) / BK;
  const int shift = p.T > 0 ? p.shift0 + ztap * p.shift_step : 0;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B) + (long)shift * p.ldb, 0, 0x7FFFFFF0, 0x00020000);
  constexpr unsigned kOob = 0x80000000u;
  auto buf_load = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
  };

  // staging: thread (j = tid >> 5, c = 4 (tid & 31)) holds columns c .. c + 3 of rows 2j and 2j + 1 of a tile
  const int kpair = tid >> 5, kcol = (tid & 31) * 4;
  const unsigned a_goff = (unsigned)(2 * kpair * p.lda + m0 + kcol) * 4u, b_goff = (unsigned)(2 * kpair * p.ldb + n0 + kcol) * 4u;
  const unsigned a_row = (unsigned)p.lda * 4u, b_row = (unsigned)p.ldb * 4u;
  const unsigned st_off = (unsigned)(kpair * LD + kcol);
  float4 ra[2], rb[2];
  const int a_step = BK * p.lda * 4, b_step = BK * p.ldb * 4;
  int a_s = k_begin * p.lda * 4, b_s = k_begin * p.ldb * 4;
  const int a_last = (k_end - BK) * p.lda * 4;
  int ts = p.T > 0 ? k_begin % p.T : 0;                    // as in gemm_tn_kernel: per-row time indices
  const int Tw = p.T > 0 ? p.T : 0x3FFFFFFF;
  auto in_sample = [&](int r) {
    int t = ts + r;
    if (t >= Tw) t -= Tw;
    if (t >= Tw) t -= Tw;
    return (unsigned)(t + shift) < (unsigned)Tw;
  };
  auto load_tile = [&]() {
    ra[0] = buf_load(rsrc_a, a_goff, a_s);
    ra[1] = buf_load(rsrc_a, a_goff + a_row, a_s);
    rb[0] = buf_load(rsrc_b, in_sample(2 * kpair) ? b_goff : kOob, b_s);
    rb[1] = buf_load(rsrc_b, in_sample(2 * kpair + 1) ? b_goff + b_row : kOob, b_s);
    if (a_s < a_last) {
      a_s += a_step;
      b_s += b_step;
      ts += BK;
      if (ts >= Tw) ts -= Tw;
      if (ts >= Tw) ts -= Tw;
    }
  };
  auto store_tile = [&](int slot) {
    uint4 hi, lo;
    split_pair4(ra[0], ra[1], hi, lo);
    *reinterpret_cast<uint4*>(As + slot * SLOT + st_off) = hi;
    *reinterpret_cast<uint4*>(As + slot * SLOT + HALF + st_off) = lo;
    split_pair4(rb[0], rb[1], hi, lo);
    *reinterpret_cast<uint4*>(Bs + slot * SLOT + st_off) = hi;
    *reinterpret_cast<uint4*>(Bs + slot * SLOT + HALF + st_off) = lo;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tile();
  store_tile(0);
  load_tile();
  __syncthreads();

  const unsigned* a_rd = As + 4 * h * LD + wm * 64 + lm;
  const unsigned* b_rd = Bs + 4 * h * LD + wn * 64 + lm;
  // One tile: fragments read, the registers holding tile it+1 to the next ring slot and reloaded with tile it+2,
  // twelve matrix instructions, barrier.  (Ring slot (it+1) % 3 was last read as tile it-2.)
  auto tile_step = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value, NS = (S + 1) % 3;
    uint4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned* r = a_rd + S * SLOT + pl * HALF + i * 32;
        fa[pl][i] = make_uint4(r[0], r[LD], r[2 * LD], r[3 * LD]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned* r = b_rd + S * SLOT + pl * HALF + j * 32;
        fb[pl][j] = make_uint4(r[0], r[LD], r[2 * LD], r[3 * LD]);
      }
    }
    store_tile(NS);
    load_tile();
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      const int qa = term == 0 ? 1 : 0, qb = term == 1 ? 1 : 0;        // lo.hi, hi.lo, hi.hi
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[qa][i]), __builtin_bit_cast(bf16x8, fb[qb][j]),
                                                             acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int it = 0;
  for (; it + 3 <= n_iter; it += 3) { tile_step(I0{}); tile_step(I1{}); tile_step(I2{}); }
  if (it < n_iter) {
    tile_step(I0{});
    if (it + 1 < n_iter) tile_step(I1{});
  }

  const EpiArgs e = make_epi_args(p);
  float* Cz = e.C + (long)ztap * e.tap_stride_c;
  epilogue_full<BM, BN, true, false>(acc, e, Cz, DropoutP{0.f, 0, nullptr}, m0, n0, tile_m, wave, lane);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// One row reduction whose output is whole 128x128 tiles, over at least 8192 rows in whole 16-row tiles, samples at
// least 8 rows long, no bias-gradient side output.
bool ocean_gemm_tn_applies(const GemmBatch& b) {
  static const bool off = getenv("OCEAN_GEMM_NO_TN") != nullptr;          // measurement switch
  if (off || b.n != 1) return false;
  const GemmP& p = b.g[0];
  if (!p.a_kmajor || !p.b_kn || p.a_colsum != nullptr || p.bias != nullptr) return false;
  if (p.M % BM != 0 || p.N % BN != 0 || p.K % BK != 0 || p.K < 8192) return false;
  if (!ocean_split_bf16() && ocean_wgrad_small_tiles(p)) return false;      // (the split-operand form exists in this kernel only)
  if (p.T > 0 && p.T < 8) return false;                  // a row's time index is wrapped at most twice per 16-row tile
  if (p.taps > 1 && p.T <= 0) return false;
  if (!al16(p.A) || !al16(p.B) || p.lda % 4 != 0 || p.ldb % 4 != 0) return false;
  if ((double)p.K * p.lda >= 5.0e8 || (double)p.K * p.ldb >= 5.0e8) return false;     // 32-bit byte offsets
  return true;
}

int ocean_launch_gemm_tn(const GemmBatch& b, hipStream_t stream) {
  GemmBatch g = b;
  GemmP& p = g.g[0];
  const long tiles = (long)(p.M / BM) * (p.N / BN);
  int split = p.split_k;
  if (split <= 0) {
    // as many row slices as give ~3 blocks per compute unit, at least 128 rows each, and - every slice adds its tile
    // into the same addresses - no more than sqrt(K / 2) of them, nor slices under 256 rows (measured at 16384 rows:
    // 64 slices beat 90 on every layer shape, 38 -> 32 us for one 128x128 tile of three taps)
    long want = (768 + tiles * p.taps - 1) / (tiles * p.taps);
    const long max_split = (p.K + 127) / 128;
    long cap = (long)sqrt((double)p.K / 2.0);
    if (cap > p.K / 256) cap = p.K / 256;
    if (want > max_split) want = max_split;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    split = (int)want;
  }
  p.split_k = split;
  const dim3 grid((unsigned)tiles, 1, (unsigned)(p.taps * split));
  void* tok = nullptr;
  if (ocean_prof_on()) {
    const double bytes = 4.0 * ((double)p.M * p.K + (double)p.N * p.K * p.taps + (double)p.M * p.N * p.taps);
    ocean_prof_begin(ocean_split_bf16() ? "gemm_tn_bf3_kernel" : "gemm_tn_kernel", 2.0 * p.M * p.N * (double)p.K * p.taps, bytes, stream, &tok);
  }
  if (ocean_split_bf16()) hipLaunchKernelGGL(gemm_tn_bf3_kernel, grid, dim3(kThreads), 0, stream, g);
  else hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(kThreads), 0, stream, g);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
