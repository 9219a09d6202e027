// fp32 GEMM for the large products whose operands are both stored with the reduction index contiguous
// (activations [rows][channels] times weights [out][in]: the convolutions' forward and, against a transposed copy of
// the weights, their data gradients).
//
// What differs from the general kernel of gemm.hip:
//  * operand tiles keep their k-contiguous storage in LDS (64-byte rows of 16 k-values, the four 16-byte quads of a
//    row XOR-swizzled by (row >> 2) & 3), so a staged float4 is written with one ds_write_b128 and lane (m, h) reads
//    the four values k = 8q + 4h .. + 3 of row m with one ds_read_b128: MFMA t of a group then multiplies k = 8q + t
//    (lanes h = 0) and k = 8q + 4 + t (h = 1) on both operands alike.  8 fragment reads and 4 staging writes per 16-deep
//    tile where the k-major layout needs 32 and 10-16;
//  * a dilated convolution's three taps read ONE staged window of rows: a tile of 128 rows is 1, 2 or 4 whole samples
//    (T = 128, 64, 32), each sample's rows sit in LDS between bands of zero rows (the convolution's padding), and tap t
//    reads the fragments `shift_t` rows up or down.  The activation tile is loaded and stored once per 16 channels
//    instead of once per tap.
// Tiles are always full (the host sends only such problems here), so there is no bounds test anywhere.
#include "gemm_epilogue.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int kPad = 32;                 // zero rows on either side of the window = largest row shift served

// byte offset of 16-byte quad `kq` of row `r` inside a tile image
__device__ __forceinline__ unsigned img_off(int r, int kq) { return (unsigned)r * 64u + (unsigned)((kq ^ ((r >> 2) & 3)) << 4); }

// SEG: samples per 128-row tile (three-tap launches; 1 otherwise).  SEG = 0 is the window for samples of ANY length: a
// tile's 128 rows start and end anywhere inside samples, so the window also holds the |shift| rows before and after the
// tile (fetched from the neighbouring tiles' rows: the halo), there are no zero bands, and a tap's contribution to an
// output row whose shifted time index leaves its sample is dropped where the fragment is used (one select per
// matrix instruction operand of taps 0 and 2, from four lane masks worked out once).
template <int BN, int TAPS, bool BNS, int SEG = 1>
__global__ __launch_bounds__(kThreads, SEG == 4 ? 2 : 3) void gemm_nt_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[blockIdx.y];
  constexpr int BM = 128, BK = 16;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int A_SLOTS = TAPS == 3 ? 2 : 3, B_SLOTS = 3;
  constexpr bool GEN = TAPS == 3 && SEG == 0;
  constexpr int TS = GEN ? BM : BM / (SEG > 0 ? SEG : 1);     // rows per sample
  constexpr int A_ROWS = BM + (TAPS == 3 ? (GEN ? 2 : SEG + 1) * kPad : 0);
  // window position of tile row r: every sample is preceded by a zero band (and the last one followed by one)
  auto win_row = [](int r) { return TAPS == 3 ? r + kPad * (1 + r / TS) : r; };      // GEN: TS = BM, one band's worth of halo on either side
  constexpr int A_SLOT_BYTES = A_ROWS * 64, B_SLOT_BYTES = BN * 64;
  constexpr int A_PIECES = GEN ? 3 : 2, B_PIECES = BN / 64; // float4 per thread per tile (GEN: the third is a halo row's)
  __shared__ __align__(16) char smem[A_SLOTS * A_SLOT_BYTES + B_SLOTS * B_SLOT_BYTES];
  char* const As = smem;
  char* const Bs = smem + A_SLOTS * A_SLOT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, h = lane >> 5;

  const int tiles_n = p.N / BN, tiles_m = p.M / BM;
  const int nwg = tiles_m * tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int bid = xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int n_chunks = p.K / BK;
  const int n_iter = TAPS * n_chunks;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7FFFFFF0, 0x00020000);
  constexpr unsigned kOob = 0x80000000u;
  auto buf_load = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
  };

  // staging coordinates: piece i of a tile is quad (idx & 3) of row (idx >> 2), idx = tid + 256 i
  unsigned a_goff[A_PIECES], a_loff[A_PIECES], b_goff[B_PIECES], b_loff[B_PIECES];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * kThreads, row = idx >> 2, kq = idx & 3;
    a_goff[i] = (unsigned)((m0 + row) * p.lda + kq * 4) * 4u;
    a_loff[i] = img_off(win_row(row), kq);
  }
  if (GEN) {
    // halo: P = |shift| rows above the tile and P below it, one quad per thread (threads 0 .. 8 P - 1).  Rows outside the
    // matrix are not fetched; the threads left over write zeros to window row 0, which no tap reads while P < kPad
    const int P = p.shift0 < 0 ? -p.shift0 : p.shift0;
    const int hr = tid >> 2, kq = tid & 3;
    int grow = -1, wrow = 0;
    if (hr < P) { grow = m0 - P + hr; wrow = kPad - P + hr; }
    else if (hr < 2 * P) { grow = m0 + BM + hr - P; wrow = kPad + BM + hr - P; }
    a_goff[A_PIECES - 1] = (grow >= 0 && grow < p.M) ? (unsigned)(grow * p.lda + kq * 4) * 4u : kOob;
    a_loff[A_PIECES - 1] = img_off(wrow, kq);
  }
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int idx = tid + i * kThreads, row = idx >> 2, kq = idx & 3;
    b_goff[i] = (unsigned)((n0 + row) * p.ldb + kq * 4) * 4u;
    b_loff[i] = img_off(row, kq);
  }
  float4 ra[A_PIECES], rb[B_PIECES];
  auto load_a = [&](int chunk) {      // activation rows of reduction chunk `chunk` (zeros past the end)
    const bool ok = chunk < n_chunks;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) ra[i] = buf_load(rsrc_a, ok ? a_goff[i] : kOob, chunk * (BK * 4));      // (an out-of-range offset stays out of range)
  };
  auto load_b = [&](int tap, int chunk) {
    const bool ok = chunk < n_chunks;
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) rb[i] = buf_load(rsrc_b, ok ? b_goff[i] : kOob, (int)(tap * p.tap_stride_b + chunk * BK) * 4);
  };
  auto store_a = [&](int slot) {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) *reinterpret_cast<float4*>(As + slot * A_SLOT_BYTES + a_loff[i]) = ra[i];
  };
  auto store_b = [&](int slot) {
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) *reinterpret_cast<float4*>(Bs + slot * B_SLOT_BYTES + b_loff[i]) = rb[i];
  };

  // fragment addresses: this lane's row of sub-tile 0 (sub-tile 1 is 32 rows further - 64 when a zero band lies between
  // them - with the same swizzle), one address per 8-deep half q of a tile and, for the window, per tap
  constexpr int A_SUB = (TAPS == 3 && SEG == 4) ? 4096 : 2048;      // (GEN: no band between the sub-tiles)
  unsigned a_rd[TAPS][2], b_rd[2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int shift = TAPS == 3 ? p.shift0 + t * p.shift_step : 0;
    const int r = win_row(wm * 64 + lm) + shift;
#pragma unroll
    for (int q = 0; q < 2; ++q) a_rd[t][q] = img_off(r, 2 * q + h);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) b_rd[q] = img_off(wn * (BN / 2) + lm, 2 * q + h);

  // GEN: whether tap 0 / tap 2 of this lane's row of sub-tile i stays inside the row's sample
  bool tap_ok[2][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { tap_ok[0][i] = true; tap_ok[1][i] = true; }
  if (GEN) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int tpos = (m0 + wm * 64 + i * 32 + lm) % p.T;
      tap_ok[0][i] = (unsigned)(tpos + p.shift0) < (unsigned)p.T;
      tap_ok[1][i] = (unsigned)(tpos + p.shift0 + 2 * p.shift_step) < (unsigned)p.T;
    }
  }

  if (TAPS == 3 && !GEN) {
    // the zero bands of both window slots (band b: rows [b (TS + kPad), b (TS + kPad) + kPad)), never written again
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int kBandQuads = (SEG + 1) * kPad * 4;
    for (int idx = tid; idx < kBandQuads * A_SLOTS; idx += kThreads) {
      const int slot = idx / kBandQuads, w = idx % kBandQuads;
      const int band = (w >> 2) / kPad, r = band * (TS + kPad) + (w >> 2) % kPad;
      *reinterpret_cast<float4*>(As + slot * A_SLOT_BYTES + r * 64 + (w & 3) * 16) = z;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: tile 0 into slot 0, tile 1's operands into the registers
  load_a(0);
  load_b(0, 0);
  store_a(0);
  store_b(0);
  if (TAPS == 3) load_b(1, 0);
  else { load_a(1); load_b(0, 1); }
  __syncthreads();

  float4 fa[2][TM], fb[2][TN];         // fragments of the two halves of a tile
  auto read_frags = [&](int half, int a_slot, int b_slot, int tap, int q) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[half][i] = *reinterpret_cast<const float4*>(As + a_slot * A_SLOT_BYTES + a_rd[tap][q] + i * A_SUB);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[half][j] = *reinterpret_cast<const float4*>(Bs + b_slot * B_SLOT_BYTES + b_rd[q] + j * 2048);
  };
  auto mfma_group = [&](int half, int t, auto tap_c) {
    constexpr int TAP = decltype(tap_c)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float av = t == 0 ? fa[half][i].x : t == 1 ? fa[half][i].y : t == 2 ? fa[half][i].z : fa[half][i].w;
        if (GEN && TAP != 1) av = tap_ok[TAP / 2][i] ? av : 0.f;
        const float bv = t == 0 ? fb[half][j].x : t == 1 ? fb[half][j].y : t == 2 ? fb[half][j].z : fb[half][j].w;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
      }
  };
  read_frags(0, 0, 0, 0, 0);

  // One 16-deep tile = two halves of 4 MFMA groups.  During the first half the registers holding tile it+1 go to
  // their ring slots and are reloaded with tile it+2; the barrier sits between the halves; the second half starts by
  // reading the first fragments of tile it+1.  With three taps the activation registers move only once per chunk:
  // loaded during tap 1, stored during tap 2 (into the other window slot, last read two barriers ago).
  int chunk = 0;
  auto tile_step = [&](auto tap_c, auto aslot_c, auto bslot_c, const int it) {
    constexpr int TAP = decltype(tap_c)::value, ASLOT = decltype(aslot_c)::value, BSLOT = decltype(bslot_c)::value;
    constexpr int NB = (BSLOT + 1) % B_SLOTS;
    constexpr bool NEXT_CHUNK = TAPS == 1 || TAP == 2;              // tile it+1 starts a new reduction chunk
    constexpr int NA = NEXT_CHUNK ? (ASLOT + 1) % A_SLOTS : ASLOT;
    constexpr int NTAP = TAPS == 1 ? 0 : (TAP + 1) % 3;
    const bool more = it + 1 < n_iter;
    read_frags(1, ASLOT, BSLOT, TAP, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0, 0, tap_c);
    // tile it+1's weights -> ring, tile it+2's weights -> registers
    store_b(NB);
    if (TAPS == 3) load_b((TAP + 2) % 3, chunk + (TAP + 2) / 3);
    else load_b(0, chunk + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0, 1, tap_c);
    if (TAPS == 1) { store_a(NA); load_a(chunk + 2); }
    else if (TAP == 2) store_a(NA);
    else if (TAP == 1) load_a(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(0, 2, tap_c);
    mfma_group(0, 3, tap_c);
    __syncthreads();
    if (more) read_frags(0, NA, NB, NTAP, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1, 0, tap_c);
    mfma_group(1, 1, tap_c);
    mfma_group(1, 2, tap_c);
    mfma_group(1, 3, tap_c);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT_CHUNK) ++chunk;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int it = 0;
  if (TAPS == 3) {
    for (; it + 6 <= n_iter; it += 6) {
      tile_step(I0{}, I0{}, I0{}, it);     tile_step(I1{}, I0{}, I1{}, it + 1); tile_step(I2{}, I0{}, I2{}, it + 2);
      tile_step(I0{}, I1{}, I0{}, it + 3); tile_step(I1{}, I1{}, I1{}, it + 4); tile_step(I2{}, I1{}, I2{}, it + 5);
    }
    if (it < n_iter) {                     // an odd number of chunks: one more, in window slot 0
      tile_step(I0{}, I0{}, I0{}, it); tile_step(I1{}, I0{}, I1{}, it + 1); tile_step(I2{}, I0{}, I2{}, it + 2);
    }
  } else {
    for (; it + 3 <= n_iter; it += 3) {
      tile_step(I0{}, I0{}, I0{}, it); tile_step(I0{}, I1{}, I1{}, it + 1); tile_step(I0{}, I2{}, I2{}, it + 2);
    }
    if (it < n_iter) {
      tile_step(I0{}, I0{}, I0{}, it);
      if (it + 1 < n_iter) tile_step(I0{}, I1{}, I1{}, it + 1);
    }
  }

  const EpiArgs e = make_epi_args(p);
  const DropoutP bn_drop = BNS ? dropout_resolve(p.bn_drop) : DropoutP{0.f, 0, nullptr};
  epilogue_full<BM, BN, false, BNS, GEN>(acc, e, e.C, bn_drop, m0, n0, tile_m, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// The same products with each fp32 operand split into two bf16 values, x = hi + lo (hi = bf16(x) rounded to nearest,
// lo = bf16(x - hi)), and three bf16 matrix instructions per fp32 one's worth of work: hi.hi + hi.lo + lo.hi,
// accumulated in fp32 (lo.lo, 2^-16 of the product, is dropped).  Opt-in (ocean_set_split_bf16): the results differ
// from the fp32 kernel's in about the 16th bit of every product.
//
// Operands are split once, on their way into LDS.  A tile image keeps 64-byte rows: quads {hi k0-7, hi k8-15,
// lo k0-7, lo k8-15}, swizzled as above; lane (m, h) reads quad h (and 2 + h) of row m with one ds_read_b128 each
// and holds k = 8h .. 8h + 7 - the operand layout of v_mfma_f32_32x32x16_bf16 - so a 16-deep tile is one k-step:
// 3 instructions of 8 passes per accumulator where the fp32 kernel issues 8 of 16 passes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  hi.x = pack_bf16(v.x, v.y);
  hi.y = pack_bf16(v.z, v.w);
  lo.x = pack_bf16(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pack_bf16(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
}

template <int BN, int TAPS, bool BNS, int SEG = 1>
__global__ __launch_bounds__(kThreads, SEG == 4 ? 2 : 3) void gemm_nt_bf3_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[blockIdx.y];
  constexpr int BM = 128, BK = 16;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int A_SLOTS = TAPS == 3 ? 2 : 3, B_SLOTS = 3;
  constexpr int TS = BM / SEG;
  constexpr int A_ROWS = BM + (TAPS == 3 ? (SEG + 1) * kPad : 0);
  auto win_row = [](int r) { return TAPS == 3 ? r + kPad * (1 + r / TS) : r; };
  constexpr int A_SLOT_BYTES = A_ROWS * 64, B_SLOT_BYTES = BN * 64;
  constexpr int A_PIECES = 2, B_PIECES = BN / 64;
  __shared__ __align__(16) char smem[A_SLOTS * A_SLOT_BYTES + B_SLOTS * B_SLOT_BYTES];
  char* const As = smem;
  char* const Bs = smem + A_SLOTS * A_SLOT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, h = lane >> 5;

  const int tiles_n = p.N / BN, tiles_m = p.M / BM;
  const int nwg = tiles_m * tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int bid = xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int n_chunks = p.K / BK;
  const int n_iter = TAPS * n_chunks;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7FFFFFF0, 0x00020000);
  constexpr unsigned kOob = 0x80000000u;
  auto buf_load = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
  };

  // staging: piece i of a tile is k-values 4 kq .. + 3 of row (idx >> 2), idx = tid + 256 i; its hi halves go to
  // 8 bytes of quad kq >> 1, its lo halves to the same place two quads further
  unsigned a_goff[A_PIECES], a_loff[A_PIECES][2], b_goff[B_PIECES], b_loff[B_PIECES][2];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int idx = tid + i * kThreads, row = idx >> 2, kq = idx & 3;
    a_goff[i] = (unsigned)((m0 + row) * p.lda + kq * 4) * 4u;
    a_loff[i][0] = img_off(win_row(row), kq >> 1) + (unsigned)(kq & 1) * 8u;
    a_loff[i][1] = img_off(win_row(row), 2 + (kq >> 1)) + (unsigned)(kq & 1) * 8u;
  }
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int idx = tid + i * kThreads, row = idx >> 2, kq = idx & 3;
    b_goff[i] = (unsigned)((n0 + row) * p.ldb + kq * 4) * 4u;
    b_loff[i][0] = img_off(row, kq >> 1) + (unsigned)(kq & 1) * 8u;
    b_loff[i][1] = img_off(row, 2 + (kq >> 1)) + (unsigned)(kq & 1) * 8u;
  }
  float4 ra[A_PIECES], rb[B_PIECES];
  auto load_a = [&](int chunk) {
    const bool ok = chunk < n_chunks;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) ra[i] = buf_load(rsrc_a, ok ? a_goff[i] : kOob, chunk * (BK * 4));
  };
  auto load_b = [&](int tap, int chunk) {
    const bool ok = chunk < n_chunks;
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) rb[i] = buf_load(rsrc_b, ok ? b_goff[i] : kOob, (int)(tap * p.tap_stride_b + chunk * BK) * 4);
  };
  auto store_a = [&](int slot) {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      uint2 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<uint2*>(As + slot * A_SLOT_BYTES + a_loff[i][0]) = hi;
      *reinterpret_cast<uint2*>(As + slot * A_SLOT_BYTES + a_loff[i][1]) = lo;
    }
  };
  auto store_b = [&](int slot) {
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      uint2 hi, lo;
      split4(rb[i], hi, lo);
      *reinterpret_cast<uint2*>(Bs + slot * B_SLOT_BYTES + b_loff[i][0]) = hi;
      *reinterpret_cast<uint2*>(Bs + slot * B_SLOT_BYTES + b_loff[i][1]) = lo;
    }
  };

  constexpr int A_SUB = (TAPS == 3 && SEG == 4) ? 4096 : 2048;
  unsigned a_rd[TAPS][2], b_rd[2];         // [.][0]: hi quad, [.][1]: lo quad
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int shift = TAPS == 3 ? p.shift0 + t * p.shift_step : 0;
    const int r = win_row(wm * 64 + lm) + shift;
#pragma unroll
    for (int q = 0; q < 2; ++q) a_rd[t][q] = img_off(r, 2 * q + h);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) b_rd[q] = img_off(wn * (BN / 2) + lm, 2 * q + h);

  if (TAPS == 3) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int kBandQuads = (SEG + 1) * kPad * 4;
    for (int idx = tid; idx < kBandQuads * A_SLOTS; idx += kThreads) {
      const int slot = idx / kBandQuads, w = idx % kBandQuads;
      const int band = (w >> 2) / kPad, r = band * (TS + kPad) + (w >> 2) % kPad;
      *reinterpret_cast<float4*>(As + slot * A_SLOT_BYTES + r * 64 + (w & 3) * 16) = z;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_a(0);
  load_b(0, 0);
  store_a(0);
  store_b(0);
  if (TAPS == 3) load_b(1, 0);
  else { load_a(1); load_b(0, 1); }
  __syncthreads();

  // One 16-deep tile: its fragments are read, the registers holding tile it+1 go to their ring slots and are reloaded
  // with tile it+2, twelve matrix instructions, barrier.  (The other blocks of the compute unit cover the LDS latency.)
  int chunk = 0;
  auto tile_step = [&](auto tap_c, auto aslot_c, auto bslot_c) {
    constexpr int TAP = decltype(tap_c)::value, ASLOT = decltype(aslot_c)::value, BSLOT = decltype(bslot_c)::value;
    constexpr int NB = (BSLOT + 1) % B_SLOTS;
    constexpr bool NEXT_CHUNK = TAPS == 1 || TAP == 2;
    constexpr int NA = NEXT_CHUNK ? (ASLOT + 1) % A_SLOTS : ASLOT;
    uint4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const uint4*>(As + ASLOT * A_SLOT_BYTES + a_rd[TAP][q] + i * A_SUB);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const uint4*>(Bs + BSLOT * B_SLOT_BYTES + b_rd[q] + j * 2048);
    }
    store_b(NB);
    if (TAPS == 3) load_b((TAP + 2) % 3, chunk + (TAP + 2) / 3);
    else load_b(0, chunk + 2);
    if (TAPS == 1) { store_a(NA); load_a(chunk + 2); }
    else if (TAP == 2) store_a(NA);
    else if (TAP == 1) load_a(chunk + 1);
    // small terms first; the four accumulators separate dependent instructions
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
    if (NEXT_CHUNK) ++chunk;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int it = 0;
  if constexpr (TAPS == 3) {
    for (; it + 6 <= n_iter; it += 6) {
      tile_step(I0{}, I0{}, I0{}); tile_step(I1{}, I0{}, I1{}); tile_step(I2{}, I0{}, I2{});
      tile_step(I0{}, I1{}, I0{}); tile_step(I1{}, I1{}, I1{}); tile_step(I2{}, I1{}, I2{});
    }
    if (it < n_iter) { tile_step(I0{}, I0{}, I0{}); tile_step(I1{}, I0{}, I1{}); tile_step(I2{}, I0{}, I2{}); }
  } else {
    for (; it + 3 <= n_iter; it += 3) { tile_step(I0{}, I0{}, I0{}); tile_step(I0{}, I1{}, I1{}); tile_step(I0{}, I2{}, I2{}); }
    if (it < n_iter) {
      tile_step(I0{}, I0{}, I0{});
      if (it + 1 < n_iter) tile_step(I0{}, I1{}, I1{});
    }
  }

  const EpiArgs e = make_epi_args(p);
  const DropoutP bn_drop = BNS ? dropout_resolve(p.bn_drop) : DropoutP{0.f, 0, nullptr};
  epilogue_full<BM, BN, false, BNS>(acc, e, e.C, bn_drop, m0, n0, tile_m, wave, lane);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int BN, int TAPS, int SEG = 1>
void launch_nt(const GemmBatch& b, dim3 grid, hipStream_t stream) {
  if constexpr (SEG != 0) {            // the split-operand kernel has no general window (ocean_gemm_nt_applies)
    if (ocean_split_bf16()) {
      if (b.g[0].bn_u != nullptr) hipLaunchKernelGGL((gemm_nt_bf3_kernel<BN, TAPS, true, SEG>), grid, dim3(kThreads), 0, stream, b);
      else hipLaunchKernelGGL((gemm_nt_bf3_kernel<BN, TAPS, false, SEG>), grid, dim3(kThreads), 0, stream, b);
      return;
    }
  }
  if (b.g[0].bn_u != nullptr) hipLaunchKernelGGL((gemm_nt_kernel<BN, TAPS, true, SEG>), grid, dim3(kThreads), 0, stream, b);
  else hipLaunchKernelGGL((gemm_nt_kernel<BN, TAPS, false, SEG>), grid, dim3(kThreads), 0, stream, b);
}
template <int BN>
void launch_nt_taps(const GemmBatch& b, dim3 grid, hipStream_t stream) {
  const GemmP& p = b.g[0];
  if (p.taps != 3) launch_nt<BN, 1>(b, grid, stream);
  else if (p.T == 128) launch_nt<BN, 3, 1>(b, grid, stream);
  else if (p.T == 64) launch_nt<BN, 3, 2>(b, grid, stream);
  else if (p.T == 32) launch_nt<BN, 3, 4>(b, grid, stream);
  else launch_nt<BN, 3, 0>(b, grid, stream);
}

}  // namespace

// One problem, both operands k-contiguous, every tile full, enough tiles to fill the chip; a three-tap problem must have
// symmetric shifts no longer than a zero band (samples of 128, 64 or 32 rows sit whole in a row tile; any other length
// takes the window with halo rows and per-row tap masks).
bool ocean_gemm_nt_applies(const GemmBatch& b) {
  static const bool off = getenv("OCEAN_GEMM_NO_NT") != nullptr;          // measurement switch
  if (off || b.n != 1) return false;
  const GemmP& p = b.g[0];
  if (p.a_kmajor || p.b_kn || (p.taps != 1 && p.taps != 3)) return false;
  if (p.taps == 1 && p.T > 0 && p.shift0 != 0) return false;              // the one-tap body reads its rows unshifted
  if (p.M % 128 != 0 || p.N % 64 != 0 || p.K % 16 != 0 || p.K < 32) return false;
  static const long min_tiles = [] { const char* e = getenv("OCEAN_GEMM_NT_MIN_TILES"); return e ? atol(e) : 256L; }();   // tuning knob
  if ((long)(p.M / 128) * (p.N % 128 == 0 ? p.N / 128 : p.N / 64) < min_tiles) return false;
  if (!al16(p.A) || !al16(p.B) || p.lda % 4 != 0 || p.ldb % 4 != 0 || p.tap_stride_b % 4 != 0) return false;
  if ((double)p.M * p.lda >= 5.0e8 || (double)p.N * p.ldb + (double)(p.taps - 1) * (double)p.tap_stride_b >= 5.0e8) return false;
  if (p.taps == 3) {
    if (p.T < 1 || p.shift_step != -p.shift0 || p.shift0 > kPad || p.shift0 < -kPad) return false;
    const bool whole = p.T == 128 || p.T == 64 || p.T == 32;
    if (!whole && (ocean_split_bf16() || p.M % p.T != 0)) return false;       // the general window exists in fp32 only
  }
  if (p.drop.p > 0.f || p.relu_mask != nullptr || p.a_colsum != nullptr) return false;
  if (p.r_bcast_T > 0) {
    if (p.bn_u == nullptr || p.R == nullptr) return false;
    // a broadcast residual row is block-uniform when tiles lie inside samples; the general window looks it up per row
    // (rows below 2^24: the epilogue divides in fp32)
    const bool general = p.taps == 3 && p.T != 128 && p.T != 64 && p.T != 32;
    if (p.r_bcast_T % 128 != 0 && !(general && p.r_bcast_T == p.T && p.M < (1 << 24))) return false;
  }
  if (p.bn_u != nullptr && (p.bias != nullptr || (double)p.M * (double)p.N >= 4294967296.0)) return false;
  return true;
}

int ocean_launch_gemm_nt(const GemmBatch& b, hipStream_t stream) {
  const GemmP& p = b.g[0];
  const int bn = p.N % 128 == 0 ? 128 : 64;
  const dim3 grid((p.M / 128) * (p.N / bn), 1, 1);
  void* tok = nullptr;
  if (ocean_prof_on()) {
    char name[96];
    const bool halo = p.taps == 3 && p.T != 128 && p.T != 64 && p.T != 32;
    snprintf(name, sizeof(name), "gemm_nt%s_kernel<%d,%d%s%s>", ocean_split_bf16() ? "_bf3" : "", bn, p.taps, p.bn_u != nullptr ? ",bn-sums" : "", halo ? ",halo" : "");
    double bytes = 4.0 * ((double)p.M * p.K + (double)p.N * p.K * p.taps + (double)p.M * p.N);
    if (p.R != nullptr) bytes += 4.0 * (p.r_bcast_T > 0 ? (double)p.M / p.r_bcast_T : (double)p.M) * p.N;
    if (p.bn_u != nullptr) bytes += 4.0 * (double)p.M * p.N;
    ocean_prof_begin(name, 2.0 * p.M * p.N * (double)p.K * p.taps, bytes, stream, &tok);
  }
  if (bn == 128) launch_nt_taps<128>(b, grid, stream);
  else launch_nt_taps<64>(b, grid, stream);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
