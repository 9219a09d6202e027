// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// multiply-accumulate, 16 accumulator registers per 32x32 tile).
//
// One kernel template serves every dense contraction of the OCEAN step:
//   nn.Linear forward (A[m][k] . W[n][k]), its data gradient (dY[m][k] . W[k][n])
//   and weight gradient (dY[k][m]^T . X[k][n], reduction over rows, split-K with
//   atomics), and the dilated Conv1d as an implicit GEMM over channels-last
//   [B*T, C] activations: tap t reads rows shifted by (t - taps/2)*dilation with
//   zero fill outside each sample's [0,T) window.
//
// Tiling: 256 threads = 4 waves in a 2x2 arrangement; each wave owns
// (BM/2)x(BN/2) of the block tile as TMxTN 32x32 MFMA tiles.  Operand tiles are
// staged k-major in LDS (As[k][m], Bs[k][n]) so that a fragment read is one
// conflict-free ds_read_b32 per lane (lane l reads element [k=2s+(l>>5)][l&31]).
// Tiles whose global storage is k-contiguous are transposed on the LDS write;
// the row pad (32/BK floats) makes those scattered writes conflict-free.
// Global loads run two tiles ahead of the MFMAs through registers into a ring of
// three LDS tiles, with one barrier per tile placed mid-tile (see the main loop).
#include "gemm_epilogue.h"
#include "rowchain.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int kRing = 3;      // LDS tiles in flight per operand

// The main loop addresses LDS with a lane-constant base plus 16-bit immediate offsets.  The backend's load/store
// merger would pair the fragment reads into ds_read2_b32, whose offsets reach only 1 KB, and then needs one address
// register - and one vector add per ring turn - per k-step; it is switched off for this kernel (the attribute
// only means something to the device pass).
#if defined(__HIP_DEVICE_COMPILE__)
#define OCEAN_SINGLE_LDS_OPS __attribute__((target("no-load-store-opt")))
#else
#define OCEAN_SINGLE_LDS_OPS
#endif

// BNS: the launch also accumulates the BatchNorm-backward column sums described at GemmP::bn_u.  A template parameter,
// not a run-time test: the extra epilogue state must not count towards the register allocation of the other
// instantiations, whose three waves per SIMD depend on staying within 168 registers.
template <int BM, int BN, int BK, bool AKM, bool BKN, bool ALIGNED, bool BNS = false>
__global__ __launch_bounds__(kThreads, 3) OCEAN_SINGLE_LDS_OPS void gemm_f32_kernel(const GemmBatch batch) {
  const GemmP& p = batch.g[blockIdx.y];
  constexpr int WN = 2;
  constexpr int TM = BM / 2 / 32, TN = BN / 2 / 32;
  constexpr int PADT = 32 / BK;
  constexpr int LDA = BM + (AKM ? 4 : PADT);
  constexpr int LDB = BN + (BKN ? 4 : PADT);
  constexpr int A_F4 = BM * BK / 4 / kThreads;
  constexpr int B_F4 = BN * BK / 4 / kThreads;
  static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for 256 threads");
  __shared__ __align__(16) float smem[kRing * BK * (LDA + LDB)];
  float* As = smem;
  float* Bs = smem + kRing * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  if ((int)blockIdx.x >= nwg) return;          // the grid is sized for the largest problem of the batch
  int bid = xcd_swizzle(blockIdx.x, nwg);
  int ztap = 0, zs = 0;
  if (AKM) {
    const int split = p.split_k > 0 ? p.split_k : 1;
    ztap = blockIdx.z / split;
    zs = blockIdx.z % split;
    if (gridDim.y == 1) {
      // Row reductions: every block of one reduction chunk (all output tiles, all taps) reads the same rows of
      // A and B, so chunks - not tiles - are what an XCD's L2 should share.  Blocks are dealt to XCDs round-robin
      // in linear-id order; give each XCD a contiguous run of (chunk, tap, tile) triples.
      const int total = nwg * (int)gridDim.z;
      const int q = xcd_swizzle((int)(blockIdx.x + gridDim.x * blockIdx.z), total);
      const int per_chunk = nwg * p.taps;
      zs = q / per_chunk;
      ztap = (q % per_chunk) / nwg;
      bid = q % nwg;
    }
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // reduction range / tap handled by this block
  int k_begin = 0, k_end = p.K;
  if (AKM) {
    const int split = p.split_k > 0 ? p.split_k : 1;
    int chunk = (p.K + split - 1) / split;
    chunk = (chunk + BK - 1) / BK * BK;
    k_begin = zs * chunk;
    k_end = min(p.K, k_begin + chunk);
    if (k_begin >= k_end) return;
  }
  const int nk = (k_end - k_begin + BK - 1) / BK;
  const int n_iter = (AKM ? 1 : p.taps) * nk;

  // per-thread load coordinates that do not change over the reduction
  int a_row[A_F4], a_col[A_F4], a_t[A_F4];
  int b_row[B_F4], b_col[B_F4], b_rt[B_F4];
  unsigned a_voff[A_F4], b_voff[B_F4];   // loop-invariant part of the byte offset (ALIGNED path)
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int idx = tid + i * kThreads;
    if (AKM) {
      a_row[i] = idx / (BM / 4); a_col[i] = (idx % (BM / 4)) * 4; a_t[i] = 0;
      a_voff[i] = (unsigned)(a_row[i] * p.lda + m0 + a_col[i]) * 4u;
    } else {
      a_row[i] = idx / (BK / 4); a_col[i] = (idx % (BK / 4)) * 4;
      a_t[i] = (p.T > 0) ? (m0 + a_row[i]) % p.T : 0;
      a_voff[i] = (unsigned)((m0 + a_row[i]) * p.lda + a_col[i]) * 4u;   // row shift and k offset are wave-uniform
    }
  }
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + i * kThreads;
    if (BKN) {
      b_row[i] = idx / (BN / 4); b_col[i] = (idx % (BN / 4)) * 4;
      b_voff[i] = (unsigned)(b_row[i] * p.ldb + n0 + b_col[i]) * 4u;
      b_rt[i] = (AKM && p.T > 0) ? b_row[i] % p.T : 0;
    } else {
      b_row[i] = idx / (BK / 4); b_col[i] = (idx % (BK / 4)) * 4;
      b_voff[i] = (unsigned)((n0 + b_row[i]) * p.ldb + b_col[i]) * 4u;
    }
  }

  float4 ra[A_F4], rb[B_F4];

  // One "piece" = one float4 per thread of one operand; a tile is A_F4 + B_F4 pieces.  Pieces are loaded and
  // stored one at a time so that the main loop can spread them over the MFMA k-steps.
  // ALIGNED operands are read with raw buffer loads: a lane that must read zero (outside the matrix, outside the
  // sample's time window, or a tile past the end of the reduction) gets an out-of-range offset instead of a
  // branch, so the loop body is straight-line and the compiler can count outstanding loads (vmcnt(N)) across it.
  // The wave-uniform part of an address travels in the scalar offset (never negative).
  constexpr unsigned kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7FFFFFF0, 0x00020000);
  auto buf_load = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int soff) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0);
    const unsigned x = r[0], y = r[1], z = r[2], w = r[3];   // (bit_cast applied to a vector element reads element 0)
    return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
  };

  // wave-uniform description of reduction tile `it`
  struct TileAt { int tap, kk0, shift, kt; bool valid; };
  const unsigned t_lim = p.T > 0 ? (unsigned)p.T : 0x7FFFFFFFu;
  // Tiles are visited strictly in order, so their coordinates are advanced, not recomputed: no division per tile.
  // (All of this is wave-uniform and runs on the scalar unit.)
  int g_it = 0, g_tap = AKM ? ztap : 0, g_kk0 = k_begin;
  int g_kt = (AKM && p.T > 0) ? k_begin % p.T : 0;
  auto next_tile = [&]() {
    TileAt t;
    t.tap = g_tap;
    t.kk0 = g_kk0;
    t.shift = (p.T > 0) ? p.shift0 + g_tap * p.shift_step : 0;
    t.kt = g_kt;
    t.valid = g_it < n_iter;
    ++g_it;
    if (AKM) {
      g_kk0 += BK;
      if (p.T > 0) { g_kt += BK; while (g_kt >= p.T) g_kt -= p.T; }
    } else if (++g_tap >= p.taps) {
      // taps innermost: the taps of one reduction tile read the same activation rows shifted by a few rows, so
      // consecutive tiles of a block re-use lines that are still in L1 / L2 (with taps outermost a row came back
      // only after a sweep over the whole reduction, by which time it had left the L2 of a busy chip: 1.8x the
      // compulsory fetch on the 512-channel layers)
      g_tap = 0;
      g_kk0 += BK;
    }
    return t;
  };

  auto load_a = [&](const TileAt& t, int i) {
    if (ALIGNED) {
      if (AKM) {
        const bool ok = t.valid && t.kk0 + a_row[i] < k_end && m0 + a_col[i] < p.M;
        ra[i] = buf_load(rsrc_a, ok ? a_voff[i] : kOob, t.kk0 * p.lda * 4);
      } else {
        const int gm = m0 + a_row[i];
        const bool ok = t.valid && gm < p.M && t.kk0 + a_col[i] < k_end && (unsigned)(a_t[i] + t.shift) < t_lim;
        const unsigned off = a_voff[i] + (unsigned)(t.shift * p.lda * 4);
        ra[i] = buf_load(rsrc_a, ok ? off : kOob, t.kk0 * 4);
      }
      return;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t.valid) {
      if (AKM) {                       // stored [k][m]: vector along m
        const int gk = t.kk0 + a_row[i], gm = m0 + a_col[i];
        if (gk < k_end) {
          const float* src = p.A + (long)gk * p.lda + gm;
          if (gm + 0 < p.M) v.x = src[0];
          if (gm + 1 < p.M) v.y = src[1];
          if (gm + 2 < p.M) v.z = src[2];
          if (gm + 3 < p.M) v.w = src[3];
        }
      } else {                         // stored [m][k]: vector along k, row-shifted
        const int gm = m0 + a_row[i], gk = t.kk0 + a_col[i];
        const bool ok = gm < p.M && (unsigned)(a_t[i] + t.shift) < t_lim;
        if (ok) {
          const float* src = p.A + (long)(gm + t.shift) * p.lda + gk;
          if (gk + 0 < k_end) v.x = src[0];
          if (gk + 1 < k_end) v.y = src[1];
          if (gk + 2 < k_end) v.z = src[2];
          if (gk + 3 < k_end) v.w = src[3];
        }
      }
    }
    ra[i] = v;
  };
  auto load_b = [&](const TileAt& t, int i) {
    // time index of reduction row gk when the reduction runs over (sample, time) rows
    auto row_time = [&](int i2) {
      const int tt = t.kt + b_rt[i2];          // both terms are below T
      return tt >= p.T ? tt - p.T : tt;
    };
    if (ALIGNED) {
      if (BKN && AKM) {
        const int gk = t.kk0 + b_row[i];
        const bool ok = t.valid && gk < k_end && n0 + b_col[i] < p.N && (unsigned)(row_time(i) + t.shift) < t_lim;
        const unsigned off = b_voff[i] + (unsigned)((t.kk0 + t.shift) * p.ldb * 4);   // lane part + wave-uniform part
        rb[i] = buf_load(rsrc_b, ok ? off : kOob, 0);
      } else if (BKN) {
        const bool ok = t.valid && t.kk0 + b_row[i] < k_end && n0 + b_col[i] < p.N;
        rb[i] = buf_load(rsrc_b, ok ? b_voff[i] : kOob, (int)(t.tap * p.tap_stride_b + (long)t.kk0 * p.ldb) * 4);
      } else {
        const bool ok = t.valid && n0 + b_row[i] < p.N && t.kk0 + b_col[i] < k_end;
        rb[i] = buf_load(rsrc_b, ok ? b_voff[i] : kOob, (int)(t.tap * p.tap_stride_b + t.kk0) * 4);
      }
      return;
    }
    const float* Bt = p.B + (AKM ? 0 : (long)t.tap * p.tap_stride_b);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t.valid) {
      if (BKN) {                       // stored [k][n]: vector along n; rows shifted only when reducing over rows
        const int gk = t.kk0 + b_row[i], gn = n0 + b_col[i];
        bool ok = gk < k_end;
        int srow = gk;
        if (AKM && p.T > 0) {
          ok = ok && (unsigned)(row_time(i) + t.shift) < (unsigned)p.T;
          srow = gk + t.shift;
        }
        if (ok) {
          const float* src = Bt + (long)srow * p.ldb + gn;
          if (gn + 0 < p.N) v.x = src[0];
          if (gn + 1 < p.N) v.y = src[1];
          if (gn + 2 < p.N) v.z = src[2];
          if (gn + 3 < p.N) v.w = src[3];
        }
      } else {                         // stored [n][k]: vector along k
        const int gn = n0 + b_row[i], gk = t.kk0 + b_col[i];
        if (gn < p.N) {
          const float* src = Bt + (long)gn * p.ldb + gk;
          if (gk + 0 < k_end) v.x = src[0];
          if (gk + 1 < k_end) v.y = src[1];
          if (gk + 2 < k_end) v.z = src[2];
          if (gk + 3 < k_end) v.w = src[3];
        }
      }
    }
    rb[i] = v;
  };
  auto store_a = [&](int buf, int i) {
    float* Ad = As + buf * BK * LDA;
    if (AKM) *reinterpret_cast<float4*>(Ad + a_row[i] * LDA + a_col[i]) = ra[i];
    else {
      Ad[(a_col[i] + 0) * LDA + a_row[i]] = ra[i].x;
      Ad[(a_col[i] + 1) * LDA + a_row[i]] = ra[i].y;
      Ad[(a_col[i] + 2) * LDA + a_row[i]] = ra[i].z;
      Ad[(a_col[i] + 3) * LDA + a_row[i]] = ra[i].w;
    }
  };
  auto store_b = [&](int buf, int i) {
    float* Bd = Bs + buf * BK * LDB;
    if (BKN) *reinterpret_cast<float4*>(Bd + b_row[i] * LDB + b_col[i]) = rb[i];
    else {
      Bd[(b_col[i] + 0) * LDB + b_row[i]] = rb[i].x;
      Bd[(b_col[i] + 1) * LDB + b_row[i]] = rb[i].y;
      Bd[(b_col[i] + 2) * LDB + b_row[i]] = rb[i].z;
      Bd[(b_col[i] + 3) * LDB + b_row[i]] = rb[i].w;
    }
  };
  auto load_tiles = [&]() {
    const TileAt t = next_tile();
#pragma unroll
    for (int i = 0; i < A_F4; ++i) load_a(t, i);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) load_b(t, i);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) store_a(buf, i);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) store_b(buf, i);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline over a ring of three LDS tiles.  While tile `it` is multiplied, the registers holding tile
  // it+1 (loaded one iteration earlier) are written to the next ring slot, the global loads of tile it+2 are issued
  // into the freed registers, and - after the single barrier in the middle of the iteration - the first fragments
  // of tile it+1 are read, so the matrix pipe never waits on a load, an LDS write or a fragment read at an
  // iteration boundary.  Ring slot (it+1)%3 was last read as tile it-2, which every wave finished before it
  // passed the barrier of iteration it-1.
  load_tiles();
  store_tiles(0);
  load_tiles();
  __syncthreads();

  const int a_frag = wm * (BM / 2) + (lane & 31);
  const int b_frag = wn * (BN / 2) + (lane & 31);
  const int khalf = lane >> 5;
  constexpr int KS = BK / 2;          // MFMA k-steps per tile (even, so the fragment double buffer lines up)
  static_assert(KS % 2 == 0 && KS >= 4, "tile depth");
  constexpr int PPS = (A_F4 + B_F4 + KS / 2 - 1) / (KS / 2);   // staging pieces per k-step of the first half-tile

  float a[2][TM], b[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[0][i] = As[khalf * LDA + a_frag + i * 32];
#pragma unroll
  for (int j = 0; j < TN; ++j) b[0][j] = Bs[khalf * LDB + b_frag + j * 32];

  // One reduction tile.  The ring slot is a compile-time constant (the loop below is unrolled three times), so every
  // LDS address of the body is a lane-constant base plus an immediate offset: no vector instruction is spent on
  // ring pointers.  Vector instructions take issue slots from the MFMAs of the same SIMD: measured, 24 extra
  // v_add per tile cost 3 % of the C = 512 convolutions' rate.
  // Row reductions can deliver the column sums of A on the side (a linear layer's bias gradient next to its weight
  // gradient): the blocks of the first tile column add up the A tile they stage anyway, one thread per column.
  const bool colsum_on = AKM && p.a_colsum != nullptr && tile_n == 0 && ztap == 0 && tid < BM;
  float a_csum = 0.f;
  auto tile_step = [&](auto slot, const int it) {
    constexpr int buf = decltype(slot)::value;
    constexpr int nbuf = (buf + 1 == kRing) ? 0 : buf + 1;
    if (AKM && colsum_on) {      // slot `buf` holds tile `it` (zero beyond the reduction range) until the next barrier
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) a_csum += As[(buf * BK + kk) * LDA + tid];
    }
    const float* Ac = As + buf * BK * LDA + a_frag;
    const float* Bc = Bs + buf * BK * LDB + b_frag;
    const float* An = As + nbuf * BK * LDA + a_frag;
    const float* Bn = Bs + nbuf * BK * LDB + b_frag;
    const bool more = it + 1 < n_iter;
    const TileAt t2 = next_tile();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < KS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = Ac[(2 * (s + 1) + khalf) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = Bc[(2 * (s + 1) + khalf) * LDB + j * 32];
      } else if (more) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = An[khalf * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = Bn[khalf * LDB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the fragment prefetch ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      // staging work of this k-step, issued behind its MFMAs: piece q of tile it+1 goes from registers to the next
      // ring slot and the same registers are reloaded with piece q of tile it+2 (a full iteration ahead of its use)
      if (s < KS / 2) {
#pragma unroll
        for (int q = s * PPS; q < (s + 1) * PPS; ++q) {
          if (q < A_F4) {
            store_a(nbuf, q);
            load_a(t2, q);
          } else if (q < A_F4 + B_F4) {
            store_b(nbuf, q - A_F4);
            load_b(t2, q - A_F4);
          }
        }
      }
      if (s == KS / 2) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int it = 0;
  for (; it + kRing <= n_iter; it += kRing) {     // whole turns of the ring: a single-exit loop
    tile_step(std::integral_constant<int, 0>{}, it);
    tile_step(std::integral_constant<int, 1>{}, it + 1);
    tile_step(std::integral_constant<int, 2>{}, it + 2);
  }
  if (it < n_iter) {
    tile_step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < n_iter) tile_step(std::integral_constant<int, 1>{}, it + 1);
  }

  if (AKM && colsum_on && m0 + tid < p.M) unsafeAtomicAdd(p.a_colsum + m0 + tid, a_csum);

  // ------------------------------------------------------------ epilogue
  const EpiArgs e = make_epi_args(p);
  const DropoutP bn_drop = BNS ? dropout_resolve(p.bn_drop) : DropoutP{0.f, 0, nullptr};
  // acc register r of lane l holds C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of its 32x32 tile.
  float* Cz = e.C + (AKM ? (long)ztap * e.tap_stride_c : 0);
  const DropoutP drop = dropout_resolve(e.drop);
  const int drop_group = e.drop_group;
  const bool atomic_out = AKM;   // row-reduction outputs always accumulate (C += ...)
  // The output dropout / ReLU-mask options get an epilogue instance of their own, chosen by a block-uniform test:
  // with the tests inside the element loop every store of the common case sat behind two extra branches, which
  // cost the short-reduction convolutions 5-20 %.
  auto epilogue = [&](auto extra_c) {
  constexpr bool EXTRA = decltype(extra_c)::value;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    const bool col_ok = col < e.N;
    if (e.colstats != nullptr) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lane < 32 && col_ok) {
        double* cs = e.colstats + (size_t)(tile_m % e.colstat_slots) * 2 * e.N;
        unsafeAtomicAdd(cs + col, (double)s);
        unsafeAtomicAdd(cs + e.N + col, (double)q);
      }
    }
    const float bias = (e.bias != nullptr && col_ok) ? e.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row_base = m0 + wm * (BM / 2) + i * 32 + 4 * khalf;
      // residual values are fetched four rows at a time with loads that do not depend on each other, so that
      // they are in flight together instead of costing one memory round trip per element
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (!atomic_out && e.R != nullptr) {
          if (e.r_bcast_T > 0) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int row = row_base + r4 + 8 * rq;
              const bool ok = row < e.M && col_ok;
              rv[r4] = e.r_scale * e.R[ok ? (long)(row / e.r_bcast_T) * e.ldr + col : 0];
            }
          } else {
            const float* rp = e.R + (long)(row_base + 8 * rq) * e.ldr + col;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const bool ok = row_base + 8 * rq + r4 < e.M && col_ok;
              rv[r4] = ok ? rp[(long)r4 * e.ldr] : 0.f;
            }
          }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int r = rq * 4 + r4;
          const int row = row_base + r4 + 8 * rq;
          if (row < e.M && col_ok) {
            float v = acc[i][j][r] + bias;
            if (atomic_out) {
              unsafeAtomicAdd(Cz + (long)row * e.ldc + col, v);
            } else {
              v += rv[r4];
              if (e.relu) v = fmaxf(v, 0.f);
              if (EXTRA) {
                if (drop.p > 0.f) {
                  const uint64_t di = drop_group > 1 ? (uint64_t)(e.drop_row0 + row) * (e.N / drop_group) + col / drop_group
                                                     : (uint64_t)(e.drop_row0 + row) * e.N + col;
                  v *= dropout_keep(drop, di);
                }
                if (e.relu_mask != nullptr && !(e.relu_mask[(long)row * e.ld_mask + col] > 0.f)) v = 0.f;
              }
              Cz[(long)row * e.ldc + col] = v;
            }
          }
        }
      }
    }
  }
  };
  // Blocks whose whole tile lies inside C (all of them for the encoder convolutions) take the epilogue without bounds tests.
  if constexpr (BNS) {
    epilogue_full<BM, BN, AKM, BNS>(acc, e, Cz, bn_drop, m0, n0, tile_m, wave, lane);   // the host checked: every tile is full, no output dropout / mask (ocean_gemm_bn_rides)
  } else {
    const bool extra = !AKM && (drop.p > 0.f || e.relu_mask != nullptr);
    if (extra) epilogue(std::true_type{});
    else if (m0 + BM <= e.M && n0 + BN <= e.N && e.r_bcast_T <= 0) epilogue_full<BM, BN, AKM, BNS>(acc, e, Cz, bn_drop, m0, n0, tile_m, wave, lane);
    else epilogue(std::false_type{});
  }
}

template <int BM, int BN, int BK, bool AKM, bool BKN>
int launch_cfg(const GemmBatch& b, bool aligned, hipStream_t stream) {
  int tiles = 0, zdim = 1;
  double flops = 0.0, bytes = 0.0;
  for (int q = 0; q < b.n; ++q) {
    const GemmP& p = b.g[q];
    const int t = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    if (t > tiles) tiles = t;
    if (AKM) zdim = p.taps * (p.split_k > 0 ? p.split_k : 1);
    flops += 2.0 * p.M * p.N * (double)p.K * p.taps;
    bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K * p.taps + (double)p.M * p.N * (AKM ? p.taps : 1));
    if (!AKM && p.R != nullptr) bytes += 4.0 * (p.r_bcast_T > 0 ? (double)p.M / p.r_bcast_T : (double)p.M) * p.N;   // residual read
    if (!AKM && p.bn_u != nullptr) bytes += 4.0 * (double)p.M * p.N;                                               // saved u read
  }
  dim3 grid(tiles, b.n, zdim);
  void* tok = nullptr;
  if (ocean_prof_on()) {
    char name[96];
    const bool bns = !AKM && BKN && b.g[0].bn_u != nullptr;
    snprintf(name, sizeof(name), "gemm_f32_kernel<%d,%d,%d,%s,%s,%s%s>", BM, BN, BK, AKM ? "true" : "false", BKN ? "true" : "false",
             aligned ? "true" : "false", bns ? ",bn-sums" : "");
    ocean_prof_begin(name, flops, bytes, stream, &tok);
  }
  static const int dyn_lds = [] { const char* e = getenv("OCEAN_GEMM_DYNLDS"); return e ? atoi(e) : 0; }();   // occupancy experiments
  bool launched = false;
  if constexpr (!AKM && BKN) {       // the data-gradient form may carry BatchNorm-backward sums in its epilogue
    if (b.g[0].bn_u != nullptr) {
      if (aligned) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, AKM, BKN, true, true>), grid, dim3(kThreads), dyn_lds, stream, b);
      else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, AKM, BKN, false, true>), grid, dim3(kThreads), dyn_lds, stream, b);
      launched = true;
    }
  }
  if (launched) {}
  else if (aligned) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, AKM, BKN, true>), grid, dim3(kThreads), dyn_lds, stream, b);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, AKM, BKN, false>), grid, dim3(kThreads), dyn_lds, stream, b);
  ocean_prof_end(tok, stream);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// development knobs (read once): OCEAN_GEMM_WG_CHUNK (minimum
// reduction rows per block of a split row reduction), OCEAN_GEMM_WG_BIGK (row reductions at least this long use
// 128x128 tiles), OCEAN_GEMM_WG_ATOMDIV (split limit sqrt(K / this); 0 = no limit)
inline int wg_chunk() { static const int v = env_int("OCEAN_GEMM_WG_CHUNK", 128); return v < 16 ? 16 : v; }
inline int wg_atom_div() { static const int v = env_int("OCEAN_GEMM_WG_ATOMDIV", 2); return v; }
inline long wg_big_k() { static const int v = env_int("OCEAN_GEMM_WG_BIGK", 8192); return v; }

// tile choice from the largest problem of a batch: wide tiles only when they still fill the chip
void pick_tile(const GemmBatch& b, bool akm, int* bm_out, int* bn_out, int* k_out) {
  int M = 0, N = 0, K = 0;
  long tiles128 = 0;
  for (int q = 0; q < b.n; ++q) {
    if (b.g[q].M > M) M = b.g[q].M;
    if (b.g[q].N > N) N = b.g[q].N;
    if (b.g[q].K > K) K = b.g[q].K;
    tiles128 += (long)ceil_div(b.g[q].M, 128) * ceil_div(b.g[q].N, 128);
  }
  const bool big_m = tiles128 >= 256;
  int bm, bn;
  if (akm && M >= 128 && N >= 128 && K >= wg_big_k() && !(b.n == 1 && ocean_wgrad_small_tiles(b.g[0]))) { bm = 128; bn = 128; }   // long row reductions fill the chip through split-K
  else if (!akm && N > 64 && big_m) { bm = 128; bn = 128; }
  else if (!akm && N <= 64 && (long)ceil_div(M, 128) * b.n >= 256) { bm = 128; bn = 64; }
  else { bm = 64; bn = 64; }
  *bm_out = bm; *bn_out = bn; *k_out = K;
}

template <bool AKM, bool BKN>
int launch_mode(GemmBatch b, bool aligned, hipStream_t stream) {
  int bm, bn, K;
  const int taps = b.g[0].taps;
  pick_tile(b, AKM, &bm, &bn, &K);
  if (AKM) {
    int split = b.g[0].split_k;
    if (split <= 0) {
      // reduction over rows: split so that the launch has >= ~3 blocks per CU
      long blocks = 0;
      for (int q = 0; q < b.n; ++q) blocks += (long)ceil_div(b.g[q].M, bm) * ceil_div(b.g[q].N, bn) * taps;
      long want = (768 + blocks - 1) / blocks;
      const long max_split = (K + wg_chunk() - 1) / wg_chunk();    // >= wg_chunk reduction rows per block
      if (want > max_split) want = max_split;
      // every block of a split adds its tile into the same addresses, and same-address atomics serialise: with s
      // blocks the loop costs ~K/s and the adds ~s, so outputs of one or two tiles stop splitting near sqrt(K)
      if (wg_atom_div() > 0) {
        const long cap = (long)sqrt((double)K / wg_atom_div());
        if (want > cap) want = cap;
      }
      if (want < 1) want = 1;
      split = (int)want;
    }
    for (int q = 0; q < b.n; ++q) b.g[q].split_k = split;
  }
  if (bm == 64) {
    return launch_cfg<64, 64, 16, AKM, BKN>(b, aligned, stream);
  }
  if (bm == 128 && bn == 128) return launch_cfg<128, 128, 16, AKM, BKN>(b, aligned, stream);
  return launch_cfg<128, 64, 16, AKM, BKN>(b, aligned, stream);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int check_problem(const GemmP& p, bool* aligned_out) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.taps < 1) return OCEAN_ERR_BAD_SHAPE;
  if (!p.A || !p.B || !p.C) return OCEAN_ERR_NULL;
  if (p.a_kmajor && !p.b_kn) return OCEAN_ERR_UNSUPPORTED;
  if (p.taps > 1 && p.T <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (p.colstats && p.colstat_slots < 1) return OCEAN_ERR_BAD_SHAPE;
  // vector (16-byte) loads need every row start and every in-range 4-group aligned
  bool aligned = al16(p.A) && al16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0 &&
                 p.tap_stride_b % 4 == 0;
  if (p.a_kmajor) aligned = aligned && p.M % 4 == 0 && p.N % 4 == 0;
  else if (p.b_kn) aligned = aligned && p.K % 4 == 0 && p.N % 4 == 0;
  else aligned = aligned && p.K % 4 == 0;
  // the vector path addresses both operands with 32-bit byte offsets
  const double a_floats = p.a_kmajor ? (double)p.K * p.lda : (double)p.M * p.lda;
  const double b_floats = (p.b_kn ? (double)p.K * p.ldb : (double)p.N * p.ldb) + (double)(p.taps - 1) * (double)p.tap_stride_b;
  if (a_floats >= 5.0e8 || b_floats >= 5.0e8) aligned = false;
  *aligned_out = aligned;
  return OCEAN_OK;
}

}  // namespace

int ocean_launch_gemm_batch(const GemmP* ps, int n, hipStream_t stream) {
  if (!ps || n < 1 || n > kGemmBatchMax) return OCEAN_ERR_BAD_SHAPE;
  GemmBatch b;
  b.n = n;
  bool aligned = true;
  for (int q = 0; q < n; ++q) {
    bool al = false;
    OCEAN_TRY(check_problem(ps[q], &al));
    aligned = aligned && al;
    if (ps[q].a_kmajor != ps[0].a_kmajor || ps[q].b_kn != ps[0].b_kn || ps[q].taps != ps[0].taps) return OCEAN_ERR_UNSUPPORTED;
    if (ps[q].a_kmajor && (ps[q].K != ps[0].K || ps[q].split_k != ps[0].split_k)) return OCEAN_ERR_UNSUPPORTED;
    b.g[q] = ps[q];
  }
  if (rc_collecting()) {
    // a scope of rowchain.h is alive on this thread: row reductions join its batched launch, and - when a stage kernel
    // is being recorded - the other problems become operations of its program; one that does not fit is launched as
    // usual, in order
    int taken = 0;
    while (taken < n && rc_gemm(ps[taken], taken > 0)) ++taken;
    if (taken == n) return OCEAN_OK;
    if (taken > 0) {
      int rest = OCEAN_OK;
      for (int q = taken; q < n && rest == OCEAN_OK; ++q) rest = ocean_launch_gemm_batch(ps + q, 1, stream);
      return rest;
    }
  }
  for (int q = n; q < kGemmBatchMax; ++q) b.g[q] = ps[0];
  if (ocean_gemm_skinny_applies(b)) return ocean_launch_gemm_skinny(b, stream);
  if (aligned && ocean_gemm_nt_applies(b)) return ocean_launch_gemm_nt(b, stream);
  if (aligned && ocean_gemm_tn_applies(b)) return ocean_launch_gemm_tn(b, stream);
  if (ps[0].a_kmajor) return launch_mode<true, true>(b, aligned, stream);
  if (ps[0].b_kn) return launch_mode<false, true>(b, aligned, stream);
  return launch_mode<false, false>(b, aligned, stream);
}

int ocean_launch_gemm(const GemmP& p, hipStream_t stream) { return ocean_launch_gemm_batch(&p, 1, stream); }

bool ocean_gemm_bn_rides(const GemmP& p) {
  if (p.a_kmajor || p.drop.p > 0.f || p.relu_mask != nullptr || p.bias != nullptr) return false;
  if (!p.b_kn) {                       // both operands k-contiguous: only the kernel of gemm_nt.hip carries the sums in that form
    GemmBatch nb;
    nb.n = 1; nb.g[0] = p;
    if (nb.g[0].bn_u == nullptr) nb.g[0].bn_u = p.A;       // eligibility as it will be once the caller has set the bn_* fields
    return ocean_gemm_nt_applies(nb) || ocean_gemm_skinny_applies(nb);     // (the few-tile kernel has no full-tile condition)
  }
  GemmBatch b;
  b.n = 1; b.g[0] = p;
  int bm, bn, K;
  pick_tile(b, false, &bm, &bn, &K);
  if (p.M % bm != 0 || p.N % bn != 0) return false;                      // every tile full: the variant has no bounds tests
  if (p.r_bcast_T > 0 && (p.R == nullptr || p.r_bcast_T % bm != 0)) return false;   // a tile inside one sample
  if ((double)p.M * (double)p.N >= 4294967296.0) return false;          // 32-bit dropout element index
  return true;
}
