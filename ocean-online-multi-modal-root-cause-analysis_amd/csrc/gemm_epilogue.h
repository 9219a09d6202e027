// Pieces shared by the MFMA GEMM kernels (gemm.hip, gemm_nt.hip): the accumulator type, the XCD-aware tile order
// and the epilogue of full tiles.
#pragma once
#include "common.h"
#include <type_traits>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int xcd_swizzle(int orig, int nwg) {
  // Blocks b and b+8 share an XCD (round-robin dispatch).  Give each XCD a
  // contiguous run of tile ids so neighbouring tiles (same A panel) share an L2.
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// Every field an epilogue reads, copied out of the kernel-argument segment once: the compiler re-reads fields of the
// argument block after each global store (it cannot prove the stores do not alias), which put hundreds of dependent
// scalar loads and waits into the unrolled element loop.
struct EpiArgs {
  float* C; int M, N; const float* R; const float* bias; int colstat_slots; double* colstats; DropoutP drop;
  int drop_group, drop_row0, ld_mask, ldc, ldr, r_bcast_T; float r_scale; int relu; const float* relu_mask; long tap_stride_c;
  const float* bn_u; int ld_bn_u; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  double* bn_slots; int bn_nslots;
};
__device__ __forceinline__ EpiArgs make_epi_args(const GemmP& p) {
  return EpiArgs{p.C, p.M, p.N, p.R, p.bias, p.colstat_slots, p.colstats, p.drop, p.drop_group, p.drop_row0, p.ld_mask, p.ldc, p.ldr,
                 p.r_bcast_T, p.r_scale, p.relu, p.relu_mask, (long)p.tap_stride_c,
                 p.bn_u, p.ld_bn_u, p.bn_scale, p.bn_shift, p.bn_mean, p.bn_invstd, p.bn_slots, p.bn_nslots};
}

// Epilogue of a block whose whole BM x BN tile lies inside C: no bounds test; every element is addressed as a
// wave-uniform row pointer (scalar registers, advanced by scalar adds) plus one lane-constant 32-bit offset, so an
// element costs its arithmetic and one store and nothing else.  acc register r of lane l holds
// C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of its 32x32 tile; the four waves sit 2 x 2 over the block tile.
// ATOMIC: the output accumulates (row reductions).  BNS: the BatchNorm-backward column sums described at GemmP::bn_u.
// RBROWS (with BNS): a broadcast residual (r_bcast_T > 0) is looked up per output row, R[row / r_bcast_T], for tiles
// that straddle samples; without it the row is taken to be the same for the whole tile.
template <int BM, int BN, bool ATOMIC, bool BNS, bool RBROWS = false, int TM, int TN>
__device__ __forceinline__ void epilogue_full(f32x16 (&acc)[TM][TN], const EpiArgs& e, float* Cz, const DropoutP bn_drop,
                                              const int m0, const int n0, const int tile_m, const int wave, const int lane) {
  constexpr int WN = 2;
  constexpr bool atomic_out = ATOMIC;
  constexpr bool bn_on = BNS && !ATOMIC;
  const int khalf = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int row_w = m0 + (wave_u / WN) * (BM / 2), col_w = n0 + (wave_u % WN) * (BN / 2);
  const int lcol = lane & 31;
  // byte offsets of this lane inside any 8-row group (they fit 32 bits: a tile spans at most BM rows)
  const uint32_t c_lane = ((uint32_t)(4 * khalf) * (uint32_t)e.ldc + lcol) * 4u;
  const uint32_t r_lane = ((uint32_t)(4 * khalf) * (uint32_t)e.ldr + lcol) * 4u;
  const bool has_r = !atomic_out && e.R != nullptr;
  const float floor_v = e.relu ? 0.f : -__builtin_inff();
  const char* r_w = (const char*)(e.R + ((has_r && e.r_bcast_T <= 0) ? (long)row_w * e.ldr + col_w : 0));
  char* c_w = (char*)(Cz + (long)row_w * e.ldc + col_w);
  const uint32_t c_pitch = (uint32_t)e.ldc * 4u, r_pitch = (uint32_t)e.ldr * 4u;
  // buffer instructions take the wave-uniform part of the address as a scalar offset operand
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(c_w, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r_w), 0, 0x7FFFFFF0, 0x00020000);
  const char* u_w = (const char*)(bn_on ? e.bn_u + (long)row_w * e.ld_bn_u + col_w : e.C);
  const uint32_t u_lane = ((uint32_t)(4 * khalf) * (uint32_t)e.ld_bn_u + lcol) * 4u, u_pitch = (uint32_t)e.ld_bn_u * 4u;
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(u_w), 0, 0x7FFFFFF0, 0x00020000);
  if constexpr (bn_on) {
    // With the BatchNorm sums riding along every output element needs its saved u (and its residual) from memory.
    // The elements go in groups of sixteen (one 32x32 sub-tile), and the loads of group g+1 are issued before group g is worked on, so
    // the block waits for memory once, not once per group; the per-column constants of all column groups are
    // fetched up front for the same reason.  (By now the main loop's staging and fragment registers are free.)
    constexpr int G = 16, GPS = 16 / G, NG = TN * TM * GPS;
    const bool bcast_r = has_r && e.r_bcast_T > 0;
    // row / r_bcast_T for rows below 2^24: the fp32 quotient is off by one at most, and is put right
    const float rcp_T = bcast_r ? 1.0f / (float)e.r_bcast_T : 0.f;
    auto sample_of = [&](uint32_t row) {
      uint32_t q = (uint32_t)((float)row * rcp_T);
      const int rem = (int)(row - q * (uint32_t)e.r_bcast_T);
      if (rem < 0) --q; else if (rem >= e.r_bcast_T) ++q;
      return q;
    };
    float sc_[TN], sh_[TN], mu_[TN], is_[TN], rb_[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = col_w + j * 32 + lcol;
      sc_[j] = e.bn_scale[c]; sh_[j] = e.bn_shift[c]; mu_[j] = e.bn_mean[c]; is_[j] = e.bn_invstd[c];
      // residual broadcast from one row per sample: the host sends such a launch here only when a tile lies inside one
      // sample (T % BM == 0), so the row is block-uniform and the value depends on the column alone
      rb_[j] = (bcast_r && !RBROWS) ? e.r_scale * e.R[(long)(m0 / e.r_bcast_T) * e.ldr + c] : 0.f;
    }
    float rv[2][G], uv[2][G];
    auto issue = [&](const int gi, const int buf) {
      const int j = gi / (TM * GPS), i = (gi / GPS) % TM, r0 = (gi % GPS) * G;
#pragma unroll
      for (int x = 0; x < G; ++x) {
        const int r = r0 + x;
        const uint32_t roff = (uint32_t)(i * 32 + 8 * (r >> 2) + (r & 3));
        if (RBROWS && bcast_r) rv[buf][x] = e.R[(size_t)sample_of((uint32_t)row_w + roff + 4u * (uint32_t)khalf) * e.ldr + (col_w + j * 32 + lcol)];
        else rv[buf][x] = bcast_r ? rb_[j] : (has_r ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)r_lane, (int)(roff * r_pitch + (uint32_t)(j * 128)), 0)) : 0.f);
        uv[buf][x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_u, (int)u_lane, (int)(roff * u_pitch + (uint32_t)(j * 128)), 0));
      }
    };
    float bn_sg = 0.f, bn_sx = 0.f;
    issue(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int buf = gi & 1;
      const int j = gi / (TM * GPS), i = (gi / GPS) % TM, r0 = (gi % GPS) * G;
      if (gi + 1 < NG) issue(gi + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t bn_idx_lane = 4u * (uint32_t)khalf * (uint32_t)e.N + (uint32_t)(col_w + j * 32 + lcol);
#pragma unroll
      for (int x = 0; x < G; ++x) {
        const int r = r0 + x;
        const uint32_t roff = (uint32_t)(i * 32 + 8 * (r >> 2) + (r & 3));
        // (the looked-up broadcast row is scaled here, not where its load is issued: the product would wait for the load)
        const float res = (RBROWS && bcast_r) ? e.r_scale * rv[buf][x] : rv[buf][x];
        const float out = fmaxf(acc[i][j][r] + res, floor_v);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out), rs_c, (int)c_lane, (int)(roff * c_pitch + (uint32_t)(j * 128)), 0);
        float g = (fmaf(uv[buf][x], sc_[j], sh_[j]) > 0.f) ? out : 0.f;
        if (bn_drop.p > 0.f) {
          // element index row * N + col; the host sends a launch here only when M * N < 2^32
          g *= dropout_keep_lo(bn_drop, ((uint32_t)row_w + roff) * (uint32_t)e.N + bn_idx_lane);
        }
        bn_sg += g;
        bn_sx += g * (uv[buf][x] - mu_[j]) * is_[j];
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((gi + 1) % (TM * GPS) == 0) {        // column group j is complete
        bn_sg += __shfl_xor(bn_sg, 32);
        bn_sx += __shfl_xor(bn_sx, 32);
        if (lane < 32) {
          double* bs = e.bn_slots + (size_t)(tile_m % e.bn_nslots) * 2 * e.N + col_w + j * 32;
          unsafeAtomicAdd(bs + lcol, (double)bn_sg);
          unsafeAtomicAdd(bs + e.N + lcol, (double)bn_sx);
        }
        bn_sg = 0.f; bn_sx = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col_u = col_w + j * 32;
      if (e.colstats != nullptr) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (lane < 32) {
          double* cs = e.colstats + (size_t)(tile_m % e.colstat_slots) * 2 * e.N + col_u;
          unsafeAtomicAdd(cs + lcol, (double)s);
          unsafeAtomicAdd(cs + e.N + lcol, (double)q);
        }
      }
      const float bias = e.bias != nullptr ? e.bias[col_u + lcol] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // the sixteen residual values of a 32x32 sub-tile are requested together: one memory round trip, not four
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        if (has_r) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rs_r, (int)r_lane, (int)((uint32_t)(i * 32 + 8 * (r >> 2) + (r & 3)) * r_pitch + (uint32_t)(j * 128)), 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t c_soff = (uint32_t)(i * 32 + 8 * (r >> 2) + (r & 3)) * c_pitch + (uint32_t)(j * 128);
          const float v = acc[i][j][r] + bias;
          if (atomic_out) unsafeAtomicAdd((float*)(c_w + (size_t)c_soff + c_lane), v);
          else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(v + rv[r], floor_v)), rs_c, (int)c_lane, (int)c_soff, 0);
        }
      }
    }
  }
}

}  // namespace
