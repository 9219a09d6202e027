// The stage kernel (a chain of row-local layers in one launch) and the batched row-reduction kernel behind rowchain.h.
//
// rowchain_kernel: block t carries samples [16 t, 16 t + 16) through every operation of the program.  An operation's
// matrices have `rps` rows per sample (1, or 3 for the fusion stage's token-major [B*3, .] matrices), so the block's rows
// are [16 t rps, (16 t + 16) rps): up to three 16-row tiles.
//
// Activations travel between operations through LDS.  A dependent write -> read through memory costs 1-2 us inside one
// block (store acknowledged, barrier, load from L2) - as much as a whole small layer launched on its own - so the
// tile an operation produces is written to an LDS slot the host planned (and, from there or alongside, to its place in
// the workspace, where the backward pass and later launches find it: stores nobody waits for), and the operations that
// consume it read the slot.  The host plans the slots while it records the program: a ring allocator over the tile
// arena, the tiles the running operation touches pinned; a tile that does not fit (or an input nobody kept) is read
// from / written to memory, with a full barrier in front of the reader.  Linear layers run on v_mfma_f32_16x16x4_f32
// (exact fp32): the eight waves split the (layer, 16-column tile) items of a group of up to three independent layers; a
// lane's four consecutive reduction elements of its activation row and of its weight row go straight into the operand
// registers of four MFMAs, both requested eight 16-deep steps at a time.  Row operations give one wave per row / sample.
//
// rc_wgrad_kernel: all weight gradients dW += dy^T x (and bias gradients db += column sums of dy) of a stage in one
// launch: one block per 32x32 tile of one layer's dW, its four waves splitting the batch rows (32x32x2 MFMAs on
// operands loaded straight from memory), partial tiles summed through LDS and added to dW by that one block - no atomics.
#include "rowchain.h"
#include "rowbodies.h"
#include <cstddef>
#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileSamples = 16;
constexpr int kRcWaves = 8, kRcThreads = kRcWaves * 64;
constexpr int kRcMaxOps = 21;
constexpr int kArgs = 8;
constexpr int kLdsFloats = 14336;             // 56 KB of activation tiles (the program itself takes 4 KB of LDS)
constexpr int kMaxTileFloats = 6400;          // larger tiles are not kept in LDS

enum RcKind : int {
  RC_LIN = 0, RC_LN_FWD, RC_LN_BWD, RC_L2N_FWD, RC_L2N_BWD, RC_CROSS_FWD, RC_CROSS_BWD, RC_GATE_FWD, RC_GATE_BWD,
  RC_MHA3_FWD, RC_MHA3_BWD, RC_MEAN_MID, RC_SIGMOID, RC_SIGMOID_BWD, RC_RELU_BWD, RC_COPY2D
};
// RC_NOSYNC: independent of the operations since the last barrier.  RC_FULLSYNC: reads memory that an earlier operation
// of this program wrote (the barrier in front also waits for this block's stores).
enum RcFlag : int { RC_NOSYNC = 1, RC_RELU = 2, RC_BKN = 4, RC_ACC = 8, RC_FULLSYNC = 16 };

struct RcOp {
  int kind, flags, rps, M;          // M: rows of the operation's matrices (batch rows x rps)
  int N, K, drop_group, drop_row0;
  int ld[kArgs];                    // leading dimension in memory of pointer operand k (where it has one)
  short loff[kArgs];                // LDS slot of operand k's tile in units of 4 floats, or -1: in memory only
  short lld[kArgs];                 // leading dimension of the slot
  short lbs[kArgs];                 // stacked operands: distance between the modalities' slots, units of 4 floats
  float f0, drop_p;
  uint64_t drop_seed;
  const void* p[kArgs];
};
struct RcProgram {
  int n_ops, B;
  const uint64_t* epoch;            // dropout epoch counter shared by every site of the stage (or null)
  long long* trace;                 // diagnostics (ocean_rc_debug_trace): block 0 stamps the clock at every operation, or null
  RcOp ops[kRcMaxOps];
};
static_assert(sizeof(RcProgram) <= 4096, "the program travels as a kernel argument");

__device__ __forceinline__ const float* cpf(const void* p) { return static_cast<const float*>(p); }
__device__ __forceinline__ float* mpf(const void* p) { return const_cast<float*>(static_cast<const float*>(p)); }

// barrier that orders LDS traffic only: stores to memory stay in flight
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Operand k as (pointer, leading dimension) such that element (row r of the operation's matrix, column c) is
// ptr[r * ld + c]: the LDS slot (rebased by the tile's first row) when the host planned one, else memory.
// The rebased slot address lies below the slot (by the rows of the blocks in front of this one) until a row index is
// added back, so it is worked out on the 64-bit flat address: as LDS pointer arithmetic it wraps at 32 bits, and the
// carry of the later 64-bit add then lands in the aperture half of the flat address (a memory fault in every block
// but the first).
struct Opnd { float* ptr; int ld; bool lds; };
__device__ __forceinline__ Opnd opnd(const RcOp& op, const int k, float* lds, const int rows0) {
  if (op.loff[k] >= 0) {
    const uintptr_t slot = reinterpret_cast<uintptr_t>(lds) + 16ull * (uintptr_t)op.loff[k];
    return Opnd{reinterpret_cast<float*>(slot - 4ull * (uintptr_t)((long)rows0 * op.lld[k])), op.lld[k], true};
  }
  return Opnd{mpf(op.p[k]), op.ld[k], false};
}
// one row of an LDS-resident output to its place in memory (by the wave that has just written it)
__device__ __forceinline__ void spill_row(const RcOp& op, const int k, const Opnd& o, const long row, const int width, const int lane) {
  if (!o.lds) return;
  float* g = mpf(op.p[k]) + row * op.ld[k];
  const float* l = o.ptr + row * o.ld;
  for (int c = lane; c < width; c += 64) g[c] = l[c];
}

// ---------------------------------------------------------------- linear layers on a tile of rows
template <int RT>
__device__ __forceinline__ void lin_group(const RcProgram& prog, const int i0, const int g, float* lds, const uint64_t seed_add,
                                          const int b0, const int nb, const int wave, const int lane) {
  constexpr int kChunk = RT == 1 ? 8 : RT == 2 ? 4 : 2;      // 16-deep reduction steps per operand-fragment chunk
  const int rps = prog.ops[i0].rps;
  const int rows0 = b0 * rps, nrows = nb * rps;
  int tiles[3], total = 0;
  for (int j = 0; j < g; ++j) { tiles[j] = (prog.ops[i0 + j].N + 15) >> 4; total += tiles[j]; }
  const int c = lane & 15, q = lane >> 4;
  for (int it = wave; it < total; it += kRcWaves) {
    int j = 0, ct = it;
    while (ct >= tiles[j]) { ct -= tiles[j]; ++j; }
    const RcOp& op = prog.ops[i0 + j];
    const float* __restrict__ Bm = cpf(op.p[1]);
    const int ldb = op.ld[1], N = op.N, K = op.K, K16 = (K + 15) & ~15;
    const bool bkn = (op.flags & RC_BKN) != 0;
    const int n = ct * 16 + c;
    const bool n_ok = n < N;
    const float* brow = bkn ? Bm + (n_ok ? n : 0) + (long)(4 * q) * ldb : Bm + (long)(n_ok ? n : 0) * ldb + 4 * q;
    // activation rows: lane (c, q) reads row c of each 16-row tile, elements k0 + 4 q .. + 3
    const Opnd A = opnd(op, 0, lds, rows0);
    const float* arow[RT];
    bool a_ok[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = rt * 16 + c;
      a_ok[rt] = r < nrows;
      arow[rt] = A.ptr + (long)(rows0 + (a_ok[rt] ? r : 0)) * A.ld + 4 * q;
    }
    auto load_b = [&](const int ch, float4 (&dst)[kChunk]) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int k0 = (ch * kChunk + u) * 16;
        dst[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n_ok && k0 + 4 * q < K) {
          if (bkn) {
            const float* b4 = brow + (long)k0 * ldb;
            dst[u].x = b4[0]; dst[u].y = b4[ldb]; dst[u].z = b4[2 * (long)ldb]; dst[u].w = b4[3 * (long)ldb];
          } else {
            dst[u] = *reinterpret_cast<const float4*>(brow + k0);
          }
        }
      }
    };
    auto load_a = [&](const int ch, float4 (&dst)[kChunk][RT]) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int k0 = (ch * kChunk + u) * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          dst[u][rt] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a_ok[rt] && k0 + 4 * q < K) dst[u][rt] = *reinterpret_cast<const float4*>(arow[rt] + k0);
        }
      }
    };
    f32x4 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const int ch, const float4 (&af)[kChunk][RT], const float4 (&bf)[kChunk]) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if ((ch * kChunk + u) * 16 < K16) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][rt].x, bf[u].x, acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][rt].y, bf[u].y, acc[rt][1], 0, 0, 0);
            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][rt].z, bf[u].z, acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][rt].w, bf[u].w, acc[rt][1], 0, 0, 0);
          }
        }
      }
    };
    const int nch = (K16 / 16 + kChunk - 1) / kChunk;
    float4 bf[kChunk], af[kChunk][RT];
    // the bias and the residual / mask values are requested with the first chunk, not after the last MFMA
    const float* __restrict__ bias = cpf(op.p[3]);
    const float* R = cpf(op.p[4]);
    const float* __restrict__ mask = cpf(op.p[5]);
    load_b(0, bf);
    load_a(0, af);
    const float bv = (bias != nullptr && n_ok) ? bias[n] : 0.f;
    float rv[RT][4], mv[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = rt * 16 + 4 * q + v;
        const bool ok = row < nrows && n_ok;
        rv[rt][v] = (R != nullptr && ok) ? R[(long)(rows0 + row) * op.ld[4] + n] : 0.f;
        mv[rt][v] = (mask != nullptr && ok) ? mask[(long)(rows0 + row) * op.ld[5] + n] : 1.f;
      }
    for (int ch = 0; ch < nch; ++ch) {
      compute(ch, af, bf);
      if (ch + 1 < nch) { load_b(ch + 1, bf); load_a(ch + 1, af); }
    }
    // lane (c, q) holds rows 4 q + v (v = 0..3) of column n of each 16-row tile
    float* C = mpf(op.p[2]);
    const int ldc = op.ld[2];
    const Opnd CL = opnd(op, 2, lds, rows0);
    const bool relu = (op.flags & RC_RELU) != 0;
    DropoutP dp;
    dp.p = op.drop_p; dp.seed = op.drop_seed + (op.drop_p > 0.f ? seed_add : 0); dp.epoch = nullptr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = rt * 16 + 4 * q + v;
        if (row >= nrows || !n_ok) continue;
        const long grow = rows0 + row;
        float val = acc[rt][0][v] + acc[rt][1][v] + bv + rv[rt][v];
        if (relu) val = fmaxf(val, 0.f);
        if (dp.p > 0.f) {
          const uint64_t di = op.drop_group > 1 ? (uint64_t)(op.drop_row0 + grow) * (N / op.drop_group) + n / op.drop_group
                                                : (uint64_t)(op.drop_row0 + grow) * N + n;
          val *= dropout_keep(dp, di);
        }
        if (!(mv[rt][v] > 0.f)) val = 0.f;
        C[grow * ldc + n] = val;
        if (CL.lds) CL.ptr[grow * CL.ld + n] = val;
      }
    }
  }
}

__global__ __launch_bounds__(kRcThreads) void rowchain_kernel(const RcProgram prog_arg) {
  // The program is read once, by all threads at once, into LDS: read field by field from the argument segment each
  // operation's descriptor costs dependent misses of the scalar cache (the segment is touched by nobody else).
  __shared__ __align__(16) RcProgram prog;
  __shared__ __align__(16) float lds[kLdsFloats];
  {
    typedef const __attribute__((address_space(4))) uint32_t* KernargWords;
    const KernargWords ka = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t* dst = reinterpret_cast<uint32_t*>(&prog);
    const int words = (int)(offsetof(RcProgram, ops) / 4) + prog_arg.n_ops * (int)(sizeof(RcOp) / 4);
    for (int w = threadIdx.x; w < words; w += kRcThreads) dst[w] = ka[w];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b0 = blockIdx.x * kTileSamples;
  const int nb = min(prog.B - b0, kTileSamples);            // samples of this block
  if (prog.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    prog.trace[2 * prog.n_ops + 2] = (long long)clock64();
    prog.trace[2 * prog.n_ops + 4] = (long long)wall_clock64();
  }
  const int B = prog.B;
  uint64_t seed_add = 0;
  if (prog.epoch != nullptr) seed_add = prog.epoch[0] * 0xD1B54A32D192ED03ull;
  for (int i = 0; i < prog.n_ops; ++i) {
    const RcOp& op = prog.ops[i];
    if (prog.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
      prog.trace[2 * i] = (long long)op.kind * 1000 + op.N;
      prog.trace[2 * i + 1] = (long long)wall_clock64();
    }
    if (i > 0 && !(op.flags & RC_NOSYNC)) {
      if (op.flags & RC_FULLSYNC) __syncthreads(); else lds_barrier();
    }
    const int rps = op.rps;
    const int rows0 = b0 * rps, nrows = nb * rps;
    if (op.kind == RC_LIN) {
      const int g = op.ld[7];                 // layers of the group (ld[7] of the group's first layer)
      if (nrows <= 16) lin_group<1>(prog, i, g, lds, seed_add, b0, nb, wave, lane);
      else if (nrows <= 32) lin_group<2>(prog, i, g, lds, seed_add, b0, nb, wave, lane);
      else lin_group<3>(prog, i, g, lds, seed_add, b0, nb, wave, lane);
      i += g - 1;
      continue;
    }
    DropoutP dp;
    dp.p = op.drop_p; dp.seed = op.drop_seed + (op.drop_p > 0.f ? seed_add : 0); dp.epoch = nullptr;
    switch (op.kind) {
      case RC_LN_FWD: {           // p: x, gamma, beta, y, stats
        const Opnd x = opnd(op, 0, lds, rows0), y = opnd(op, 3, lds, rows0);
        for (int r = wave; r < nrows; r += kRcWaves) {
          rowbody::ln_fwd_row(x.ptr, x.ld, cpf(op.p[1]), cpf(op.p[2]), y.ptr, y.ld, mpf(op.p[4]), rows0 + r,
                              (uint64_t)op.drop_row0 + rows0 + r, op.N, op.f0, dp, lane);
          spill_row(op, 3, y, rows0 + r, op.N, lane);
        }
        break;
      }
      case RC_LN_BWD: {           // p: dy, x, stats, gamma, dx, dgamma, dbeta
        float pg[rowbody::kMaxColsPerLane], pb[rowbody::kMaxColsPerLane];
#pragma unroll
        for (int k = 0; k < rowbody::kMaxColsPerLane; ++k) { pg[k] = 0.f; pb[k] = 0.f; }
        const Opnd dy = opnd(op, 0, lds, rows0), dx = opnd(op, 4, lds, rows0);
        for (int r = wave; r < nrows; r += kRcWaves) {
          rowbody::ln_bwd_row(dy.ptr, dy.ld, cpf(op.p[1]), op.ld[1], cpf(op.p[2]), cpf(op.p[3]), dx.ptr, dx.ld, rows0 + r,
                              (uint64_t)op.drop_row0 + rows0 + r, op.N, dp, lane, pg, pb);
          spill_row(op, 4, dx, rows0 + r, op.N, lane);
        }
        if (wave < nrows) {
          float* dgamma = mpf(op.p[5]);
          float* dbeta = mpf(op.p[6]);
#pragma unroll
          for (int k = 0; k < rowbody::kMaxColsPerLane; ++k) {
            const int c = lane + 64 * k;
            if (c < op.N) {
              if (dgamma) unsafeAtomicAdd(dgamma + c, pg[k]);
              if (dbeta) unsafeAtomicAdd(dbeta + c, pb[k]);
            }
          }
        }
        break;
      }
      case RC_L2N_FWD: {          // p: x, y, norms
        const Opnd x = opnd(op, 0, lds, rows0);
        for (int r = wave; r < nrows; r += kRcWaves)
          rowbody::l2norm_fwd_row(x.ptr, x.ld, mpf(op.p[1]), op.ld[1], mpf(op.p[2]), rows0 + r, op.N, op.f0, lane);
        break;
      }
      case RC_L2N_BWD: {          // p: dy, y, norms, dx
        const Opnd dx = opnd(op, 3, lds, rows0);
        for (int r = wave; r < nrows; r += kRcWaves) {
          rowbody::l2norm_bwd_row(cpf(op.p[0]), op.ld[0], cpf(op.p[1]), op.ld[1], cpf(op.p[2]), dx.ptr, dx.ld, rows0 + r, op.N, op.f0, lane);
          spill_row(op, 3, dx, rows0 + r, op.N, lane);
        }
        break;
      }
      case RC_CROSS_FWD: {        // p: Nn (stacked), mw, G (stacked), cw
        const int W = op.N;
        const Opnd nn = opnd(op, 0, lds, b0), gg = opnd(op, 2, lds, b0);
        const long bsn = nn.lds ? 4L * op.lbs[0] : (long)B * W, bsg = gg.lds ? 4L * op.lbs[2] : (long)B * W;
        for (int s = wave; s < nb; s += kRcWaves) {
          rowbody::cross_fwd_row(nn.ptr, nn.ld, bsn, cpf(op.p[1]), gg.ptr, gg.ld, bsg, mpf(op.p[3]), W, b0 + s, lane);
          if (gg.lds)
            for (int m = 0; m < 3; ++m) {
              float* g = mpf(op.p[2]) + ((long)m * B + b0 + s) * W;
              const float* l = gg.ptr + m * bsg + (long)(b0 + s) * gg.ld;
              for (int c = lane; c < W; c += 64) g[c] = l[c];
            }
        }
        break;
      }
      case RC_CROSS_BWD: {        // p: dG (stacked), Nn (stacked, memory), cw, mw, dN (stacked), dpi
        const int W = op.N;
        const Opnd dg = opnd(op, 0, lds, b0), dn = opnd(op, 4, lds, b0);
        const long bsg = dg.lds ? 4L * op.lbs[0] : (long)B * W, bsdn = dn.lds ? 4L * op.lbs[4] : (long)B * W;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int s = wave; s < nb; s += kRcWaves) {
          float d0, d1, d2;
          rowbody::cross_bwd_row(dg.ptr, dg.ld, bsg, cpf(op.p[1]), W, (long)B * W, cpf(op.p[2]), cpf(op.p[3]), dn.ptr, dn.ld, bsdn, W,
                                 b0 + s, lane, d0, d1, d2);
          t0 += d0; t1 += d1; t2 += d2;
          if (dn.lds)
            for (int m = 0; m < 3; ++m) {
              float* g = mpf(op.p[4]) + ((long)m * B + b0 + s) * W;
              const float* l = dn.ptr + m * bsdn + (long)(b0 + s) * dn.ld;
              for (int c = lane; c < W; c += 64) g[c] = l[c];
            }
        }
        if (wave < nb && lane < 3) unsafeAtomicAdd(mpf(op.p[5]) + lane, lane == 0 ? t0 : lane == 1 ? t1 : t2);
        break;
      }
      case RC_GATE_FWD: {         // p: hid (stacked), w2, b2, G (stacked), ig, cat
        const int W = op.N, Wh = op.K;
        const Opnd hd = opnd(op, 0, lds, b0), gg = opnd(op, 3, lds, b0), ct = opnd(op, 5, lds, b0);
        const long bsh = hd.lds ? 4L * op.lbs[0] : (long)B * Wh, bsg = gg.lds ? 4L * op.lbs[3] : (long)B * W;
        for (int j = wave; j < 3 * nb; j += kRcWaves) {
          const int m = j / nb, b = b0 + j % nb;
          rowbody::gate_fwd_row(hd.ptr, hd.ld, bsh, cpf(op.p[1]), cpf(op.p[2]), gg.ptr, gg.ld, bsg, mpf(op.p[4]), ct.ptr, ct.ld, B, W, Wh, m, b, lane);
          if (ct.lds) {
            float* g = mpf(op.p[5]) + (long)b * op.ld[5] + m * W;
            const float* l = ct.ptr + (long)b * ct.ld + m * W;
            for (int c = lane; c < W; c += 64) g[c] = l[c];
          }
        }
        break;
      }
      case RC_GATE_BWD: {         // p: dcat, G (stacked, memory), ig, hid (stacked, memory), w2, dG (stacked), dz, dhid (stacked)
        const int W = op.N, Wh = op.K;
        const Opnd dc = opnd(op, 0, lds, b0), dg = opnd(op, 5, lds, b0), dh = opnd(op, 7, lds, b0);
        const long bsdg = dg.lds ? 4L * op.lbs[5] : (long)B * W, bsdh = dh.lds ? 4L * op.lbs[7] : (long)B * Wh;
        for (int j = wave; j < 3 * nb; j += kRcWaves) {
          const int m = j / nb, b = b0 + j % nb;
          rowbody::gate_bwd_row(dc.ptr, dc.ld, cpf(op.p[1]), W, (long)B * W, cpf(op.p[2]), cpf(op.p[3]), Wh, (long)B * Wh, cpf(op.p[4]),
                                dg.ptr, dg.ld, bsdg, mpf(op.p[6]), dh.ptr, dh.ld, bsdh, B, W, Wh, m, b, lane);
          if (dg.lds) {
            float* g = mpf(op.p[5]) + ((long)m * B + b) * W;
            const float* l = dg.ptr + m * bsdg + (long)b * dg.ld;
            for (int c = lane; c < W; c += 64) g[c] = l[c];
          }
          if (dh.lds) {
            float* g = mpf(op.p[7]) + ((long)m * B + b) * Wh;
            const float* l = dh.ptr + m * bsdh + (long)b * dh.ld;
            for (int c = lane; c < Wh; c += 64) g[c] = l[c];
          }
        }
        break;
      }
      case RC_MHA3_FWD: {
        const int H = op.K, hlog = op.drop_group;
        for (int s = wave; s < nb; s += kRcWaves) {
          if (op.N == 256) rowbody::mha3_fwd_sample<4>(cpf(op.p[0]), mpf(op.p[1]), mpf(op.p[2]), H, hlog, dp, b0 + s, lane);
          else if (op.N == 128) rowbody::mha3_fwd_sample<2>(cpf(op.p[0]), mpf(op.p[1]), mpf(op.p[2]), H, hlog, dp, b0 + s, lane);
          else rowbody::mha3_fwd_sample<1>(cpf(op.p[0]), mpf(op.p[1]), mpf(op.p[2]), H, hlog, dp, b0 + s, lane);
        }
        break;
      }
      case RC_MHA3_BWD: {
        const int H = op.K, hlog = op.drop_group;
        for (int s = wave; s < nb; s += kRcWaves) {
          if (op.N == 256) rowbody::mha3_bwd_sample<4>(cpf(op.p[0]), cpf(op.p[1]), cpf(op.p[2]), mpf(op.p[3]), H, hlog, dp, b0 + s, lane);
          else if (op.N == 128) rowbody::mha3_bwd_sample<2>(cpf(op.p[0]), cpf(op.p[1]), cpf(op.p[2]), mpf(op.p[3]), H, hlog, dp, b0 + s, lane);
          else rowbody::mha3_bwd_sample<1>(cpf(op.p[0]), cpf(op.p[1]), cpf(op.p[2]), mpf(op.p[3]), H, hlog, dp, b0 + s, lane);
        }
        break;
      }
      case RC_MEAN_MID: {         // out[g, c] = mean over t < n of a[g, t, c];  p: a, out
        const float* a = cpf(op.p[0]);
        float* out = mpf(op.p[1]);
        const Opnd ol = opnd(op, 1, lds, b0);
        const int n = op.K, C = op.N;
        for (int e = threadIdx.x; e < nb * C; e += kRcThreads) {
          const int g = b0 + e / C, c = e % C;
          float s = 0.f;
          for (int t = 0; t < n; ++t) s += a[((long)g * n + t) * C + c];
          s /= (float)n;
          out[(long)g * C + c] = s;
          if (ol.lds) ol.ptr[(long)g * ol.ld + c] = s;
        }
        break;
      }
      case RC_SIGMOID: {
        const Opnd x = opnd(op, 0, lds, rows0);
        float* y = mpf(op.p[1]);
        const int N = op.N;
        for (int e = threadIdx.x; e < nrows * N; e += kRcThreads) {
          const long r = rows0 + e / N;
          const int c = e % N;
          y[r * N + c] = 1.0f / (1.0f + __expf(-x.ptr[r * x.ld + c]));
        }
        break;
      }
      case RC_SIGMOID_BWD: {      // out = dlogits_in + dprobs * p * (1 - p);  p: dlogits_in, dprobs, p, out
        const float* dl = cpf(op.p[0]);
        const float* dpr = cpf(op.p[1]);
        const float* pr = cpf(op.p[2]);
        float* out = mpf(op.p[3]);
        const Opnd ol = opnd(op, 3, lds, rows0);
        const int N = op.N;
        for (int e = threadIdx.x; e < nrows * N; e += kRcThreads) {
          const long r = rows0 + e / N;
          const int c = e % N;
          const long gi = r * N + c;
          float v = dl ? dl[gi] : 0.f;
          if (dpr) v += dpr[gi] * pr[gi] * (1.0f - pr[gi]);
          out[gi] = v;
          if (ol.lds) ol.ptr[r * ol.ld + c] = v;
        }
        break;
      }
      case RC_RELU_BWD: {
        float* dy = mpf(op.p[0]);
        const float* y = cpf(op.p[1]);
        const int N = op.N;
        for (int e = threadIdx.x; e < nrows * N; e += kRcThreads) {
          const long r = rows0 + e / N;
          const int c = e % N;
          if (!(y[r * op.ld[1] + c] > 0.f)) dy[r * op.ld[0] + c] = 0.f;
        }
        break;
      }
      case RC_COPY2D: {
        float* dst = mpf(op.p[0]);
        const float* src = cpf(op.p[1]);
        const int N = op.N;
        const bool acc = (op.flags & RC_ACC) != 0;
        for (int e = threadIdx.x; e < nrows * N; e += kRcThreads) {
          const long r = rows0 + e / N;
          const int c = e % N;
          const float v = src[r * op.ld[1] + c];
          if (acc) dst[r * op.ld[0] + c] += v; else dst[r * op.ld[0] + c] = v;
        }
        break;
      }
      default: break;
    }
  }
  if (prog.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    prog.trace[2 * prog.n_ops] = -1;
    prog.trace[2 * prog.n_ops + 1] = (long long)wall_clock64();
    prog.trace[2 * prog.n_ops + 3] = (long long)clock64();          // shader clock at the end (the start is stamped below)
  }
}

// ---------------------------------------------------------------- batched row reductions
constexpr int kWgMax = 56;
struct WgProb { const float* dy; const float* x; float* dW; float* db; int lddy, ldx, ldw, M, N, K, tile0, pad; };
struct WgProgram { int n, total; WgProb p[kWgMax]; };
static_assert(sizeof(WgProgram) <= 4096, "the problem list travels as a kernel argument");

__global__ __launch_bounds__(256) void rc_wgrad_kernel(const WgProgram prog) {
  __shared__ float part[4][32][33];
  int pi = 0;
  while (pi + 1 < prog.n && (int)blockIdx.x >= prog.p[pi + 1].tile0) ++pi;
  const WgProb& P = prog.p[pi];
  const int t = blockIdx.x - P.tile0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, h = lane >> 5;
  const float* __restrict__ dy = P.dy;
  const int M = P.M, N = P.N, lddy = P.lddy;
  if (P.x == nullptr) {
    // column sums only: db[n0 + c] += sum over rows of dy[., n0 + c], 32 columns per block
    const int n0 = t * 32, col = n0 + (tid & 31);
    float s = 0.f;
    if (col < N)
      for (int r = tid >> 5; r < M; r += 8) s += dy[(long)r * lddy + col];
    float* red = &part[0][0][0];
    red[tid] = s;
    __syncthreads();
    if (tid < 32 && col < N) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[g * 32 + tid];
      P.db[col] += v;
    }
    return;
  }
  const float* __restrict__ x = P.x;
  const int K = P.K, ldx = P.ldx;
  const int tiles_k = (K + 31) >> 5;
  const int n0 = (t / tiles_k) * 32, k0 = (t % tiles_k) * 32;
  const bool a_ok = n0 + lm < N, b_ok = k0 + lm < K;
  const float* ap = dy + n0 + lm;
  const float* bp = x + k0 + lm;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // wave w takes row pairs w, w + 4, ...: rows 2 pair + h
  for (int r = 2 * wave + h; r < M + h; r += 8) {
    const bool r_ok = r < M;
    const float a = (a_ok && r_ok) ? ap[(long)r * lddy] : 0.f;
    const float b = (b_ok && r_ok) ? bp[(long)r * ldx] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // acc register r of lane (lm, h) is C[(r&3) + 8*(r>>2) + 4*h][lm]
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][lm] = acc[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = (tid >> 5) + 8 * i, lc = tid & 31;
    const int n = n0 + lr, k = k0 + lc;
    if (n < N && k < K) {
      const float v = part[0][lr][lc] + part[1][lr][lc] + part[2][lr][lc] + part[3][lr][lc];
      P.dW[(long)n * P.ldw + k] += v;
    }
  }
  // the bias gradient rides with the first column tile of each row band: column sums of dy over 32 of its columns
  if (P.db != nullptr && k0 == 0) {
    __syncthreads();
    const int col = n0 + (tid & 31);
    float s = 0.f;
    if (col < N)
      for (int r = tid >> 5; r < M; r += 8) s += dy[(long)r * lddy + col];
    float* red = &part[0][0][0];
    red[tid] = s;
    __syncthreads();
    if (tid < 32 && col < N) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[g * 32 + tid];
      P.db[col] += v;
    }
  }
}

// ---------------------------------------------------------------- recorder (host)
// A tile of an activation matrix that the running program keeps in LDS: rows [16 t rps, 16 (t + 1) rps) x [0, width) of
// the matrix at `g` with leading dimension `ld`, in a slot of 16 rps rows of `lld` floats at float offset `off`.
struct Tile { const float* g; int ld, rps, width, off, lld, size; unsigned epoch; };
constexpr int kMaxTiles = 24;

struct RcState {
  bool on = false;              // a scope is alive: row reductions are collected
  bool chain = false;           // ... and the other operations are recorded into a stage kernel
  int B = 0;
  hipStream_t st = nullptr;
  int err = OCEAN_OK;
  bool token_major = false;
  int lin_head = -1;            // first operation of the group of linear layers the next independent one may join
  int sync_op = 0;              // the operation that carries the barrier of the current run of independent operations
  bool drop_seen = false;       // a dropout site has been recorded into the current program
  // LDS plan of the current program
  Tile tiles[kMaxTiles];
  int n_tiles = 0;
  int top = 0;                  // ring allocator: next free float of the arena
  unsigned epoch = 1;           // advances at every barrier of the program: tiles touched in the current epoch are pinned
  RcProgram prog;
  WgProgram wg;
};
thread_local RcState g_rc;

// the batched row-reduction launch replaces the per-layer weight-gradient launches up to this many batch rows
int defer_max_rows() {
  static const int v = [] { const char* e = getenv("OCEAN_WGRAD_BATCH_MAX_ROWS"); return e ? atoi(e) : 256; }();
  return v;
}
// The stage kernel itself is opt-in (OCEAN_STAGE_FUSE=1, for batches up to OCEAN_STAGE_FUSE_MAX_ROWS rows): measured on
// MI355X it does not yet beat the per-layer launches it replaces (see rowchain.h).
int chain_max_rows() {
  static const int v = [] {
    const char* on = getenv("OCEAN_STAGE_FUSE");
    if (on == nullptr || atoi(on) == 0) return 0;
    const char* e = getenv("OCEAN_STAGE_FUSE_MAX_ROWS");
    return e ? atoi(e) : 256;
  }();
  return v;
}
bool lds_plan_on() { static const bool v = getenv("OCEAN_STAGE_NO_LDS") == nullptr; return v; }      // measurement switch

void note(int status) { if (status != OCEAN_OK && g_rc.err == OCEAN_OK) g_rc.err = status; }

long long* g_trace = nullptr;          // ocean_rc_debug_trace

void reset_program() {
  RcState& s = g_rc;
  s.prog.n_ops = 0;
  s.prog.epoch = nullptr;
  s.drop_seen = false;
  s.lin_head = -1;
  s.sync_op = 0;
  s.n_tiles = 0;
  s.top = 0;
  s.epoch = 1;
}

void launch_chain() {
  RcState& s = g_rc;
  if (s.prog.n_ops == 0) return;
  s.prog.B = s.B;
  s.prog.trace = g_trace;
  hipLaunchKernelGGL(rowchain_kernel, dim3(ceil_div(s.B, kTileSamples)), dim3(kRcThreads), 0, s.st, s.prog);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) note(OCEAN_ERR_HIP + (int)e);
  reset_program();
}

void launch_wgrads() {
  RcState& s = g_rc;
  if (s.wg.n == 0) return;
  hipLaunchKernelGGL(rc_wgrad_kernel, dim3(s.wg.total), dim3(256), 0, s.st, s.wg);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) note(OCEAN_ERR_HIP + (int)e);
  s.wg.n = 0;
  s.wg.total = 0;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
void drop_tile(int i) { RcState& s = g_rc; s.tiles[i] = s.tiles[--s.n_tiles]; }

// a fresh operation slot (the chain is launched first when the program is full), or null when `rows` does not map onto
// the scope's samples (one row per sample, or three token-major ones inside an RcTokenMajor)
RcOp* new_op(int kind, int rows, bool independent) {
  RcState& s = g_rc;
  if (rows != s.B && !(s.token_major && rows == 3 * s.B)) return nullptr;
  if (s.prog.n_ops == kRcMaxOps) launch_chain();
  RcOp* op = &s.prog.ops[s.prog.n_ops];
  memset(op, 0, sizeof(RcOp));
  for (int k = 0; k < kArgs; ++k) op->loff[k] = -1;
  op->kind = kind;
  op->rps = rows / s.B;
  op->M = rows;
  op->flags = (independent && s.prog.n_ops > 0) ? RC_NOSYNC : 0;
  op->drop_group = 1;
  if (!(op->flags & RC_NOSYNC)) {
    ++s.epoch; s.sync_op = s.prog.n_ops;
    for (int i = 0; i < s.n_tiles;)           // nameless slots of earlier epochs are free again
      if (s.tiles[i].g == nullptr) drop_tile(i); else ++i;
  }
  return op;
}
void commit() { ++g_rc.prog.n_ops; }

// the barrier in front of the current run of operations must also wait for this block's stores to memory
void need_full_sync() {
  RcState& s = g_rc;
  if (s.sync_op > 0) s.prog.ops[s.sync_op].flags |= RC_FULLSYNC;
}

// ---- LDS plan.  Inputs are looked up first (a hit pins the tile for the current epoch), then outputs are placed.
Tile* find_tile(const float* g, int ld, int rps, int width) {
  RcState& s = g_rc;
  for (int i = 0; i < s.n_tiles; ++i) {
    Tile& t = s.tiles[i];
    if (t.g != nullptr && t.g == g && t.ld == ld && t.rps == rps && t.width >= width) return &t;
  }
  return nullptr;
}
// whether a write of rows x [0, width) of the matrix (g, ld) may touch what tile t mirrors
bool may_alias(const Tile& t, const float* g, int ld, int width, int rows) {
  if (t.ld == ld) {
    // same row pitch: the column intervals decide (matrices that interleave in one wider matrix do not alias)
    const long d = (long)(g - t.g);
    const long dm = ((d % ld) + ld) % ld;
    const bool cols = dm < t.width || ld - dm < width;
    return cols;
  }
  const float* t1 = t.g + (long)(g_rc.B * t.rps - 1) * t.ld + t.width;
  const float* g1 = g + (long)(rows - 1) * ld + width;
  return g < t1 && t.g < g1;
}
// an operation writes rows x [0, width) of (g, ld): every tile that mirrors any of it is stale from here on
// (a tile the running operations still read keeps its slot until the next barrier: it only loses its name)
void invalidate(const float* g, int ld, int width, int rows) {
  RcState& s = g_rc;
  for (int i = 0; i < s.n_tiles;) {
    Tile& t = s.tiles[i];
    if (t.g != nullptr && may_alias(t, g, ld, width, rows)) {
      if (t.epoch == s.epoch) { t.g = nullptr; ++i; } else drop_tile(i);
    } else {
      ++i;
    }
  }
}
// operand k of `op` reads rows x [0, width) of the matrix (g, ld): from its LDS tile when the program holds one, else
// from memory behind a full barrier (an earlier operation of this program may have written it)
void plan_in(RcOp* op, int k, const float* g, int ld, int width) {
  RcState& s = g_rc;
  if (g == nullptr) return;
  Tile* t = lds_plan_on() ? find_tile(g, ld, op->rps, width) : nullptr;
  if (t != nullptr) {
    t->epoch = s.epoch;
    op->loff[k] = (short)(t->off / 4);
    op->lld[k] = (short)t->lld;
    return;
  }
  need_full_sync();
}
// a stacked operand: three tiles (g + m * bs), equally spaced in LDS, or memory
void plan_in_stack(RcOp* op, int k, const float* g, int ld, long bs, int width) {
  RcState& s = g_rc;
  Tile* t[3] = {nullptr, nullptr, nullptr};
  if (lds_plan_on())
    for (int m = 0; m < 3; ++m) t[m] = find_tile(g + m * bs, ld, op->rps, width);
  if (t[0] && t[1] && t[2] && t[1]->off - t[0]->off == t[2]->off - t[1]->off && t[1]->off > t[0]->off && t[0]->lld == t[1]->lld &&
      t[1]->lld == t[2]->lld) {
    for (int m = 0; m < 3; ++m) t[m]->epoch = s.epoch;
    op->loff[k] = (short)(t[0]->off / 4);
    op->lld[k] = (short)t[0]->lld;
    op->lbs[k] = (short)((t[1]->off - t[0]->off) / 4);
    return;
  }
  need_full_sync();
}
// place `n` equally spaced tiles (16 rps rows x width each) for the matrices g + m * bs this operation writes; returns
// the float offset of the first or -1 (too large, or no room that spares the tiles pinned in this epoch)
int place(RcOp* op, const float* g, int ld, long bs, int width, int n) {
  RcState& s = g_rc;
  for (int m = 0; m < n; ++m) invalidate(g + m * bs, ld, width, op->M);
  if (!lds_plan_on()) return -1;
  const int lld = ((width + 3) & ~3) + 4, size = 16 * op->rps * lld;
  if (size > kMaxTileFloats || n * size > kLdsFloats || s.n_tiles + n > kMaxTiles) return -1;
  int off = s.top;
  if (off + n * size > kLdsFloats) off = 0;
  for (int i = 0; i < s.n_tiles; ++i) {
    const Tile& t = s.tiles[i];
    if (t.off < off + n * size && off < t.off + t.size && t.epoch == s.epoch) return -1;       // would overwrite a pinned tile
  }
  for (int i = 0; i < s.n_tiles;)
    if (s.tiles[i].off < off + n * size && off < s.tiles[i].off + s.tiles[i].size) drop_tile(i); else ++i;
  for (int m = 0; m < n; ++m) s.tiles[s.n_tiles++] = Tile{g + m * bs, ld, op->rps, width, off + m * size, lld, size, s.epoch};
  s.top = off + n * size;
  return off;
}
void plan_out(RcOp* op, int k, const float* g, int ld, int width) {
  const int off = place(op, g, ld, 0, width, 1);
  if (off >= 0) { op->loff[k] = (short)(off / 4); op->lld[k] = (short)(((width + 3) & ~3) + 4); }
}
void plan_out_stack(RcOp* op, int k, const float* g, int ld, long bs, int width) {
  const int off = place(op, g, ld, bs, width, 3);
  if (off >= 0) {
    const int lld = ((width + 3) & ~3) + 4;
    op->loff[k] = (short)(off / 4); op->lld[k] = (short)lld; op->lbs[k] = (short)(16 * op->rps * lld / 4);
  }
}

// Every dropout site of one program shares one epoch counter (or none).  A site that differs from those recorded so far
// closes the program: what was recorded is launched and the operation becomes the first of a new program.
RcOp* set_drop(RcOp* op, const DropoutP& d) {
  if (!(d.p > 0.f)) return op;
  RcState& s = g_rc;
  if (s.drop_seen && s.prog.epoch != d.epoch) {
    const RcOp keep = *op;
    launch_chain();
    op = &s.prog.ops[0];
    *op = keep;
    op->flags &= ~RC_NOSYNC;
    ++s.epoch;
  }
  s.drop_seen = true;
  s.prog.epoch = d.epoch;
  op->drop_p = d.p;
  op->drop_seed = d.seed;
  return op;
}

}  // namespace

RcScope::RcScope(int batch_rows, hipStream_t stream) : on(false) {
  RcState& s = g_rc;
  if (s.on || batch_rows <= 0) return;
  const bool chain = batch_rows <= chain_max_rows(), defer = batch_rows <= defer_max_rows();
  if (!chain && !defer) return;
  s.chain = chain;
  s.on = true; s.B = batch_rows; s.st = stream; s.err = OCEAN_OK; s.token_major = false;
  reset_program();
  s.wg.n = 0; s.wg.total = 0;
  on = true;
}

int RcScope::finish() {
  if (!on) return OCEAN_OK;
  launch_chain();
  launch_wgrads();
  g_rc.on = false;
  g_rc.chain = false;
  on = false;
  return g_rc.err;
}

RcScope::~RcScope() { finish(); }

RcTokenMajor::RcTokenMajor() { g_rc.token_major = true; }
RcTokenMajor::~RcTokenMajor() { g_rc.token_major = false; }

bool rc_active() { return g_rc.on && g_rc.chain; }
bool rc_collecting() { return g_rc.on; }
int rc_batch() { return g_rc.B; }
void rc_flush() { if (g_rc.on) launch_chain(); }

bool rc_gemm(const GemmP& p, bool independent) {
  RcState& s = g_rc;
  if (!s.on) return false;
  const bool plain = p.taps == 1 && p.T == 0 && p.colstats == nullptr && p.bn_u == nullptr && p.r_bcast_T == 0 && p.M > 0 && p.N > 0 && p.K > 0;
  if (p.a_kmajor) {
    // row reduction C[M,N] += A[K,M]^T B[K,N] (a weight gradient: A = dy, B = x), a_colsum = the bias gradient
    const bool ok = plain && p.b_kn && p.bias == nullptr && p.R == nullptr && !p.relu && !(p.drop.p > 0.f) && p.relu_mask == nullptr &&
                    s.wg.n < kWgMax;
    if (!ok) { launch_chain(); return false; }
    WgProb& w = s.wg.p[s.wg.n++];
    w.dy = p.A; w.lddy = p.lda; w.x = p.B; w.ldx = p.ldb; w.dW = p.C; w.ldw = p.ldc; w.db = p.a_colsum;
    w.M = p.K; w.N = p.M; w.K = p.N; w.tile0 = s.wg.total; w.pad = 0;
    s.wg.total += ceil_div(w.N, 32) * ceil_div(w.K, 32);
    return true;
  }
  if (!s.chain) return false;
  bool ok = plain && p.a_colsum == nullptr && p.K % 4 == 0 && p.lda % 4 == 0 && al16(p.A) && (p.b_kn || (p.ldb % 4 == 0 && al16(p.B)));
  if (p.drop.p > 0.f && p.drop_group > 1 && p.N % p.drop_group != 0) ok = false;
  const int rps = (ok && p.M % s.B == 0) ? p.M / s.B : 1;
  // A block gives one compute unit's matrix rate (~0.6 TFLOP/s) to its 16 samples however large the layer: a layer that
  // would hold a block for more than ~12 us is better off as its own launch across the chip
  if (2.0 * 16.0 * rps * (double)p.N * (double)p.K > 12.0e-6 * 0.6e12) ok = false;
  RcOp* op = ok ? new_op(RC_LIN, p.M, independent) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op = set_drop(op, p.drop);
  // join the group of the preceding linear layers when independent of them, else start a group
  RcOp* head = s.lin_head >= 0 ? &s.prog.ops[s.lin_head] : nullptr;
  if ((op->flags & RC_NOSYNC) && head != nullptr && s.lin_head + head->ld[7] == s.prog.n_ops && head->ld[7] < 3 && head->rps == op->rps) {
    head->ld[7] += 1;
  } else {
    if (op->flags & RC_NOSYNC) { op->flags &= ~RC_NOSYNC; ++s.epoch; s.sync_op = s.prog.n_ops; }
    s.lin_head = s.prog.n_ops;
    op->ld[7] = 1;
  }
  op->N = p.N; op->K = p.K;
  op->p[0] = p.A; op->p[1] = p.B; op->p[2] = p.C; op->p[3] = p.bias; op->p[4] = p.R; op->p[5] = p.relu_mask;
  op->ld[0] = p.lda; op->ld[1] = p.ldb; op->ld[2] = p.ldc; op->ld[4] = p.ldr; op->ld[5] = p.ld_mask;
  if (p.relu) op->flags |= RC_RELU;
  if (p.b_kn) op->flags |= RC_BKN;
  op->drop_group = p.drop_group > 1 ? p.drop_group : 1;
  op->drop_row0 = p.drop_row0;
  // LDS plan: the activation operand; a residual is read from memory (it may have been written earlier in this program)
  plan_in(op, 0, p.A, p.lda, p.K);
  if (p.R != nullptr) need_full_sync();
  plan_out(op, 2, p.C, p.ldc, p.N);
  commit();
  return true;
}

bool rc_colsum(const float* a, int lda, int M, int N, float* out) {
  RcState& s = g_rc;
  if (!s.on) return false;
  if (s.wg.n >= kWgMax || M <= 0 || N <= 0) { launch_chain(); return false; }
  WgProb& w = s.wg.p[s.wg.n++];
  w.dy = a; w.lddy = lda; w.x = nullptr; w.ldx = 0; w.dW = nullptr; w.ldw = 0; w.db = out;
  w.M = M; w.N = N; w.K = 0; w.tile0 = s.wg.total; w.pad = 0;
  s.wg.total += ceil_div(N, 32);
  return true;
}

bool rc_ln_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats, int rows, int W,
               float eps, DropoutP in_drop, int drop_row0, bool independent) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_LN_FWD, rows, independent);
  if (op == nullptr || op->rps != 1) { launch_chain(); return false; }
  op = set_drop(op, in_drop);
  op->N = W; op->f0 = eps; op->drop_row0 = drop_row0;
  op->p[0] = x; op->p[1] = gamma; op->p[2] = beta; op->p[3] = y; op->p[4] = stats;
  op->ld[0] = ldx; op->ld[3] = ldy;
  plan_in(op, 0, x, ldx, W);
  plan_out(op, 3, y, ldy, W);
  commit();
  return true;
}

bool rc_ln_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma, float* dx, int lddx,
               float* dgamma, float* dbeta, int rows, int W, DropoutP in_drop, int drop_row0, bool independent) {
  if (!rc_active()) return false;
  RcOp* op = W <= 64 * rowbody::kMaxColsPerLane ? new_op(RC_LN_BWD, rows, independent) : nullptr;
  if (op == nullptr || op->rps != 1) { launch_chain(); return false; }
  op = set_drop(op, in_drop);
  op->N = W; op->drop_row0 = drop_row0;
  op->p[0] = dy; op->p[1] = x; op->p[2] = stats; op->p[3] = gamma; op->p[4] = dx; op->p[5] = dgamma; op->p[6] = dbeta;
  op->ld[0] = lddy; op->ld[1] = ldx; op->ld[4] = lddx;
  plan_in(op, 0, dy, lddy, W);
  plan_out(op, 4, dx, lddx, W);
  commit();
  return true;
}

bool rc_l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* norms, int rows, int W, float eps, bool independent) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_L2N_FWD, rows, independent);
  if (op == nullptr || op->rps != 1) { launch_chain(); return false; }
  op->N = W; op->f0 = eps;
  op->p[0] = x; op->p[1] = y; op->p[2] = norms;
  op->ld[0] = ldx; op->ld[1] = ldy;
  plan_in(op, 0, x, ldx, W);
  invalidate(y, ldy, W, rows);
  commit();
  return true;
}

bool rc_l2norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* norms, float* dx, int lddx, int rows, int W,
                   float eps, bool independent) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_L2N_BWD, rows, independent);
  if (op == nullptr || op->rps != 1) { launch_chain(); return false; }
  op->N = W; op->f0 = eps;
  op->p[0] = dy; op->p[1] = y; op->p[2] = norms; op->p[3] = dx;
  op->ld[0] = lddy; op->ld[1] = ldy; op->ld[3] = lddx;
  need_full_sync();             // dy and y are read from memory
  plan_out(op, 3, dx, lddx, W);
  commit();
  return true;
}

bool rc_cross_fwd(const float* Nn, const float* mw, float* G, float* cw, int B, int W) {
  if (!rc_active()) return false;
  RcOp* op = B == g_rc.B ? new_op(RC_CROSS_FWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op->N = W;
  op->p[0] = Nn; op->p[1] = mw; op->p[2] = G; op->p[3] = cw;
  op->ld[0] = W; op->ld[2] = W;
  plan_in_stack(op, 0, Nn, W, (long)B * W, W);
  plan_out_stack(op, 2, G, W, (long)B * W, W);
  commit();
  return true;
}

bool rc_cross_bwd(const float* dG, const float* Nn, const float* cw, const float* mw, float* dN, float* dpi, int B, int W) {
  if (!rc_active()) return false;
  RcOp* op = B == g_rc.B ? new_op(RC_CROSS_BWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op->N = W;
  op->p[0] = dG; op->p[1] = Nn; op->p[2] = cw; op->p[3] = mw; op->p[4] = dN; op->p[5] = dpi;
  op->ld[0] = W; op->ld[4] = W;
  plan_in_stack(op, 0, dG, W, (long)B * W, W);
  plan_out_stack(op, 4, dN, W, (long)B * W, W);
  commit();
  return true;
}

bool rc_gate_fwd(const float* hid, const float* w2, const float* b2, const float* G, float* ig, float* cat, int B, int W, int Wh) {
  if (!rc_active()) return false;
  RcOp* op = B == g_rc.B ? new_op(RC_GATE_FWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op->N = W; op->K = Wh;
  op->p[0] = hid; op->p[1] = w2; op->p[2] = b2; op->p[3] = G; op->p[4] = ig; op->p[5] = cat;
  op->ld[0] = Wh; op->ld[3] = W; op->ld[5] = 3 * W;
  plan_in_stack(op, 0, hid, Wh, (long)B * Wh, Wh);
  plan_in_stack(op, 3, G, W, (long)B * W, W);
  plan_out(op, 5, cat, 3 * W, 3 * W);
  commit();
  return true;
}

bool rc_gate_bwd(const float* dcat, const float* G, const float* ig, const float* hid, const float* w2, float* dG, float* dz,
                 float* dhid, int B, int W, int Wh) {
  if (!rc_active()) return false;
  RcOp* op = B == g_rc.B ? new_op(RC_GATE_BWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op->N = W; op->K = Wh;
  op->p[0] = dcat; op->p[1] = G; op->p[2] = ig; op->p[3] = hid; op->p[4] = w2; op->p[5] = dG; op->p[6] = dz; op->p[7] = dhid;
  op->ld[0] = 3 * W; op->ld[5] = W; op->ld[7] = Wh;
  plan_in(op, 0, dcat, 3 * W, 3 * W);
  plan_out_stack(op, 5, dG, W, (long)B * W, W);
  plan_out_stack(op, 7, dhid, Wh, (long)B * Wh, Wh);
  commit();
  return true;
}

bool rc_mha3_fwd(const float* qkv, float* probs, float* o, int B, int E, int H, DropoutP dp) {
  if (!rc_active()) return false;
  RcOp* op = (B == g_rc.B && rowbody::mha3_vec(E, H) != 0 && al16(qkv) && al16(o)) ? new_op(RC_MHA3_FWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op = set_drop(op, dp);
  op->N = E; op->K = H; op->drop_group = rowbody::mha3_hlog(H);
  op->p[0] = qkv; op->p[1] = probs; op->p[2] = o;
  need_full_sync();             // operands in memory only
  invalidate(o, E, E, 3 * B);
  commit();
  return true;
}

bool rc_mha3_bwd(const float* dOm, const float* qkv, const float* probs, float* dqkv, int B, int E, int H, DropoutP dp) {
  if (!rc_active()) return false;
  RcOp* op = (B == g_rc.B && rowbody::mha3_vec(E, H) != 0 && al16(qkv) && al16(dqkv) && al16(dOm)) ? new_op(RC_MHA3_BWD, B, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op = set_drop(op, dp);
  op->N = E; op->K = H; op->drop_group = rowbody::mha3_hlog(H);
  op->p[0] = dOm; op->p[1] = qkv; op->p[2] = probs; op->p[3] = dqkv;
  need_full_sync();
  invalidate(dqkv, 3 * E, 3 * E, 3 * B);
  commit();
  return true;
}

bool rc_mean_mid(const float* a, float* out, int G, int n, int C) {
  if (!rc_active()) return false;
  RcOp* op = G == g_rc.B ? new_op(RC_MEAN_MID, G, false) : nullptr;
  if (op == nullptr) { launch_chain(); return false; }
  op->N = C; op->K = n;
  op->p[0] = a; op->p[1] = out;
  op->ld[1] = C;
  need_full_sync();             // `a` is read from memory
  plan_out(op, 1, out, C, C);
  commit();
  return true;
}

bool rc_sigmoid(const float* x, float* y, int rows, int cols) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_SIGMOID, rows, false);
  if (op == nullptr) { launch_chain(); return false; }
  op->N = cols;
  op->p[0] = x; op->p[1] = y;
  op->ld[0] = cols;
  plan_in(op, 0, x, cols, cols);
  invalidate(y, cols, cols, rows);
  commit();
  return true;
}

bool rc_sigmoid_bwd(const float* dlogits_in, const float* dprobs, const float* p, float* out, int rows, int cols) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_SIGMOID_BWD, rows, false);
  if (op == nullptr) { launch_chain(); return false; }
  op->N = cols;
  op->p[0] = dlogits_in; op->p[1] = dprobs; op->p[2] = p; op->p[3] = out;
  op->ld[3] = cols;
  need_full_sync();
  plan_out(op, 3, out, cols, cols);
  commit();
  return true;
}

bool rc_relu_bwd(float* dy, int lddy, const float* y, int ldy, int M, int N, bool independent) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_RELU_BWD, M, false);
  if (op == nullptr) { launch_chain(); return false; }
  op->N = N;
  op->p[0] = dy; op->p[1] = y;
  op->ld[0] = lddy; op->ld[1] = ldy;
  need_full_sync();             // in place, in memory: a tile of it in LDS is stale afterwards
  invalidate(dy, lddy, N, M);
  commit();
  return true;
}

bool rc_copy2d(float* dst, int ldd, const float* src, int lds, int M, int N, int accumulate) {
  if (!rc_active()) return false;
  RcOp* op = new_op(RC_COPY2D, M, false);
  if (op == nullptr) { launch_chain(); return false; }
  op->N = N;
  if (accumulate) op->flags |= RC_ACC;
  op->p[0] = dst; op->p[1] = src;
  op->ld[0] = ldd; op->ld[1] = lds;
  need_full_sync();
  invalidate(dst, ldd, N, M);
  commit();
  return true;
}

// Diagnostics: while `buffer` (device memory, at least 2 * 30 64-bit words) is set, block 0 of every stage kernel
// launched from any thread writes (operation kind * 1000 + width, 100 MHz clock) per operation into it, then (-1, clock).
extern "C" void ocean_rc_debug_trace(void* buffer) { g_trace = static_cast<long long*>(buffer); }
