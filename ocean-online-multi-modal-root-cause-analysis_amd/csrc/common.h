// Shared declarations for the OCEAN gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// Status codes returned through the C ABI (0 = success).
enum OceanStatus {
  OCEAN_OK = 0,
  OCEAN_ERR_BAD_SHAPE = 1,      // a dimension is <= 0 or inconsistent
  OCEAN_ERR_UNSUPPORTED = 2,    // a size beyond a kernel's compiled limit
  OCEAN_ERR_WORKSPACE = 3,      // caller workspace too small
  OCEAN_ERR_NULL = 4,           // required pointer is null
  OCEAN_ERR_SCRATCH = 5,        // the problem needs more than the library's built-in scratch array holds
  OCEAN_ERR_HIP = 1000          // + hipError_t
};

#define OCEAN_LAUNCH_CHECK()                                  \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return OCEAN_ERR_HIP + (int)e__;   \
  } while (0)

#define OCEAN_TRY(x)                \
  do {                              \
    int s__ = (x);                  \
    if (s__ != OCEAN_OK) return s__; \
  } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- dropout
// Counter-based dropout: element `idx` of a site is kept iff hash(seed, idx) >= p,
// so forward and backward regenerate the same mask without storing it.
// `epoch` (device pointer or null) is added into the seed inside the kernel: a captured HIP graph replays the same
// kernel arguments, so a launch-time seed alone would repeat the masks; the caller bumps *epoch between replays.
struct DropoutP { float p; uint64_t seed; const uint64_t* epoch; };

// 32-bit multiply-xorshift mixer over (seed, idx); integer multiplies run at a quarter of the fp32 rate on
// CDNA, so the hash is kept to four of them.
__host__ __device__ inline uint32_t ocean_hash32(uint64_t seed, uint64_t idx) {
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= h >> 15;
  h += (uint32_t)(idx >> 32) * 0x85EBCA77u + (uint32_t)(seed >> 32);
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}
#ifdef __HIPCC__
// fold *epoch into the seed once, at kernel entry (kernels take DropoutP by value)
__device__ __forceinline__ DropoutP dropout_resolve(DropoutP d) {
  if (d.p > 0.f && d.epoch != nullptr) d.seed += d.epoch[0] * 0xD1B54A32D192ED03ull;
  d.epoch = nullptr;
  return d;
}
#endif
// the same hash for element indices below 2^32 (idx >> 32 == 0), without 64-bit index arithmetic
__host__ __device__ inline uint32_t ocean_hash32_lo(uint64_t seed, uint32_t idx) {
  uint32_t h = idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= h >> 15;
  h += (uint32_t)(seed >> 32);
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}
__host__ __device__ inline float dropout_keep_lo(const DropoutP& d, uint32_t idx) {
  const float u = (float)(ocean_hash32_lo(d.seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= d.p ? 1.0f / (1.0f - d.p) : 0.0f;
}
// multiplier applied to a surviving / dropped activation: 1/(1-p) or 0
__host__ __device__ inline float dropout_keep(const DropoutP& d, uint64_t idx) {
  const float u = (float)(ocean_hash32(d.seed, idx) >> 8) * (1.0f / 16777216.0f);   // [0,1)
  return u >= d.p ? 1.0f / (1.0f - d.p) : 0.0f;
}

// ---------------------------------------------------------------- GEMM
// C[M,N] (+)= sum over taps t, sum_k  A_t[m,k] * B_t[k,n]     fp32 MFMA (32x32x2).
//
// Operand storage:
//   a_kmajor == 0 : A stored [m][k] (k contiguous, leading dim lda)
//   a_kmajor == 1 : A stored [k][m] (m contiguous)
//   b_kn     == 0 : B stored [n][k] (k contiguous, e.g. nn.Linear weight [out,in])
//   b_kn     == 1 : B stored [k][n] (n contiguous)
//
// Row shifting (dilated-convolution taps over a [B*T, C] channels-last matrix):
//   rows of the *M-like* index are (sample, time) pairs with time = row % T.
//   For tap t the shift is s = shift0 + t*shift_step; a shifted row is read
//   from row + s when 0 <= time + s < T and is zero otherwise.
//   - a_kmajor == 0: the shift applies to A's rows (m); B advances by
//     tap_stride_b per tap; all taps accumulate into one C.
//   - a_kmajor == 1 (reduction over rows): the shift applies to B's rows (k);
//     each tap writes its own C + t*tap_stride_c (grid.z covers taps*split_k).
struct GemmP {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int a_kmajor, b_kn;
  int taps, T, shift0, shift_step;
  long tap_stride_b, tap_stride_c;
  int split_k;              // only with a_kmajor == 1; C is then accumulated with atomics (must be pre-zeroed or hold the running sum)
  const float* bias;        // [N] added to every row, or null
  const float* R; int ldr;  // residual added elementwise, or null (R == C allowed: in-place accumulate)
  int r_bcast_T; float r_scale;  // r_bcast_T > 0: residual row is R[row / r_bcast_T] (one row per sample), times r_scale
  int relu;
  double* colstats;         // [slots][2][N] running (sum, sum of squares) of the pre-epilogue value, atomically accumulated, or null
  int colstat_slots;        // row-tile t adds into copy t % slots (same-address atomics serialise; the caller sums the copies)
  // epilogue of non-reducing launches, applied after bias / residual / ReLU:
  DropoutP drop;            // p > 0: out *= keep(index); index = (drop_row0 + row) * (N / drop_group) + col / drop_group, the
  int drop_group;           // numbering dropout_inplace uses on the [rows, N] matrix this output is a row block of
  int drop_row0;
  const float* relu_mask;   // non-null: out = relu_mask[row * ld_mask + col] > 0 ? out : 0   (ReLU backward on a saved output)
  int ld_mask;
  float* a_colsum;          // row reductions (a_kmajor) with one tap only: a_colsum[m] += sum_k A[k][m], the bias gradient of a
                            // linear layer whose weight gradient this launch computes; or null
  // BatchNorm-backward column sums on the way out (non-reducing launches; null bn_u: off).  The output C[row][col] is
  // the gradient dy of a layer y = dropout(relu(u * scale + shift)) whose pre-normalisation values u [M, N] the
  // forward pass kept: with g = dy * [u*scale+shift > 0] * keep(row*N + col), the launch adds sum_rows g and
  // sum_rows g * (u - mean) * invstd per column into bn_slots [slots][2][N] (fp64, slot = row tile % slots), which
  // is what a separate pass over dy and u would otherwise compute.  C itself is stored unmasked.
  const float* bn_u; int ld_bn_u;
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  double* bn_slots; int bn_nslots;
  DropoutP bn_drop;
};

int ocean_launch_gemm(const GemmP& p, hipStream_t stream);
// whether a launch of `p` (data-gradient form, bn_* fields aside) can carry the BatchNorm-backward column sums in its
// epilogue: every output tile full, a broadcast residual constant per tile, element indices within 32 bits.  Set the
// bn_* fields only when this holds.
bool ocean_gemm_bn_rides(const GemmP& p);

// Up to kGemmBatchMax independent problems of the same operand storage mode (a_kmajor / b_kn) and tap count in ONE
// launch (blockIdx.y selects the problem): the independent modality branches of the attention and fusion
// stages are each too small to fill the chip on their own.
constexpr int kGemmBatchMax = 3;
struct GemmBatch { GemmP g[kGemmBatchMax]; int n; };
int ocean_launch_gemm_batch(const GemmP* ps, int n, hipStream_t stream);

// the kernel for outputs of few tiles (gemm_skinny.hip): ocean_launch_gemm_batch sends eligible batches there
bool ocean_gemm_skinny_applies(const GemmBatch& b);
int ocean_launch_gemm_skinny(const GemmBatch& b, hipStream_t stream);
// a linear layer's weight-gradient problems `w` and data-gradient problems `d` (same dy, independent outputs): one launch
// when both are few-tile work, else ocean_launch_gemm_batch(w) then ocean_launch_gemm_batch(d)
int ocean_launch_gemm_pair(const GemmP* w, int nw, const GemmP* d, int nd, hipStream_t stream);

// whether the large product about to be launched by this thread runs as three bf16 matrix instructions on split
// operands instead of one fp32 one (opt-in: ocean_set_split_bf16 / OCEAN_GEMM_SPLIT_BF16; api.hip)
bool ocean_split_bf16();
// alive for the duration of a backward driver call: mode 2 splits only the products launched under one
struct OceanBackwardScope { OceanBackwardScope(); ~OceanBackwardScope(); };

// the kernel for large products of two k-contiguous operands with full tiles (gemm_nt.hip), likewise
bool ocean_gemm_nt_applies(const GemmBatch& b);
int ocean_launch_gemm_nt(const GemmBatch& b, hipStream_t stream);

// the kernel for long row reductions into full 128x128 tiles (gemm_tn.hip), likewise
bool ocean_gemm_tn_applies(const GemmBatch& b);
// A row reduction (weight gradient) whose output is a dozen 128x128 tile-taps or fewer over fewer than 65536 rows is
// bound by the atomic adds of its row slices, and 64x64 tiles need a quarter of the slices to fill the chip: measured at
// 16384 rows, 128x128x3 taps 25 against 34 us, 256x256x3 68 against 78, 256x128 19 against 27, 512x256 46 against 58
// (at 32768 rows 36 / 47, 120 / 124, 28 / 38, 85 / 93).  Such problems take the ring kernel's 64x64 tiles.
inline bool ocean_wgrad_small_tiles(const GemmP& p) {
  return p.a_kmajor && p.K < 65536 && (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.taps <= 12;
}
int ocean_launch_gemm_tn(const GemmBatch& b, hipStream_t stream);

// optional kernel timing (prof.hip); token is null while profiling is off
bool ocean_prof_on();
void ocean_prof_begin(const char* name, double flops, double bytes, hipStream_t s, void** token);
void ocean_prof_end(void* token, hipStream_t s);

static inline GemmP gemm_defaults() {
  GemmP p;
  p.A = p.B = nullptr; p.C = nullptr; p.M = p.N = p.K = 0; p.lda = p.ldb = p.ldc = 0;
  p.a_kmajor = 0; p.b_kn = 0; p.taps = 1; p.T = 0; p.shift0 = 0; p.shift_step = 0;
  p.tap_stride_b = 0; p.tap_stride_c = 0; p.split_k = 1; p.bias = nullptr; p.a_colsum = nullptr; p.drop = DropoutP{0.f, 0, nullptr}; p.drop_group = 1; p.drop_row0 = 0;
  p.relu_mask = nullptr; p.ld_mask = 0; p.R = nullptr; p.ldr = 0; p.r_bcast_T = 0; p.r_scale = 1.f;
  p.relu = 0; p.colstats = nullptr; p.colstat_slots = 1;
  p.bn_u = nullptr; p.ld_bn_u = 0; p.bn_scale = p.bn_shift = p.bn_mean = p.bn_invstd = nullptr; p.bn_slots = nullptr; p.bn_nslots = 1;
  p.bn_drop = DropoutP{0.f, 0, nullptr};
  return p;
}
