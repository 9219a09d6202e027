// Small shared device routines: linear-layer wrappers over the MFMA GEMM,
// column sums, ReLU backward, fills and strided copies.
#include "ops.h"
#include "rowchain.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

static void apply_epi(GemmP& p, const Epi* e) {
  if (!e) return;
  if (e->drop) { p.drop = *e->drop; p.drop_group = e->group; p.drop_row0 = e->row0; }
  p.relu_mask = e->relu_mask; p.ld_mask = e->ld_mask;
}

int linear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy,
               int M, int N, int K, int relu, hipStream_t s, const Epi* e) {
  GemmP p = gemm_defaults();
  p.A = x; p.lda = ldx; p.B = W; p.ldb = K; p.C = y; p.ldc = ldy;
  p.M = M; p.N = N; p.K = K; p.bias = b; p.relu = relu;
  apply_epi(p, e);
  return ocean_launch_gemm(p, s);
}

int linear_bwd_data(const float* dy, int lddy, const float* W, float* dx, int lddx,
                    int M, int N, int K, int accumulate, hipStream_t s, const Epi* e) {
  GemmP p = gemm_defaults();
  p.A = dy; p.lda = lddy; p.B = W; p.ldb = K; p.b_kn = 1; p.C = dx; p.ldc = lddx;
  p.M = M; p.N = K; p.K = N;
  if (accumulate) { p.R = dx; p.ldr = lddx; }
  apply_epi(p, e);
  return ocean_launch_gemm(p, s);
}

int linear_bwd_weight(const float* dy, int lddy, const float* x, int ldx, float* dW,
                      int M, int N, int K, hipStream_t s, float* db) {
  GemmP p = gemm_defaults();
  p.a_colsum = db;
  p.A = dy; p.lda = lddy; p.a_kmajor = 1; p.B = x; p.ldb = ldx; p.b_kn = 1; p.C = dW; p.ldc = K;
  p.M = N; p.N = K; p.K = M; p.split_k = 0;
  return ocean_launch_gemm(p, s);
}

int linear_fwd_batch(const LinFwd* l, int n, int relu, hipStream_t s, const DropoutP* drop, int drop_group) {
  if (n < 1 || n > kGemmBatchMax) return OCEAN_ERR_BAD_SHAPE;
  GemmP g[kGemmBatchMax];
  for (int q = 0; q < n; ++q) {
    g[q] = gemm_defaults();
    g[q].A = l[q].x; g[q].lda = l[q].ldx; g[q].B = l[q].W; g[q].ldb = l[q].K; g[q].C = l[q].y; g[q].ldc = l[q].ldy;
    g[q].M = l[q].M; g[q].N = l[q].N; g[q].K = l[q].K; g[q].bias = l[q].b; g[q].relu = relu;
    if (drop) { g[q].drop = *drop; g[q].drop_group = drop_group; g[q].drop_row0 = q * l[q].M; }
  }
  return ocean_launch_gemm_batch(g, n, s);
}

int linear_bwd_pair(const float* x, int ldx, const float* dy, int lddy, const float* W, float* dW, float* db, float* dx, int lddx,
                    int M, int N, int K, int accumulate, hipStream_t s, const Epi* e) {
  GemmP w = gemm_defaults();
  w.a_colsum = db;
  w.A = dy; w.lda = lddy; w.a_kmajor = 1; w.B = x; w.ldb = ldx; w.b_kn = 1; w.C = dW; w.ldc = K;
  w.M = N; w.N = K; w.K = M; w.split_k = 0;
  GemmP d = gemm_defaults();
  d.A = dy; d.lda = lddy; d.B = W; d.ldb = K; d.b_kn = 1; d.C = dx; d.ldc = lddx;
  d.M = M; d.N = K; d.K = N;
  if (accumulate) { d.R = dx; d.ldr = lddx; }
  apply_epi(d, e);
  return ocean_launch_gemm_pair(&w, 1, &d, 1, s);
}

int linear_bwd_batch(const LinBwd* l, int n, int accumulate_dx, hipStream_t s, const DropoutP* dx_drop, int drop_group,
                     const float* const* relu_masks, int ld_mask) {
  if (n < 1 || n > kGemmBatchMax) return OCEAN_ERR_BAD_SHAPE;
  GemmP gw[kGemmBatchMax], gd[kGemmBatchMax];
  int kw = 0, kd = 0;
  for (int q = 0; q < n; ++q) {
    if (!l[q].dW) continue;
    GemmP& g = gw[kw++];
    g = gemm_defaults();
    g.A = l[q].dy; g.lda = l[q].lddy; g.a_kmajor = 1; g.B = l[q].x; g.ldb = l[q].ldx; g.b_kn = 1;
    g.C = l[q].dW; g.ldc = l[q].K; g.M = l[q].N; g.N = l[q].K; g.K = l[q].M; g.split_k = 0;
    g.a_colsum = l[q].db;       // the bias gradient rides along with the weight gradient
  }
  for (int q = 0; q < n; ++q) {
    if (!l[q].dx) continue;
    GemmP& g = gd[kd++];
    g = gemm_defaults();
    g.A = l[q].dy; g.lda = l[q].lddy; g.B = l[q].W; g.ldb = l[q].K; g.b_kn = 1; g.C = l[q].dx; g.ldc = l[q].lddx;
    g.M = l[q].M; g.N = l[q].K; g.K = l[q].N;
    if (accumulate_dx) { g.R = l[q].dx; g.ldr = l[q].lddx; }
    if (dx_drop) { g.drop = *dx_drop; g.drop_group = drop_group; g.drop_row0 = q * l[q].M; }
    if (relu_masks) { g.relu_mask = relu_masks[q]; g.ld_mask = ld_mask; }
  }
  // reductions of different lengths cannot share a weight-gradient launch, nor can the pair launch take them
  bool same_rows = true;
  for (int q = 1; q < kw; ++q) same_rows = same_rows && gw[q].K == gw[0].K;
  if (kw > 0 && kd > 0 && same_rows) OCEAN_TRY(ocean_launch_gemm_pair(gw, kw, gd, kd, s));
  else if (kw > 0) OCEAN_TRY(ocean_launch_gemm_batch(gw, kw, s));
  ColSum c[kGemmBatchMax];
  int k = 0;
  for (int q = 0; q < n; ++q) {
    if (!l[q].db || l[q].dW) continue;
    c[k].a = l[q].dy; c[k].lda = l[q].lddy; c[k].M = l[q].M; c[k].N = l[q].N; c[k].out = l[q].db;
    ++k;
  }
  if (k > 0) OCEAN_TRY(colsum_add_batch(c, k, s));
  if (kd > 0 && !(kw > 0 && same_rows)) OCEAN_TRY(ocean_launch_gemm_batch(gd, kd, s));
  return OCEAN_OK;
}

namespace {

struct ColSumBatch { ColSum c[kGemmBatchMax]; };

// Column sums: a block is W column groups (VEC consecutive columns each, W a power of two) by 256/W row lanes
// over `rows` rows; consecutive lanes read consecutive addresses, eight rows are in flight per thread, and each
// block issues one atomic per column.
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const ColSumBatch batch, int wlog, int rows) {
  const ColSum& cs = batch.c[blockIdx.z];
  const float* __restrict__ a = cs.a;
  float* __restrict__ out = cs.out;
  const int lda = cs.lda, M = cs.M, N = cs.N;
  __shared__ float red[256 * VEC];
  const int W = 1 << wlog, RL = 256 >> wlog;
  const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> wlog;
  const int c0 = (blockIdx.x * W + tx) * VEC;
  const int r0 = blockIdx.y * rows, r1 = min(M, r0 + rows);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (c0 < N) {
#pragma unroll 8
    for (int r = r0 + ty; r < r1; r += RL) {
      const float* p = a + (long)r * lda + c0;
      if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4*>(p);
        acc[0] += x.x; acc[1 % VEC] += x.y; acc[2 % VEC] += x.z; acc[3 % VEC] += x.w;
      } else {
        acc[0] += *p;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = acc[v];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float t = 0.f;
      for (int q = 0; q < RL; ++q) t += red[(q * W + tx) * VEC + v];
      unsafeAtomicAdd(out + c0 + v, t);
    }
  }
}

__global__ void relu_bwd_kernel(float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int M, int N) {
  const long n = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), c = (int)(i % N);
    if (!(y[(long)r * ldy + c] > 0.f)) dy[(long)r * lddy + c] = 0.f;
  }
}

template <typename T>
__global__ void fill_zero_kernel(T* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = T(0);
}

__global__ void copy2d_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int M, int N, int acc) {
  const long n = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), c = (int)(i % N);
    const float v = src[(long)r * lds + c];
    if (acc) dst[(long)r * ldd + c] += v; else dst[(long)r * ldd + c] = v;
  }
}

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

int colsum_add_batch(const ColSum* c, int n, hipStream_t s) {
  if (n < 1 || n > kGemmBatchMax) return OCEAN_ERR_BAD_SHAPE;
  if (rc_collecting()) {
    int taken = 0;
    while (taken < n && rc_colsum(c[taken].a, c[taken].lda, c[taken].M, c[taken].N, c[taken].out)) ++taken;
    if (taken == n) return OCEAN_OK;
    c += taken; n -= taken;
  }
  ColSumBatch b;
  bool v4 = true;
  int M = 0, N = 0;
  for (int q = 0; q < n; ++q) {
    if (c[q].M <= 0 || c[q].N <= 0) return OCEAN_ERR_BAD_SHAPE;
    v4 = v4 && (c[q].N % 4 == 0) && (c[q].lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(c[q].a) & 15) == 0);
    if (c[q].M > M) M = c[q].M;
    if (c[q].N > N) N = c[q].N;
    b.c[q] = c[q];
  }
  for (int q = n; q < kGemmBatchMax; ++q) b.c[q] = c[0];
  const int vec = v4 ? 4 : 1;
  const int cg = (N + vec - 1) / vec;
  int wlog = 0;
  while ((1 << wlog) < cg && wlog < 8) ++wlog;
  const int RL = 256 >> wlog;
  // Same-address atomics serialise at the memory side (~tens of ns each): keep them to <= 256 per column.
  int rows = RL * 4;
  if (rows < ceil_div(M, 256)) rows = ceil_div(M, 256);
  dim3 grid(ceil_div(cg, 1 << wlog), ceil_div(M, rows), n);
  if (v4) hipLaunchKernelGGL(colsum_kernel<4>, grid, dim3(256), 0, s, b, wlog, rows);
  else hipLaunchKernelGGL(colsum_kernel<1>, grid, dim3(256), 0, s, b, wlog, rows);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int colsum_add(const float* a, int lda, int M, int N, float* out, hipStream_t s) {
  ColSum c;
  c.a = a; c.lda = lda; c.M = M; c.N = N; c.out = out;
  return colsum_add_batch(&c, 1, s);
}

int relu_bwd_inplace(float* dy, int lddy, const float* y, int ldy, int M, int N, hipStream_t s) {
  if (rc_active() && rc_relu_bwd(dy, lddy, y, ldy, M, N, false)) return OCEAN_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, s, dy, lddy, y, ldy, M, N);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int zero_f32(float* dst, size_t n, hipStream_t s) {
  if (n == 0) return OCEAN_OK;
  rc_flush();
  hipLaunchKernelGGL(fill_zero_kernel<float>, dim3(grid_for((long)n)), dim3(256), 0, s, dst, n);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int zero_f64(double* dst, size_t n, hipStream_t s) {
  if (n == 0) return OCEAN_OK;
  rc_flush();
  hipLaunchKernelGGL(fill_zero_kernel<double>, dim3(grid_for((long)n)), dim3(256), 0, s, dst, n);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int copy2d(float* dst, int ldd, const float* src, int lds, int M, int N, int accumulate, hipStream_t s) {
  if (rc_active() && rc_copy2d(dst, ldd, src, lds, M, N, accumulate)) return OCEAN_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, s, dst, ldd, src, lds, M, N, accumulate);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
