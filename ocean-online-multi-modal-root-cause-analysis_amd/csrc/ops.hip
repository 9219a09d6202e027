// Small shared device routines: linear-layer wrappers over the MFMA GEMM,
// column sums, ReLU backward, fills and strided copies.
#include "ops.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

int linear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy,
               int M, int N, int K, int relu, hipStream_t s) {
  GemmP p = gemm_defaults();
  p.A = x; p.lda = ldx; p.B = W; p.ldb = K; p.C = y; p.ldc = ldy;
  p.M = M; p.N = N; p.K = K; p.bias = b; p.relu = relu;
  return ocean_launch_gemm(p, s);
}

int linear_bwd_data(const float* dy, int lddy, const float* W, float* dx, int lddx,
                    int M, int N, int K, int accumulate, hipStream_t s) {
  GemmP p = gemm_defaults();
  p.A = dy; p.lda = lddy; p.B = W; p.ldb = K; p.b_kn = 1; p.C = dx; p.ldc = lddx;
  p.M = M; p.N = K; p.K = N;
  if (accumulate) { p.R = dx; p.ldr = lddx; }
  return ocean_launch_gemm(p, s);
}

int linear_bwd_weight(const float* dy, int lddy, const float* x, int ldx, float* dW,
                      int M, int N, int K, hipStream_t s) {
  GemmP p = gemm_defaults();
  p.A = dy; p.lda = lddy; p.a_kmajor = 1; p.B = x; p.ldb = ldx; p.b_kn = 1; p.C = dW; p.ldc = K;
  p.M = N; p.N = K; p.K = M; p.split_k = 0;
  return ocean_launch_gemm(p, s);
}

namespace {

// Each block reduces ROWS rows of up to 256 columns; threads run along columns
// (coalesced row reads), one atomic per column per block.
// Block = 64 column lanes x 4 row lanes over kColsumRows rows: each wave reads 256 contiguous bytes per row,
// four independent accumulators per thread keep loads in flight; one atomic per column per block.
constexpr int kColsumRows = 1024;

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int lda, int M, int N, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * kColsumRows;
  const int r1 = min(M, r0 + kColsumRows);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < N) {
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
      s0 += a[(long)r * lda + col]; s1 += a[(long)(r + 4) * lda + col];
      s2 += a[(long)(r + 8) * lda + col]; s3 += a[(long)(r + 12) * lda + col];
    }
    for (; r < r1; r += 4) s0 += a[(long)r * lda + col];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && col < N) unsafeAtomicAdd(out + col, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
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

int colsum_add(const float* a, int lda, int M, int N, float* out, hipStream_t s) {
  if (M <= 0 || N <= 0) return OCEAN_ERR_BAD_SHAPE;
  dim3 grid(ceil_div(N, 64), ceil_div(M, kColsumRows));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, a, lda, M, N, out);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int relu_bwd_inplace(float* dy, int lddy, const float* y, int ldy, int M, int N, hipStream_t s) {
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, s, dy, lddy, y, ldy, M, N);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int zero_f32(float* dst, size_t n, hipStream_t s) {
  if (n == 0) return OCEAN_OK;
  hipLaunchKernelGGL(fill_zero_kernel<float>, dim3(grid_for((long)n)), dim3(256), 0, s, dst, n);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int zero_f64(double* dst, size_t n, hipStream_t s) {
  if (n == 0) return OCEAN_OK;
  hipLaunchKernelGGL(fill_zero_kernel<double>, dim3(grid_for((long)n)), dim3(256), 0, s, dst, n);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int copy2d(float* dst, int ldd, const float* src, int lds, int M, int N, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, s, dst, ldd, src, lds, M, N, accumulate);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}
