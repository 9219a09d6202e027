// Generic entry points of the C ABI (see include/ocean_hip.h).
#include "../../include/ocean_hip.h"
#include "ops.h"
#include <atomic>
#include <cstdlib>

namespace {
// 0: fp32 products; 1: split-bf16 products in every large GEMM; 2: only inside the backward drivers
std::atomic<int>& split_flag() {
  static std::atomic<int> f{[] {
    const char* e = getenv("OCEAN_GEMM_SPLIT_BF16");
    const int v = e ? atoi(e) : 0;
    return v == 1 || v == 2 ? v : 0;
  }()};
  return f;
}
thread_local int backward_depth = 0;
}  // namespace
OceanBackwardScope::OceanBackwardScope() { ++backward_depth; }
OceanBackwardScope::~OceanBackwardScope() { --backward_depth; }
bool ocean_split_bf16() {
  const int m = split_flag().load(std::memory_order_relaxed);
  return m == 1 || (m == 2 && backward_depth > 0);
}

extern "C" {

int ocean_abi_version(void) { return 1; }

int ocean_set_split_bf16(int mode) {
  if (mode < 0 || mode > 2) return -1;
  return split_flag().exchange(mode, std::memory_order_relaxed);
}
int ocean_get_split_bf16(void) { return split_flag().load(std::memory_order_relaxed); }

int ocean_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                   int a_kmajor, int b_kn, const float* bias, const float* R, int ldr, int relu, void* stream) {
  GemmP p = gemm_defaults();
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.a_kmajor = a_kmajor; p.b_kn = b_kn;
  p.bias = bias; p.R = R; p.ldr = ldr; p.relu = relu;
  p.split_k = a_kmajor ? 0 : 1;
  return ocean_launch_gemm(p, (hipStream_t)stream);
}

int ocean_conv_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                        int a_kmajor, int b_kn, int taps, int T, int shift0, int shift_step, long tap_stride_b,
                        long tap_stride_c, int split_k, void* stream) {
  GemmP p = gemm_defaults();
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.a_kmajor = a_kmajor; p.b_kn = b_kn;
  p.taps = taps; p.T = T; p.shift0 = shift0; p.shift_step = shift_step;
  p.tap_stride_b = tap_stride_b; p.tap_stride_c = tap_stride_c; p.split_k = split_k;
  return ocean_launch_gemm(p, (hipStream_t)stream);
}

int ocean_linear_bwd_f32(const float* x, int ldx, const float* dy, int lddy, const float* W, float* dW, float* db, float* dx, int lddx,
                         int M, int N, int K, int accumulate_dx, void* stream) {
  if (!x || !dy || !W || !dW || !dx) return OCEAN_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return OCEAN_ERR_BAD_SHAPE;
  return linear_bwd_pair(x, ldx, dy, lddy, W, dW, db, dx, lddx, M, N, K, accumulate_dx, (hipStream_t)stream);
}

int ocean_mean_mid_f32(const float* a, float* out, int G, int n, int C, void* stream) {
  if (!a || !out) return OCEAN_ERR_NULL;
  if (G <= 0 || n <= 0 || C <= 0) return OCEAN_ERR_BAD_SHAPE;
  return mean_mid(a, out, G, n, C, (hipStream_t)stream);
}

}  // extern "C"
