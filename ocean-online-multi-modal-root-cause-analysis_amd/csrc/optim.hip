// Gradient clipping + Adam in two launches over a chunk table.
//
// The update step of OnlineLearner / Trainer (online_learner.py:325-335,402-412;
// trainer.py:416-426) is clip_grad_norm_(max_norm) followed by torch.optim.Adam
// (coupled L2 weight decay).  Here: one launch accumulates the global squared
// gradient norm in fp64, one launch applies clip coefficient + weight decay +
// moment updates + parameter update, 28 bytes of HBM traffic per parameter.
// The table has one row per chunk of at most kChunk consecutive elements of one
// tensor: {param, grad, exp_avg, exp_avg_sq} addresses and the element count.
#include "../../include/ocean_hip.h"
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// A block walks every gridDim.x-th chunk and ends with ONE fp64 atomic: the adds all go to one address and serialise at
// the memory side (~10 ns each), so a block per chunk (1700 of them for 3.3 M parameters) spent most of its 20 us there.
// Chunks whose addresses are 16-byte aligned are read four elements at a time.
constexpr int kNormBlocks = 512, kAdamBlocks = 1024;
__global__ __launch_bounds__(256) void sqnorm_kernel(const ocean_adam_chunk* __restrict__ tab, int n_chunks, double* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int q = blockIdx.x; q < n_chunks; q += gridDim.x) {
    const ocean_adam_chunk c = tab[q];
    const float* g = reinterpret_cast<const float*>(c.grad);
    const long n4 = (c.grad & 15) == 0 ? c.n / 4 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const float4 v = reinterpret_cast<const float4*>(g)[i];
      s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
    }
    for (long i = 4 * n4 + threadIdx.x; i < c.n; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (double)(red[0] + red[1] + red[2] + red[3]));
}

__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, float coef, float wd, float b1, float b2, float step,
                                          float rs2, float eps) {
  g *= coef;
  const float gd = fmaf(wd, p, g);
  m = b1 * m + (1.f - b1) * gd;
  v = b2 * v + (1.f - b2) * gd * gd;
  p = p - step * m / (sqrtf(v) * rs2 + eps);
}

// sqnorm and step_dev may point into / be written through `tail` by the block that finishes last, so neither they nor
// `tail` are declared restrict or const-restrict, and both are read once, explicitly, before any chunk is touched.
__global__ __launch_bounds__(256) void clip_adam_kernel(const ocean_adam_chunk* __restrict__ tab, int n_chunks, const double* sqnorm,
                                                        float max_norm, float lr, float b1, float b2, float eps, float wd,
                                                        float bc1, float bc2, int write_clipped,
                                                        const float* __restrict__ lr_dev, int* step_dev,
                                                        ocean_adam_tail* tail) {
  int t_now = 0;
  if (lr_dev != nullptr) {     // learning rate and step count live on the device (HIP-graph replays)
    lr = lr_dev[0];
    t_now = __hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (tail != nullptr ? 1 : 0);      // with a tail the count is advanced here, not by the caller
    const float t = (float)t_now;
    bc1 = 1.0f - powf(b1, t);
    bc2 = 1.0f - powf(b2, t);
  }
  float coef = 1.f;
  const double sq_now = sqnorm ? __hip_atomic_load(sqnorm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  if (max_norm > 0.f && sqnorm) coef = fminf(1.f, max_norm / ((float)sqrt(sq_now) + 1e-6f));
  const float step = lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  for (int q = blockIdx.x; q < n_chunks; q += gridDim.x) {
    const ocean_adam_chunk c = tab[q];
    float* p = reinterpret_cast<float*>(c.param);
    float* g = reinterpret_cast<float*>(c.grad);
    float* m = reinterpret_cast<float*>(c.exp_avg);
    float* v = reinterpret_cast<float*>(c.exp_avg_sq);
    const long n4 = ((c.param | c.grad | c.exp_avg | c.exp_avg_sq) & 15) == 0 ? c.n / 4 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      adam_elem(pv.x, gv.x, mv.x, vv.x, coef, wd, b1, b2, step, rs2, eps);
      adam_elem(pv.y, gv.y, mv.y, vv.y, coef, wd, b1, b2, step, rs2, eps);
      adam_elem(pv.z, gv.z, mv.z, vv.z, coef, wd, b1, b2, step, rs2, eps);
      adam_elem(pv.w, gv.w, mv.w, vv.w, coef, wd, b1, b2, step, rs2, eps);
      if (write_clipped) reinterpret_cast<float4*>(g)[i] = gv;
      reinterpret_cast<float4*>(m)[i] = mv;
      reinterpret_cast<float4*>(v)[i] = vv;
      reinterpret_cast<float4*>(p)[i] = pv;
    }
    for (long i = 4 * n4 + threadIdx.x; i < c.n; i += 256) {
      float pv = p[i], gv = g[i], mv = m[i], vv = v[i];
      adam_elem(pv, gv, mv, vv, coef, wd, b1, b2, step, rs2, eps);
      if (write_clipped) g[i] = gv;
      m[i] = mv; v[i] = vv; p[i] = pv;
    }
  }
  if (tail == nullptr) return;
  // The block that finishes last leaves the state the next step starts from: the squared norm published, its
  // accumulator back at zero, the step count advanced.  Every block has read the accumulator and the count (and used
  // the values) before it signs off, and the last one writes only after all have signed off; nothing the blocks STORE
  // has to be ordered against the counter, so there is no fence here (one in front of the sign-off made every wave
  // wait for its parameter stores to drain: 20 -> 72 us for 3.3 M parameters).
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(&tail->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      tail->done = 0u;
      tail->sq_last = sq_now;
      tail->sq_acc = 0.0;
      step_dev[0] = t_now;
    }
  }
}

}  // namespace

extern "C" {

int ocean_grad_sqnorm(const ocean_adam_chunk* table, int n_chunks, double* sqnorm, void* stream) {
  if (!table || !sqnorm) return OCEAN_ERR_NULL;
  if (n_chunks <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(n_chunks < kNormBlocks ? n_chunks : kNormBlocks), dim3(256), 0, (hipStream_t)stream, table, n_chunks, sqnorm);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_clip_adam_step(const ocean_adam_chunk* table, int n_chunks, const double* sqnorm, float max_norm, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, int write_clipped, void* stream) {
  if (!table) return OCEAN_ERR_NULL;
  if (n_chunks <= 0 || step < 1) return OCEAN_ERR_BAD_SHAPE;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(n_chunks < kAdamBlocks ? n_chunks : kAdamBlocks), dim3(256), 0, (hipStream_t)stream, table, n_chunks, sqnorm, max_norm, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, write_clipped, (const float*)nullptr, (int*)nullptr, (ocean_adam_tail*)nullptr);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_clip_adam_step_dev(const ocean_adam_chunk* table, int n_chunks, const double* sqnorm, float max_norm, const float* lr_dev,
                             const int* step_dev, float beta1, float beta2, float eps, float weight_decay, int write_clipped,
                             void* stream) {
  if (!table || !lr_dev || !step_dev) return OCEAN_ERR_NULL;
  if (n_chunks <= 0) return OCEAN_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(n_chunks < kAdamBlocks ? n_chunks : kAdamBlocks), dim3(256), 0, (hipStream_t)stream, table, n_chunks, sqnorm, max_norm, 0.f, beta1, beta2, eps,
                     weight_decay, 1.f, 1.f, write_clipped, lr_dev, const_cast<int*>(step_dev), (ocean_adam_tail*)nullptr);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

int ocean_clip_adam_step_auto(const ocean_adam_chunk* table, int n_chunks, ocean_adam_tail* state, float max_norm, const float* lr_dev,
                              int* step_dev, float beta1, float beta2, float eps, float weight_decay, int write_clipped,
                              void* stream) {
  if (!table || !lr_dev || !step_dev || !state) return OCEAN_ERR_NULL;
  if (n_chunks <= 0) return OCEAN_ERR_BAD_SHAPE;
  if (max_norm > 0.f) {
    hipLaunchKernelGGL(sqnorm_kernel, dim3(n_chunks < kNormBlocks ? n_chunks : kNormBlocks), dim3(256), 0, (hipStream_t)stream, table, n_chunks,
                       &state->sq_acc);
    OCEAN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(clip_adam_kernel, dim3(n_chunks < kAdamBlocks ? n_chunks : kAdamBlocks), dim3(256), 0, (hipStream_t)stream, table, n_chunks, (const double*)&state->sq_acc, max_norm, 0.f,
                     beta1, beta2, eps, weight_decay, 1.f, 1.f, write_clipped, lr_dev, step_dev, state);
  OCEAN_LAUNCH_CHECK();
  return OCEAN_OK;
}

}  // extern "C"
