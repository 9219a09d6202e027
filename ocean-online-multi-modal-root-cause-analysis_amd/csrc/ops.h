// Internal (C++) building blocks shared by the component drivers.
#pragma once
#include "common.h"

// Optional epilogue of a linear-layer GEMM: dropout on the output (numbered like dropout_inplace on the [rows, N]
// matrix the output is rows [row0, row0 + M) of) and / or ReLU backward through a saved forward output.
struct Epi { const DropoutP* drop = nullptr; int group = 1; int row0 = 0; const float* relu_mask = nullptr; int ld_mask = 0; };
// y[M,N] = x[M,K] . W[N,K]^T (+ b) (ReLU)                (nn.Linear forward)
int linear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy,
               int M, int N, int K, int relu, hipStream_t s, const Epi* e = nullptr);
// dx[M,K] (+= if accumulate) = dy[M,N] . W[N,K]          (nn.Linear data gradient)
int linear_bwd_data(const float* dy, int lddy, const float* W, float* dx, int lddx,
                    int M, int N, int K, int accumulate, hipStream_t s, const Epi* e = nullptr);
// dW[N,K] += dy[M,N]^T . x[M,K]                          (nn.Linear weight gradient, atomic accumulate)
int linear_bwd_weight(const float* dy, int lddy, const float* x, int ldx, float* dW,
                      int M, int N, int K, hipStream_t s, float* db = nullptr);
// both gradients of one layer (dW += dy^T x, db += column sums of dy when non-null, dx (+)= dy W): one launch when the
// layer is small enough for the few-tile kernels, else the two calls above in that order
int linear_bwd_pair(const float* x, int ldx, const float* dy, int lddy, const float* W, float* dW, float* db, float* dx, int lddx,
                    int M, int N, int K, int accumulate, hipStream_t s, const Epi* e = nullptr);
// Batched forms: up to kGemmBatchMax independent layers (the three modality branches) per launch.
struct LinFwd { const float* x; int ldx; const float* W; const float* b; float* y; int ldy; int M, N, K; };
// drop: one dropout site over the entries' outputs stacked row-wise (entry q covers rows [q*M, (q+1)*M))
int linear_fwd_batch(const LinFwd* l, int n, int relu, hipStream_t s, const DropoutP* drop = nullptr, int drop_group = 1);
// one layer's backward: dW += dy^T x, db += colsum(dy), dx (+)= dy W; null outputs are skipped
struct LinBwd { const float* x; int ldx; const float* dy; int lddy; int M, N, K; const float* W; float* dW; float* db; float* dx; int lddx; };
// dx_drop / relu_masks: applied to the dx outputs (dropout site stacked as in linear_fwd_batch; relu_masks[q] is entry q's
// saved forward output, leading dimension ld_mask)
int linear_bwd_batch(const LinBwd* l, int n, int accumulate_dx, hipStream_t s, const DropoutP* dx_drop = nullptr, int drop_group = 1,
                     const float* const* relu_masks = nullptr, int ld_mask = 0);
struct ColSum { const float* a; int lda; int M, N; float* out; };
int colsum_add_batch(const ColSum* c, int n, hipStream_t s);
// out[N] += column sums of a[M,N]                         (bias gradients)
int colsum_add(const float* a, int lda, int M, int N, float* out, hipStream_t s);
// dy[i] = (y[i] > 0) ? dy[i] : 0   over an [M,N] matrix with leading dims    (ReLU backward on the saved output)
int relu_bwd_inplace(float* dy, int lddy, const float* y, int ldy, int M, int N, hipStream_t s);
// dst = 0
int zero_f32(float* dst, size_t n, hipStream_t s);
int zero_f64(double* dst, size_t n, hipStream_t s);
// dst[M,N] (ld) = src[M,N] (ld) ; or dst += src
int copy2d(float* dst, int ldd, const float* src, int lds, int M, int N, int accumulate, hipStream_t s);

// ---------------------------------------------------------------- row-wise ops (one 64-lane wave per row)
// y = LayerNorm(dropout(x)) * gamma + beta; stats[row] = {mean, rstd}.  in_drop.p == 0: no dropout.
int layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats,
                  int rows, int W, float eps, DropoutP in_drop, hipStream_t s);
// dx = d/dx of the above (dropout mask re-applied); dgamma/dbeta accumulated.  dx may alias dy.
int layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma,
                  float* dx, int lddx, float* dgamma, float* dbeta, int rows, int W, DropoutP in_drop, hipStream_t s);
// Batched LayerNorm: set q has its own tensors and parameters, all sets share the shape and leading dimensions;
// the dropout index of set q, row r, column c is (q*rows + r)*W + c.
struct LnFwdSet { const float* x; const float* gamma; const float* beta; float* y; float* stats; };
int layernorm_fwd_batch(const LnFwdSet* sets, int n, int ldx, int ldy, int rows, int W, float eps, DropoutP in_drop, hipStream_t s);
struct LnBwdSet { const float* dy; const float* x; const float* stats; const float* gamma; float* dx; float* dgamma; float* dbeta; };
int layernorm_bwd_batch(const LnBwdSet* sets, int n, int lddy, int ldx, int lddx, int rows, int W, DropoutP in_drop, hipStream_t s);
// y = x / max(||x||_2, eps) per row; norms[row] saved
int l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* norms, int rows, int W, float eps, hipStream_t s);
// dx = (dy - y * (y . dy)) / max(norm, eps); dx may alias dy
int l2norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* norms, float* dx, int lddx,
               int rows, int W, float eps, hipStream_t s);
// x *= keep(mask)/(1-p) elementwise over [M,N] (ld); idx = row*N + col.  `group` > 1: one mask value per
// `group` consecutive columns (per-head attention-weight dropout on a length-1 sequence).
int dropout_inplace(float* x, int ld, int M, int N, int group, DropoutP d, hipStream_t s);
// out[G, C] = mean over the middle axis of a[G, n, C]
int mean_mid(const float* a, float* out, int G, int n, int C, hipStream_t s);
