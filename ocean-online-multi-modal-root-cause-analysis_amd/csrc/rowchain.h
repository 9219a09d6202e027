// Two things small batches get from an RcScope (alive for the duration of one stage driver call on one thread):
//
// 1. Batched row reductions (default for batches up to OCEAN_WGRAD_BATCH_MAX_ROWS = 256 rows).  The weight and bias
//    gradients of a stage's linear layers are reductions over the batch rows; launched per layer each costs a dependent
//    launch however few rows there are.  While a scope is alive the GEMM launchers of gemm.hip do not launch row
//    reductions: they append them to a list, and the scope's end runs the whole list as ONE launch (rowchain.hip:
//    one block per 32x32 tile of one layer's gradient, its four waves splitting the rows, no atomics - the result does
//    not depend on scheduling).  A layer's data gradient then runs alone (6-7 us instead of 9-13 us for the pair).
//    The drivers keep every layer's output gradient in a buffer of its own for this (nothing the list reads is reused).
//
// 2. Stage kernels (opt-in: OCEAN_STAGE_FUSE=1).  The attention, fusion and prediction-head stages are chains of small
//    layers (Linear, LayerNorm, the cross-modal softmax, the feature gate, the three-token attention, ...) in which a
//    sample's outputs depend on that sample's rows only.  With the switch on, the helpers the stage drivers call (the
//    GEMM launchers, the row kernels of rowops.hip, the stage-specific row kernels) append operations to a program
//    instead of launching, and the scope's end launches ONE kernel whose blocks each carry a tile of 16 samples through
//    every operation, activations handed from operation to operation through LDS slots the host plans while recording
//    (and written through to their workspace buffers for the backward pass).  It passes every test, and it is not the
//    default because it is not faster: a block is one compute unit working through ~20 dependent steps, each of which
//    waits for its weights from L2 - all of the stage's ~8 MB pass through that one unit's load path - (tools/rc_trace.py
//    prints the per-operation timeline), so a chain costs more than
//    the launches it replaces (attention forward 155 us against 85 us at batch 128; update step 2.8 against 2.2 ms).
//    What would change that is streaming the weights of upcoming layers into LDS ahead of the layers that use them
//    (they do not depend on the activations) with dedicated producer waves; not built.
//
// An operation the program cannot express (odd alignment, a shape outside the kernel's limits) is not recorded: what
// has been recorded so far is launched, then the operation runs as its usual stand-alone launch - stream order is
// kept either way.
#pragma once
#include "common.h"

// Alive for the duration of one stage driver call; does nothing when another scope is alive on this thread or the batch
// is beyond both limits above.
struct RcScope {
  RcScope(int batch_rows, hipStream_t stream);
  ~RcScope();
  int finish();          // launches the chain and the collected reductions; returns the first error of the scope
  bool on;
};

// While alive, operations whose matrices have THREE rows per sample, token-major (row = 3 * sample + token: the fusion
// stage's [B*3, .] matrices), are accepted; without it only matrices with one row per sample are, so that a matrix of 3 B
// rows stacked modality-major ([3][B][.]: a sample's rows are B apart) can never be mistaken for one.
struct RcTokenMajor { RcTokenMajor(); ~RcTokenMajor(); };

bool rc_active();                                 // a stage kernel is being recorded (mode 2)
bool rc_collecting();                             // row reductions are being collected (modes 1 and 2)
int rc_batch();                                   // the scope's batch rows
// launch what has been recorded so far (the chain, not the collected reductions); call before any launch that is made
// directly while a scope is alive
void rc_flush();

// A GEMM problem: non-reducing forms (a_kmajor == 0: forward and data-gradient products) become chain operations,
// row reductions (a_kmajor == 1: weight gradients, with the bias gradient in a_colsum) are collected.  `independent`:
// the operation does not read what the previous recorded operation wrote (members of one batched launch).
// Returns true when taken; false after flushing the chain (the caller then launches as usual).
bool rc_gemm(const GemmP& p, bool independent);
// column sums added into out[N] (bias gradients that do not ride with a weight gradient)
bool rc_colsum(const float* a, int lda, int M, int N, float* out);

bool rc_ln_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats, int rows, int W,
               float eps, DropoutP in_drop, int drop_row0, bool independent);
bool rc_ln_bwd(const float* dy, int lddy, const float* x, int ldx, const float* stats, const float* gamma, float* dx, int lddx,
               float* dgamma, float* dbeta, int rows, int W, DropoutP in_drop, int drop_row0, bool independent);
bool rc_l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* norms, int rows, int W, float eps, bool independent);
bool rc_l2norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* norms, float* dx, int lddx, int rows, int W,
                   float eps, bool independent);
bool rc_cross_fwd(const float* Nn, const float* mw, float* G, float* cw, int B, int W);
bool rc_cross_bwd(const float* dG, const float* Nn, const float* cw, const float* mw, float* dN, float* dpi, int B, int W);
bool rc_gate_fwd(const float* hid, const float* w2, const float* b2, const float* G, float* ig, float* cat, int B, int W, int Wh);
bool rc_gate_bwd(const float* dcat, const float* G, const float* ig, const float* hid, const float* w2, float* dG, float* dz,
                 float* dhid, int B, int W, int Wh);
bool rc_mha3_fwd(const float* qkv, float* probs, float* o, int B, int E, int H, DropoutP dp);
bool rc_mha3_bwd(const float* dOm, const float* qkv, const float* probs, float* dqkv, int B, int E, int H, DropoutP dp);
bool rc_mean_mid(const float* a, float* out, int G, int n, int C);
bool rc_sigmoid(const float* x, float* y, int rows, int cols);
bool rc_sigmoid_bwd(const float* dlogits_in, const float* dprobs, const float* p, float* out, int rows, int cols);
bool rc_relu_bwd(float* dy, int lddy, const float* y, int ldy, int M, int N, bool independent);
bool rc_copy2d(float* dst, int ldd, const float* src, int lds, int M, int N, int accumulate);
