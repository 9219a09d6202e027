/*
 * C ABI of libocean_hip.so - the gfx950 (MI355X) kernels behind the OCEAN
 * per-step model hot path.
 *
 * The reference has no FFI layer: its boundary is the Python class API
 * (ocean/models/__init__.py:3-15).  The Python package in this repository
 * keeps those classes and binds these entry points with ctypes; each entry
 * point cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 unless stated;
 *  - nothing here allocates, frees or synchronises: the caller owns every
 *    buffer (outputs and workspaces included) and the work is enqueued on
 *    `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *  - return value: 0 on success, else an OceanStatus code (1 bad shape,
 *    2 unsupported size, 3 workspace too small, 4 null pointer, 5 beyond the
 *    built-in scratch array (ocean_infonce_fwd_static only),
 *    1000 + hipError_t for a launch failure);
 *  - gradient outputs named d<param> are ACCUMULATED into (+=); activation
 *    gradients are overwritten unless stated.
 */
#ifndef OCEAN_HIP_H
#define OCEAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCEAN_MAX_CONV_LAYERS 8

/* Library/ABI version: bumped when a signature changes. */
int ocean_abi_version(void);

/* Opt-in arithmetic for the large (full-tile) products of the convolutions: each fp32 operand is split into two bf16
 * values (x = hi + lo) on its way into LDS and the product formed as hi.hi + hi.lo + lo.hi by three bf16 matrix
 * instructions with fp32 accumulation - about 16 bits of each product instead of 24, at 2-2.5 times the fp32 rate.
 * mode 0 (default): fp32 products.  mode 1: every large product.  mode 2: only the products launched by the backward
 * drivers (ocean_dcnn_bwd*, ocean_gnn_bwd): the forward pass, and with it the sign of every ReLU input, stays fp32.
 * Process-wide (also: environment OCEAN_GEMM_SPLIT_BF16=1|2 at load time); returns the previous mode, or -1 for a mode
 * outside 0..2.  Set it before a step is recorded into a HIP graph - a recording keeps the kernels it
 * was made with.  No counterpart in the reference (torch fp32 on its device's default matmul path). */
int ocean_set_split_bf16(int mode);
int ocean_get_split_bf16(void);

/* ------------------------------------------------------------------ GEMM
 * C[M,N] = A . B (+bias)(+R)(ReLU).  fp32 in, fp32 accumulate on the matrix
 * cores.  a_kmajor: A stored [K][M] instead of [M][K]; b_kn: B stored [K][N]
 * instead of [N][K].  With a_kmajor the result is atomically ADDED to C.
 * Replaces torch.nn.functional.linear / torch.mm wherever the reference calls
 * them on the model path (e.g. multi_factor_attention.py:332-334). */
int ocean_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   int M, int N, int K, int a_kmajor, int b_kn,
                   const float* bias, const float* R, int ldr, int relu, void* stream);

/* The same GEMM with the dilated-convolution row shifts exposed (taps accumulate; rows are (sample,time)
 * pairs with time = row % T; tap t reads rows shifted by shift0 + t*shift_step, zero outside [0,T)).
 * a_kmajor == 0: shift on A's rows, B advances tap_stride_b per tap.  a_kmajor == 1: shift on B's rows,
 * each tap accumulates into C + t*tap_stride_c, reduction split split_k ways (0 = automatic).
 * This is the implicit-GEMM form of nn.Conv1d in DilatedConvBlock (dilated_cnn.py:48-55) and its two
 * gradients; exported for kernel-level tests and benchmarks. */
int ocean_conv_gemm_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                        int a_kmajor, int b_kn, int taps, int T, int shift0, int shift_step, long tap_stride_b,
                        long tap_stride_c, int split_k, void* stream);

/* Backward of y[M,N] = x[M,K] . W[N,K]^T + b: dW[N,K] += dy^T . x, db[N] += column sums of dy (NULL: skipped),
 * dx[M,K] = dy . W (accumulate_dx: +=).  A small layer's two products share one launch; dx must not overlap x, dy or
 * the gradients.  Replaces autograd's backward of every nn.Linear on the model path (e.g. the head,
 * ocean/models/ocean_model.py:112-121); exported for kernel-level tests. */
int ocean_linear_bwd_f32(const float* x, int ldx, const float* dy, int lddy, const float* W, float* dW, float* db, float* dx,
                         int lddx, int M, int N, int K, int accumulate_dx, void* stream);

/* out[G,C] = mean over the middle axis of a[G,n,C].  Replaces torch.mean(logs, dim=1) for
 * [B,L,768] log embeddings (ocean/models/ocean_model.py:179-181). */
int ocean_mean_mid_f32(const float* a, float* out, int G, int n, int C, void* stream);

/* ------------------------------------------------------------------ DilatedCNN
 * Replaces DilatedCNN.forward / DilatedConvBlock.forward
 * (ocean/models/components/dilated_cnn.py:76-99, 195-241) and their autograd
 * backward.  Activations are channels-last [B*T, C]; every Conv1d is an
 * implicit GEMM with symmetric zero padding (k-1)*d/2 (dilated_cnn.py:45).
 */
typedef struct {
  int B, T, F;                               /* input metrics [B,T,F]                      */
  int L;                                     /* number of dilated blocks (1..8)            */
  int channels[OCEAN_MAX_CONV_LAYERS];       /* block widths                               */
  int dilations[OCEAN_MAX_CONV_LAYERS];
  int kernel_size;                           /* odd                                        */
  int out_dim;                               /* output_projection width                    */
  int training;                              /* 1: batch statistics + running-stat update  */
  float eps, momentum;                       /* BatchNorm1d: 1e-5, 0.1                     */
  long bn_count;                             /* rows behind the BN statistics; B*T, or the
                                                global row count when ranks all-reduce them */
  float dropout;                             /* p of every nn.Dropout (dilated_cnn.py:58,175);
                                                applied only when training                  */
  uint64_t seed;                             /* dropout mask seed of this forward/backward   */
  const uint64_t* seed_epoch;                /* device counter folded into the seed inside the
                                                kernels, or NULL: lets a captured HIP graph draw
                                                fresh masks on every replay (bump it in between);
                                                same meaning in every *_shape below            */
} ocean_dcnn_shape;

/* Parameter table order (pointers to device fp32):
 *   [0] input_projection.weight [C0,F]  [1] input_projection.bias [C0]
 *   then for block i in 0..L-1:   conv.weight [Ci,Ci,k], bn.weight, bn.bias, bn.running_mean, bn.running_var
 *   then for transition i in 0..L-2: conv.weight [Ci+1,Ci,1], bn.weight, bn.bias, bn.running_mean, bn.running_var
 *   then output_projection.weight [out,C_last], output_projection.bias [out].
 * The gradient table has the same order; running_mean/var slots are ignored. */
size_t ocean_dcnn_param_count(const ocean_dcnn_shape* s);

/* Workspace sizes in floats.  The forward region must stay untouched between
 * ocean_dcnn_fwd and ocean_dcnn_bwd (it holds the saved activations). */
size_t ocean_dcnn_fwd_workspace(const ocean_dcnn_shape* s);
size_t ocean_dcnn_bwd_workspace(const ocean_dcnn_shape* s);

/* Number of phases of the forward / backward drivers.  A phase boundary is a
 * point where BatchNorm column statistics are complete on this rank; a
 * data-parallel caller all-reduces them there.  phase range [lo,hi). */
int ocean_dcnn_num_phases(const ocean_dcnn_shape* s);
/* Offset (in floats, into the fwd / bwd workspace) and length (in doubles) of
 * the statistics block [2][C] (fp64) that is complete after phase `phase`. */
int ocean_dcnn_stats_block(const ocean_dcnn_shape* s, int phase, int backward,
                           size_t* offset_floats, int* n_doubles);

/* Offset (in floats, into the forward workspace) and width C of dilated block `block`'s output activation
 * [B*T, C] (after BatchNorm, ReLU, dropout and the residual), valid after ocean_dcnn_fwd.  It is what
 * DilatedCNN.get_temporal_features pools per layer (dilated_cnn.py:243-277).  The LAST block's activation is not kept
 * when its width is a multiple of 4 (the pass that would write it pools it over time instead): status 2, use
 * ocean_dcnn_pooled_output. */
int ocean_dcnn_block_output(const ocean_dcnn_shape* s, int block, size_t* offset_floats, int* width);

/* Offset (in floats, into the forward workspace) and width of the last block's output averaged over time, [B, C]:
 * the input of DilatedCNN's output projection (dilated_cnn.py:221-223, global_pooling == 'mean'), valid after
 * ocean_dcnn_fwd. */
int ocean_dcnn_pooled_output(const ocean_dcnn_shape* s, size_t* offset_floats, int* width);

int ocean_dcnn_fwd(const ocean_dcnn_shape* s, const float* x, float* const* params,
                   float* ws, size_t ws_floats, float* out /* [B,out_dim] */,
                   int phase_lo, int phase_hi, void* stream);
int ocean_dcnn_bwd(const ocean_dcnn_shape* s, const float* x, float* const* params,
                   float* ws_fwd, float* ws_bwd, size_t ws_bwd_floats,
                   const float* dout /* [B,out_dim] */, float* const* grads,
                   float* dx /* [B,T,F] or NULL */, int phase_lo, int phase_hi, void* stream);

/* One backward phase in two calls, for data-parallel callers that all-reduce the statistics block between phases:
 * part 0 launches what the next phase's statistics wait for (the layer's BatchNorm-backward apply pass, its data
 * gradient, the column sums of the layer below), part 1 the rest (the layer's weight gradient; in the last phase the
 * input projection's gradients and the unpacking of the per-tap weight gradients).  Start the all-reduce after part 0,
 * run part 1 under it, wait before the next phase's part 0.  Phase 0 is all part 0.  No counterpart in the reference
 * (which has no distributed code); ocean_dcnn_bwd(phase, phase + 1) == part 0 then part 1. */
int ocean_dcnn_bwd_part(const ocean_dcnn_shape* s, const float* x, float* const* params,
                        float* ws_fwd, float* ws_bwd, size_t ws_bwd_floats,
                        const float* dout, float* const* grads, float* dx, int phase, int part, void* stream);

/* ------------------------------------------------------------------ GraphNeuralNetwork (GAT)
 * Replaces GraphNeuralNetwork.forward with conv_type='gat', pooling='mean',
 * batch_norm=False (ocean/models/components/graph_neural_network.py:236-302),
 * the torch_geometric GATConv / global_mean_pool it calls (:187-193, :258, :290)
 * and the O(B^2) batched-graph construction of ocean_model.py:155-171, plus
 * their backward.  One graph, R replicas: node tensors are [R*N, .].
 *
 * graph[] = { rowptr_in[N+1], src_in[E], rowptr_out[N+1], dst_out[E], eid_out[E] }
 * device int32 arrays: incoming edges grouped by target (self loops already
 * removed and re-added once per node), outgoing edges grouped by source, and
 * for each outgoing edge its position in the by-target order.
 */
typedef struct {
  int N, R;                 /* nodes per graph, replicas computed (1 when every replica is identical) */
  int G, D, H;              /* node-feature width, hidden width (= H * per-head width), heads          */
  int layers;               /* GAT layers; all concatenate heads except the last, which averages them  */
  int E;                    /* edges in the CSRs, self loops included                                  */
  int pool_n;               /* nodes per mean-pooled group (0 = N: one output row per replica)         */
  int training;
  float dropout;            /* feature and attention dropout p (graph_neural_network.py:191,218)       */
  float negative_slope;     /* LeakyReLU slope of GATConv (0.2)                                        */
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_gnn_shape;

/* params: [0] input_projection.weight [D,G] [1] .bias [D]; per layer l: lin.weight [D,D], att_src [D],
 * att_dst [D], bias [D or D/H]; then output_projection.weight [D, D/H], .bias [D].  grads: same order. */
size_t ocean_gnn_fwd_workspace(const ocean_gnn_shape* s);
size_t ocean_gnn_bwd_workspace(const ocean_gnn_shape* s);
/* Offset (floats, into the forward workspace) and width of GAT layer `layer`'s output [R*N, width] (after ReLU, dropout
 * and the residual), valid after ocean_gnn_fwd: the node embeddings GraphNeuralNetwork.get_node_embeddings returns
 * (graph_neural_network.py:320-356). */
int ocean_gnn_layer_output(const ocean_gnn_shape* s, int layer, size_t* offset_floats, int* width);
/* Which kernel family the edge phase (scores, edge softmax, aggregation) of this shape runs on: 2 = one block per
 * replica / node range with the per-node tables and CSR in LDS (small graphs), 1 = one node per wave with edge-parallel
 * lanes reading CSR from memory (graphs whose tables exceed 64 KB of LDS), 0 = generic one-edge-at-a-time kernels
 * (head counts / widths the other two do not cover); -1 = bad shape.  No counterpart in the reference. */
int ocean_gnn_edge_kernel_family(const ocean_gnn_shape* s);
int ocean_gnn_fwd(const ocean_gnn_shape* s, const float* node_features /* [N,G] */, const int* const* graph,
                  float* const* params, float* ws, size_t ws_floats, float* out /* [R*N/pool_n, D] */, void* stream);
int ocean_gnn_bwd(const ocean_gnn_shape* s, const float* node_features, const int* const* graph,
                  float* const* params, float* ws_fwd, float* ws_bwd, size_t ws_bwd_floats,
                  const float* dout /* [R*N/pool_n, D] */, float* const* grads, void* stream);

/* ------------------------------------------------------------------ MultiFactorAttention
 * Replaces MultiFactorAttention.forward with fusion_strategy='concatenate'
 * (ocean/models/components/multi_factor_attention.py:312-440) and backward.
 * params (37 pointers): temporal/spatial/log_projection (w,b) x3;
 * {temporal,spatial,log}_self_attention (w_v.w, w_v.b, w_o.w, w_o.b) x3;
 * layer_norms temporal/spatial/log/output (weight,bias) x4; modality_weights[3];
 * feature_attention.0 (w,b), feature_attention.2 (w,b); fusion_layer (w,b);
 * output_projection.0 (w,b), output_projection.3 (w,b).  grads: same order. */
typedef struct {
  int B, td, sd, ld;        /* rows; widths of the temporal / spatial / log inputs */
  int W, heads;             /* attention_dim, num_heads                              */
  int training;
  float dropout, eps;       /* attention_dropout; LayerNorm eps (1e-5)               */
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_mfa_shape;
size_t ocean_mfa_fwd_workspace(const ocean_mfa_shape* s);
size_t ocean_mfa_bwd_workspace(const ocean_mfa_shape* s);
int ocean_mfa_fwd(const ocean_mfa_shape* s, const float* temporal, const float* spatial, const float* logs,
                  float* const* params, float* ws, size_t ws_floats, float* out /* [B,W] */, void* stream);
int ocean_mfa_bwd(const ocean_mfa_shape* s, const float* temporal, const float* spatial, const float* logs,
                  float* const* params, float* ws_fwd, float* ws_bwd, size_t ws_bwd_floats, const float* dout,
                  float* const* grads, float* d_temporal, float* d_spatial, float* d_logs /* each may be NULL */,
                  void* stream);

/* ------------------------------------------------------------------ DilatedConvBlock on its own
 * Replaces DilatedConvBlock.forward (dilated_cnn.py:76-99) for callers outside DilatedCNN (whose blocks ocean_dcnn_*
 * runs): out = Dropout(ReLU(BatchNorm(Conv1d(x)))) + residual, residual = x when Cin == Cout (res_conv = 0), else
 * BatchNorm(Conv1d 1x1 (x)) (res_conv = 1).  x, out, dx: channels-last [B*T, C] rows (sample, time).  Odd kernel sizes
 * (symmetric padding (k-1)*dilation/2 keeps the length).  Training mode needs B*T >= 2 and updates the running statistics.
 * params / grads: {conv.weight [Cout,Cin,k], batch_norm.weight, .bias, .running_mean, .running_var,
 * residual_conv.weight [Cout,Cin,1], residual_bn.weight, .bias, .running_mean, .running_var} (the last five only with
 * res_conv); gradients are ADDED to grads (entries 3, 4, 8, 9 - the buffers - are ignored; NULL entries skipped). */
typedef struct {
  int B, T, Cin, Cout, kernel, dilation;
  int training, res_conv;
  float dropout, eps, momentum;
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_convblock_shape;
size_t ocean_convblock_fwd_workspace(const ocean_convblock_shape* s);
size_t ocean_convblock_bwd_workspace(const ocean_convblock_shape* s);
int ocean_convblock_fwd(const ocean_convblock_shape* s, const float* x, float* const* params, float* ws, size_t ws_floats,
                        float* out, void* stream);
int ocean_convblock_bwd(const ocean_convblock_shape* s, const float* x, float* const* params, const float* ws_fwd,
                        const float* dy, float* ws_bwd, size_t ws_bwd_floats, float* const* grads, float* dx /* or NULL */,
                        void* stream);

/* ------------------------------------------------------------------ one GATConv layer on its own
 * torch_geometric.nn.GATConv(in, C, heads = H, concat, negative_slope, dropout)(x, edge_index) as the reference's GNN
 * uses it (graph_neural_network.py:82-90; the published layer: one shared linear map without bias, attention
 * logits from att_src / att_dst, LeakyReLU, softmax over each target's incoming edges incl. its self loop, dropout on
 * the coefficients, sum, heads concatenated or averaged, + bias) for callers outside GraphNeuralNetwork (whose layers
 * ocean_gnn_* runs).  graph[]: the five CSR arrays described at ocean_gnn_shape, E edges incl. self loops.  H*C <= 512.
 * params / grads: {lin.weight [H*C, in], att_src [H*C], att_dst [H*C], bias [H*C or C] (or NULL)}; gradients are ADDED
 * to grads (the first three are required); dx [N, in] may be NULL. */
typedef struct {
  int N, in_dim, H, C, E;
  int concat, training;
  float negative_slope, dropout;
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_gatconv_shape;
size_t ocean_gatconv_fwd_workspace(const ocean_gatconv_shape* s);
size_t ocean_gatconv_bwd_workspace(const ocean_gatconv_shape* s);
int ocean_gatconv_fwd(const ocean_gatconv_shape* s, const float* x, const int* const* graph, float* const* params, float* ws,
                      size_t ws_floats, float* out /* [N, H*C] or [N, C] */, void* stream);
int ocean_gatconv_bwd(const ocean_gatconv_shape* s, const float* x, const int* const* graph, float* const* params, float* ws_fwd,
                      const float* dy, float* ws_bwd, size_t ws_bwd_floats, float* const* grads, float* dx, void* stream);

/* ------------------------------------------------------------------ MultiHeadAttention on its own
 * Replaces MultiHeadAttention.forward and ScaledDotProductAttention.forward (multi_factor_attention.py:108-148,
 * 34-65) for callers outside MultiFactorAttention (whose length-1 sequences ocean_mfa_* handles): query / key / value
 * [B, L, D] of equal length, mask [B, L, L] (0 = masked: score -1e9) or NULL, dropout on the attention weights and on
 * the output.  attn [B, heads, L, L] holds the weights as the reference returns them (after dropout).  L <= 1024.
 * params / grads: {w_q.weight, w_q.bias, w_k.weight, w_k.bias, w_v.weight, w_v.bias, w_o.weight, w_o.bias}; gradients are
 * ADDED to grads (NULL entries skipped); d_out or d_attn may be NULL (not both); dq / dk / dv may be NULL.
 * project == 0 is ScaledDotProductAttention.forward alone (:34-65): query / key / value are the heads' vectors laid out
 * [B, L, heads * d], no parameters (params / grads may be NULL), `out` is the attended values, dq / dk / dv are required. */
typedef struct {
  int B, L, D, heads;
  int training;
  int project;              /* 1: the four linear layers around the attention; 0: the attention alone */
  float dropout;
  float scale;              /* factor on the scores; 0: 1/sqrt(D/heads) */
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_mha_shape;
size_t ocean_mha_fwd_workspace(const ocean_mha_shape* s);
size_t ocean_mha_bwd_workspace(const ocean_mha_shape* s);
int ocean_mha_fwd(const ocean_mha_shape* s, const float* query, const float* key, const float* value, const float* mask,
                  float* const* params, float* ws, size_t ws_floats, float* out /* [B,L,D] */,
                  float* attn /* [B,heads,L,L] */, void* stream);
int ocean_mha_bwd(const ocean_mha_shape* s, const float* query, const float* key, const float* value, float* const* params,
                  const float* ws_fwd, const float* attn, const float* d_out, const float* d_attn, float* ws_bwd,
                  size_t ws_bwd_floats, float* const* grads, float* dq, float* dk, float* dv, void* stream);

/* The rest of CrossModalAttention.forward after its MultiHeadAttention (multi_factor_attention.py:205-215):
 * y1 = LayerNorm(query + attended); out = LayerNorm(y1 + Dropout(Linear(Dropout(ReLU(Linear(y1)))))), rows = B * L.
 * params / grads: {layer_norm.weight, .bias, feed_forward.0.weight [F,D], .0.bias, feed_forward.3.weight [D,F], .3.bias,
 * ff_layer_norm.weight, .bias}; gradients are ADDED to grads; d_sum receives the gradient of (query + attended),
 * which is the gradient of each of the two.  D <= 1024. */
typedef struct {
  int rows, D, F;
  int training;
  float dropout, eps;
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_xattn_tail_shape;
size_t ocean_xattn_tail_fwd_workspace(const ocean_xattn_tail_shape* s);
size_t ocean_xattn_tail_bwd_workspace(const ocean_xattn_tail_shape* s);
int ocean_xattn_tail_fwd(const ocean_xattn_tail_shape* s, const float* query, const float* attended, float* const* params,
                         float* ws, size_t ws_floats, float* out, void* stream);
int ocean_xattn_tail_bwd(const ocean_xattn_tail_shape* s, float* const* params, const float* ws_fwd, const float* d_out,
                         float* ws_bwd, size_t ws_bwd_floats, float* const* grads, float* d_sum, void* stream);

/* ------------------------------------------------------------------ GraphFusionModule
 * Replaces GraphFusionModule.forward with fusion_strategy='attention'
 * (ocean/models/components/graph_fusion.py:388-473) including
 * nn.MultiheadAttention over the 3 modality tokens (:353-358) and, when
 * `contrastive`, the three ProjectionHeads (:77-92); and backward.
 * params (34 pointers): {temporal,spatial,log}_processor (0.weight, 0.bias, 3.weight, 3.bias) x3;
 * attention_fusion in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias; attention_norm (w,b);
 * output_projection.0 (w,b), .3 (w,b); projection_heads.{temporal,spatial,log}.projection (0.w,0.b,3.w,3.b) x3. */
typedef struct {
  int B, td, sd, ld;
  int E, Pd, heads;         /* fusion_dim, projection_dim, attention heads (8) */
  int training, contrastive;
  float dropout, eps;
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_fusion_shape;
size_t ocean_fusion_fwd_workspace(const ocean_fusion_shape* s);
size_t ocean_fusion_bwd_workspace(const ocean_fusion_shape* s);
int ocean_fusion_fwd(const ocean_fusion_shape* s, const float* temporal, const float* spatial, const float* logs,
                     float* const* params, float* ws, size_t ws_floats, float* fused /* [B,E] */,
                     float* processed /* [B,3,E] */, float* projections /* [3,B,Pd] or NULL */, void* stream);
int ocean_fusion_bwd(const ocean_fusion_shape* s, const float* temporal, const float* spatial, const float* logs,
                     float* const* params, float* ws_fwd, const float* processed, const float* projections,
                     float* ws_bwd, size_t ws_bwd_floats, const float* d_fused /* or NULL */,
                     const float* d_processed /* or NULL */, const float* d_projections /* or NULL */,
                     float* const* grads, float* d_temporal, float* d_spatial, float* d_logs, void* stream);

/* One ProjectionHead called on its own (ProjectionHead.forward, graph_fusion.py:77-92, as
 * ContrastiveLearningModule.forward calls it per modality, :217-221): y = normalize(W1 . dropout(relu(W0 x + b0)) + b1),
 * two layers with ReLU (the reference's default); normalize == 0 leaves the last Linear's output as it is.
 * params / grads: {0.weight [hidden,in], 0.bias, 3.weight [out,hidden], 3.bias}; gradients are ADDED to grads
 * (a NULL entry skips that gradient alone); dx may be NULL (the input's gradient is then not formed). */
typedef struct {
  int B, in_dim, hidden, out_dim;
  int training, normalize;
  float dropout;
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_proj_shape;
size_t ocean_projection_head_fwd_workspace(const ocean_proj_shape* s);
size_t ocean_projection_head_bwd_workspace(const ocean_proj_shape* s);
int ocean_projection_head_fwd(const ocean_proj_shape* s, const float* x, float* const* params, float* ws, size_t ws_floats,
                              float* y /* [B,out] */, void* stream);
int ocean_projection_head_bwd(const ocean_proj_shape* s, const float* x, float* const* params, const float* ws_fwd,
                              const float* y, const float* dy, float* ws_bwd, size_t ws_bwd_floats, float* const* grads,
                              float* dx /* [B,in] or NULL */, void* stream);

/* ------------------------------------------------------------------ InfoNCE
 * Replaces InfoNCELoss.forward (graph_fusion.py:112-162) and backward.
 * K > 0: memory-bank branch, logits = [a.p, a.n_1 .. a.n_K]/tau, label 0 (Bp == B).
 * K == 0: in-batch branch, logits = a . p^T / tau over Bp positives (all ranks'
 * rows when data-parallel), label of row b = label_offset + b.
 * *loss is ACCUMULATED (+= mean over mean_rows, 0 -> B). */
typedef struct {
  int B, Bp, P, K;
  int label_offset, mean_rows;
  float tau;
} ocean_infonce_shape;
size_t ocean_infonce_workspace(const ocean_infonce_shape* s);
size_t ocean_infonce_bwd_workspace(const ocean_infonce_shape* s);
int ocean_infonce_fwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                      float* ws, size_t ws_floats, float* loss, void* stream);
int ocean_infonce_bwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, float* ws, float* ws_bwd,
                      size_t ws_bwd_floats, const float* dloss /* device scalar or NULL (=1) */, float* d_anchor /* [B,P] */,
                      float* d_positive /* [Bp,P] */, void* stream);

/* reduction='none' of InfoNCELoss (graph_fusion.py:100, 160: F.cross_entropy(..., reduction=self.reduction)): the same
 * computation with one loss value per anchor row written to row_loss[B] (overwritten, not accumulated; mean_rows is
 * ignored), and a backward that takes one incoming gradient per row. */
int ocean_infonce_rows_fwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                           float* ws, size_t ws_floats, float* row_loss /* [B] */, void* stream);
int ocean_infonce_rows_bwd(const ocean_infonce_shape* s, const float* anchor, const float* positive, float* ws, float* ws_bwd,
                           size_t ws_bwd_floats, const float* d_row_loss /* [B] */, float* d_anchor, float* d_positive,
                           void* stream);

/* ocean_infonce_fwd for callers that cannot allocate device memory (code under HIP-graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the same computation and result, with the workspace taken from an array
 * built into the library instead of a caller-provided buffer.  No counterpart in the reference.  Neither this call nor
 * anything it uses allocates or frees.  Limits and sharing:
 *  - ocean_infonce_workspace(s) must not exceed OCEAN_STATIC_SCRATCH_FLOATS (e.g. B = Bp = 1024, P = 128 in-batch needs
 *    1.32 M floats and fits); larger problems are refused with status 5 (use ocean_infonce_fwd);
 *  - each GPU holds one copy of the array (6 MB) from the library's load to the end of the process, shared by every call
 *    on that GPU: calls that overlap on one GPU (two streams, two threads) race on it - serialise them;
 *  - the backward pass needs the state the forward leaves in its workspace, so this variant is forward-only;
 *  - `stream` must belong to the CURRENT device (the array's copy on the current device is the one used);
 *  - the first call on a device looks the array's address up (hipGetSymbolAddress), which a stream capture in progress
 *    may refuse: before recording this call into a HIP graph, call ocean_static_scratch_warmup() (or this entry point)
 *    once on that device outside the capture. */
#define OCEAN_STATIC_SCRATCH_FLOATS (1536 * 1024)
int ocean_infonce_fwd_static(const ocean_infonce_shape* s, const float* anchor, const float* positive, const float* negatives,
                             float* loss /* ACCUMULATED, as in ocean_infonce_fwd */, void* stream);
/* Resolves the built-in array's address for the current device; launches nothing.  No counterpart in the reference. */
int ocean_static_scratch_warmup(void);

/* Diagnostics of the stage kernels (the attention / fusion / head chains that small batches run as one launch each; no
 * counterpart in the reference): while `buffer` - device memory of at least 60 64-bit words, or NULL to stop - is set,
 * block 0 of every such kernel writes (operation kind * 1000 + width, 100 MHz clock stamp) per operation, then
 * (-1, stamp at the end).  Not thread-safe; for measurement scripts. */
void ocean_rc_debug_trace(void* buffer);

/* The three in-batch pairs of ContrastiveLearningModule.forward in one call (graph_fusion.py:231-249 builds the pairs
 * (temporal, spatial), (temporal, log), (spatial, log) and calls InfoNCELoss.forward, :112-162, on each).
 * Z [3, B, P] holds the three projections; losses[q] is ACCUMULATED (+= mean over B rows) for pair q in that order.
 * Bp == 0: the positives are Z's own rows (one process); same values as three ocean_infonce_fwd calls with K == 0.
 * Bp > 0 (data parallel): G [3, Bp, P] holds the projections gathered from every rank; local row r is matched with
 * gathered row label_offset + r; the backward call then also returns dG [3, Bp, P] (for the gather's reduce-scatter).
 * The forward workspace carries the state the backward call needs (it is modified by it). */
typedef struct {
  int B, P;
  float tau;
  int Bp, label_offset;
} ocean_infonce3_shape;
size_t ocean_infonce3_workspace(const ocean_infonce3_shape* s);
size_t ocean_infonce3_bwd_workspace(const ocean_infonce3_shape* s);
int ocean_infonce3_fwd(const ocean_infonce3_shape* s, const float* Z, const float* G /* NULL iff Bp == 0 */, float* ws,
                       size_t ws_floats, float* losses /* [3] */, void* stream);
int ocean_infonce3_bwd(const ocean_infonce3_shape* s, float* ws, float* ws_bwd, size_t ws_bwd_floats,
                       const float* dlosses /* device [3] or NULL (= 1, 1, 1) */, float* dZ /* [3, B, P] */,
                       float* dG /* [3, Bp, P]; NULL iff Bp == 0 */, void* stream);

/* ContrastiveLearningModule._update_memory_bank (graph_fusion.py:251-273) as a ring, pointer on the device: rows [R, P]
 * go to bank[(ptr + r) % K] (bank [K, P]; of more than K rows the last K survive), then *ptr (int64, the module's
 * memory_ptr buffer) advances by the reference's formula: (ptr+R) % K while ptr+R <= K, ptr+R-K while <= 2K (which
 * yields K itself at ptr+R == 2K, as the reference stores it), (ptr+R) % K beyond (where the reference raises).
 * Keeping the pointer off the host lets the write sit in a captured HIP graph. */
int ocean_bank_ring_write(float* bank, long long* ptr, const float* rows, int R, int K, int P, void* stream);

/* out [n_perm, B] int64: n_perm independent uniform random permutations of 0..B-1 (B <= 4096; larger B returns
 * OCEAN_ERR_UNSUPPORTED and the caller draws them itself), a function of seed (+ *seed_epoch, see ocean_dcnn_shape).
 * Replaces torch.randperm(batch_size) in _compute_mutual_information (graph_fusion.py:494), one launch for the three
 * modalities. */
int ocean_draw_permutations(int n_perm, int B, uint64_t seed, const uint64_t* seed_epoch, long long* out, void* stream);

/* MINE estimates of GraphFusionModule._compute_mutual_information (graph_fusion.py:475-501) for the three modalities:
 * mi[m] = mean_b f(cat(fused_b, X_{b,m})) - log(mean_b exp f(cat(fused_b, X_{perms[m][b],m})) + 1e-8),
 * f = Linear(E -> 1)(ReLU(Linear(2E -> E)(.))) with weights w0 [E, 2E], b0 [E], w1 [E], b1 [1].
 * X is [B, 3, E] (the processed temporal / spatial / log features side by side), perms [3, B] int64 row permutations
 * drawn by the caller.  Diagnostics only: no gradient. */
typedef struct {
  int B, E;
} ocean_mi_shape;
size_t ocean_mi_workspace(const ocean_mi_shape* s);
int ocean_mi_estimates(const ocean_mi_shape* s, const float* fused, const float* X, const long long* perms, const float* w0,
                       const float* b0, const float* w1, const float* b1, float* ws, size_t ws_floats, float* mi /* [3] */,
                       void* stream);

/* ------------------------------------------------------------------ prediction head + sigmoid
 * Replaces prediction_head + output_activation (ocean/models/ocean_model.py:112-125, 205-206).
 * params: 0.weight [h1,E], 0.bias, 3.weight [h2,h1], 3.bias, 6.weight [S,h2], 6.bias. */
typedef struct {
  int B, E, h1, h2, S;
  int training;
  float dropout;            /* 0.1 hard-coded in the reference (ocean_model.py:117,120) */
  uint64_t seed;
  const uint64_t* seed_epoch;
} ocean_head_shape;
size_t ocean_head_workspace(const ocean_head_shape* s);
size_t ocean_head_bwd_workspace(const ocean_head_shape* s);
int ocean_head_fwd(const ocean_head_shape* s, const float* x, float* const* params, float* ws, size_t ws_floats,
                   float* logits, float* probs, void* stream);
int ocean_head_bwd(const ocean_head_shape* s, const float* x, float* const* params, float* ws_fwd, float* ws_bwd,
                   size_t ws_bwd_floats, const float* probs, const float* d_logits /* or NULL */,
                   const float* d_probs /* or NULL */, float* const* grads, float* dx /* or NULL */, void* stream);

/* OCEANModel.compute_loss's arithmetic on the device scalars (ocean_model.py:257-266), in its order of operations:
 * out3 = [wp * pred + wc * (0 + c[0] + ... + c[n_c-1]),  pred,  0 + c[0] + ...]   (n_c <= 64; c may be NULL when 0).
 * Backward: d_pred = wp * d_total, d_c[q] = wc * d_total + d_csum (a NULL upstream gradient counts as zero). */
int ocean_loss_combine_fwd(const float* pred, const float* c, int n_c, float wp, float wc, float* out3, void* stream);
int ocean_loss_combine_bwd(const float* d_total, const float* d_csum, int n_c, float wp, float wc, float* d_pred,
                           float* d_c, void* stream);

/* The prediction loss and compute_loss's arithmetic together (ocean_model.py:249-266): pred = BCE(probs, targets)
 * summed and divided by mean_n (0 -> n), then out3 as ocean_loss_combine_fwd writes it; one launch each way instead
 * of three and two.  The per-block sums are added in block order by the block that finishes last, so the value does
 * not vary from run to run.  `state` is zero-initialised device memory that the caller keeps for these calls (one per
 * device; calls that overlap on one GPU must not share it).  Backward: d_probs = (wp * d_total + d_pred) * dBCE,
 * d_c[q] = wc * d_total + d_csum (a NULL upstream gradient counts as zero). */
#define OCEAN_LOSS_STATE_BLOCKS 512
typedef struct {
  float partial[OCEAN_LOSS_STATE_BLOCKS];   /* per-block sums of the call under way */
  unsigned done;                            /* blocks that have finished; zero between calls */
  unsigned pad_[3];
} ocean_loss_state;
int ocean_pred_total_fwd(const float* probs, const float* targets, long n, long mean_n, const float* c, int n_c,
                         float wp, float wc, ocean_loss_state* state, float* out3, void* stream);
int ocean_pred_total_bwd(const float* probs, const float* targets, const float* d_total, const float* d_csum,
                         const float* d_pred, long n, long mean_n, int n_c, float wp, float wc, float* d_probs,
                         float* d_c, void* stream);

/* F.binary_cross_entropy(probs, targets, reduction='mean') (ocean_model.py:249-253; trainer.py:79).
 * *loss is ACCUMULATED; mean over mean_n elements (0 -> n). */
int ocean_bce_fwd(const float* probs, const float* targets, long n, long mean_n, float* loss, void* stream);
int ocean_bce_bwd(const float* probs, const float* targets, const float* dloss /* device scalar or NULL */,
                  long n, long mean_n, float* d_probs, void* stream);

/* F.binary_cross_entropy(probs, targets, reduction='none') (ocean_model.py:249-253 with compute_loss(reduction='none')):
 * out[i] = -(y log p + (1-y) log(1-p)), log terms clamped at -100; d_probs[i] = d_out[i] * dBCE/dp. */
int ocean_bce_elem_fwd(const float* probs, const float* targets, long n, float* out, void* stream);
int ocean_bce_elem_bwd(const float* probs, const float* targets, const float* d_out, long n, float* d_probs, void* stream);

/* Extra terms of the Trainer's LossFunction (ocean/models/training/trainer.py:90-110):
 * nn.MSELoss between two [n] tensors (gradient to `a` only), and the feature-diversity loss
 * mean_{i != j} | normalize(x)_i . normalize(x)_j | over x [B,E].  *loss is ACCUMULATED. */
int ocean_mse_fwd(const float* a, const float* b, long n, float* loss, void* stream);
int ocean_mse_bwd(const float* a, const float* b, const float* dloss, long n, float* da, void* stream);
size_t ocean_diversity_workspace(int B, int E);
int ocean_diversity_fwd(const float* x, int B, int E, float* ws, size_t ws_floats, float* loss, void* stream);
int ocean_diversity_bwd(int B, int E, float* ws /* from fwd */, const float* dloss, float* dx, void* stream);

/* ------------------------------------------------------------------ clip + Adam
 * Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step as used at
 * ocean/models/training/online_learner.py:325-335, 402-412 (Adam constructed
 * at :256-261: coupled L2 weight decay, no amsgrad).  `table` is a DEVICE array
 * with one row per chunk of consecutive elements of one tensor. */
typedef struct {
  uint64_t param, grad, exp_avg, exp_avg_sq;   /* device addresses of fp32 data */
  int64_t n;                                    /* elements in this chunk        */
} ocean_adam_chunk;
/* *sqnorm (device fp64, pre-zeroed) += sum of squares of every gradient in the table */
int ocean_grad_sqnorm(const ocean_adam_chunk* table, int n_chunks, double* sqnorm, void* stream);
/* g *= min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) when max_norm > 0 and sqnorm != NULL; then the Adam update
 * with bias corrections for `step` (1-based).  write_clipped: store the clipped gradient back. */
int ocean_clip_adam_step(const ocean_adam_chunk* table, int n_chunks, const double* sqnorm, float max_norm,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         int write_clipped, void* stream);
/* Same update with the learning rate (float[1]) and the 1-based step count (int[1]) read from device memory, so
   that the launch can sit in a captured HIP graph and still follow the host's learning-rate policy and the bias
   correction of torch.optim.Adam (online_learner.py:256-261): the caller rewrites lr_dev / increments step_dev
   between replays.  No counterpart in the reference (it launches every kernel eagerly). */
int ocean_clip_adam_step_dev(const ocean_adam_chunk* table, int n_chunks, const double* sqnorm, float max_norm,
                             const float* lr_dev, const int* step_dev, float beta1, float beta2, float eps,
                             float weight_decay, int write_clipped, void* stream);
/* The whole clip + Adam step (online_learner.py:325-335: clip_grad_norm_ then optimizer.step) as two launches that need
   no preparation between steps: the squared norm is accumulated into state->sq_acc (zero on entry), the update runs
   for step *step_dev + 1, and the block of the update kernel that finishes last publishes the squared norm in
   state->sq_last, puts sq_acc back to zero and stores the advanced count in *step_dev.  `state` is 24 zero-initialised
   bytes of device memory that belong to this optimizer.  max_norm <= 0: no clipping, no norm launch. */
typedef struct {
  double sq_acc;      /* sum of squared gradients of the step under way; zero between steps */
  double sq_last;     /* the finished step's sum (its square root is clip_grad_norm_'s return value) */
  unsigned done;      /* blocks of the update kernel that have finished; zero between steps */
  unsigned pad_;
} ocean_adam_tail;
int ocean_clip_adam_step_auto(const ocean_adam_chunk* table, int n_chunks, ocean_adam_tail* state, float max_norm,
                              const float* lr_dev, int* step_dev, float beta1, float beta2, float eps,
                              float weight_decay, int write_clipped, void* stream);

/* ------------------------------------------------------------------ kernel timing (measurement aid)
 * No counterpart in the reference.  While enabled, every GEMM launch is bracketed by HIP events on its
 * stream (this is the one place the library creates runtime objects).  ocean_prof_summary waits for the
 * recorded events and writes a JSON array, one object per kernel instantiation:
 * {"kernel", "launches", "total_ms", "flops", "bytes"} (algorithmic flops / bytes summed over launches). */
int ocean_prof_enable(int on);
int ocean_prof_summary(char* buf, size_t cap);

/* Every entry point reports a failed launch by reading the runtime's last-error slot, which also holds errors left by
 * the caller's own HIP calls (a stream capture that was invalidated, for one).  A caller that has handled such an error
 * calls this before its next library call; returns the cleared HIP error code (0: none).  No counterpart in the
 * reference. */
int ocean_clear_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_HIP_H */
