/* lib3dinfomax_hip.so - C ABI of the MI355X (gfx950) kernels behind the 3DInfomax pre-training hot path.
 *
 * The reference (HannesStark/3DInfomax) is pure Python: its "FFI" for this path is the set of ATen/DGL
 * calls its nn.Modules make.  Every entry point below replaces one such call site (cited as
 * reference file:line) and is what a maintainer binds with ctypes (the in-tree
 * binding is 3dinfomax_amd/_lib.py + ops.py).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless the parameter comment says
 *    "host"; tensors are dense row-major fp32 unless stated; index arrays are int32.
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is enqueued
 *    asynchronously on it, nothing synchronises the device, no hidden allocations except where a
 *    `workspace` pointer is taken explicitly.
 *  - return value: I3D_OK (0) or a negative I3D_ERR_*; i3d_last_error() returns a thread-local message.
 *  - edge-sized tensors are in DESTINATION-SORTED ("epos") order: the in-edges of node v are the
 *    contiguous rows [in_ptr[v], in_ptr[v+1]) (stable w.r.t. edge id = DGL's mailbox order).
 *  - thread safety: the error string is thread-local.  PROCESS-WIDE state exists and is set through explicit setters only:
 *    the fp32 product form (i3d_set_fp32_products), the one-launch BatchNorm backward (i3d_set_bn_bwd_one_launch), the
 *    collective table of synchronised BatchNorm (i3d_set_collectives / the peer exchange) and the per-stream side-stream
 *    table; call the setters before the first compute call of the other threads.
 */
#ifndef INFOMAX3D_HIP_H
#define INFOMAX3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I3D_OK 0
#define I3D_ERR_INVALID (-1)
#define I3D_ERR_LAUNCH (-2)

/* activations: reference models/base_layers.py:9-20 (get_activation) */
#define I3D_ACT_NONE 0
#define I3D_ACT_RELU 1
#define I3D_ACT_SILU 2
#define I3D_ACT_SIGMOID 3
#define I3D_ACT_LEAKY_RELU 4 /* negative slope 0.01 = nn.LeakyReLU() default, reference models/pna_original.py:291 */
/* the rest of the reference's SUPPORTED_ACTIVATION_MAP (models/base_layers.py:5) with torch's default parameters; GLU (halves the
 * feature dimension) is not an elementwise activation and is not offered */
#define I3D_ACT_TANH 5
#define I3D_ACT_ELU 6      /* alpha 1 */
#define I3D_ACT_SELU 7
#define I3D_ACT_SOFTPLUS 8 /* beta 1, threshold 20 */

/* aggregators: reference models/pna.py:71-81 (PNA_AGGREGATORS); readout ops: dgl.readout_nodes */
#define I3D_AGG_MEAN 0
#define I3D_AGG_SUM 1
#define I3D_AGG_MAX 2
#define I3D_AGG_MIN 3
#define I3D_AGG_STD 4
#define I3D_AGG_VAR 5

/* scalers: reference models/pna.py:83-87 (PNA_SCALERS) */
#define I3D_SCALE_IDENTITY 0
#define I3D_SCALE_AMPLIFICATION 1
#define I3D_SCALE_ATTENUATION 2

int i3d_abi_version(void);
const char* i3d_last_error(void);

/* ---- K1: fused multi-table embedding sum -----------------------------------------------------------
 * replaces AtomEncoder.forward / BondEncoder.forward, reference commons/mol_encoder.py:34-42, 65-73
 * (n_cols nn.Embedding lookups + adds).  out[r,:] = sum_k tables[k][idx[r,k], :].
 *   idx     int64 [rows, n_cols]          tables  host array of n_cols device pointers, table k is [dim_k, feat]
 *   row_perm int32 [rows] or NULL: output row r uses index row row_perm[r] (bond features are stored in edge-id
 *            order, the kernels want destination-sorted edges: the permutation is folded into the lookup)
 * bwd: grad_tables[k][idx[r,k], :] += grad_out[r,:]  (caller zero-fills; dims = host array of table sizes,
 * sum(dims)*feat*4 bytes must fit the 160 KiB LDS of a CU for the LDS-privatised path, else global atomics). */
int i3d_embedding_sum_fwd(const int64_t* idx, const int* row_perm, int rows, int n_cols, const float* const* tables,
                          int feat, float* out, void* stream);
int i3d_embedding_sum_bwd(const int64_t* idx, const int* row_perm, int rows, int n_cols, const float* grad_out,
                          int feat, float* const* grad_tables, const int* dims, void* stream);

/* ---- K4: PNA aggregation (the HBM-roofline kernel) -------------------------------------------------
 * replaces DGL update_all(message_func, reduce_func) + aggregators + scalers,
 * reference models/pna.py:206, 215-235, 17-37, 57-68.
 *   e [E, feat] messages in epos order;  in_ptr [N+1];  aggregators/scalers: host int arrays (I3D_AGG_*, I3D_SCALE_*)
 *   out [N, n_scalers_eff * n_aggregators * feat], scaler-major, zero rows for in-degree 0;
 *   n_scalers_eff = n_scalers if n_scalers > 1 else 1 with NO scaling (reference quirk, models/pna.py:232).
 *   force_scalers != 0: apply the scalers also when only one is configured (reference models/pna_original.py:235,
 *   422 - the original PNA layers have no such quirk).
 *   avg_d_log: the reference hard-codes 1.0 (models/pna.py:153); pna_original.py passes the real value.
 * bwd: grad_e [E, feat] (fully overwritten).  max/min ties route to the first (lowest edge id) slot. */
int i3d_pna_aggregate_fwd(const float* e, const int* in_ptr, int num_nodes, int feat, const int* aggregators,
                          int n_aggregators, const int* scalers, int n_scalers, int force_scalers, float avg_d_log,
                          float* out, void* stream);
int i3d_pna_aggregate_bwd(const float* grad_out, const float* e, const int* in_ptr, int num_nodes, int feat,
                          const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                          int force_scalers, float avg_d_log, float* grad_e, void* stream);

/* ---- K6: per-graph readout -------------------------------------------------------------------------
 * replaces dgl.readout_nodes(graph,'feat',op) for op in readout_aggregators + torch.cat,
 * reference models/pna.py:133-134, models/net3d.py:73-74.
 *   x [N, feat]; graph_ptr [B+1] node offsets; ops host array of I3D_AGG_{MEAN,SUM,MAX,MIN}
 *   out [B, n_ops*feat].  bwd: grad_x [N, feat] (fully overwritten), ties -> first node. */
int i3d_segment_readout_fwd(const float* x, const int* graph_ptr, int num_graphs, int feat, const int* ops,
                            int n_ops, float* out, void* stream);
int i3d_segment_readout_bwd(const float* grad_out, const float* x, const int* graph_ptr, int num_graphs, int feat,
                            const int* ops, int n_ops, float* grad_x, void* stream);

/* ---- dense towers: fp32 MFMA GEMM ------------------------------------------------------------------
 * replaces nn.Linear forward/backward inside FCLayer, reference models/base_layers.py:101 (aten::addmm/mm).
 *   C[M,N] (ldc) = (accumulate ? C : 0) + opA(A)[M,K] * opB(B)[K,N] + (bias ? bias[N] : 0)
 *   trans_a = 0: A stored [M,K] (lda >= K);  1: A stored [K,M] (lda >= M)
 *   trans_b = 0: B stored [K,N] (ldb >= N);  1: B stored [N,K] (ldb >= K)
 * exact fp32 (v_mfma_f32_16x16x4_f32), fp32 accumulate.
 *   Linear fwd  Y = X W^T + b : trans_a=0, trans_b=1, B=W[N,K]
 *   dX = dY W               : trans_a=0, trans_b=0, B=W
 *   dW = dY^T X             : trans_a=1, trans_b=0, A=dY[M',N'] (K := rows) */
int i3d_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int accumulate, void* stream);
/* tuning entry: same, with the tile configuration BMxBNxBK forced (0: 128x128x16, 1: 256x32x16, 2: 64x64x16, 3: 32x64x32 on
 * v_mfma_f32_16x16x4_f32; 4: 64x64x16, 5: 128x128x16 on v_mfma_f32_32x32x2_f32; -1 = auto) and the split-K factor
 * (0 = auto) - used by tools/gemm_bench.py to pick the dispatch heuristics */
int i3d_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int accumulate, int tile_cfg, int splits, void* workspace,
                    long workspace_bytes, void* stream);
/* Precision of every GEMM of the library (process-level; the reference's switch is the trainer's `dtype`, configs[3] of
 * BASELINE.json): 0 (default) = exact fp32 products on the fp32 matrix pipe; 1 = the operands are rounded to bf16
 * (round-to-nearest-even) as a lane reads its fragments and multiplied with v_mfma_f32_*_bf16, accumulation / bias /
 * epilogue / BatchNorm statistics / tensors in memory stay fp32 (the master weights are the fp32 parameters). */
int i3d_set_matmul_precision(int bf16);
int i3d_get_matmul_precision(void);
/* fp32 mode only - how the tiled forward / data-gradient GEMMs (32x32 MFMA tiles, 16-byte aligned operands) form the product of
 * two fp32 operands: 0 = v_mfma_f32_32x32x2_f32; 1 = both operands split exactly into three bf16 parts (x = hi + mid + lo) and
 * the six part products of order <= 2 taken on the bf16 matrix pipe with fp32 accumulation (each part product is exact; what is
 * dropped is <= 3 x 2^-24 |a b|, the size of one fp32 rounding) - the same nn.Linear arithmetic class as
 * /root/reference/models/base_layers.py:101, 2.7x fewer matrix-pipe cycles.  Process-level; I3D_FP32_PRODUCTS=native|split
 * sets it at load.  THE LIBRARY DEFAULT IS 1 (split) - every "fp32" result of this library (tests, bench.py's headline) is formed
 * this way unless 0 is set; the weight-gradient panels (wgrad.hip) and the small-tile / unaligned kernels always use the fp32
 * pipe.  Non-finite and huge operands: the split is exact for |x| <= 3.39e38 (above it the bf16 rounding of `hi` overflows); an
 * operand that is +-Inf or beyond that bound gives hi = +-Inf and a NaN remainder, so the outputs that depend on it are NaN where
 * the fp32 pipe gives +-Inf (or NaN): they are non-finite in exactly the positions where the native product is non-finite, but
 * isinf() does not tell overflow from invalid any more (tests/test_gpu_ops.py: test_split_products_with_non_finite_operands). */
int i3d_set_fp32_products(int split);
int i3d_get_fp32_products(void);

/* Row-panel form of the chain's Linear products (csrc/panel.hip; reference models/base_layers.py:101 and its data gradient): the
 * weight is split ONCE per optimisation step into the three bf16 images of the split form (i3d_set_fp32_products) and laid out as the
 * LDS image of the product kernel - i3d_panel_pack: trans 1 = B[n][k] = W[n * ldw + k] (forward, W stored [out, in]), 0 = B[n][k] =
 * W[k * ldw + n] (data gradient); `packed`: i3d_panel_packed_bytes(N, K) bytes, 16-byte aligned - and i3d_panel_gemm forms
 * C[M, N] (+)= A[M, K] B^T (+ bias) with every 64-row slab of A read and split once per 208-column block.  K % 8 == 0, N % 4 == 0,
 * 16-byte aligned rows.  Same arithmetic as the split form of i3d_gemm_f32 (six bf16 part products, fp32 accumulation). */
long i3d_panel_packed_bytes(int N, int K);
int i3d_panel_pack(const float* W, int ldw, int N, int K, int trans, void* packed, void* stream);
typedef struct {
    const float* W;
    int ldw, N, K, trans;
    void* packed;
} I3dPanelPack;
int i3d_panel_pack_multi(const I3dPanelPack* weights, int n, void* stream); /* up to 8 weights (a layer's) in ONE launch */
int i3d_panel_gemm(int M, int N, int K, const float* A, int lda, const void* packed, float* C, int ldc, const float* bias,
                   int accumulate, void* stream);
/* the fused forward of a block behind a never-materialised BatchNorm (i3d_gemm_f32_fused in row-panel form): A read as
 * (A - mean) * scale + shift (a_aff [3 K], may be NULL), the stored value = act(product + bias), and - stats != NULL - per 32-row
 * tile and column {sum, M2 about the tile mean, row count} of the stored values: stats [i3d_panel_stats_tiles(M)][3][N] */
int i3d_panel_stats_tiles(int M);
int i3d_panel_gemm_fused(int M, int N, int K, const float* A, int lda, const void* packed, float* C, int ldc, const float* bias,
                         const float* a_aff, int epi_act, float* stats, void* stream);

/* i3d_gemm_f32 with scratch: when the reduction dimension is split over workgroups (weight gradients), the slices are
 * written to workspace[slices][M][N] and summed in a fixed order by a second kernel (deterministic, no zero-fill, no
 * atomics).  workspace NULL or too small: fp32 atomics as i3d_gemm_f32. */
int i3d_gemm_f32_ws(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int accumulate, void* workspace, long workspace_bytes,
                    void* stream);

/* i3d_gemm_f32_ws where B and/or C consist of two blocks of one parent matrix (the [W_s | W_d] column blocks of an
 * edge-MLP weight, P trick of i3d_edge_combine_fwd: P = h [W_s | W_d]^T, dh = dP [W_s; W_d], d[W_s | W_d] = dP^T h as ONE
 * GEMM each instead of two):  B index (n if trans_b, k otherwise) >= b_split adds b_delta floats to the address and
 * b_view_floats is the number of floats addressable behind B; C rows >= c_split add c_delta floats.  b_split / c_split
 * <= 0: plain operand. */
int i3d_gemm_f32_blocks(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        int b_split, long b_delta, long b_view_floats, float* C, int ldc, int c_split, long c_delta,
                        int accumulate, void* workspace, long workspace_bytes, void* stream);

/* K4 with a tower-major row layout (the tower variant's stacked layers, csrc/tower.hip): the feat message columns are feat /
 * tower_feat towers; a node's output row is [tower][block (scaler, aggregator)][feature of the tower] instead of
 * [block][feature] - the B blocks of one tower are one contiguous column range.  feat % 4 == tower_feat % 4 == 0; otherwise as
 * i3d_pna_aggregate_fwd / _bwd (grad_out in the same layout). */
int i3d_pna_aggregate_fwd_towers(const float* e, const int* in_ptr, int num_nodes, int feat, int tower_feat,
                                 const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                                 int force_scalers, float avg_d_log, float* out, void* stream);
int i3d_pna_aggregate_bwd_towers(const float* grad_out, const float* e, const int* in_ptr, int num_nodes, int feat,
                                 int tower_feat, const int* aggregators, int n_aggregators, const int* scalers,
                                 int n_scalers, int force_scalers, float avg_d_log, float* grad_e, void* stream);
/* n_batch products of ONE shape in one launch: batch b multiplies A + b a_batch by B + b b_batch into C + b c_batch (strides in
 * floats; layouts and `accumulate` as i3d_gemm_f32, no bias; K-slices of weight-gradient layouts go through `workspace` as in
 * i3d_gemm_f32_ws and are required to: n_batch <= 32).  The diagonal blocks of a block-diagonal product: the posttrans Linear of
 * the `towers` PNATowers of a layer (reference models/pna_original.py:209-211, 250: tower t reads only ITS aggregated columns),
 * forward, data gradient and weight gradient, without the zero blocks of the stacked weight (csrc/tower.hip). */
int i3d_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, long a_batch, const float* B,
                         int ldb, long b_batch, float* C, int ldc, long c_batch, int n_batch, int accumulate, void* workspace,
                         long workspace_bytes, void* stream);

/* i3d_gemm_f32_grouped (below) whose per-group weight is block-diagonal: batch b multiplies the columns A + b a_batch of the
 * group's rows by B_g + b b_batch into the columns C + b c_batch - the towers' blocks of the per-degree posttrans weights of the
 * tower variant (csrc/tower.hip with I3dTowerLayerArgs.n_towers > 1 and n_deg_groups > 0) */
int i3d_gemm_f32_grouped_batched(int trans_b, int m_padded, int N, int K, const float* A, int lda, long a_batch, long a_rows_total,
                                 const int* m_rows, const int* tile_group, const float* B, int ldb, long b_group_stride, long b_batch,
                                 float* C, int ldc, long c_batch, int n_batch, int accumulate, void* stream);

/* ---- degree-grouped posttrans of the PNA layer ------------------------------------------------------
 * replaces cat([h, agg]) -> posttrans Linear of reference models/pna.py:207-209 for the aggregated part:  the three
 * scaler blocks of agg are per-node multiples (functions of the in-degree D only) of the same aggregator block a,
 * so  [a | amp(D) a | att(D) a] W_agg^T = a W_D^T  with  W_D = sum_s c_s(D) W_s  (re-association only): K drops
 * from n_scalers*A to A, and the aggregation kernel only has to write the identity block.
 *  combine_weights_fwd: WD[g, n, k] = sum_s coef[g*n_scalers+s] * W[n, f_in + s*agg_width + k]   (coef: host array)
 *  combine_weights_bwd: dW[n, f_in + s*agg_width + k] = sum_g coef[g*n_scalers+s] * dWD[g, n, k]
 *  gemm_f32_grouped:    C[r, :] (+)= A[r, :] * op(B_g) for r = m_rows[m] >= 0, g = tile_group[m / 64]; m_rows lists
 *                       the nodes grouped by in-degree, every group padded with -1 to a multiple of 64 rows;
 *                       trans_b = 1: B_g stored [N, K] (forward, B_g = WD[g]);  0: B_g stored [K, N] (data gradient)
 *  gemm_f32_rowsubset:  C[M, N] = sum_j A[k_rows[j], 0:M]^T B[k_rows[j], 0:N]   (weight gradient of one group)
 *  gemm_f32_rowsubset_multi: the same for n_groups disjoint ranges [group_start[g], +group_count[g]) of k_rows in ONE
 *                       launch, C_g = C + g * c_group_stride (host arrays for the ranges; tile_cfg -1 / seg_rows 0 =
 *                       automatic; the row segments are combined through `workspace` like i3d_gemm_f32_ws, or with
 *                       fp32 atomics when it is NULL / too small) */
int i3d_pna_combine_weights_fwd(const float* W, int ldw, int f_in, int f_out, int agg_width, int n_groups,
                                int n_scalers, const float* coef, float* WD, void* stream);
int i3d_pna_combine_weights_bwd(const float* dWD, int ldw, int f_in, int f_out, int agg_width, int n_groups,
                                int n_scalers, const float* coef, float* dW, void* stream);
int i3d_gemm_f32_grouped(int trans_b, int m_padded, int N, int K, const float* A, int lda, long a_rows_total,
                         const int* m_rows, const int* tile_group, const float* B, int ldb, long b_group_stride,
                         float* C, int ldc, int accumulate, void* stream);
int i3d_gemm_f32_rowsubset(int M, int N, int n_rows, const float* A, int lda, const float* B, int ldb,
                           const int* k_rows, long rows_total, float* C, int ldc, int accumulate, void* stream);
int i3d_gemm_f32_rowsubset_multi(int M, int N, int n_groups, const int* group_start, const int* group_count,
                                 const float* A, int lda, const float* B, int ldb, const int* k_rows, long rows_total,
                                 float* C, long c_group_stride, int ldc, int accumulate, int tile_cfg, int seg_rows,
                                 void* workspace, long workspace_bytes, void* stream);

/* ---- column statistics / BatchNorm1d ---------------------------------------------------------------
 * replaces nn.BatchNorm1d in FCLayer (train: batch statistics, momentum m, unbiased running_var; eval: running
 * statistics), reference models/base_layers.py:87, 106-110, plus the activation in front of it (:102-103).
 *
 * i3d_act_stats_fwd: x = act(pre) written to `x` (may alias pre; pass act NONE and x == pre for stats only),
 *   then mean[feat], invstd[feat] = 1/sqrt(biased var + eps); if running_mean != NULL the running statistics are
 *   updated in place (running = (1-m)*running + m*batch, var unbiased).  `count_extra`/`stats_extra` support
 *   synchronised BN: when sums_out != NULL the kernel only writes the fp64 sums_out[2*feat+1] = {sum x,
 *   sum x^2, rows} and does NOT finalise (the caller all-reduces them and calls i3d_bn_finalize_stats).
 *   workspace: device scratch of i3d_colreduce_workspace_bytes(rows, feat) bytes, ZERO-INITIALISED once by the caller
 *   and used by one stream at a time: the column reductions (this function, i3d_bn_bwd, i3d_bn_eval_bwd, i3d_colsum)
 *   are two-stage and deterministic.  With I3D_FUSED_FINAL=1 in the environment the second stage runs inside the first
 *   stage's launch (the last workgroups to arrive reduce the partial rows), synchronised through two counters at the
 *   head of the workspace that every launch leaves at zero; by default it is a separate small launch. */
long i3d_colreduce_workspace_bytes(int rows, int feat);
int i3d_act_stats_fwd(const float* pre, int rows, int feat, int act, float* x, float eps, float momentum,
                      float* mean, float* invstd, float* running_mean, float* running_var, double* sums_out,
                      void* workspace, void* stream);
/* as i3d_act_stats_fwd; additionally increments *num_batches_tracked (BatchNorm1d's int64 counter, may be NULL) in the
 * kernel that updates the running statistics - reference models/base_layers.py:107-108 runs nn.BatchNorm1d in training
 * mode, which bumps it once per forward */
int i3d_act_stats_fwd_counted(const float* pre, int rows, int feat, int act, float* x, float eps, float momentum,
                      float* mean, float* invstd, float* running_mean, float* running_var, double* sums_out, long long* num_batches_tracked,
                      void* workspace, void* stream);
/* fp64 sums [2*feat+1] = {sum x, sum x^2, count} (already all-reduced) -> mean, invstd, running stats update */
int i3d_bn_finalize_stats(const double* sums, int feat, float eps, float momentum, float* mean, float* invstd,
                          float* running_mean, float* running_var, void* stream);
/* y = post_act( (x-mean)*invstd*gamma + beta ) + (residual ? residual : 0).   y may alias x. */
int i3d_bn_apply_fwd(const float* x, int rows, int feat, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int post_act, const float* residual, float* y,
                     void* stream);
/* eval mode: invstd computed from running_var on the fly */
int i3d_bn_eval_fwd(const float* x, int rows, int feat, const float* running_mean, const float* running_var,
                    float eps, const float* gamma, const float* beta, int post_act, const float* residual, float* y,
                    void* stream);
/* backward of  y = post_act(BN(x)), x = act(pre)   (train mode):
 *   grad_y, x [rows,feat]; pre may be NULL when act is NONE or RELU (x itself decides relu');
 *   outputs: grad_gamma[feat], grad_beta[feat], grad_pre [rows,feat] (may alias grad_y).
 *   If sums_out != NULL: phase 1 only - writes the LOCAL grad_gamma/grad_beta and the fp64 sums_out[2*feat] =
 *   {sum dy, sum dy*xhat} for the sync-BN all-reduce (the caller stores its row count in sums[2*feat] before
 *   reducing); then call again with sums_in != NULL (phase 2: the reduced sums and the count sums_in[2*feat]
 *   drive grad_pre - total_rows is ignored, nothing is read back to the host; grad_gamma/grad_beta untouched).   grad_bias (may be NULL): column sums of grad_pre - the bias gradient of the Linear in
 * front of the block - from the same pass that writes grad_pre. */
int i3d_bn_bwd(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act, int post_act,
               const float* mean, const float* invstd, const float* gamma, const float* beta, float* grad_gamma,
               float* grad_beta, float* grad_pre, float* grad_bias, double* sums_out, const double* sums_in,
               long total_rows, void* workspace, void* stream);
/* eval-mode backward (statistics are constants): grad_pre = grad_y * post_act' * gamma*invstd * act' */
int i3d_bn_eval_bwd(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                    int post_act, const float* running_mean, const float* running_var, float eps,
                    const float* gamma, const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre,
                    void* workspace, void* stream);
/* out[feat] = sum over rows of x[r,:] * (w ? w[r] : 1)   (bias gradients, weighted column sums) */
int i3d_colsum(const float* x, const float* w, int rows, int feat, float* out, void* workspace, void* stream);
/* y = act(x) elementwise (n elements); bwd: grad_x = grad_y * act'(x) */
int i3d_act_fwd(const float* x, long n, int act, float* y, void* stream);
int i3d_act_bwd(const float* grad_y, const float* x, long n, int act, float* grad_x, void* stream);
/* The gates of ONE GRU step (csrc/gru.hip) - reference models/pna_original.py:64-84 (`GRU`: nn.GRU run for one step between the layers
 * of PNAGNNOriginal with gru_enable=True, :190-193).  GI = x W_ih^T + b_ih, GH = h0 W_hh^T + b_hh [rows, 3 hidden] (gate order r | z | n,
 * torch's); forward: out = (1 - z) n + z h0, saved [rows, 3 hidden] = r | z | n; backward: gradients w.r.t. GI, GH and the direct
 * share of h0 (the caller adds grad_GH W_hh). */
int i3d_gru_gates_fwd(const float* GI, const float* GH, const float* h0, int rows, int hidden, float* out, float* saved, void* stream);
int i3d_gru_gates_bwd(const float* grad_out, const float* saved, const float* GH, const float* h0, int rows, int hidden, float* grad_GI,
                      float* grad_GH, float* grad_h0, void* stream);
/* dst += src;  out = a + b;  out[r, :] = row for r < rows (Net3D's broadcast node embedding, models/net3d.py:61) */
int i3d_add_inplace(float* dst, const float* src, long n, void* stream);
int i3d_add(const float* a, const float* b, long n, float* out, void* stream);
/* out = a * b elementwise (out may alias a): nn.Dropout of FCLayer / the tower variants with the scaled mask as b (reference
 * models/base_layers.py:84-85, 104-105; models/pna_original.py:260, 428) */
int i3d_mul(const float* a, const float* b, long n, float* out, void* stream);
int i3d_broadcast_row(const float* row, long rows, int feat, float* out, void* stream);

/* ---- edge kernels ----------------------------------------------------------------------------------
 * i3d_edge_combine_fwd replaces the gather + concat + first Linear of the edge MLPs,
 *   reference models/pna.py:237-252 (pretrans_edges: cat[h_src, h_dst, e_feat] -> Linear) and
 *   models/net3d.py:113-115, using  [h_s|h_d|q] W^T = h_s W_s^T + h_d W_d^T + q W_q^T :
 *   pre[j,:] = P[src_s[j], 0:feat] + P[dst_s[j], feat:2feat] + Q[q_code ? q_code[j] : j,:] + bias
 *   (P is [N, ldp >= 2*feat]; Q, q_code, bias may be NULL)
 * i3d_edge_codes: categorical edge features with a small joint vocabulary V (bonds, commons/mol_encoder.py:45-73:
 *   5 x 6 x 2 = 60): codes[j] = sum_c idx[row_perm ? row_perm[j] : j, c] * strides[c] (host array), and the one-hot
 *   matrix onehot[rows, v_pad] (v_pad >= V, multiple of 4).  The bond embedding of an edge is then row codes[j] of the
 *   [V, F] table of all combinations, so the per-layer  e_feat W_q^T  is a gather from (table W_q^T) (q_code above)
 *   and its gradients are  onehot^T dpre  ([V, F] instead of two [E, F] x [F, F] products).
 * i3d_segment_sum: out[v,:] = scale(v) * sum_{j in [ptr[v],ptr[v+1])} x[idx ? idx[j] : j, :]
 *   scale_mode 0: 1;  1: 1/max(count,1)  (DGL fn.mean, reference models/net3d.py:95-96)
 *   used for the backward of the gathers (by in_ptr, and by out_ptr/out_epos) and for Net3D's mean reduce.
 * i3d_segment_bcast: grad of segment mean/sum: out[j,:] = scale(seg(j)) * g[seg(j),:], seg given by dst_s. */
int i3d_edge_combine_fwd(const float* P, int ldp, const float* Q, const int* q_code, const float* bias,
                         const int* src_s, const int* dst_s, int num_edges, int feat, float* pre, void* stream);
/* multi-hot encoding of categorical columns: out[j, offsets[c] + idx[row_perm ? row_perm[j] : j, c]] = 1, else 0
 * (out [rows, v_pad], offsets: host prefix sums of the table sizes).  out^T dY is the gradient of all embedding tables of
 * an encoder (commons/mol_encoder.py:34-42) as one weight-gradient GEMM - deterministic, unlike the atomics of
 * i3d_embedding_sum_bwd. */
int i3d_multihot(const int64_t* idx, const int* row_perm, int rows, int n_cols, const int* offsets, int v_pad, float* out,
                 void* stream);
int i3d_edge_codes(const int64_t* idx, const int* row_perm, int rows, int n_cols, const int* strides, int v_pad,
                   int* codes, float* onehot, void* stream);
int i3d_segment_sum(const float* x, int ldx, const int* ptr, const int* idx, int num_segments, int feat,
                    int scale_mode, float* out, int ldo, void* stream);
/* two segmentations of the same rows in one launch (sums, no scaling): out0 over (ptr0, idx0), out1 over (ptr1, idx1) */
/* i3d_segment_sum over rows of bf16 values (x: row r at (bf16*)x + r * ldx; feat, ldx multiples of 4, 8-byte aligned), fp32 sums */
int i3d_segment_sum_bf16(const void* x, int ldx, const int* ptr, const int* idx, int num_segments, int feat, int scale_mode,
                         float* out, int ldo, void* stream);
int i3d_segment_sum_pair(const float* x, int ldx, const int* ptr0, const int* idx0, float* out0, const int* ptr1,
                         const int* idx1, float* out1, int num_segments, int feat, int ldo, void* stream);
int i3d_segment_bcast(const float* g, const int* ptr, const int* seg_of_row, int rows, int feat, int scale_mode,
                      float* out, void* stream);
/* rows gather: out[j,:] = x[idx[j],:]  (edge permutation edge-id -> epos order) */
int i3d_gather_rows(const float* x, const int* idx, int rows, int feat, float* out, void* stream);

/* ---- Net3D specifics -------------------------------------------------------------------------------
 * fourier: replaces fourier_encode_dist, reference commons/utils.py:103-110, models/net3d.py:63-64:
 *   out[j,:] = [sin(d/2^k)]_{k<n} | [cos(d/2^k)]_{k<n} | d        out [E, 2n+1]
 * soft edge gate: reference models/net3d.py:106, 117-118:  w = sigmoid(m . ws + bs);  msg = m * w
 *   bwd: grad_m [E,feat], g_gate[E] = (sum_f grad_msg*m) * w(1-w)  (so that grad_ws = sum_e g_gate*m, grad_bs = sum g_gate) */
int i3d_fourier_encode(const float* d, int num_edges, int n_enc, float* out, void* stream);
int i3d_soft_edge_fwd(const float* m, const float* ws, const float* bs, int num_edges, int feat, float* msg,
                      float* w, void* stream);
int i3d_soft_edge_bwd(const float* grad_msg, const float* m, const float* w, const float* ws, int num_edges,
                      int feat, float* grad_m, float* g_gate, void* stream);

/* ---- NT-Xent ---------------------------------------------------------------------------------------
 * replaces NTXent.forward / NTXentMultiplePositives.forward, reference commons/losses.py:143-155, 225-247.
 *   z1 [b1, dim] (local 2D-view rows), z2 [b2*conf, dim] (all 3D-view rows, conformer-minor); the positive columns
 *   of row i are (pos_offset+i)*conf .. +conf-1 (pos_offset = rank*b1 when z2 is the all-gathered batch).
 *   eps: 1e-8 for NTXent, 0 for the multiple-positives variant (reference :150 vs :239).
 *   sim  [b1, b2*conf] = z1 z2^T from i3d_gemm_f32;  dsim [b1, b2*conf] receives dL/dS
 *   fwd: loss_sum[0] = sum_i -log(pos_i / (rowsum_i - pos_i))   (caller divides by the GLOBAL batch size)
 *   bwd: writes dsim = dL/dS (scaled by grad_scale = upstream/global_batch) and the norm-path
 *        coefficients ca[b1], cb[b2*conf] such that dz1 = dS z2 + ca*z1, dz2 = dS^T z1 + cb*z2. */
int i3d_row_norms(const float* z, int rows, int dim, float* norms, void* stream);
/* loss_sum[0] = loss_scale * sum_i l_i (loss_scale = 1/global batch);  the backward multiplies by grad_scale and, when
 * grad_scale_dev != NULL, by the device scalar grad_scale_dev[0] (the upstream gradient of the loss: no host read-back) */
int i3d_ntxent_fwd(const float* sim, const float* n1, const float* n2, int b1, int b2, int conf, int pos_offset,
                   float tau, float eps, float loss_scale, float* row_sum, float* row_pos, float* loss_sum, void* stream);
int i3d_ntxent_bwd(const float* sim, const float* n1, const float* n2, const float* row_sum, const float* row_pos,
                   int b1, int b2, int conf, int pos_offset, float tau, float eps, float grad_scale,
                   const float* grad_scale_dev, float* dsim, float* ca, float* cb, void* stream);
/* the whole loss from one call per direction (row norms, similarity GEMM, i3d_ntxent_fwd / i3d_ntxent_bwd, the two
 * gradient GEMMs and their normalisation terms; reference commons/losses.py:143-155, 225-247).  scratch
 * (i3d_ntxent_loss_scratch_floats(b1, b2 * conf) floats) is written by the forward and read by the backward; work:
 * b1 * b2c + b1 + b2c (+ alignment) floats.  z1 [b1, dim] local rows, z2 [b2 * conf, dim] the (gathered) other view. */
long i3d_ntxent_loss_scratch_floats(int b1, int b2c);
int i3d_ntxent_loss_fwd(const float* z1, const float* z2, int b1, int b2, int conf, int dim, int pos_offset, float tau,
                        float eps, float loss_scale, float* scratch, float* loss, void* stream);
int i3d_ntxent_loss_bwd(const float* z1, const float* z2, int b1, int b2, int conf, int dim, int pos_offset, float tau,
                        float eps, float loss_scale, const float* scratch, const float* grad_scale_dev, float* work, float* dz1,
                        float* dz2, void* stream);
/* out[r,:] += coef[r] * z[r,:] */
int i3d_row_axpy(const float* z, const float* coef, int rows, int dim, float* out, void* stream);
/* out[r,:] = coef[r] * z[r,:]   (graph-size normalisation h * snorm_n, reference models/pna_original.py:258-259) */
int i3d_row_scale(const float* z, const float* coef, int rows, int dim, float* out, void* stream);
/* ---- the tower variant's stacked parameters (csrc/pack.hip; reference models/pna_original.py:264-319: `towers` PNATowers per
 * layer, outputs concatenated) - n_blocks 2-D copies src[rows, cols] (pitch ld_src) -> dst (pitch ld_dst) in ONE launch, the
 * table in device memory; reverse != 0: dst -> src (stacked gradients / running statistics back to the towers' tensors) */
typedef struct {
    const float* src;
    float* dst;
    int rows, cols;
    long ld_src, ld_dst;
} I3dCopyBlock;
int i3d_block_copy(const I3dCopyBlock* table /* device */, int n_blocks, int reverse, void* stream);
/* dst[rows, cols_dst] (contiguous) = the first cols_dst columns of src[rows, cols_src] (contiguous), zeros where cols_dst >
 * cols_src: the stacked towers run on activations whose width is rounded up to 4 floats (hidden_dim 90 of the reference's
 * configs/pna_original.yml -> 92: every kernel then takes its 16-byte form); the embedding's output is widened once, the
 * last layer's output cropped once, their gradients the other way round */
int i3d_copy_cols(const float* src, int rows, int cols_src, float* dst, int cols_dst, void* stream);
/* out[j * ld + col] = sum_k (x[src[j], k] - x[dst[j], k])^2 over the three coordinates of x [N, 3] - the squared distance that
 * PNALayer.pretrans_edges appends to an edge's features with pairwise_distances=True (reference models/pna.py:243-245);
 * src / dst: the edge list in the order of the rows of `out` (the kernels' destination-sorted order).  take_sqrt: the distance
 * itself, torch.norm(x_src - x_dst, dim=-1) - PNATower.pretrans_edges with use_3d=True (reference models/pna_original.py:224-226) */
int i3d_edge_sqdist(const float* x, const int* src, const int* dst, int num_edges, float* out, int ld, int col, int take_sqrt,
                    void* stream);

/* ---- one PNALayer of the tower variant from ONE call per direction (csrc/tower.hip) --------------------------------------
 * Replaces PNALayer.forward of reference models/pna_original.py:296-319 (all `towers` PNATower.forward, :239-261, the
 * concatenation, the mixing network, LeakyReLU and the residual) and its autograd backward, on the STACKED form of the
 * towers: Wp [f_msg, 2 f_in + f_edge] (pitch ldp) = every tower's pretrans Linear, Wq [f_out, f_in + B f_msg] (pitch ldq) =
 * every tower's posttrans Linear (B = n_aggregators * n_scalers column blocks of f_msg), bq / gamma / beta / running
 * statistics [f_out] side by side, Wm [f_mix, f_out] / bm the mixing network.  h [N, f_in], e [E, f_edge] (destination-
 * sorted, or NULL), snorm [N] or NULL (graph norm), out [N, f_mix]; residual needs f_mix == f_in.  gamma NULL: no BatchNorm.
 * saved / scratch: i3d_tower_layer_saved_floats / _scratch_floats floats (the forward's `saved` is the backward's).
 * workspace: i3d_colreduce_workspace_bytes, zero-initialised; gemm_workspace: the split-K scratch of the weight gradients.
 * Backward: grad_out [N, f_mix] -> grad_h [N, f_in] (written), grad_e [E, f_edge] (written, or added to when
 * grad_e_accumulate; NULL: not wanted), parameter gradients (grad_Wp / grad_Wq with pitches ldgp / ldgq). */
typedef struct {
    int num_nodes, num_edges, f_in, f_edge, f_msg, f_out, f_mix, ldp, ldq, ldgp, ldgq;
    int n_aggregators, n_scalers, residual, training, grad_e_accumulate;
    int aggregators[8], scalers[4];
    float avg_d_log, eps, momentum;
    const float* h;
    const float* e;
    const float* snorm;
    const float* Wp;
    const float* bp;
    const float* Wq;
    const float* bq;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    const float* Wm;
    const float* bm;
    const int* src_s;
    const int* dst_s;
    const int* in_ptr;
    const int* out_ptr;
    const int* out_epos;
    float* saved;
    float* scratch;
    void* workspace;
    void* gemm_workspace;
    long gemm_workspace_bytes;
    float* out;
    const float* grad_out;
    float* grad_h;
    float* grad_e;
    float* grad_Wp;
    float* grad_bp;
    float* grad_Wq;
    float* grad_bq;
    float* grad_gamma;
    float* grad_beta;
    float* grad_Wm;
    float* grad_bm;
    /* > 1: the posttrans products as n_towers diagonal blocks (i3d_gemm_f32_batched) - tower t's rows of Wq hold its f_msg / n_towers
     * x B aggregated columns at f_in + t B f_msg / n_towers .. (zeros elsewhere are never read) and the aggregation is written /
     * its gradient read tower-major (i3d_pna_aggregate_fwd_towers); needs f_msg / n_towers and f_out / n_towers multiples of 4.
     * 0 / 1: one dense product on [block][tower][feature] columns */
    int n_towers;
    /* > 0: the scalers folded into per-degree weights as in the 2D network's posttrans block (above:
     * "degree-grouped posttrans") - the aggregation writes its identity blocks only ([N, n_aggregators f_msg]), the product on it
     * is grouped by in-degree with W_D = sum_s coef[g][s] W_s: K of the dominant products n_scalers times shorter, the aggregated
     * tensor n_scalers times smaller.  group_start / group_count / coef / deg_rows / deg_tile_group / m_padded as in
     * I3dGroupedFcArgs (graph.py: GraphIndex.degree_groups).  With n_towers > 1 as well: the columns behind f_in are
     * [scaler][tower][aggregator][feature], the aggregation is written tower-major and the per-degree weights' diagonal blocks
     * are multiplied (i3d_gemm_f32_grouped_batched); the weight gradient stays one dense grouped product */
    int n_deg_groups, m_padded;
    int group_start[32];
    int group_count[32];
    float coef[128];
    const int* deg_rows;
    const int* deg_tile_group;
} I3dTowerLayerArgs;
long i3d_tower_layer_saved_floats(const I3dTowerLayerArgs* a);
long i3d_tower_layer_scratch_floats(const I3dTowerLayerArgs* a);
int i3d_tower_layer_fwd(const I3dTowerLayerArgs* a, void* stream);
int i3d_tower_layer_bwd(const I3dTowerLayerArgs* a, void* stream);

/* ---- BatchNorm out of the memory path (csrc/fused_bn.hip, csrc/gemm.hip; reference models/base_layers.py:100-111) ----
 * Column statistics as per-row-tile partials  partial[tile][3][feat] = {sum, M2 about the tile mean, row count},
 * produced by the kernel that writes the activation; i3d_bn_finalize_partials merges them (fp64, fixed order) into
 * mean / invstd / running statistics (momentum, unbiased running_var, num_batches_tracked += 1) and
 * aff[3 feat] = mean | gamma * invstd | beta, which the CONSUMER of the activation applies while loading it. */
int i3d_bn_finalize_partials(const float* partial, int n_tiles, int feat, float eps, float momentum, const float* gamma,
                             const float* beta, float* mean, float* invstd, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float* aff, void* stream);
/* x[j,:] = act(P[src_s[j], 0:feat] + P[dst_s[j], feat:2 feat] + Q[q_code ? q_code[j] : j, :] + bias) and the partials of x;
 * a tile is i3d_edge_stats_rows_per_tile(feat) consecutive rows (reference models/pna.py:237-252 + base_layers.py:102-108) */
int i3d_edge_stats_rows_per_tile(int feat);
int i3d_edge_combine_act_stats(const float* P, int ldp, const float* Q, const int* q_code, const float* bias,
                               const int* src_s, const int* dst_s, int num_edges, int feat, int act, float* x,
                               float* partial, void* stream);
/* C[M,N] = epi_act(A' W^T + bias (+ C)),  A' = (A - mean) * scale + shift per column k when a_aff ([3K]) is given (the
 * BatchNorm of the block in front applied while A is staged), W [N,K] row-major; stats ([ceil(M/64)][3][N]) receives the
 * partials of the stored values per 64-row tile.  Grouped form as i3d_gemm_f32_grouped (m_rows padded to 64 per group with
 * -1, M = padded count, a_rows_total = physical rows of A / C).  Needs 16-byte aligned operands, K % 4 == 0.  epi_act: none, ReLU
 * or LeakyReLU (the activations of the blocks this product sits in; any other runs as a pass of its own behind a plain product). */
int i3d_gemm_f32_fused(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb, float* C,
                       int ldc, const float* bias, int accumulate, const float* a_aff, int epi_act, float* stats,
                       const int* m_rows, const int* tile_group, long b_group_stride, void* stream);
/* i3d_gemm_f32_fused whose `accumulate` addend is read from c_in (row pitch ldcin) instead of C */
int i3d_gemm_f32_fused_src(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb, float* C,
                           int ldc, const float* c_in, int ldcin, const float* bias, int accumulate, const float* a_aff,
                           int epi_act, float* stats, const int* m_rows, const int* tile_group, long b_group_stride,
                           void* stream);
/* i3d_gemm_f32_fused (statistics variant) with the output stored as bf16 - row r at (bf16*)C + r * ldc, rounded where it is
 * stored, statistics from the fp32 values (the bf16 matmul mode's storage form of a PNA layer's messages) */
int i3d_gemm_f32_fused_bf16out(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb, void* C, int ldc,
                               const float* bias, const float* a_aff, int epi_act, float* stats, void* stream);
/* Wcat [2 f_out_edge + f_out_post, f_h] = [W_s ; W_d ; W_h], bcat = [0 | 0 | bias_post] (I3dPnaLayerArgs.merge_h) */
int i3d_pna_pack_h_weights(const float* W_edge, int ldw_edge, int f_out_edge, const float* W_post, int ldw_post, int f_out_post,
                           const float* bias_post, int f_h, float* Wcat, float* bcat, void* stream);
/* i3d_bn_bwd_deferred_bias (local statistics or the process-wide collectives) with grad_pre as a column block of a wider
 * matrix: row pitch ld_out floats */
int i3d_bn_bwd_strided(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act, int post_act,
                       const float* mean, const float* invstd, const float* gamma, const float* beta, float* grad_gamma,
                       float* grad_beta, float* grad_pre, int ld_out, float* grad_bias, void* workspace, float* bias_partial,
                       void* stream);
/* i3d_bn_bwd with the BatchNorm input x stored as bf16 (row r at (bf16*)x + r * feat) */
int i3d_bn_bwd_x_bf16(const float* grad_y, const void* x, int rows, int feat, int act, int post_act, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre,
                      float* grad_bias, void* workspace, float* bias_partial, void* stream);
/* dW[f_out,f_in] = dY^T y for y = (x - mean) * scale + shift (aff over f_in) computed from the raw x; grad_bias[f_out] =
 * column sums of dY (already computed) */
int i3d_gemm_f32_wgrad_bn(int f_out, int f_in, int rows, const float* dY, int ldy, const float* x, int ldx, float* dW,
                          int ldw, const float* grad_bias, const float* aff, void* workspace, long workspace_bytes,
                          void* stream);
/* ---- synchronised BatchNorm from inside the sequencers: process-wide collectives (csrc/comm.hip) ---------------------
 * The reference's BatchNorm statistics are over the whole batch (models/base_layers.py:87, 100-111); with the batch
 * sharded over ranks, i3d_bn_finalize_partials, i3d_act_stats_fwd[_counted] (sums_out NULL) and i3d_bn_bwd[_deferred_bias]
 * (sums_out / sums_in NULL) synchronise their statistics over the ranks WHILE a collective table is set: local merge ->
 * collective on the caller's stream -> finalisation over all ranks (running statistics = global, grad_gamma / grad_beta =
 * this rank's share: the gradient all-reduce sums them).  world 1 runs the same sequence (self-test).
 *   all_gather_f32: recv[world][count] <- every rank's send[count];  all_reduce_f64: buf[count] summed in place
 *   scratch: device memory of the table's own (>= (4 + 3 world) * widest BatchNorm * 8 bytes; 1 MiB is plenty)
 * Providers: callbacks (i3d_set_collectives), RCCL (below), peer-write exchange (further below: the production one).
 * RCCL provider: rank 0 i3d_rccl_unique_id -> (caller broadcasts the 128 bytes) -> every rank i3d_rccl_init ->
 * i3d_set_collectives_rccl.  i3d_set_collectives(NULL): off.  One training per process. */
typedef struct {
    int world;
    int (*all_gather_f32)(void* user, const float* send, float* recv, long count, void* stream);
    int (*all_reduce_f64)(void* user, double* buf, long count, void* stream);
    void* user;
    void* scratch;
    long scratch_bytes;
} I3dCollectives;
int i3d_set_collectives(const I3dCollectives* c /* host; copied */);
int i3d_collectives_world(void); /* 0: none set */
/* The installed provider's collectives as the BatchNorm entry points call them when no fused exchange applies: recv[world][count]
 * <- every rank's send[count]; buf[count] <- sum over the ranks (the peer provider: in rank order).  The vectors are staged through the
 * table's scratch as the BatchNorm entry points stage theirs ((count + world count) floats / count doubles must fit).  For a binding's
 * self-test (tests/test_gpu_dist.py); every rank makes the same calls in the same order on one stream.  (No reference counterpart:
 * SURVEY.md C3.) */
int i3d_collectives_all_gather_f32(const float* send, float* recv, long count, void* stream);
int i3d_collectives_all_reduce_f64(double* buf, long count, void* stream);
int i3d_rccl_available(void);
int i3d_rccl_unique_id(char* out128 /* host */);
int i3d_rccl_init(const char* id128 /* host */, int rank, int world, void** comm);
int i3d_rccl_destroy(void* comm);
int i3d_set_collectives_rccl(void* comm, int world, void* scratch, long scratch_bytes);
/* ---- provider 3: one-shot peer-write exchange over IPC-mapped mailboxes (csrc/peer.h, peer.hip) -----------------------
 * What the reference's whole-batch BatchNorm (models/base_layers.py:87, 100-111: nn.BatchNorm1d over every row the model is
 * given) costs once the batch is sharded over ranks: ~50 vectors of <= [3F] floats per step, each needed by every rank, on
 * the step's dependent chain.  Every rank allocates a mailbox (i3d_peer_alloc: uncached device memory + its IPC handle),
 * the caller distributes the handles (torch.distributed all_gather_object), every rank maps the others' mailboxes
 * (i3d_peer_open) and installs the provider (i3d_set_collectives_peer).  While it is installed
 *   - i3d_bn_finalize_partials exchanges its {sum, M2, count} triples INSIDE the finalisation kernel (one launch, as
 *     without synchronisation);
 *   - i3d_bn_bwd*, i3d_act_stats_fwd* and the 3D network's edge stage exchange their fp64 sums in one launch
 *     (append row count -> sum over ranks in rank order -> fp32 vectors + 1 / rows);
 *   - the table's generic all_gather_f32 / all_reduce_f64 are one-launch kernels of the same protocol.
 * A kernel writes payload + sequence flag into every rank's mailbox and waits (bounded: timeout_s, default 30 s,
 * I3D_PEER_TIMEOUT_S) for the `world` flags in its own; a wait that times out is reported by the NEXT call (error return,
 * i3d_peer_status) - no hung GPU.  One stream issues the collectives, in the same order on every rank.  Results are
 * bit-identical on all ranks (sums in rank order).  world <= 16, BatchNorm width <= 4095. */
long i3d_peer_mailbox_bytes(void);
int i3d_peer_handle_bytes(void); /* sizeof(hipIpcMemHandle_t) = 64 */
int i3d_peer_alloc(void** mailbox /* out: device */, char* handle_out /* host, i3d_peer_handle_bytes() */);
int i3d_peer_free(void* mailbox); /* a mailbox that did not end up in a context (i3d_peer_close frees the one it owns) */
int i3d_peer_open(void* mailbox, const char* handles /* host [world][handle bytes], rank order */, int rank, int world,
                  double timeout_s /* <= 0: default */, void** ctx /* out */);
int i3d_set_collectives_peer(void* ctx, void* scratch, long scratch_bytes);
/* a second context (own mailbox, sequence and scratch) for the collectives issued on `stream`: the 3D network's side stream keeps
 * running beside the 2D network under synchronised BatchNorm; every rank binds the same streams in the same roles */
int i3d_peer_bind_stream(void* ctx, void* stream, void* scratch, long scratch_bytes);
int i3d_peer_status(void* ctx); /* 0: healthy, else the sequence number of the collective whose wait timed out */
long long i3d_peer_sequence(void* ctx); /* collectives issued so far */
int i3d_peer_close(void* ctx); /* unmaps the peers, frees the mailbox; every rank must have stopped issuing collectives */

/* ---- all weight gradients of a layer in ONE launch + one fixed-order reduction (csrc/wgrad.hip) ----------------------
 * Replaces autograd's dW = dY^T X of every nn.Linear of a PNA layer (reference models/base_layers.py:101; the Linears of
 * models/pna.py:186-197) - round 2 issued them as ~10 launches per layer.  A problem is one product
 *     P[M,N] = sum_{k_begin <= k < k_begin + k_count} A[row(k), 0:M]^T B[row(k), 0:N],   row(k) = rows ? rows[k] : k
 * (rows[k] = -1: padding, contributes nothing; rows_total = physical rows of A and B).  An output takes the products of
 * n_groups consecutive problems (first_problem ...), all of one shape:
 *   I3D_WGRAD_PLAIN   (n_groups 1)  C[m, n] = P, rows m >= c_split displaced by c_delta floats (c_split 0: none)
 *   I3D_WGRAD_BN      (n_groups 1)  C = (P - row[m] mean[n]) scale[n] + row[m] shift[n], aff = mean | scale | shift [3N]:
 *                     the product against a BatchNorm output that was never materialised (i3d_gemm_f32_wgrad_bn)
 *   I3D_WGRAD_COMBINE               C[m, s * scaler_stride + n] = sum_g coef[g * n_scalers + s] P_g  (host coef): the
 *                     per-degree posttrans gradients folded into the scaler blocks (i3d_pna_combine_weights_bwd); at
 *                     most one such output per call
 * Needs M, N, lda, ldb, ldc multiples of 4 and 16-byte aligned pointers (i3d_wgrad_multi_supported), and
 * i3d_wgrad_multi_workspace_bytes(units) of scratch for `units` K-slice panels (more scratch = shorter slices, up to
 * I3D_WGRAD_UNITS = 256 by default).  Deterministic: every slice is summed in a fixed order. */
#define I3D_WGRAD_PLAIN 0
#define I3D_WGRAD_BN 1
#define I3D_WGRAD_COMBINE 2
typedef struct {
    const float* A;
    const float* B;
    const int* rows;
    long rows_total;
    int lda, ldb, M, N;
    int k_begin, k_count;
} I3dWgradProblem;
typedef struct {
    int kind, n_groups, first_problem;
    int ldc, c_split, n_scalers;
    long c_delta, scaler_stride;
    float* C;
    const float* aff;
    const float* row;
    const float* coef; /* host */
} I3dWgradOutput;
int i3d_wgrad_multi_supported(const I3dWgradProblem* problems, int n_problems, const I3dWgradOutput* outputs, int n_outputs);
long i3d_wgrad_multi_workspace_bytes(int max_units);
long i3d_wgrad_multi_min_workspace_bytes(const I3dWgradProblem* problems, int n_problems); /* with the longest slices */
int i3d_wgrad_multi(const I3dWgradProblem* problems, int n_problems, const I3dWgradOutput* outputs, int n_outputs,
                    void* workspace, long workspace_bytes, void* stream);

/* i3d_pna_aggregate_fwd / _bwd with the messages read as (e - mean) * scale + shift (aff [3 feat], may be NULL) */
int i3d_pna_aggregate_fwd_aff(const float* e, const float* aff, const int* in_ptr, int num_nodes, int feat,
                              const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                              int force_scalers, float avg_d_log, float* out, void* stream);
int i3d_pna_aggregate_bwd_aff(const float* grad_out, const float* e, const float* aff, const int* in_ptr, int num_nodes,
                              int feat, const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                              int force_scalers, float avg_d_log, float* grad_e, void* stream);
/* the same two with the messages stored as bf16 (e_bf16 != 0: row r of the [E, feat] message matrix at (bf16*)e + r * feat, feat a
 * multiple of 4, 8-byte aligned; the bf16 matmul mode's storage form of the last pretrans block's activation): K4 reads half the
 * message bytes; outputs and gradients stay fp32 */
int i3d_pna_aggregate_fwd_ex(const void* e, int e_bf16, const float* aff, const int* in_ptr, int num_nodes, int feat,
                             const int* aggregators, int n_aggregators, const int* scalers, int n_scalers, int force_scalers,
                             float avg_d_log, float* out, void* stream);
int i3d_pna_aggregate_bwd_ex(const float* grad_out, const void* e, int e_bf16, const float* aff, const int* in_ptr, int num_nodes,
                             int feat, const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                             int force_scalers, float avg_d_log, float* grad_e, void* stream);

/* The BatchNorm backward (i3d_bn_bwd and its variants; reference models/base_layers.py:100-111 under autograd) as ONE launch for
 * tensors of up to 256 * (4096 / feat) * 4 rows, feat % 4 == 0, feat <= 512, activations none / ReLU / LeakyReLU: a workgroup
 * keeps its row chunk in registers across an in-launch reduction (every workgroup of the launch is resident: <= 256 workgroups,
 * two per CU).  Process-wide, on by default (environment I3D_BN_BWD_ONE_LAUNCH=0); switch it OFF when several processes share
 * one GPU (their launches compete for the CUs and the residency argument no longer holds).  Returns the previous setting. */
int i3d_set_bn_bwd_one_launch(int on);
int i3d_bn_bwd_one_launch_supported(int rows, int feat); /* 1: a local (not synchronised) i3d_bn_bwd of this shape takes it */
/* i3d_bn_bwd with the finalisation of grad_bias deferred (bias_partial != NULL): see I3dBnTail.bias_partial */
long i3d_bn_bias_partial_floats(int feat);
int i3d_bn_bwd_deferred_bias(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act, int post_act,
                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float* grad_gamma, float* grad_beta, float* grad_pre, float* grad_bias, double* sums_out,
                             const double* sums_in, long total_rows, void* workspace, float* bias_partial, void* stream);
int i3d_bn_bias_finalize(const float* bias_partial, int rows, int feat, float* grad_bias, void* stream);
/* The BatchNorm backward of a PNA layer's edge block (the FCLayer inside pretrans_edges, reference models/pna.py:237-252 with
 * models/base_layers.py:100-111) fused with the two segmented sums behind it: column sums of dy and dy xhat as i3d_bn_bwd (one
 * launch; grad_gamma / grad_beta), then ONE launch that forms the data gradient g of every edge row inside the sums that consume
 * it - out_src[v] = sum of g over the out-edges of v (rows out_epos[out_ptr[v] .. out_ptr[v+1])), out_dst[v] = sum over its
 * in-edges (rows in_ptr[v] .. in_ptr[v+1]) - and stores g to grad_pre [rows, feat].  The same bits as i3d_bn_bwd followed by
 * i3d_segment_sum_pair.  act: none / ReLU / LeakyReLU; feat % 4 == 0, feat <= 512; synchronised statistics as i3d_bn_bwd.
 * The bias gradient of the Linear in front is the column sum of out_dst: i3d_colsum_strided (any stream). */
int i3d_bn_bwd_edge_sums(const float* grad_y, const float* x, int rows, int feat, int act, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre,
                         const int* in_ptr, const int* out_ptr, const int* out_epos, int num_nodes, float* out_src, float* out_dst,
                         int ldo, void* workspace, void* stream);
/* out[c] = sum over rows of x[r * ldx + c] (two launches, fixed order: deterministic); partial: i3d_bn_bias_partial_floats(feat) floats
 * of the caller's (the reduction's row-chunk partials - not the BatchNorm workspace, which another stream may be using) */
int i3d_colsum_strided(const float* x, int ldx, int rows, int feat, float* out, float* partial, void* stream);

/* ---- composites: one call enqueues a whole FCLayer-shaped block (forward or backward) ------------------------
 * "input operator -> Linear -> activation -> BatchNorm1d (training, local statistics) -> post-activation (+ residual)",
 * reference models/base_layers.py:100-111, with the three input operators of the PNA / Net3D layers.  Same kernels
 * as the per-kernel entry points above, sequenced in C++ so the host pays one transition per block instead of ~7-13.
 * All pointers are device pointers except the struct itself (host).  Forward fills xact (activation output, saved),
 * pre_keep (Linear output, only when the activation needs it: SiLU/Sigmoid; NULL otherwise: activation in place),
 * tail.mean / tail.invstd, y.  Backward reads them and fills the grad_* members (grad_x / grad_q may be NULL). */
typedef struct {
    int act, post_act;
    float eps, momentum;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* mean;
    float* invstd;
    void* workspace; /* i3d_colreduce_workspace_bytes(rows, f_out) */
    void* gemm_workspace; /* scratch of the weight-gradient GEMMs of the backward (i3d_gemm_f32_ws), may be NULL */
    long gemm_workspace_bytes;
    long long* num_batches_tracked; /* BatchNorm1d's int64 counter, incremented by the forward (may be NULL) */
    float* bias_partial; /* backward, optional (i3d_bn_bias_partial_floats(f_out) floats): the bias gradient is finalised
                          * from these partials next to the weight gradients instead of inside the data-gradient pass */
} I3dBnTail;

typedef struct { /* y = tail(x W^T + b) */
    I3dBnTail tail;
    int rows, f_in, f_out, ldw;
    const float* x;
    const float* W;
    const float* bias;
    const float* residual;
    float* xact;
    float* pre_keep;
    float* y;
    const float* grad_y;
    float* grad_pre; /* scratch [rows, f_out] */
    float* grad_gamma;
    float* grad_beta;
    float* grad_W;
    float* grad_bias;
    float* grad_x;
    void* W_dgrad_panel; /* optional: the block's weight packed for the row-panel data gradient (i3d_panel_pack(W, ldw, f_in, f_out,
                          * trans 0): i3d_panel_packed_bytes(f_in, f_out) bytes) - filled by i3d_pna_layer_weights_fwd / the layer's
                          * forward pass, read by its backward pass (dX = dZ W through i3d_panel_gemm).  NULL: the tiled product */
    void* W_fwd_panel;   /* optional: the same weight packed for the row-panel FORWARD product of the fused-BatchNorm layer path
                          * (i3d_panel_pack(W, ldw, f_out, f_in, trans 1): i3d_panel_packed_bytes(f_out, f_in) bytes) */
} I3dFcArgs;

typedef struct { /* y = tail(P[src,:F] + P[dst,F:] + q W_q^T + b),  P = h [W_s|W_d]^T  (reference models/pna.py:237-252) */
    I3dBnTail tail;
    int num_nodes, num_edges, f_h, f_q, f_out, ldw;
    int q_rows;  /* 0: q is [E, f_q];  V > 0: q is the [V, f_q] table of all categorical combinations (i3d_edge_codes) */
    int v_pad;   /* leading dimension of onehot */
    const int* q_code;   /* [E] row of q for every edge (table mode) */
    const float* onehot; /* [E, v_pad] */
    float* grad_Q;       /* scratch [v_pad, f_out] (table mode) */
    const float* h;
    const float* q;
    const float* W;
    const float* bias;
    const int* src_s;
    const int* dst_s;
    const int* in_ptr;
    const int* out_ptr;
    const int* out_epos;
    float* P; /* [N, 2*f_out] scratch */
    float* Q; /* [E, f_out] ([V, f_out] in table mode) scratch (NULL when q is NULL) */
    float* xact;
    float* pre_keep;
    float* y;
    const float* grad_y;
    float* grad_pre; /* scratch [E, f_out] */
    float* grad_P;   /* scratch [N, 2*f_out] */
    float* grad_gamma;
    float* grad_beta;
    float* grad_W;
    float* grad_bias;
    float* grad_h;
    float* grad_q;
    int grad_q_accumulate; /* grad_q += ... instead of = (a q shared by several layers: the bond table) */
} I3dEdgeFcArgs;

typedef struct { /* y = tail(h W_h^T + b + agg[deg group] W_D^T),  W_D = sum_s coef[g][s] W_s  (models/pna.py:207-209, 229-233) */
    I3dBnTail tail;
    int num_nodes, f_h, f_out, agg_width, ldw, n_groups, n_scalers, m_padded;
    int group_start[32];
    int group_count[32];
    float coef[128]; /* [n_groups][n_scalers] */
    const float* h;
    const float* agg;
    const float* W;
    const float* bias;
    const float* residual;
    const int* deg_rows;
    const int* deg_tile_group;
    float* WD; /* [n_groups, f_out, agg_width] */
    float* xact;
    float* pre_keep;
    float* y;
    const float* grad_y;
    float* grad_pre; /* scratch [N, f_out] */
    float* grad_WD;  /* scratch like WD */
    float* grad_gamma;
    float* grad_beta;
    float* grad_W;
    float* grad_bias;
    float* grad_h;
    float* grad_agg;
} I3dGroupedFcArgs;

#define I3D_MAX_EXTRA_FC 3
typedef struct { /* one PNA layer, reference models/pna.py:199-216: pretrans edge MLP (edge block + n_pre_extra plain FC
                  * blocks) -> aggregation (i3d_pna_aggregate_*) -> posttrans (degree-grouped block + n_post_extra plain FC
                  * blocks) + residual.  The member blocks are chained by the caller (pre[0].x = edge.y, ...); forward
                  * and backward are the member composites back to back, the three contributions to dL/dh (posttrans,
                  * edge MLP, residual) are summed into post.grad_h. */
    I3dEdgeFcArgs edge;
    int n_pre_extra;
    I3dFcArgs pre[I3D_MAX_EXTRA_FC];
    int n_aggregators, n_scalers, force_scalers;
    int aggregators[8];
    int scalers[4];
    float avg_d_log;
    const float* msg;      /* [E, F] output of the pretrans MLP (= y of its last block) */
    float* grad_msg;       /* [E, F] scratch: gradient of the messages */
    I3dGroupedFcArgs post;
    int n_post_extra;
    I3dFcArgs postx[I3D_MAX_EXTRA_FC];
    int residual;          /* h_new += h (fused into the last block's BN apply by the caller via its residual pointer) */
    const float* grad_out; /* [N, F] incoming gradient of the layer output (added to post.grad_h when residual) */
    void* agg_event_start; /* optional (i3d_event_create): recorded on the stream right before / after the forward */
    void* agg_event_stop;  /* aggregation kernel - the roofline measurement of bench.py */
    /* fused BatchNorm (csrc/fused_bn.hip): statistics out of the producers' epilogues, BatchNorm-apply in the consumers'
     * loads.  Then edge.y / pre[i].y are not written (may be NULL), msg = the xact of the last pretrans block, and the
     * degree groups of `post` must cover every node (in-degree 0 included, zero coefficients).  Needs activations whose
     * derivative follows from the output (none / ReLU / LeakyReLU), no post-activation, n_post_extra == 0. */
    int fused_bn;
    int defer_join; /* backward: do not wait for the weight-gradient stream at the end of the layer; the caller keeps every
                     * buffer that stream reads or writes alive and calls i3d_wgrad_stream_join before it consumes them */
    float* stats_ws;                   /* scratch, i3d_pna_layer_stats_floats(...) floats */
    float* aff[I3D_MAX_EXTRA_FC + 1];  /* [3 f_out] mean | gamma invstd | beta of the edge block and of pre[i] (saved) */
    int weights_ready; /* forward (fused_bn): edge.Q and post.WD - products of parameters and the bond table only - are already
                        * there (i3d_pna_layer_weights_fwd, e.g. on the side stream while the layers before this one run) */
    /* round 3: the products that read the node features h as ONE GEMM per direction.  Forward: PL [N, 2 f_out(edge) + f_out(post)]
     * = h Wcat^T + bcat (Wcat = [W_s ; W_d ; W_h] packed by i3d_pna_pack_h_weights) - its first 2 f_out columns are the edge
     * block's P, the rest the posttrans block's h-product, which the grouped GEMM takes as its addend.  Backward: DL (same
     * shape) = [dP | dlin], dL/dh (+)= DL Wcat in one GEMM at the end of the layer.  merge_h = 1 needs fused_bn,
     * n_post_extra == 0 and the four buffers (Wcat, bcat, PL saved by the forward pass; DL backward scratch that the
     * weight-gradient stream reads). */
    int merge_h;
    float* Wcat;
    float* bcat;
    float* PL;
    float* DL;
    int wgrad_split;   /* backward: issue the posttrans weight gradients as soon as the posttrans chain is done and the rest at the
                        * end, as two launches (the last layer of a backward pass: nothing runs next to its weight gradients
                        * once the chain has ended); 0: one launch at the end of the layer */
    int eval_mode;     /* forward (fused_bn) in eval(): BatchNorm with the running statistics (reference: nn.BatchNorm1d in eval
                        * mode, trainer/trainer.py:75 model.eval()) - aff[i] already hold mean | gamma / sqrt(var + eps) | beta
                        * (i3d_bn_eval_aff_multi), no statistics are finalised, no running statistic is touched */
    int msg_bf16;      /* fused_bn with at least one later pretrans block, bf16 matmul mode: `msg` (= the last pretrans block's xact,
                        * [E, f_msg]) holds bf16 - written by that block's GEMM epilogue, read by the aggregation kernels and by the
                        * block's BatchNorm backward; the buffer keeps its fp32 size.  Same value forward and backward. */
    float* edge_bias_partial; /* backward, optional (i3d_bn_bias_partial_floats(edge.f_out) floats): with it (and merge_h, an
                        * activation of the none / ReLU / LeakyReLU class, 16-byte rows) the edge block's BatchNorm backward runs
                        * as i3d_bn_bwd_edge_sums - the data gradient formed inside the two segmented sums behind it - and its
                        * bias gradient is taken from dP[dst] on the weight-gradient stream through this buffer */
    void* Wcat_panel;  /* optional (merge_h): Wcat packed for the row-panel forward product PL = h Wcat^T + bcat (i3d_panel_pack, trans 1:
                        * i3d_panel_packed_bytes(2 f_out(edge) + f_out(post), f_h) bytes), packed where Wcat is.  NULL: the tiled product */
    void* Wcat_dgrad_panel; /* optional (merge_h): Wcat packed for the backward product dL/dh (+)= DL Wcat (trans 0:
                        * i3d_panel_packed_bytes(f_h, 2 f_out(edge) + f_out(post)) bytes) */
} I3dPnaLayerArgs;

/* eval-mode affine vector of one BatchNorm: aff [3 feat] = running_mean | gamma / sqrt(running_var + eps) | beta */
typedef struct {
    const float* running_mean;
    const float* running_var;
    const float* gamma;
    const float* beta;
    float* aff;
    int feat;
    float eps;
} I3dBnEvalAff;
int i3d_bn_eval_aff_multi(const I3dBnEvalAff* entries /* host */, int n, void* stream);

/* floats of I3dPnaLayerArgs.stats_ws for a layer with these dimensions (f = widest block output) */
long i3d_pna_layer_stats_floats(int num_nodes, int num_edges, int m_padded, int f);

/* ---- the edge stage of the 3D network, one lane per edge (net3d_edge.hip) ---------------------------------------
 * reference models/net3d.py:57-81 + 100-118 for propagation_depth 1, one message block, broadcast node embedding:
 *   e0 = post(BN_in(act(W_in fourier(d) + b_in))),  m = BN_msg(act(W_msg [emb | emb | e0] + b_msg)),
 *   m_sum[v] = mean/sum over the in-edges of m * sigmoid(w_gate . m + b_gate)
 * Forward fills d_out (e0 in edge-id order: what the reference leaves in graph.edata['d']), x_msg, aff_*, tail_*.mean /
 * invstd (running statistics updated), m_sum.  Backward reads them and grad_m_sum and fills the grad_* members
 * (grad_emb is accumulated into: the caller has put the node-level part there). */
typedef struct {
    I3dBnTail tail_in;   /* edge-input block: act, post_act, BatchNorm over [E, hidden] (workspace members unused) */
    I3dBnTail tail_msg;  /* message block: act, BatchNorm (post_act must be none) */
    int num_nodes, num_edges, hidden, n_enc, reduce_mean;
    int ld_w_in, ld_w_msg;
    const float* d_raw;  /* [E] distances, edge-id order */
    const int* perm;     /* [E] edge id of the j-th edge in destination-sorted order */
    const int* dst_s;    /* [E] destination node, destination-sorted */
    const int* in_ptr;   /* [N + 1] */
    const float* emb;    /* [hidden] */
    const float* W_in;   /* [hidden, 2 n_enc + 1 (1 when n_enc == 0)] */
    const float* b_in;
    const float* W_msg;  /* [hidden, 3 hidden] */
    const float* b_msg;
    const float* w_gate; /* [hidden] */
    const float* b_gate; /* [1] */
    float* stats;        /* scratch, i3d_net3d_edge_stats_floats(E, hidden) floats */
    float* aff_in;       /* [3 hidden] saved */
    float* aff_msg;      /* [3 hidden] saved */
    float* x_msg;        /* [E, hidden] saved: activation output of the message block, destination-sorted */
    float* d_out;        /* [E, hidden] saved + graph side effect, edge-id order */
    float* msg;          /* [E, hidden] scratch (forward) */
    float* m_sum;        /* [N, hidden] out */
    const float* grad_m_sum; /* [N, hidden] */
    float* grad_ya;      /* scratch [E, hidden] (backward) */
    float* grad_lin;     /* unused since round 5 (the gradient of the message block's pre-activation stays in registers); may be null */
    float* partial;      /* scratch, i3d_net3d_edge_bwd_floats(E, hidden, n_enc) floats (backward) */
    float* grad_W_in;
    float* grad_b_in;
    float* grad_gamma_in;
    float* grad_beta_in;
    float* grad_W_msg;
    float* grad_b_msg;
    float* grad_gamma_msg;
    float* grad_beta_msg;
    float* grad_w_gate;
    float* grad_b_gate;
    float* grad_emb;     /* += */
    int store_bf16;      /* x_msg and msg - the [E, hidden] activations only this stage reads and writes - hold bf16 (the buffers
                          * keep their fp32 size): the bf16 matmul mode's storage form of the stage.  x_msg is stored about
                          * x_center (its columns can be nearly constant over a batch: see n3_center_kernel).  The same value in
                          * the forward and the backward call of a pass. */
    float* x_center;     /* [hidden] saved (store_bf16 only) */
} I3dNet3dEdgeArgs;
int i3d_net3d_edge_supported(int hidden, int n_enc);   /* 1 when the kernels are built for this combination */
long i3d_net3d_edge_stats_floats(int num_edges, int hidden);
long i3d_net3d_edge_bwd_floats(int num_edges, int hidden, int n_enc);
int i3d_net3d_edge_fwd(const I3dNet3dEdgeArgs* args, void* stream);
int i3d_net3d_edge_bwd(const I3dNet3dEdgeArgs* args, void* stream);

/* timing events for measurements around a kernel inside a composite (thin wrappers of hipEvent_t) */
int i3d_event_create(void** event);
int i3d_event_destroy(void* event);
int i3d_event_record(void* event, void* stream);
int i3d_event_elapsed_ms(void* start, void* stop, float* ms);   /* both events must have completed */

/* makes `stream` wait for everything its weight-gradient side stream holds (I3dPnaLayerArgs.defer_join) */
int i3d_wgrad_stream_join(void* stream);
/* the side stream of `stream`, ordered behind everything `stream` holds now (*side == stream when there is none) */
int i3d_wgrad_stream_fork(void* stream, void** side);
int i3d_wgrad_stream_peek(void* stream, void** side);      /* that stream without a new fork (`stream` itself if there is none) */
/* the parameter-only products of a fused_bn layer's forward: edge.Q = q W_q^T and post.WD = sum_s coef W_s */
int i3d_pna_layer_weights_fwd(const I3dPnaLayerArgs* args, void* stream);
int i3d_pna_layer_fwd(const I3dPnaLayerArgs* args, void* stream);
int i3d_pna_layer_bwd(const I3dPnaLayerArgs* args, void* stream);
int i3d_fc_bn_fwd(const I3dFcArgs* args, void* stream);
int i3d_fc_bn_bwd(const I3dFcArgs* args, void* stream);
/* the two halves of i3d_fc_bn_bwd: BatchNorm backward + data gradient (what the block in front waits for), and the weight
 * gradients (leaves: they may run on another stream that is ordered after the chain half, e.g. i3d_wgrad_stream_fork) */
int i3d_fc_bn_bwd_chain(const I3dFcArgs* args, void* stream);
int i3d_fc_bn_bwd_wgrad(const I3dFcArgs* args, void* stream);
int i3d_edge_fc_bn_fwd(const I3dEdgeFcArgs* args, void* stream);
int i3d_edge_fc_bn_bwd(const I3dEdgeFcArgs* args, void* stream);
int i3d_grouped_fc_bn_fwd(const I3dGroupedFcArgs* args, void* stream);
int i3d_grouped_fc_bn_bwd(const I3dGroupedFcArgs* args, void* stream);

/* ---- whole-model sequencing (csrc/model.hip): the PNA forward / backward of a training step from one C call each ----
 * replaces PNA.forward -> PNAGNN.forward -> [PNALayer.forward]* -> readout -> head (reference models/pna.py:131-135,
 * 161-166, 199-213, 127-129) and its autograd backward for the pre-training configuration: every block Linear ->
 * activation (none / ReLU / LeakyReLU) -> BatchNorm1d in training mode with local statistics, >= 2 degree scalers
 * (degree-grouped posttrans), categorical bond features (bond table), one posttrans block.  The memory layout of the
 * saved activations and of the backward scratch is computed here (i3d_pna_model_saved_floats / _scratch_floats). */
#define I3D_MAX_LAYERS 16
#define I3D_MAX_HEAD_FC 4
typedef struct { /* one FCLayer: Linear + optional BatchNorm1d (reference models/base_layers.py:23-111) */
    const float* W;     /* [f_out, f_in] row-major, contiguous */
    const float* bias;
    const float* gamma; /* NULL: no BatchNorm */
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* num_batches_tracked;
    float* grad_W; /* outputs of the backward pass (written, not accumulated) */
    float* grad_bias;
    float* grad_gamma;
    float* grad_beta;
    int f_in, f_out, act;
    float eps, momentum;
} I3dFcParams;

typedef struct {
    int training; /* 1: BatchNorm with batch statistics (running statistics updated); 0: eval mode, forward only (the
                   * validation pass of trainer/trainer.py:72-78) */
    int n_layers, hidden, n_pre, residual;
    int n_aggregators, aggregators[8];
    int n_scalers, scalers[4];
    float avg_d_log;
    I3dFcParams pre[I3D_MAX_LAYERS][I3D_MAX_EXTRA_FC + 1]; /* pretrans blocks (block 0: [h_src | h_dst | e] -> f_out) */
    I3dFcParams post[I3D_MAX_LAYERS];                      /* posttrans block ([h | agg x scalers] -> hidden) */
    int n_atom_tables, atom_dims[16];
    const float* atom_tables[16];
    float* grad_atom_tables; /* [sum atom_dims, hidden]: the table gradients, contiguous in table order */
    int n_bond_tables, bond_dims[16];
    const float* bond_tables[16];
    float* grad_bond_tables;
    int n_readout, readout_ops[4];
    int n_head;
    I3dFcParams head[I3D_MAX_HEAD_FC];
} I3dPnaModel;

typedef struct {
    int num_nodes, num_edges, num_graphs;
    const int64_t* atom_feat; /* [N, n_atom_tables] */
    const int64_t* bond_feat; /* [E, n_bond_tables], edge-id order */
    const int* in_ptr;        /* the destination-sorted index of 3dinfomax_amd/graph.py: GraphIndex */
    const int* perm;
    const int* src_s;
    const int* dst_s;
    const int* out_ptr;
    const int* out_epos;
    const int* graph_ptr;
    const int* deg_rows; /* nodes grouped by in-degree (0 included), groups padded to 64 with -1 */
    const int* deg_tile_group;
    int m_padded, n_groups;
    int group_degree[32], group_start[32], group_count[32];
    const int64_t* comb; /* [n_comb, n_bond_tables]: every combination of the categorical bond features (device) */
    int n_comb, v_pad;
} I3dPnaBatch;

long i3d_pna_model_saved_floats(const I3dPnaModel* model, const I3dPnaBatch* batch);   /* -1: not supported */
long i3d_pna_model_scratch_floats(const I3dPnaModel* model, const I3dPnaBatch* batch);
/* out [B, f_out of the last head block]; node_emb [N, hidden] (the graph's ndata['feat'] side effect); edge_emb [E, hidden]
 * or NULL (edata['feat'] side effect); agg_events: NULL or 2 * n_layers events recorded around the aggregation kernels;
 * *ctx_out: host-side context for i3d_pna_model_bwd (valid while `saved` is), freed with i3d_pna_model_ctx_free */
int i3d_pna_model_fwd(const I3dPnaModel* model, const I3dPnaBatch* batch, float* saved, float* node_emb, float* edge_emb,
                      float* out, void* bn_workspace, void* const* agg_events, void* stream, void** ctx_out);
/* grads_from: the same model with its grad_* members set (they are chosen at backward time), or NULL to use the forward's */
int i3d_pna_model_bwd(void* ctx, const I3dPnaModel* grads_from, const float* grad_out, float* scratch, void* bn_workspace,
                      void* gemm_workspace, long gemm_workspace_bytes, void* stream);
/* the backward pass in two calls (data parallel: the all-reduce of the first part's gradients runs next to the second):
 * part 1 = head + layers [split, n_layers), part 2 = layers [0, split) + encoders; part 0 = everything */
int i3d_pna_model_bwd_part(void* ctx, const I3dPnaModel* grads_from, const float* grad_out, float* scratch, void* bn_workspace,
                           void* gemm_workspace, long gemm_workspace_bytes, int part, int split, void* stream);
int i3d_pna_model_ctx_free(void* ctx);
/* test entries: out [rows, feat] = (e - mean) * scale + shift exactly as i3d_pna_aggregate_*_aff read the messages
 * (aff = mean | scale | shift, NULL: a copy); the messages of one layer of a forward pass from its context, [E, f_msg] in
 * destination-sorted order - the tests derive the kernels' arg-max / arg-min routing from them */
int i3d_pna_messages_normalized(const float* e, const float* aff, long rows, int feat, float* out, void* stream);
int i3d_pna_messages_normalized_ex(const void* e, int e_bf16, const float* aff, long rows, int feat, float* out, void* stream);
int i3d_pna_model_debug_messages(void* ctx, int layer, float* out, void* stream);

/* ---- Adam step of all parameter tensors in one launch (csrc/adam.hip; reference: torch.optim.Adam built by name,
 * train.py:189, stepped at trainer/trainer.py:120).  chunk_table: device array of n_chunks records {float* param,
 * const float* grad, float* exp_avg, float* exp_avg_sq, int n} (i3d_adam_chunk_bytes() bytes each, n <=
 * i3d_adam_chunk_elems()); steps: the n_steps per-parameter step counters (fp32), incremented by the launch;
 * bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) for the step t being taken.
 * The update is torch's fused Adam expression for expression (L2 weight decay; no amsgrad / maximize / grad scaling). */
int i3d_adam_chunk_elems(void);
int i3d_adam_chunk_bytes(void);
int i3d_adam_step(const void* chunk_table, int n_chunks, float* steps, int n_steps, double lr, double beta1, double beta2,
                  double weight_decay, double eps, double bias_correction1, double bias_correction2_sqrt, void* stream);

/* ---- contrastive monitoring metrics -------------------------------------------------
 * replaces the nine metric modules of configs_clean/pre-train_QM9.yml:15-24 (reference trainer/metrics.py:161-174,
 * 212-333, 443-463; commons/losses.py:946-959), evaluated every log_iterations steps by
 * trainer/self_supervised_trainer.py:31-50.  Inputs are products computed with i3d_gemm_f32:
 *   S = x1 x2[:B1]^T [B1,B1], G1 = x1 x1^T [B1,B1], G2 = x2 x2^T [B2,B2] (B2 >= B1: "noisy" extra rows), C = X^T X [D,D].
 * i3d_contrastive_rowstats: out[B2, 8] per row i: {sum_j cos_ij, cos_ii, [(cos_ii+1)/2 > threshold],
 *   #{j != i: (cos_ij+1)/2 <= threshold}, |x1_i - x2_i|^alpha, sum_{j>i} exp(-t|x1_i-x1_j|^2), the same for x2, 0}
 *   (cos = S / (|x1_i| |x2_j|); rows >= B1 only carry the x2 term).
 * i3d_cov_rowstats: out[D, 2] per row a of C: {sum_{b != a} ((C_ab - s_a s_b / n)/(n-1))^2, C_aa}, s = column sums. */
int i3d_contrastive_rowstats(const float* S, const float* G1, const float* G2, int B1, int B2, float threshold, float t,
                             float alpha, float* out, void* stream);
int i3d_cov_rowstats(const float* C, const float* colsum, int n, int D, float* out, void* stream);

/* ---- batch assembly -----------------------------------------------------------------
 * replaces B x QM9Dataset.get_complete_graph (reference datasets/qm9_dataset.py:233-244, :215-217) + dgl.batch
 * (datasets/custom_collate.py:108-109) for the 3D view: from coords [N,3] and the node offsets graph_ptr[B+1] (plus
 * edge_ptr[B+1], the prefix sum of n(n-1)) builds, on the device, the complete graphs in the reference's edge-id
 * order (src_id/dst_id int64 [E3], d_id fp32 [E3] = ||x_src-x_dst||) AND the destination-sorted kernel index
 * (in_ptr[N+1] (== out_ptr), src_s, dst_s, perm, inv_perm, out_epos: int32 [E3]). */
int i3d_complete_graph_build(const float* coords, const int* graph_ptr, const int* edge_ptr, int num_graphs,
                             int num_nodes, int num_edges, int* in_ptr, int* src_s, int* dst_s, int* perm,
                             int* inv_perm, int* out_epos, int64_t* src_id, int64_t* dst_id, float* d_id,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INFOMAX3D_HIP_H */
