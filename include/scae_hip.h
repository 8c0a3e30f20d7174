/*
 * scae_hip.h -- C ABI of libscae_hip.so: the MI355X (gfx950) kernels of the
 * SCAE forward/backward hot path.
 *
 * Drop-in boundary.  The reference (bdsaglam/torch-scae) has no FFI of its
 * own -- every op on the path is a stock ATen call made from Python
 * (SURVEY.md section 8b).  The entry points below are therefore what a
 * maintainer of the reference would bind (ctypes / cffi stub in
 * INTEGRATION.md) to replace each ATen op cluster, one launcher per cluster,
 * each citing the reference lines it replaces (paths relative to the
 * reference checkout).
 *
 * Conventions
 *   - plain pointers + sizes only; no torch types, no exceptions, no
 *     allocation, no global state (the one stateful object, a launch list,
 *     is a handle the caller holds), no implicit synchronisation;
 *   - every pointer is a DEVICE pointer to densely packed (contiguous,
 *     row-major, last index fastest) float32 unless stated otherwise; the
 *     caller owns every buffer, including the "saved"/"partial" workspaces;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it
 *     and the call returns immediately (safe under hipGraph capture);
 *   - launchers are re-entrant (autograd calls the *_bwd ones from its own
 *     worker thread);
 *   - return value: 0 on success, SCAE_ERR_* (negative) for rejected
 *     arguments, otherwise the positive hipError_t of the failed launch;
 *   - pointers documented "nullable" select a mode of the reference op.
 */
#ifndef SCAE_HIP_H
#define SCAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: scae_launch_list_begin(stream) -> handle; scae_conv3x3_relayout* write the packed
 * fragment-major filter copy behind wf (scae_conv3x3_wf_floats); scae_mlp_chain_desc has
 * row_tile / bf16; scae_decoder_desc has bwd_resident; scae_first_layer_desc has out_h /
 * rwfh / rwdh */
#define SCAE_ABI_VERSION 2

#define SCAE_OK 0
#define SCAE_ERR_BAD_ARG (-1)      /* null pointer / non-positive size     */
#define SCAE_ERR_UNSUPPORTED (-2)  /* shape outside the kernels' limits    */

/* limits of this build (checked by the launchers) */
#define SCAE_MAX_CHANNELS 4        /* image channels C                     */
#define SCAE_ATTN_MAX_SET 64       /* queries N and keys M per attention   */
#define SCAE_RENDER_MAX_TEMPLATE_ELEMS 4096 /* C*th*tw + th*tw per template */

int scae_abi_version(void);
const char *scae_error_string(int code);

/* Launch lists: a training step as the library's own record of its kernel launches.
 * scae_launch_list_begin(stream) opens a recording and returns its handle (NULL: out of
 * memory).  Until scae_launch_list_end(list) every kernel launch that an entry point of
 * this library SUCCESSFULLY issues on exactly that `stream` (from any host thread --
 * autograd's backward worker launches there too; also into the stream while it is being
 * captured) is appended to the list: kernel, grid, block, LDS bytes, a copy of the argument
 * bytes.  Launches on other streams are not seen, and any number of recordings -- one per
 * stream -- may be open at once: the recording is the only state, and the caller holds it.
 * scae_launch_list_run re-issues the launches, in order, on `stream`: what replaying a
 * captured HIP graph of the same launches does, with a hipLaunchKernel per launch on the
 * host and without the graph's end-of-launch cost on the device.  The list holds pointers,
 * not buffers: the caller keeps every buffer the recorded launches use alive and unmoved.
 * Only this library's kernels are recorded -- a graph captured over the same region may
 * hold other nodes (memsets, another library's kernels); whoever replays a list in place
 * of a graph checks scae_launch_list_size against the graph's node count. */
void *scae_launch_list_begin(void *stream);
int scae_launch_list_end(void *list); /* stops recording; SCAE_ERR_BAD_ARG if not open */
int scae_launch_list_size(const void *list); /* kernel launches recorded (both lanes) */
int scae_launch_list_run(const void *list, void *stream);
void scae_launch_list_free(void *list); /* (also ends a recording that is still open) */
/* Two lanes.  A HIP graph with a forked branch replays its branches one after the other on
 * this stack, while kernels of two plain streams do run side by side
 * (tools/probes/stream_overlap.cpp): a step whose backward has two independent chains is
 * therefore recorded on TWO streams and re-issued on two.
 *   _side_stream: launches on `side_stream` are recorded too, as lane 1 (once per list,
 *     while it is open).
 *   _order(later, earlier): the caller has just made `later_stream` wait for the work given
 *     to `earlier_stream` so far (an event of its own: the library issues nothing here);
 *     every open recording that holds both streams notes the edge at this position.  Returns
 *     the number of recordings that did.
 *   _run2: lane 0 on `stream`, lane 1 on `side_stream`, each noted edge re-issued as an
 *     event record + stream wait (events owned by the list; one run of a list at a time).
 *     side_stream NULL or equal to stream: everything in recorded order on `stream`
 *     (= scae_launch_list_run).
 *   _side_size: the launches of lane 1. */
int scae_launch_list_side_stream(void *list, void *side_stream);
int scae_launch_list_order(void *later_stream, void *earlier_stream);
int scae_launch_list_run2(const void *list, void *stream, void *side_stream);
int scae_launch_list_side_size(const void *list);
int scae_launch_list_lane(const void *list, int i); /* lane of launch i (0 / 1), -1 past the end */
/* Diagnostic: one run with a timing event in front of and behind every launch, on the
 * launch's own stream; out_us[2 i], out_us[2 i + 1] = start / end of launch i in microseconds
 * after the first (n_out >= 2 * size).  Synchronises both streams.  A kernel trace serialises
 * the dispatches of a process; this shows the two lanes as they run. */
int scae_launch_list_timeline(const void *list, void *stream, void *side_stream, float *out_us,
                              int n_out);

/* ------------------------------------------------------------------------
 * Presence-logit noise     replaces torch.rand_like (part_encoder.py:106,
 *     object_decoder.py:201): out[0..n) ~ U[0,1), Philox4x32-10 keyed by
 *     state[0] (seed) and counted by state[1] (launches so far), which the
 *     kernel itself advances -- so a replayed HIP graph draws fresh noise.
 *     state: 3 uint64 of device memory {seed, 0, 0} owned by the caller; one
 *     state per stream of concurrent use.
 * ---------------------------------------------------------------------- */
int scae_uniform_f32(float *out, int64_t n, uint64_t *state, void *stream);

/* ------------------------------------------------------------------------
 * K5  geometric_transform            replaces cv_ops.py:20-76
 *   pose (n,6) -> out (n,6), or (n,9) when as_matrix (third row 0,0,1).
 *   similarity / nonlinear / as_matrix: the reference's three flags.
 * ---------------------------------------------------------------------- */
int scae_geometric_transform_fwd_f32(const float *pose, float *out, int64_t n,
                                     int similarity, int nonlinear,
                                     int as_matrix, void *stream);
/* gout has the forward's output layout; gpose (n,6). */
int scae_geometric_transform_bwd_f32(const float *pose, const float *gout,
                                     float *gpose, int64_t n, int similarity,
                                     int nonlinear, int as_matrix,
                                     void *stream);

/* 3 x 3 products of the hierarchical CapsuleLayer.forward (object_decoder.py:184-191:
 * torch.matmul(cvr.repeat(1, 1, n_votes, 1, 1), cpr)): out[c][v] = left[c] right[c][v]
 *   left (n_caps,3,3)  right (n_caps,V,3,3)  out (n_caps,V,3,3); backward: gleft
 *   (n_caps,3,3) nullable (the parent's transform may be a constant), gright like right. */
int scae_mat3_mul_fwd_f32(const float *left, const float *right, float *out, int64_t n_caps,
                          int V, void *stream);
int scae_mat3_mul_bwd_f32(const float *left, const float *right, const float *gout,
                          float *gleft, float *gright, int64_t n_caps, int V, void *stream);

/* ------------------------------------------------------------------------
 * K2  qkv_attention                  replaces set_transformer.py:24-47
 *   q (HB,N,dk)  k (HB,M,dk)  v (HB,M,dv)  presence (HB,M) nullable
 *   out (HB,N,dv);  probs (HB,N,M) = softmax((q k^T - (1-presence) 1e32) /
 *   sqrt_dk) is written for the backward pass.  QK^T and PV run on the fp32
 *   MFMA (v_mfma_f32_16x16x4_f32).  N, M <= SCAE_ATTN_MAX_SET.
 *   sqrt_dk: the divisor, float(np.sqrt(dk)) in the reference (:43).
 * ---------------------------------------------------------------------- */
int scae_qkv_attention_fwd_f32(const float *q, const float *k, const float *v,
                               const float *presence, float *out, float *probs,
                               int HB, int N, int M, int dk, int dv,
                               float sqrt_dk, void *stream);
/* gq (HB,N,dk) gk (HB,M,dk) gv (HB,M,dv); gpresence (HB,M) nullable. */
/* bf16 forward (BASELINE.json configs[2]): q, k, v, out as bf16 bit patterns
 * (what torch.autocast hands the attention), both contractions on the bf16
 * matrix cores with fp32 accumulation, mask / scale / softmax in fp32; probs
 * (HB,N,M) stays fp32 -- the backward pass is scae_qkv_attention_bwd_f32 on
 * upcast operands. */
int scae_qkv_attention_fwd_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                const float *presence, uint16_t *out, float *probs, int HB,
                                int N, int M, int dk, int dv, float sqrt_dk, void *stream);
int scae_qkv_attention_bwd_f32(const float *q, const float *k, const float *v,
                               const float *probs, const float *gout,
                               float *gq, float *gk, float *gv,
                               float *gpresence, int HB, int N, int M, int dk,
                               int dv, float sqrt_dk, void *stream);

/* ------------------------------------------------------------------------
 * K2b  fused set-transformer trunk   replaces set_transformer.py:212-219
 *      (fc1 -> n_layers x SAB(x, x, presence) -> fc2; SAB/MAB :107-142 with
 *      n_heads = 1), forward and backward in one launch each.
 *   input x (B,N,Din) is given as 1..4 column segments (so the caller never
 *   materialises the reference's torch.cat, stacked_capsule_auto_encoder.py
 *   :105-124): segment s holds element (b,n,j) at seg_ptr[s][b*batch_stride +
 *   n*row_stride + j], j < seg_width[s]; sum of widths = Din.
 *   presence (B,N) nullable.  params: ONE packed buffer,
 *     [W1 (D,Din), b1 (D)] + per layer [Wq,bq,Wk,bk,Wv,bv,Wo,bo,
 *     (ln0.weight, ln0.bias), Wf,bf, (ln1.weight, ln1.bias)] + [W2 (Dout,D),
 *     b2 (Dout)], every matrix row-major (out,in) like nn.Linear.weight;
 *     scae_set_encoder_param_count() gives its length.
 *   z (B,N,Dout) = fc2 output; hsave (B,L+1,N,D) workspace kept for backward.
 *   Dout = 0 selects "trunk only": no W2/b2 in params, z (B,N,D) receives the
 *   last block's output (the caller folds fc2 into the output attention, K2c).
 *   Limits: N <= 64, D in {8,16,32}, n_heads = 1.
 *   backward: gz (B,N,Dout) -> seg_grad[s] (B,N,width) contiguous, nullable
 *   per segment; pg_partial (scae_set_encoder_grid(B), P) per-workgroup
 *   parameter gradients in the packed layout (caller sums over dim 0).
 * ---------------------------------------------------------------------- */
int scae_set_encoder_param_count(int D, int Din, int Dout, int L, int layer_norm);
int scae_set_encoder_grid(int B);
/* 1 if the shape fits the kernels (LDS budget of the backward pass), else 0 */
int scae_set_encoder_supported(int N, int D, int Din, int Dout, int L, int layer_norm);
int scae_set_encoder_fwd_f32(int nseg, const float *const *seg_ptr, const int *seg_width,
                             const int *seg_row_stride, const int64_t *seg_batch_stride,
                             const float *presence, const float *params, float *z,
                             float *hsave, int B, int N, int D, int Din, int Dout, int L,
                             int layer_norm, void *stream);
int scae_set_encoder_bwd_f32(int nseg, const float *const *seg_ptr, const int *seg_width,
                             const int *seg_row_stride, const int64_t *seg_batch_stride,
                             float *const *seg_grad, const float *presence,
                             const float *params, const float *hsave, const float *gz,
                             float *pg_partial, int B, int N, int D, int Din, int Dout,
                             int L, int layer_norm, void *stream);
/* BASELINE.json configs[2] ("bf16 ... MFMA attention path"): the same two passes with the
 * attention products of every block -- Q K^T, P V (set_transformer.py:24-47) and, in the
 * backward, dO V^T, dS K, dS^T Q, P^T dO -- on v_mfma_f32_16x16x16_bf16 (operands rounded
 * to bf16, fp32 accumulate); mask / scale / softmax, projections, LayerNorm and the saved
 * activations stay fp32.  scae_set_encoder_bf16_supported: the shapes the matrix-core
 * (one wave per 16-row tile) kernels cover; others return SCAE_ERR_UNSUPPORTED. */
int scae_set_encoder_bf16_supported(int N, int D, int Din, int Dout, int L, int layer_norm);
int scae_set_encoder_fwd_bf16(int nseg, const float *const *seg_ptr, const int *seg_width,
                              const int *seg_row_stride, const int64_t *seg_batch_stride,
                              const float *presence, const float *params, float *z,
                              float *hsave, int B, int N, int D, int Din, int Dout, int L,
                              int layer_norm, void *stream);
int scae_set_encoder_bwd_bf16(int nseg, const float *const *seg_ptr, const int *seg_width,
                              const int *seg_row_stride, const int64_t *seg_batch_stride,
                              float *const *seg_grad, const float *presence,
                              const float *params, const float *hsave, const float *gz,
                              float *pg_partial, int B, int N, int D, int Din, int Dout,
                              int L, int layer_norm, void *stream);

/* ------------------------------------------------------------------------
 * K2c  output attention with folded projections
 *      replaces set_transformer.py:218-223 (fc2 + MultiHeadQKVAttention(seeds,
 *      z, z, presence), n_heads = 1) given the trunk output h (B,N,D):
 *        K' = h wk^T + bk,  V' = h wv^T + bv   (wk = Wk W2 etc., folded by the
 *        caller; bo and the value bias ride inside bv because softmax rows sum
 *        to one),  out = softmax((q K'^T - (1-presence) 1e32)/sqrt(C)) V'
 *   q (O,C) = q_projector(seeds) (batch invariant); wk, wv (C,D); bk, bv (C);
 *   out (B,O,C); probs (B,O,N) nullable (inspection only).
 *   A set is shared out over S = scae_seed_attention_splits(B,O) workgroups
 *   (groups of queries).  backward: gout (B,O,C) -> gh (S,B,N,D), one slab
 *   per query group (caller sums over dim 0); partial
 *   (scae_seed_attention_grid(B,O), O*C + 2*C*D + 2*C) = per-workgroup
 *   [gq | gwk | gbk | gwv | gbv] (caller sums over dim 0).
 *   Limits: N, O <= 64, D in {8,16,32}, C % 8 == 0, LDS.
 * ---------------------------------------------------------------------- */
int scae_seed_attention_splits(int B, int O);
int scae_seed_attention_grid(int B, int O);
int scae_seed_attention_supported(int N, int O, int D, int C);
int scae_seed_attention_fwd_f32(const float *h, const float *q, const float *wk,
                                const float *bk, const float *wv, const float *bv,
                                const float *presence, float *out, float *probs, int B,
                                int N, int O, int D, int C, void *stream);
int scae_seed_attention_bwd_f32(const float *h, const float *q, const float *wk,
                                const float *bk, const float *wv, const float *bv,
                                const float *presence, const float *gout, float *gh,
                                float *partial, int B, int N, int O, int D, int C,
                                void *stream);

/* The same output attention on the matrix cores (seed_attention_wave.hip; D = 16,
 * N, O <= 64, C a multiple of 64), with two more steps of algebra: the logits contract
 * over D through qk = q wk (the q bk term is constant along the keys and drops out of
 * the softmax), and out = (P h) wv^T + bv.  bk is therefore not an input, and its
 * gradient is the exact zero the softmax's shift invariance implies.
 *   fwd : out (B,O,C)
 *   bwd : gh (B,N,16) and, per workgroup, one row of `partial`
 *         (scae_seed_attention_mfma_rows(B), O*16 + C*16 + C) = [d(qk) | dwv | dbv]
 *   reduce: the column sums of `partial`, expanded to the gradients of the operands
 *         of scae_seed_attention_fwd_f32: gq (O,C) = d(qk) wk^T, gwk (C,16) = q^T d(qk),
 *         gbk (C) = 0, gwv (C,16), gbv (C) -- one launch (it replaces the column-sum
 *         launch behind scae_seed_attention_bwd_f32). */
int scae_seed_attention_mfma_supported(int N, int O, int D, int C);
int scae_seed_attention_mfma_rows(int B);
int scae_seed_attention_mfma_fwd_f32(const float *h, const float *q, const float *wk,
                                     const float *wv, const float *bv, const float *presence,
                                     float *out, int B, int N, int O, int C, void *stream);
int scae_seed_attention_mfma_bwd_f32(const float *h, const float *q, const float *wk,
                                     const float *wv, const float *presence, const float *gout,
                                     float *gh, float *partial, int B, int N, int O, int C,
                                     void *stream);
/* BASELINE.json configs[2]: the attention products of the output attention (logits q.wkf h^T,
 * P h; backward dT h^T, dS h, P^T dT + dS^T qk) on v_mfma_f32_16x16x16_bf16, everything else
 * (folded projections, softmax, the C-wide output product) fp32.  Same arguments. */
int scae_seed_attention_mfma_fwd_bf16(const float *h, const float *q, const float *wk,
                                      const float *wv, const float *bv, const float *presence,
                                      float *out, int B, int N, int O, int C, void *stream);
int scae_seed_attention_mfma_bwd_bf16(const float *h, const float *q, const float *wk,
                                      const float *wv, const float *presence,
                                      const float *gout, float *gh, float *partial, int B,
                                      int N, int O, int C, void *stream);
int scae_seed_attention_mfma_reduce_f32(const float *partial, int rows, const float *q,
                                        const float *wk, float *gq, float *gwk, float *gbk,
                                        float *gwv, float *gbv, int O, int C, void *stream);

/* ------------------------------------------------------------------------
 * K2d  batch-invariant weight folding feeding K2c
 *      replaces the per-element fc2 / q,k,v,o projector Linear layers of
 *      set_transformer.py:218-223, :36-75 by parameter-only products:
 *        q   = seeds Wq^T + bq                   (O,C)
 *        wkf = Wk W2,       bkf = Wk b2 + bk     (C,D), (C)
 *        wvf = Wo Wv W2,    bvf = Wo (Wv b2 + bv) + bo
 *      wv2e (C,D+1) = [Wv W2 | Wv b2 + bv] and wowv (C,C) = Wo Wv are kept for the
 *      backward pass.
 *      backward: gradients of (q, wkf, bkf, wvf, bvf) -> gradients of all
 *      eleven parameters; gv2e, t1: unused (the workspaces of an earlier two-launch
 *      form; may be NULL).
 *      Limits: C % 64 == 0, C <= 1024, D in {8,16,32}, O <= 64
 *      (scae_seed_fold_supported).
 * ---------------------------------------------------------------------- */
typedef struct scae_seed_fold_desc {
  const float *seeds;                   /* (O,C) */
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo; /* (C,C) / (C) */
  const float *w2, *b2;                 /* fc2: (C,D), (C) */
  float *q, *wkf, *bkf, *wvf, *bvf;     /* outputs */
  float *wv2e;                          /* (C,D+1) output kept for backward */
  float *wowv;                          /* (C,C) Wo Wv, kept for backward (NULL: not kept;
                                           scae_seed_fold_bwd_f32 needs it) */
  int O, C, D;
} scae_seed_fold_desc;
typedef struct scae_seed_fold_grads {
  const float *g_q, *g_wkf, *g_bkf, *g_wvf, *g_bvf; /* incoming gradients */
  float *d_seeds, *d_wq, *d_bq, *d_wk, *d_bk, *d_wv, *d_bv, *d_wo, *d_bo, *d_w2, *d_b2;
  float *gv2e, *t1;                     /* unused, may be NULL */
} scae_seed_fold_grads;
int scae_seed_fold_supported(int O, int C, int D);
int scae_seed_fold_fwd_f32(const scae_seed_fold_desc *desc, void *stream);
int scae_seed_fold_bwd_f32(const scae_seed_fold_desc *desc, const scae_seed_fold_grads *grads,
                           void *stream);

/* ------------------------------------------------------------------------
 * LayerNorm over the last dimension      replaces nn.LayerNorm(d) of MAB
 *     (set_transformer.py:114-131) where the blocks run module by module.
 *     x, y, gy, gx: (rows, d); weight / bias (d) nullable; mean, rstd (rows)
 *     saved for the backward; d <= 1024.  bwd: gx (nullable) and, when
 *     `partial` is given, scae_layer_norm_rows(rows) partial rows [gw (d) |
 *     gb (d)] for the caller to sum (scae_sum_rows_f32).
 * ---------------------------------------------------------------------- */
int scae_layer_norm_rows(int64_t rows);
int scae_layer_norm_fwd_f32(const float *x, const float *weight, const float *bias, float *y,
                            float *mean, float *rstd, int64_t rows, int d, float eps,
                            void *stream);
int scae_layer_norm_bwd_f32(const float *x, const float *weight, const float *mean,
                            const float *rstd, const float *gy, float *gx, float *partial,
                            int64_t rows, int d, void *stream);

/* ------------------------------------------------------------------------
 * K7  batched fp32 MFMA GEMM with fused epilogue -- the per-capsule MLPs of
 *     CapsuleLayer (object_decoder.py:86-107, :137-158: a Python loop of 4*O
 *     tiny GEMMs in the reference), forward and backward:
 *       C[g][m][n] = epi( sum_k A[g](m,k) * B[g](n,k) ),   g < batch
 *     A(m,k) = A[g*a_batch + (a_kcontig ? m*lda + k : k*lda + m)], B likewise;
 *     C[g*c_batch + m*ldc + n].  epi: + bias[g*bias_batch + n*bias_ld]
 *     (nullable), ReLU if relu, then zeroed where mask[g*mask_batch +
 *     m*ldmask + n] <= 0 (nullable; the ReLU gate of the backward pass).
 *     asum (nullable, needs a_kcontig == 0): asum[g*asum_batch + m*asum_ld] =
 *     sum_k A[g](m,k), the bias gradient that goes with a weight-gradient GEMM
 *     (asum_ld <= 0 means 1; a stride lets it land in a column of C).
 *     Also serves the 1x1 attention convolution of part_encoder.py:71-73.
 * ---------------------------------------------------------------------- */
int scae_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                  const float *mask, float *asum, int batch, int M, int N, int K,
                  int a_kcontig, int lda, int64_t a_batch, int b_kcontig, int ldb,
                  int64_t b_batch, int ldc, int64_t c_batch, int bias_ld, int64_t bias_batch,
                  int ldmask, int64_t mask_batch, int64_t asum_batch, int asum_ld, int relu,
                  void *stream);

/* Two independent GEMMs of the kind above in ONE launch (e.g. the weight- and
 * the data-gradient GEMM of a layer, which wait for the same incoming
 * gradient and are each too small to fill the device).  Fields as the
 * arguments of scae_gemm_f32. */
typedef struct scae_gemm_desc {
  const float *A, *B;
  float *C;
  const float *bias, *mask;
  float *asum;
  int batch, M, N, K;
  int a_kcontig, lda;
  int64_t a_batch;
  int b_kcontig, ldb;
  int64_t b_batch;
  int ldc;
  int64_t c_batch;
  int bias_ld;
  int64_t bias_batch;
  int ldmask;
  int64_t mask_batch, asum_batch;
  int relu, asum_ld;
  float *c_nomask; /* nullable, layout of C: the values before the mask gate */
} scae_gemm_desc;
int scae_gemm_pair_f32(const scae_gemm_desc *first, const scae_gemm_desc *second, void *stream);
/* Up to 4 independent GEMMs of that kind in ONE launch (the four weight-gradient
 * GEMMs of a capsule's MLP chain once the data-gradient chain has run). */
int scae_gemm_multi_f32(const scae_gemm_desc *descs, int n, void *stream);
/* scae_seed_attention_mfma_bwd_{f32,bf16} and scae_gemm_multi_f32(descs, n) in ONE launch
 * (csrc/seed_bwd_gemm.hip): in a training step the output attention's backward and the
 * weight-gradient GEMMs of the capsule MLPs both wait for the MLPs' data-gradient chain
 * only and neither fills the device.  SCAE_ERR_UNSUPPORTED unless the GEMMs take the
 * 32 x 32 split-K tiles (then they fill the device alone: launch them separately). */
int scae_seed_attention_mfma_bwd_gemm_f32(const float *h, const float *q, const float *wk,
                                          const float *wv, const float *presence,
                                          const float *gout, float *gh, float *partial, int B,
                                          int N, int O, int C, const scae_gemm_desc *descs,
                                          int n, void *stream);
int scae_seed_attention_mfma_bwd_gemm_bf16(const float *h, const float *q, const float *wk,
                                           const float *wv, const float *presence,
                                           const float *gout, float *gh, float *partial, int B,
                                           int N, int O, int C, const scae_gemm_desc *descs,
                                           int n, void *stream);


/* ------------------------------------------------------------------------
 * K7b a chain of up to 4 per-group layers in ONE launch -- the two ReLU MLPs
 *     CapsuleLayer runs back to back per object capsule (object_decoder.py:
 *     86-107 `mlps`, :137-158 `caps_mlps`, the `cat([.., caps_exist])` of :149
 *     between them as an implicit ones column), forward and the data-gradient
 *     half of their backward.  A workgroup carries 16 batch rows of one group
 *     through all layers (activations in LDS, weights as MFMA fragments from L2).
 *     in: element (b, g, k) at in[g*in_gs + b*in_bs + k], k < in_dim.
 *   fwd: layer i: y = x W_i^T (W_i rows = N outputs, K = width of x, row
 *     stride ldw, group stride w_gs) + bias[g*bias_gs + n*bias_ld] (nullable),
 *     ReLU if relu; stored (nullable except for the last layer) at
 *     out[g*out_gs + b*out_bs + n].
 *   bwd: layer i: y = x W_i (W_i rows = the K contraction entries, N columns
 *     used), zeroed where gate[g*gate_gs + b*gate_bs + n] <= 0 (nullable);
 *     out as above.  Pass the layers in backward order; x of the first = the
 *     gradient w.r.t. the last pre-activation.
 *     Each layer's K must equal its predecessor's N (in_dim for the first);
 *     widths <= scae_mlp_chain_max_width().
 * ---------------------------------------------------------------------- */
typedef struct scae_mlp_chain_layer {
  const float *w;
  int64_t w_gs;
  int ldw, K, N;
  const float *bias;
  int64_t bias_gs;
  int bias_ld;
  const float *gate;
  int64_t gate_gs, gate_bs;
  float *out;
  int64_t out_gs, out_bs;
  int relu;
} scae_mlp_chain_layer;
typedef struct scae_mlp_chain_desc {
  scae_mlp_chain_layer layer[4];
  int n_layers;
  const float *in;
  int64_t in_gs, in_bs;
  int in_dim, B, G;
  int row_tile; /* batch rows per workgroup: 0 = chosen by the launcher from the shape
                   (the production setting); 16 | 32 force one (tests, measurements) */
  int bf16;     /* != 0: the layer products on v_mfma_f32_16x16x16_bf16 -- both operands rounded
                   to bf16 (nearest even) on their way into the matrix core, fp32 accumulate,
                   fp32 tensors in memory, epilogues unchanged (BASELINE.json configs[2]) */
} scae_mlp_chain_desc;
int scae_mlp_chain_max_width(void);
int scae_mlp_chain_fwd_f32(const scae_mlp_chain_desc *desc, void *stream);
int scae_mlp_chain_bwd_f32(const scae_mlp_chain_desc *desc, void *stream);

/* The chain with the vote kernel (K3, scae_capsule_votes_*_f32 below) riding in
 * the same launch: CapsuleLayer.forward (object_decoder.py:120-236) from the
 * object encoding to the votes as ONE launch, and the first half of its backward
 * (vote gradients -> parameter-row gradients -> the data-gradient chain) as one.
 *   fwd: the last layer's output rows (B, G, ld_param), N = 8V+7, are the vote
 *     kernel's all_param; outputs as scae_capsule_votes_fwd_f32.
 *   bwd: desc->in is ignored -- the chain's input block is the vote kernel's
 *     gall_param_gated (or gall_param when that is NULL), computed in the launch
 *     from the vote gradients and also written out (the weight-gradient GEMM of
 *     the last layer and the bias sums read them); all_param = the saved rows.
 * Fields as the arguments of scae_capsule_votes_fwd/bwd_f32; B and O are the
 * chain's B and G. */
typedef struct scae_votes_desc {
  const float *all_param; /* bwd only */
  const float *cpr_static, *bias_cvr, *bias_caps, *bias_vote, *bias_scale;
  const float *noise_caps, *noise_vote; /* nullable */
  float noise_scale;
  int V, ld_param, similarity, learn_vote_scale, allow_deformations;
  /* forward outputs */
  float *vote, *scale, *vote_presence, *logit_caps, *logit_vote, *reg_partial;
  float *caps_presence; /* nullable together with caps_arg */
  int *caps_arg;
  /* backward: incoming gradients (nullable), outputs */
  const float *gvote, *gscale, *gvote_presence, *glogit_caps, *glogit_vote, *greg;
  const float *gcaps_presence;
  float *gall_param, *gcpr_in, *gall_param_gated;
} scae_votes_desc;
int scae_mlp_chain_votes_fwd_f32(const scae_mlp_chain_desc *desc, const scae_votes_desc *votes,
                                 void *stream);
int scae_mlp_chain_votes_bwd_f32(const scae_mlp_chain_desc *desc, const scae_votes_desc *votes,
                                 void *stream);

/* bf16-operand forms of the GEMM-shaped launchers (BASELINE.json configs[2], "bs=1024
 * bf16"): same arguments, same fp32 tensors in memory; the operands are rounded to bf16
 * (nearest even) on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation, in 128 x 128 tiles.  Problems too small for those tiles (a side shorter
 * than 32; convolutions whose channel counts are not multiples of 128 or that have too few
 * pixels to fill the device) run the fp32 kernels unchanged.  Epilogues (bias, ReLU,
 * gates, column sums -- those of the values as they arrive, rounded) are identical. */
int scae_gemm_bf16(const float *A, const float *B, float *C, const float *bias,
                   const float *mask, float *asum, int batch, int M, int N, int K,
                   int a_kcontig, int lda, int64_t a_batch, int b_kcontig, int ldb,
                   int64_t b_batch, int ldc, int64_t c_batch, int bias_ld, int64_t bias_batch,
                   int ldmask, int64_t mask_batch, int64_t asum_batch, int asum_ld, int relu,
                   void *stream);
int scae_gemm_pair_bf16(const scae_gemm_desc *first, const scae_gemm_desc *second, void *stream);
/* scae_gemm_multi_f32 on the bf16 tiles when every problem has both sides >= 32 (else the
 * fp32 tiles, unchanged) */
int scae_gemm_multi_bf16(const scae_gemm_desc *descs, int n, void *stream);
/* scae_conv3x3_fwd_f32 (the 32 x 64 second-generation tiles: Cin % 32 == 0, Cout % 64 == 0)
 * with the workgroups of scae_seed_fold_fwd_f32(fold) as the tail of its grid: in a training
 * step the parameter-only folding products hide behind an early, large launch instead of
 * lengthening the prologue.  SCAE_ERR_UNSUPPORTED: launch the two separately. */
int scae_conv3x3_fwd_fold_f32(const float *in, const float *wf, const float *bias, float *out,
                              const float *post_bias, float *out_post, int B, int IH, int IW,
                              int Cin, int Cout, int stride, const scae_seed_fold_desc *fold,
                              void *stream);
/* K8r: the forward with the input images resident in LDS (csrc/conv_resident.hip), for the
 * small layers of the encoder: a workgroup owns `group` images x 32 output channels, stages
 * their input pixels once, reads the filter from its fragment-major copy `wp` (see relayout)
 * straight into MFMA fragments.  Same contract as scae_conv3x3_fwd_f32 otherwise (results agree
 * to fp32 round-off: the K sum is split by input-channel block).  Cin % 128 == 0,
 * Cout % 32 == 0, the group's pixels <= 64 KiB of LDS: scae_conv3x3_fwd_res_supported, else
 * SCAE_ERR_UNSUPPORTED.  group: 0 = by shape; > 0 forces a group size (tests). */
int scae_conv3x3_fwd_res_supported(int B, int IH, int IW, int Cin, int Cout, int stride);
int scae_conv3x3_fwd_res_f32(const float *in, const float *wp, const float *bias, float *out,
                             const float *post_bias, float *out_post, int B, int IH, int IW,
                             int Cin, int Cout, int stride, int group, void *stream);
int scae_conv3x3_fwd_bf16(const float *in, const float *wf, const float *bias, float *out,
                          const float *post_bias, float *out_post, int B, int IH, int IW, int Cin,
                          int Cout, int stride, void *stream);
int scae_conv3x3_bwd_pair_bf16(const float *dpre, const float *wd, const float *in, float *din,
                               float *partial, int B, int IH, int IW, int Cin, int Cout,
                               int stride, void *stream);

/* ------------------------------------------------------------------------
 * K8  3x3 "valid" convolutions of the CNN encoder as implicit GEMMs on the
 *     fp32 matrix cores      replaces part_encoder.py:26-44 / nn_ext.py:34-59
 *     (Conv2d(k=3, stride, padding=0) + ReLU) and their autograd backward.
 *   Activations NHWC: in (B,IH,IW,Cin) -> out (B,OH,OW,Cout), OH=(IH-3)/s+1.
 *   relayout: w (Cout,Cin,3,3) -> wf, wd (Cin,9,Cout).  `wf` holds scae_conv3x3_wf_floats(Cout,Cin)
 *     floats: the (Cout,9,Cin) layout the tile kernels read (`wf` of fwd / fwd_fold: a caller may
 *     fill that block by hand), followed -- when Cin, Cout % 32 == 0 -- by the fragment-major copy
 *     `wp` = wf + Cout*9*Cin of the image-resident forward (fwd_res): the filter split into the
 *     three bf16 planes of its exact fp32 products (csrc/bf16x6.h), 1.5 * Cout*9*Cin floats.
 *   first_*: direct kernels for the image layer (NCHW image, small Cin; w in
 *     the reference layout, Cout % 64 == 0); first_wgrad writes
 *     scae_conv3x3_first_wgrad_rows(B,Cout) partial rows, each
 *     [dW (Cout,Cin*9) | db (Cout)], for the caller to sum over the rows.
 *   fwd:   out = relu(conv(in, wf) + bias)        Cin, Cout % 64 == 0, s <= 2
 *          out_post (nullable, NHWC like out) = out + post_bias, post_bias
 *          (Cout,OH,OW) = img_embedding_bias of part_encoder.py:87 (the sum
 *          the 1x1 attention convolution reads)
 *   dgrad: din = conv^T(dpre, wd), zeroed where gate <= 0 (gate (B,IH,IW,Cin)
 *          = the producing layer's ReLU output, nullable)
 *   wgrad: dw (Cout,Cin,3,3) = sum_pixels dpre x in and (db nullable) db (Cout)
 *          = sum_pixels dpre; partial is a workspace of
 *          scae_conv3x3_wgrad_splits(B,OH,OW,Cin,Cout) * (9*Cout*Cin + Cout) floats.
 *          dw == NULL leaves the partials unreduced; wgrad_reduce_batch then
 *          reduces the partials of up to 8 layers in one launch (HOST arrays).
 * ---------------------------------------------------------------------- */
/* floats a `wf` buffer of relayout / relayout_batch / first_fwd_relayout / the step prologue
 * must hold for a (Cout, Cin) layer: Cout*9*Cin, x 2.5 when the packed copy is written */
int64_t scae_conv3x3_wf_floats(int Cout, int Cin);
int scae_conv3x3_relayout_f32(const float *w, float *wf, float *wd, int Cout, int Cin,
                              void *stream);

/* K8 with bf16-RESIDENT operands (BASELINE configs[2]; csrc/conv_bf16.hip): the same three
 * passes of part_encoder.py:26-44 / nn_ext.py:34-59 with every GEMM operand stored as bf16
 * (uint16_t bit patterns, round-to-nearest-even of the fp32 values) -- NHWC activations,
 * pre-activation gradients, the re-laid-out filters wf (Cout,9,Cin) / wd (Cin,9,Cout) --
 * multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Cin, Cout % 128 == 0,
 * stride 1 or 2, tensors below 2 GiB (scae_conv3x3_bf16r_supported).
 *   cvt_bf16_batch: dst[a][0..n[a]) = bf16(src[a]) for up to 8 arrays (n[a] % 8 == 0) in one
 *     launch; the three arrays are HOST arrays.
 *   fwd:   out_h (bf16) = relu(conv(in, wf) + bias); out_f (nullable, fp32): the same values
 *          unrounded; out_post (nullable, fp32) = relu(.) + post_bias (Cout,OH,OW).
 *   dgrad: din = conv^T(dpre, wd), zeroed where gate <= 0 (gate: bf16 (B,IH,IW,Cin),
 *          nullable), written as bf16 (din_h) and / or fp32 (din_f).
 *   wgrad: partial = scae_conv3x3_wgrad_bf16r_splits(..) x (9*Cout*Cin + Cout) fp32 partials
 *          in the layout of scae_conv3x3_wgrad_f32's (dw == NULL): the fp32 reduction
 *          launches (wgrad_reduce_batch, first_wgrad_reduce) sum them. */
int scae_conv3x3_bf16r_supported(int B, int IH, int IW, int Cin, int Cout, int stride);
/* scae_conv3x3_first_fwd_relayout_f32 for the bf16-resident encoder: the image layer's output
 * as bf16 (out_h), the re-laid-out filters as fp32 (rwf / rwd, as there) AND as bf16 (rwfh / rwdh) */
int scae_conv3x3_first_fwd_relayout_bf16(const float *img, const float *w, const float *bias,
                                         uint16_t *out_h, int B, int Cin, int IH, int IW,
                                         int Cout, int stride, int n_layers,
                                         const float *const *rw, float *const *rwf,
                                         float *const *rwd, uint16_t *const *rwfh,
                                         uint16_t *const *rwdh, const int *rCout,
                                         const int *rCin, void *stream);
int scae_cvt_bf16_batch(int n_arrays, const float *const *src, uint16_t *const *dst,
                        const int64_t *n, void *stream);
int scae_conv3x3_fwd_bf16r(const uint16_t *in, const uint16_t *wf, const float *bias,
                           uint16_t *out_h, float *out_f, const float *post_bias,
                           float *out_post, int B, int IH, int IW, int Cin, int Cout, int stride,
                           void *stream);
int scae_conv3x3_dgrad_bf16r(const uint16_t *dpre, const uint16_t *wd, const uint16_t *gate,
                             uint16_t *din_h, float *din_f, int B, int IH, int IW, int Cin,
                             int Cout, int stride, void *stream);
int scae_conv3x3_wgrad_bf16r_splits(int B, int OH, int OW, int Cin, int Cout);
int scae_conv3x3_wgrad_bf16r(const uint16_t *dpre, const uint16_t *x, float *partial, int B,
                             int IH, int IW, int Cin, int Cout, int stride, void *stream);
/* the same for n_layers <= 8 layers in one launch; the five arrays are HOST arrays */
int scae_conv3x3_relayout_batch_f32(int n_layers, const float *const *w, float *const *wf,
                                    float *const *wd, const int *Cout, const int *Cin,
                                    void *stream);
int scae_conv3x3_first_fwd_f32(const float *img, const float *w, const float *bias,
                               float *out, int B, int Cin, int IH, int IW, int Cout,
                               int stride, void *stream);
int scae_conv3x3_first_wgrad_rows(int B, int Cout);
int scae_conv3x3_first_wgrad_f32(const float *dpre, const float *img, float *partial, int B,
                                 int Cin, int IH, int IW, int Cout, int stride, void *stream);
/* first_fwd / first_wgrad with riders: the parameter-only filter re-layouts of the
 * following n_layers layers (arguments of scae_conv3x3_relayout_batch_f32), resp.
 * the split reductions of their weight-gradient partials (arguments of
 * scae_conv3x3_wgrad_reduce_batch_f32), run as extra workgroups of the same
 * launch -- they are independent of the image layer.  n_layers = 0: no riders. */
int scae_conv3x3_first_fwd_relayout_f32(const float *img, const float *w, const float *bias,
                                        float *out, int B, int Cin, int IH, int IW, int Cout,
                                        int stride, int n_layers, const float *const *rw,
                                        float *const *rwf, float *const *rwd, const int *rCout,
                                        const int *rCin, void *stream);
int scae_conv3x3_first_wgrad_reduce_f32(const float *dpre, const float *img, float *partial,
                                        int B, int Cin, int IH, int IW, int Cout, int stride,
                                        int n_layers, const float *const *rpartial,
                                        float *const *rdw, float *const *rdb, const int *rCout,
                                        const int *rCin, const int *rsplits, void *stream);
int scae_conv3x3_fwd_f32(const float *in, const float *wf, const float *bias, float *out,
                         const float *post_bias, float *out_post, int B, int IH, int IW,
                         int Cin, int Cout, int stride, void *stream);
int scae_conv3x3_dgrad_f32(const float *dpre, const float *wd, const float *gate, float *din,
                           int B, int IH, int IW, int Cin, int Cout, int stride, void *stream);
/* dgrad (gated by `in`, the layer's input = the producing layer's ReLU output)
 * and the weight-gradient partials (as scae_conv3x3_wgrad_f32 with dw == NULL)
 * of one layer in ONE launch: both only wait for dpre. */
int scae_conv3x3_bwd_pair_f32(const float *dpre, const float *wd, const float *in, float *din,
                              float *partial, int B, int IH, int IW, int Cin, int Cout,
                              int stride, void *stream);
/* scae_conv3x3_bwd_pair_f32 with the workgroups of scae_seed_fold_bwd_f32(fold, fold_grads) as
 * the head of its grid: the folding products' backward writes parameter gradients only, so in
 * a training step it may wait for this launch (256-thread workgroups, tens of microseconds
 * of MFMA tiles to hide behind).  Same results, bit for bit, as the launch of its own.
 * scae_conv3x3_bwd_pair_reduce_f32: the same for scae_seed_attention_mfma_reduce_f32
 * (arguments rpartial .. C), whose outputs only the folding products' backward reads.
 * SCAE_ERR_UNSUPPORTED (folding width != 256, another tile form of the pair): launch both. */
int scae_conv3x3_bwd_pair_reduce_f32(const float *dpre, const float *wd, const float *in,
                                     float *din, float *partial, int B, int IH, int IW, int Cin,
                                     int Cout, int stride, const float *rpartial, int rows,
                                     const float *q, const float *wk, float *gq, float *gwk,
                                     float *gbk, float *gwv, float *gbv, int O, int C,
                                     void *stream);
int scae_conv3x3_bwd_pair_fold_f32(const float *dpre, const float *wd, const float *in,
                                   float *din, float *partial, int B, int IH, int IW, int Cin,
                                   int Cout, int stride, const scae_seed_fold_desc *fold,
                                   const scae_seed_fold_grads *fold_grads, void *stream);
int scae_conv3x3_wgrad_splits(int B, int OH, int OW, int Cin, int Cout);
int scae_conv3x3_wgrad_f32(const float *dpre, const float *in, float *partial, float *dw,
                           float *db, int B, int IH, int IW, int Cin, int Cout, int stride,
                           void *stream);
int scae_conv3x3_wgrad_reduce_batch_f32(int n_layers, const float *const *partial,
                                        float *const *dw, float *const *db, const int *Cout,
                                        const int *Cin, const int *splits, void *stream);

/* ------------------------------------------------------------------------
 * K9  attention pooling of the part-capsule head
 *     replaces nn_ext.py:76-101 (multiple_soft_attention +
 *     multiple_attention_pooling_2d) as used by part_encoder.py:74.
 *   y (B,HW,A*P): NHWC output of the 1x1 attention conv (K7 GEMM); per capsule
 *   a the channels a*P .. a*P+P-2 are features, a*P+P-1 the attention logit.
 *   out (B,A,P-1) = sum_pix y[pix][a*P+p] * softmax_pix(logit)[pix]
 *   backward: g (B,A,P-1) -> dy (B,HW,A*P).
 * ---------------------------------------------------------------------- */
int scae_attention_pool_supported(int HW, int A, int P);
int scae_attention_pool_fwd_f32(const float *y, float *out, int B, int HW, int A, int P,
                                void *stream);
int scae_attention_pool_bwd_f32(const float *y, const float *g, float *dy, int B, int HW,
                                int A, int P, void *stream);
/* The same pooling with the rest of CapsuleImageEncoder.forward fused behind it
 * (part_encoder.py:75-92, n_poses = 6): the pooled row of capsule a is
 * [pose (6) | presence logit | special features (P-8)];
 *   pose (B,A,6)   = geometric_transform(pooled[..., :6], similarity) (K5 math)
 *   presence (B,A) = sigmoid(pooled[..., 6] + (noise_u - .5) * noise_scale)
 *   feature (B,A,P-8) (nullable when P == 8); noise_u (B,A) U[0,1) or NULL;
 *   absence (B,A) = 1 - presence (nullable; the set-transformer input of
 *   stacked_capsule_auto_encoder.py:113);
 *   pooled (B,A,P-1) is kept for the backward pass, whose incoming gradients
 *   g_pose / g_presence / g_feature may each be NULL (= zeros); g_feature2
 *   (nullable) is added to g_feature -- the feature feeds both the template
 *   colours and the set transformer (stacked_capsule_auto_encoder.py:103,:119),
 *   and summing here saves the caller an accumulate launch. */
int scae_capsule_head_fwd_f32(const float *y, const float *noise_u, float noise_scale,
                              int similarity, float *pooled, float *pose, float *presence,
                              float *feature, float *absence, int B, int HW, int A, int P,
                              void *stream);
/* The 1x1 attention conv (part_encoder.py:70-73) fused into the head's forward:
 * y (B,HW,A*P) = x (B,HW,C) w^T (A*P,C) + bias is computed slab by slab inside the
 * pooling workgroups (fp32 MFMA) and written to y for the backward; everything else
 * as scae_capsule_head_fwd_f32.  Supported for HW <= 32 and C a multiple of 64
 * and at most 256
 * (scae_capsule_head_conv_supported); x and w 16-byte aligned. */
int scae_capsule_head_conv_supported(int HW, int A, int P, int C);
/* ... and faster than GEMM + scae_capsule_head_fwd_f32: the fused form re-reads the
 * weights per image, so it pays for small batches only (see attention_pool.hip). */
int scae_capsule_head_conv_preferred(int B, int HW, int A, int P, int C);
int scae_capsule_head_conv_fwd_f32(const float *x, const float *w, const float *bias, int C,
                                   float *y, const float *noise_u, float noise_scale,
                                   int similarity, float *pooled, float *pose, float *presence,
                                   float *feature, float *absence, int B, int HW, int A, int P,
                                   void *stream);
/* scae_capsule_head_conv_fwd_f32 with scae_template_color_fwd_f32 (TemplateGenerator.forward,
 * part_decoder.py:78-110; arguments logits .. color_nonlin, M = A, its `feature` input is
 * this launch's `feature` output) behind it, workgroup by workgroup: the colour MLP of an
 * (image, capsule group) needs the special features of its own capsules only.
 * SCAE_ERR_UNSUPPORTED (F != P - 8, different grouping, one group per image): launch the two
 * in turn. */
int scae_capsule_head_conv_fwd_tc_f32(
    const float *x, const float *w, const float *bias, int C, float *y, const float *noise_u,
    float noise_scale, int similarity, float *pooled, float *pose, float *presence,
    float *feature, float *absence, int B, int HW, int A, int P, const float *logits,
    const float *w1, const float *b1, const float *w2, const float *b2, float *raw,
    float *templates, float *color, int Ct, int hw, int F, int H1, int template_nonlin,
    int color_nonlin, void *stream);
int scae_capsule_head_bwd_f32(const float *y, const float *pooled, const float *noise_u,
                              float noise_scale, int similarity, const float *g_pose,
                              const float *g_presence, const float *g_feature,
                              const float *g_feature2, float *dy, int B, int HW, int A, int P,
                              void *stream);
/* scae_capsule_head_bwd_f32 with scae_template_color_bwd_f32 (arguments logits ..
 * color_nonlin, M = A; tc_g_feature is both that kernel's g_feature output and this one's
 * g_feature2) in front of it, workgroup by workgroup: both run one workgroup per (image,
 * capsule group) over the same groups, and the head only needs the colour MLP's feature
 * gradient of its own capsules.  SCAE_ERR_UNSUPPORTED (F != P - 8, different grouping):
 * launch the two in turn. */
int scae_capsule_head_bwd_tc_f32(
    const float *y, const float *pooled, const float *noise_u, float noise_scale, int similarity,
    const float *g_pose, const float *g_presence, const float *g_feature, float *dy, int B,
    int HW, int A, int P, const float *logits, const float *feature, const float *w1,
    const float *b1, const float *w2, const float *b2, const float *color,
    const float *g_templates, const float *g_raw, float *g_logits, float *tc_g_feature,
    float *partial, int C, int hw, int F, int H1, int template_nonlin, int color_nonlin,
    void *stream);

/* ------------------------------------------------------------------------
 * Optimiser step on the flat parameter buffer
 *     replaces torch.optim.RMSprop(lr, momentum, eps) of
 *     base_experiment.py:44-77 (alpha = 0.99, weight_decay = 0, not centered):
 *       v <- alpha v + (1-alpha) g^2;  buf <- momentum buf + g/(sqrt(v)+eps);
 *       p <- p - lr buf        (momentum == 0: p <- p - lr g/(sqrt(v)+eps),
 *       buf may be NULL); g <- grad_scale * g + weight_decay p first (grad_scale:
 *       e.g. 1/world_size after a SUM all-reduce).  lr_dev (nullable):
 *       the learning rate in device memory, read instead of `lr` -- lets the
 *       per-epoch ExponentialLR schedule (:73-76) change it under graph replay.
 *       All buffers n floats, at the same offset within a 16-byte line
 *       (slices of identically laid out flat buffers).
 * ---------------------------------------------------------------------- */
int scae_rmsprop_step_f32(float *param, const float *grad, float *square_avg, float *buf,
                          int64_t n, float lr, const float *lr_dev, float alpha, float eps,
                          float momentum, float weight_decay, float grad_scale, void *stream);

/* A training step's last column sums (scae_sum_rows_multi_f32 over `jobs`, whose destinations
 * are parameter-gradient slots of `grad`) and the RMSprop pass above as ONE launch: the sum
 * workgroups apply the update of the elements they produce, the streaming workgroups skip
 * them; bit for bit the two launches' results.  No weight decay in this form. */
struct scae_sum_job;
int scae_rmsprop_sums_step_f32(float *param, float *grad, float *square_avg, float *buf,
                               int64_t n, float lr, const float *lr_dev, float alpha, float eps,
                               float momentum, float grad_scale, const struct scae_sum_job *jobs,
                               int n_jobs, void *stream);

/* The batch hand-over of a training step (base_experiment.py:109-112): n_image
 * floats and n_label int64 labels (device memory) into the step's resident
 * input buffers, in one launch. */
int scae_stage_batch(float *dst_image, const float *src_image, int64_t n_image,
                     int64_t *dst_label, const int64_t *src_label, int64_t n_label,
                     void *stream);

/* Prologue of a training step: the three jobs that depend on nothing but the
 * previous step's parameter update, as block ranges of ONE launch ahead of the
 * (replayed) step --
 *   the batch hand-over of scae_stage_batch (base_experiment.py:109-112),
 *   the presence-noise draws of scae_uniform_f32 (part_encoder.py:106,
 *   object_decoder.py:201: n_noise floats, same generator state, same values),
 *   the parameter-only folding products of scae_seed_fold_fwd_f32
 *   (set_transformer.py:218-223; `fold` as there).
 * A part is left out with n_image == 0 / n_noise == 0 / fold == NULL. */
int scae_step_prologue_f32(float *dst_image, const float *src_image, int64_t n_image,
                           int64_t *dst_label, const int64_t *src_label, int64_t n_label,
                           float *noise, int64_t n_noise, uint64_t *noise_state,
                           const scae_seed_fold_desc *fold, void *stream);

/* ... with a fourth job: the image layer of the CNN encoder and the filter re-layouts
 * of its other layers (the arguments of scae_conv3x3_first_fwd_relayout_f32;
 * part_encoder.py:26-44).  `img` is the batch wherever the caller holds it (the
 * hand-over's source or its destination: the layer does not wait for the copy).
 * first == NULL: scae_step_prologue_f32. */
typedef struct scae_first_layer_desc {
  const float *img;   /* (B, Cin, IH, IW) */
  const float *w;     /* (Cout, Cin, 3, 3) */
  const float *bias;  /* (Cout) */
  float *out;         /* (B, OH, OW, Cout) NHWC, ReLU applied */
  int B, Cin, IH, IW, Cout, stride;
  int n_layers;       /* re-layouts riding along (0..8) */
  const float *rw[8]; /* (rCout, rCin, 3, 3) each */
  float *rwf[8];      /* (rCout, 9, rCin) */
  float *rwd[8];      /* (rCin, 9, rCout) */
  int rCout[8], rCin[8];
  /* bf16-resident encoder (scae_conv3x3_*_bf16r): out_h non-NULL -> the layer's output is
   * written as bf16 to out_h INSTEAD of `out` (which must still be a valid pointer), and bf16
   * copies of the re-laid-out filters go to rwfh / rwdh */
  uint16_t *out_h;    /* (B, OH, OW, Cout) bf16, nullable */
  uint16_t *rwfh[8];  /* (rCout, 9, rCin) bf16 */
  uint16_t *rwdh[8];  /* (rCin, 9, rCout) bf16 */
} scae_first_layer_desc;
int scae_step_prologue_first_f32(float *dst_image, const float *src_image, int64_t n_image,
                                 int64_t *dst_label, const int64_t *src_label,
                                 int64_t n_label, float *noise, int64_t n_noise,
                                 uint64_t *noise_state, const scae_seed_fold_desc *fold,
                                 const scae_first_layer_desc *first, void *stream);

/* ------------------------------------------------------------------------
 * K10  coloured templates      replaces TemplateGenerator.forward,
 *      part_decoder.py:78-110 (colorize_templates = True):
 *        raw (M,C,hw)        = nonlin_t(logits)
 *        colour (B,M,C)      = nonlin_c'(relu(W2 relu(W1 feature + b1) + b2))
 *                              (nonlin_c' = sigmoid, or relu1(. + .99))
 *        templates (B,M,C,hw) = raw * colour
 *      nonlin codes: 0 sigmoid, 1 relu1 (nn_ext.py:139-140).
 *      w1 (H1,F), b1 (H1), w2 (C,H1), b2 (C); feature (B,M,F).
 *      backward: g_templates (B,M,C,hw), g_raw (M,C,hw, nullable) ->
 *      g_logits (M,C,hw), g_feature (B,M,F), partial
 *      (scae_template_color_partial_rows(B,M), H1*F + H1 + C*H1 + C) =
 *      per-workgroup [dW1 | db1 | dW2 | db2] for the caller to sum over dim 0.
 * ---------------------------------------------------------------------- */
int scae_template_color_supported(int M, int C, int F, int H1);
int scae_template_color_partial_rows(int B, int M);
int scae_template_color_fwd_f32(const float *logits, const float *feature, const float *w1,
                                const float *b1, const float *w2, const float *b2, float *raw,
                                float *templates, float *color, int B, int M, int C, int hw,
                                int F, int H1, int template_nonlin, int color_nonlin,
                                void *stream);
int scae_template_color_bwd_f32(const float *logits, const float *feature, const float *w1,
                                const float *b1, const float *w2, const float *b2,
                                const float *color, const float *g_templates,
                                const float *g_raw, float *g_logits, float *g_feature,
                                float *partial, int B, int M, int C, int hw, int F, int H1,
                                int template_nonlin, int color_nonlin, void *stream);

/* ------------------------------------------------------------------------
 * Column sums of a (rows, cols) matrix of partial gradients, written to up to
 * 8 contiguous destinations: column j in [begin, end) of segment i goes to
 * segments[i].dst[j - begin]; with period > 0 the window [begin, end) of
 * every period-wide block of columns is gathered instead:
 * dst[(j / period) * (end - begin) + j % period - begin] (cols must be a
 * multiple of period: SCAE_ERR_BAD_ARG otherwise).  period = -W < 0:
 * the window is an (n x W) matrix whose TRANSPOSE is written,
 * dst[((j - begin) % W) * n + (j - begin) / W] (an NHWC batch sum landing in
 * a (C,H,W) parameter).  Columns in no
 * segment are dropped; segments may overlap.  `segments` is a HOST array.  Replaces `partial.sum(0)` + per-parameter slice copies
 * behind the partial-gradient outputs of K1, K2b, K2c, K3, K8, K9 and K10.
 * ---------------------------------------------------------------------- */
typedef struct scae_sum_segment {
  float *dst;
  int64_t begin, end, period;
} scae_sum_segment;
int scae_sum_rows_f32(const float *src, int64_t rows, int64_t cols,
                      const scae_sum_segment *segments, int n_segments, void *stream);
/* Up to 16 such column-sum jobs (different matrices) in ONE launch: a backward
 * pass usually leaves two or three partial matrices behind at the same time,
 * and the sums that only feed the optimiser (parameter gradients) of a whole
 * training step can wait for one launch at its end.
 * `jobs` and the segment arrays it points to are HOST memory. */
typedef struct scae_sum_job {
  const float *src;
  int64_t rows, cols;
  const scae_sum_segment *segments;
  int n_segments;
} scae_sum_job;
int scae_sum_rows_multi_f32(const scae_sum_job *jobs, int n_jobs, void *stream);

/* Up to 8 scaled full sums in ONE launch: dst[i][0] = scale[i] * sum(src[i][0..n[i]))
 * (one workgroup each, fixed order).  The scalar outputs of the forward pass:
 * cpr_dynamic_reg_loss = sum(reg_partial)/2/B (object_decoder.py:170) and
 * log_prob = sum(log_prob_per_point)/B (:301-306). */
typedef struct scae_scaled_sum {
  const float *src;
  int64_t n;
  float scale;
  float *dst;
} scae_scaled_sum;
int scae_scaled_sums_f32(const scae_scaled_sum *jobs, int n_jobs, void *stream);

/* ------------------------------------------------------------------------
 * K3  capsule votes                  replaces object_decoder.py:160-225
 *     (+ cv_ops.py:20-76 on OPR/OVR, the batched 3x3 product :189-191)
 *   all_param (B,O,A), A = 6V+6+1+2V, the output of the per-capsule MLPs,
 *     split as [cpr_dynamic 6V | cvr 6 | caps logit 1 | vote logit V |
 *     scale V] (:160-162); consecutive capsule rows are ld_param >= A floats
 *     apart (0 = A; a multiple of 4 lets the MLP GEMMs on either side use
 *     16-byte accesses), the gradients of the backward pass likewise.
 *   cpr_static (O,V,6); bias_cvr (O,6); bias_caps (O); bias_vote (O,V);
 *   bias_scale (O,V)  (caps_bias_list, :108-111).
 *   noise_caps (B,O) / noise_vote (B,O,V): U[0,1) draws, nullable (no noise);
 *     the kernel adds (u-0.5)*noise_scale (:201).
 *   outputs: vote (B,O,V,6) = top two rows of OVR x OPR (:189-191, :413),
 *     scale (B,O,V) (:225), vote_presence (B,O,V) (:217-219),
 *     logit_caps (B,O), logit_vote (B,O,V) (noised logits, :211-212),
 *     reg_partial (B,O): per-(b,o) sum of cpr_dynamic^2 (caller: sum/2/B,
 *     :170); caps_presence (B,O) = max_v vote_presence with its first
 *     maximiser caps_arg (B,O) int32 (object_decoder.py:411; both nullable).
 * ---------------------------------------------------------------------- */
int scae_capsule_votes_fwd_f32(const float *all_param, const float *cpr_static,
                               const float *bias_cvr, const float *bias_caps,
                               const float *bias_vote, const float *bias_scale,
                               const float *noise_caps, const float *noise_vote,
                               float noise_scale, float *vote, float *scale,
                               float *vote_presence, float *logit_caps,
                               float *logit_vote, float *reg_partial,
                               float *caps_presence, int *caps_arg, int B, int O, int V,
                               int ld_param, int similarity, int learn_vote_scale,
                               int allow_deformations, void *stream);
/* incoming grads (all nullable = zero): gvote (B,O,V,6) gscale (B,O,V)
 * gvote_presence (B,O,V) glogit_caps (B,O) glogit_vote (B,O,V);
 * greg: d loss / d cpr_dynamic_reg_loss, a DEVICE scalar, nullable;
 * gcaps_presence (B,O) nullable, routed to vote caps_arg of each capsule.
 * outputs: gall_param (B,O,A); gcpr_in (B,O,V,6) = grad wrt
 * (cpr_dynamic + cpr_static) (caller sums over B for cpr_static; the bias
 * grads are the batch sums of the matching gall_param slices);
 * gall_param_gated (B,O,A), nullable: gall_param zeroed where all_param <= 0,
 * i.e. the gradient w.r.t. the pre-activation of the ReLU that produced
 * all_param (saves the caller a threshold_backward launch). */
int scae_capsule_votes_bwd_f32(const float *all_param, const float *cpr_static,
                               const float *bias_cvr, const float *bias_caps,
                               const float *bias_vote, const float *bias_scale,
                               const float *noise_caps, const float *noise_vote,
                               float noise_scale, const float *gvote,
                               const float *gscale, const float *gvote_presence,
                               const float *glogit_caps, const float *glogit_vote,
                               const float *greg, const float *gcaps_presence,
                               const int *caps_arg, float *gall_param, float *gcpr_in,
                               float *gall_param_gated, int B, int O, int V, int ld_param,
                               int similarity, int learn_vote_scale,
                               int allow_deformations, void *stream);

/* ------------------------------------------------------------------------
 * K4  capsule likelihood             replaces object_decoder.py:257-372
 *   vote (B,O,M,6) scale (B,O,M) vote_presence (B,O,M) dummy_vote (M,6)
 *   x (B,M,6) presence (B,M) nullable
 *   outputs: log_prob_per_point (B,M) (:296-300; caller: sum/B -> log_prob),
 *     vote_presence_binary (B,O,M), winner (B,M,6), winner_presence (B,M),
 *     winner_idx (B,M) int64 (:310), is_from_capsule (B,M) int64 (:334),
 *     soft_winner (B,M,6), soft_winner_presence (B,M),
 *     posterior (B,O+1,M) (softmax of :338 incl. the dummy row),
 *     mixing_log_prob (B,O+1,M), mixing_logit (B,O+1,M).
 * ---------------------------------------------------------------------- */
int scae_capsule_likelihood_fwd_f32(
    const float *vote, const float *scale, const float *vote_presence,
    const float *dummy_vote, const float *x, const float *presence,
    float *log_prob_per_point, float *vote_presence_binary, float *winner,
    float *winner_presence, int64_t *winner_idx, int64_t *is_from_capsule,
    float *soft_winner, float *soft_winner_presence, float *posterior,
    float *mixing_log_prob, float *mixing_logit, int B, int O, int M,
    void *stream);
/* incoming grads, each nullable: g_lpp (B,M), g_winner (B,M,6),
 * g_winner_presence (B,M), g_soft_winner (B,M,6), g_soft_winner_presence
 * (B,M), g_posterior (B,O+1,M), g_mixing_log_prob (B,O+1,M), g_mixing_logit
 * (B,O+1,M).  outputs: gvote, gscale, gvote_presence, gx (B,M,6),
 * gpresence (B,M) (nullable), gdummy_partial (B,M,6) (caller sums over B). */
int scae_capsule_likelihood_bwd_f32(
    const float *vote, const float *scale, const float *vote_presence,
    const float *dummy_vote, const float *x, const float *presence,
    const float *posterior, const int64_t *winner_idx, const float *g_lpp,
    const float *g_winner, const float *g_winner_presence,
    const float *g_soft_winner, const float *g_soft_winner_presence,
    const float *g_posterior, const float *g_mixing_log_prob,
    const float *g_mixing_logit, float *gvote, float *gscale,
    float *gvote_presence, float *gx, float *gpresence, float *gdummy_partial,
    int B, int O, int M, void *stream);

/* ------------------------------------------------------------------------
 * Class probabilities of SCAE.forward
 *     replaces stacked_capsule_auto_encoder.py:205-212:
 *     prior_prob (B,ncls) = softmax(w caps_presence + bias),
 *     post_prob  (B,ncls) = softmax(w sum_m posterior[:, :O, m] + bias)
 *     with caps_presence (B,O), posterior (B,O+1,M), w (ncls,O), bias (ncls)
 *     = prior_classifier.0 (the reference routes both through it).
 *     Limits: O <= 64, ncls <= 32.
 *     extra_sums (HOST array, n_extra <= 8, nullable): scaled full sums (see
 *     scae_scaled_sums_f32) that ride in the same launch -- this is the last
 *     kernel of SCAE.forward, and the forward's scalar outputs (log_prob,
 *     cpr_dynamic_reg_loss) are not read before it.
 * ---------------------------------------------------------------------- */
int scae_class_probs_supported(int O, int ncls);
int scae_class_probs_f32(const float *caps_presence, const float *posterior, const float *w,
                         const float *bias, float *prior_prob, float *post_prob, int B, int O,
                         int M, int ncls, const scae_scaled_sum *extra_sums, int n_extra,
                         void *stream);

/* ------------------------------------------------------------------------
 * K6  fused tail of SCAE.loss        replaces stacked_capsule_auto_encoder.py
 *     :238-285 + object_decoder.py:433-493 (+ math_ops.py) and their backward
 *   lpp (B,M) log_prob_per_point; posterior (B,O+1,M); caps_presence (B,O);
 *   cls_w (ncls,O), cls_b (ncls): prior_classifier.0 (both classification
 *   terms go through it, like the reference :207-212); label (B) int64
 *   nullable (no classification terms).
 *   prior_type / post_type: 0 'l2', 1 'entropy', 2 'kl'; sparsity_on: the
 *   reference's gate (prior weights > 0, :243/:258); weights5 (HOST array) =
 *   [caps_ll, prior_within, prior_between, posterior_within,
 *   posterior_between]; within_const: prior_within_example_constant or NaN
 *   (= n_caps/n_classes); n_classes_cfg: SCAE.n_classes (l2 constants).
 *   out12: [loss, log_prob, prior_within, prior_between, post_within,
 *   post_between, prior_cls_xe, posterior_cls_xe, rec_ll, -rec_ll, -log_prob,
 *   reg]; loss = -w0*log_prob + w1*pw + w2*pb + w3*qw + w4*qb + xe + xe
 *                - rec_ll + w_reg*reg.
 *   extras (nullable): the remaining terms of the training loss, so that the
 *   whole scalar (:217-287) leaves one kernel -- rec_sums (n_rec) = K1's tile
 *   sums of the reconstruction log-likelihood (rec_ll = sum / B, :222-224), reg
 *   (1) = cpr_dynamic_reg_loss with its weight w_reg (:270-272); g_rec_sums /
 *   g_reg receive their gradients in the backward pass.
 *   backward: gout12 (12) -> g_lpp, g_posterior, g_caps_presence, g_cls_w,
 *   g_cls_b (classifier inputs are detached in the reference).
 *   workspace: scae_loss_tail_workspace_floats(B,O,ncls) floats owned by the
 *   caller; the forward leaves the per-image / per-column statistics there and
 *   the backward of the same inputs reads them.
 * ---------------------------------------------------------------------- */
typedef struct scae_loss_extras {
  const float *rec_sums;
  int n_rec;
  const float *reg;
  float w_reg;
  float *g_rec_sums, *g_reg;
  float *loss;          /* forward: also receives out12[0] (a separate scalar) */
  const float *g_loss;  /* backward: gradient of that scalar, added to gout12[0];
                           gout12 may then be NULL (= zeros) */
  /* A training step in which nothing reads the forward's scalars before the backward has
   * run: with defer_combine the forward launches its per-image kernel only, and the batch
   * combine (out12, loss) is one more workgroup of the BACKWARD launch, which then needs
   * out12 (and loss) here.  scae_loss_tail_combine_f32 is that combine on its own, for a
   * deferred forward that no backward followed. */
  int defer_combine;
  float *out12;
} scae_loss_extras;
int scae_loss_tail_supported(int B, int O, int ncls);
int64_t scae_loss_tail_workspace_floats(int B, int O, int ncls);
int scae_loss_tail_fwd_f32(const float *lpp, const float *posterior,
                           const float *caps_presence, const float *cls_w,
                           const float *cls_b, const int64_t *label,
                           const scae_loss_extras *extras, float *out12, float *workspace,
                           int B, int O, int M, int ncls, int n_classes_cfg, int prior_type,
                           int post_type, int sparsity_on, const float *weights5,
                           float within_const, void *stream);
/* defer_combine costs B^2 2 O extra loads in the backward: worth it for small batches only */
int scae_loss_tail_defer_preferred(int B, int O);
int scae_loss_tail_combine_f32(const float *lpp, const float *posterior,
                               const float *caps_presence, const float *cls_w,
                               const float *cls_b, const int64_t *label,
                               const scae_loss_extras *extras, float *out12, float *workspace,
                               int B, int O, int M, int ncls, int n_classes_cfg, int prior_type,
                               int post_type, int sparsity_on, const float *weights5,
                               float within_const, void *stream);
/* scae_loss_tail_fwd_f32 with the workgroups of scae_class_probs_f32 (arguments cp_* ..
 * n_extra, same meaning) riding in its per-image launch: both are one wave per image and
 * independent, and in a training step nothing reads the class probabilities in between. */
int scae_loss_tail_fwd_class_probs_f32(
    const float *lpp, const float *posterior, const float *caps_presence, const float *cls_w,
    const float *cls_b, const int64_t *label, const scae_loss_extras *extras, float *out12,
    float *workspace, int B, int O, int M, int ncls, int n_classes_cfg, int prior_type,
    int post_type, int sparsity_on, const float *weights5, float within_const,
    const float *cp_caps_presence, const float *cp_posterior, const float *cp_w,
    const float *cp_bias, float *prior_prob, float *post_prob, int cp_B, int cp_O, int cp_M,
    int cp_ncls, const scae_scaled_sum *extra_sums, int n_extra, void *stream);
int scae_loss_tail_bwd_f32(const float *lpp, const float *posterior,
                           const float *caps_presence, const float *cls_w,
                           const float *cls_b, const int64_t *label,
                           const scae_loss_extras *extras, const float *gout12,
                           const float *workspace, float *g_lpp, float *g_posterior,
                           float *g_caps_presence, float *g_cls_w,
                           float *g_cls_b, int B, int O, int M, int ncls, int n_classes_cfg,
                           int prior_type, int post_type, int sparsity_on,
                           const float *weights5, float within_const, void *stream);

/* ------------------------------------------------------------------------
 * K1  template render + Gaussian-mixture image likelihood
 *     replaces part_decoder.py:174-237 (affine_grid + 2x grid_sample +
 *     background + presence), distributions.py:34-47 (mixture log_prob) and
 *     the rec term of stacked_capsule_auto_encoder.py:220.
 *
 * Decoder description shared by the K1 entry points:
 *   templates (B,M,C,th,tw); pose (B,M,6) = row-major 2x3 affine;
 *   presence (B,M) nullable;
 *   templates_alpha (M,th,tw) -- non-null selects the alpha-channel mode
 *     (mixing logits have ONE channel, Cm = 1), null selects the temperature
 *     mode (logits = transformed/temperature, Cm = C) and then
 *     temperature_logit (1) must be given;
 *   bg_image (B,C,H,W) nullable; when null bg_value (1) must be given;
 *   bg_mixing_logit (1) (alpha mode only);
 *   out_scale (1) nullable: sigma = softplus(out_scale)+1e-4, else sigma = 1.
 *   K = M+1 mixture components, the last one is the background.
 * ---------------------------------------------------------------------- */
typedef struct scae_decoder_desc {
  const float *templates;
  const float *templates_alpha;
  const float *pose;
  const float *presence;
  const float *bg_image;
  const float *bg_value;
  const float *bg_mixing_logit;
  const float *temperature_logit;
  const float *out_scale;
  int B, M, C, th, tw, H, W;
  int template_repeat; /* > 1: templates is (B / template_repeat, M, C, th, tw) and images
                          r*k .. r*k + r-1 share template set k (forward only) */
  int bwd_resident;    /* > 0 (scae_render_gmm_sums_bwd_f32 / _bwd_f32, alpha mode): the
                          likelihood backward runs from this many resident workgroups, each
                          walking (component, image) pairs, instead of one workgroup per
                          pair -- same results; a launch that shares the chip with another
                          stream's kernels.  0: one workgroup per pair */
} scae_decoder_desc;

/* materialise transformed_templates (B,K,C,H,W) and mixing_logits
 * (B,K,Cm,H,W): part_decoder.py:174-231. */
int scae_template_render_fwd_f32(const scae_decoder_desc *d,
                                 float *transformed_templates,
                                 float *mixing_logits, void *stream);

/* fused path: log_prob (B,C,H,W) of image x (B,C,H,W) under the mixture,
 * straight from the compact decoder inputs (nothing (B,K,..)-sized touches
 * HBM).  lse_post (B,C,H,W) and lse_prior (B,Cm,H,W) are saved for the
 * backward pass; log_prob = lse_post - lse_prior. */
int scae_render_gmm_logprob_fwd_f32(const scae_decoder_desc *d, const float *x,
                                    float *log_prob, float *lse_post,
                                    float *lse_prior, void *stream);

/* backward of either path.
 *   fused (g_tt == NULL): g_logprob (B,C,H,W) with x / lse_post / lse_prior of
 *     the forward;
 *   materialised (g_tt != NULL): g_tt (B,K,C,H,W), g_ml (B,K,Cm,H,W) (either
 *     may be NULL = zero, not both) -- x, lse_*, g_logprob ignored.
 * outputs: g_templates (B,M,C,th,tw), g_alpha_partial (B,M,th,tw) (alpha mode;
 *   caller sums over B), g_pose (B,M,6), g_presence (B,M) (nullable),
 *   g_bg_image (B,C,H,W) (nullable),
 *   g_scalar_partial (B,K,4): per-(b,k) partial grads of [bg_value,
 *   bg_mixing_logit, temperature_logit, out_scale] with the sigmoid/softplus
 *   chain already applied (caller sums over (B,K)). */
int scae_render_gmm_bwd_f32(const scae_decoder_desc *d, const float *x,
                            const float *lse_post, const float *lse_prior,
                            const float *g_logprob, const float *g_tt,
                            const float *g_ml, float *g_templates,
                            float *g_alpha_partial, float *g_pose,
                            float *g_presence, float *g_bg_image,
                            float *g_scalar_partial, void *stream);
/* Tile-sum variant for the training loss (stacked_capsule_auto_encoder.py:222-224
 * only ever needs rec_ll = mean_b sum_{c,h,w} log_prob): the forward emits one
 * partial sum per (image, pixel tile) -- tile_sums (B, scae_render_gmm_logprob_tiles(d)),
 * summed in a fixed order -- instead of the per-pixel map, and the backward
 * takes the gradient of those sums (B, tiles). */
int scae_render_gmm_logprob_tiles(const scae_decoder_desc *d);
int scae_render_gmm_logprob_sums_fwd_f32(const scae_decoder_desc *d, const float *x,
                                         float *tile_sums, float *lse_post, float *lse_prior,
                                         void *stream);
int scae_render_gmm_sums_bwd_f32(const scae_decoder_desc *d, const float *x,
                                 const float *lse_post, const float *lse_prior,
                                 const float *g_tile_sums, float *g_templates,
                                 float *g_alpha_partial, float *g_pose, float *g_presence,
                                 float *g_bg_image, float *g_scalar_partial, void *stream);


/* scae_set_encoder_fwd_f32 and scae_render_gmm_logprob_sums_fwd_f32 as ONE launch:
 * the two forward kernels are independent -- both only read the part encoder's outputs
 * (stacked_capsule_auto_encoder.py:105-124 and :146-162 / :220) -- and the trunk leaves three
 * quarters of the SIMDs idle, so the likelihood's workgroups ride as a second block range of
 * its launch.  Falls back to the two launches where the shapes do not fit; identical results. */
int scae_set_encoder_fwd_logprob_f32(int nseg, const float *const *seg_ptr,
                                     const int *seg_width, const int *seg_row_stride,
                                     const int64_t *seg_batch_stride, const float *presence,
                                     const float *params, float *z, float *hsave, int B, int N,
                                     int D, int Din, int Dout, int L, int layer_norm,
                                     const scae_decoder_desc *d, const float *x,
                                     float *tile_sums, float *lse_post, float *lse_prior,
                                     void *stream);

/* ... with the bf16 attention products of scae_set_encoder_fwd_bf16 in the trunk */
int scae_set_encoder_fwd_logprob_bf16(int nseg, const float *const *seg_ptr,
                                      const int *seg_width, const int *seg_row_stride,
                                      const int64_t *seg_batch_stride, const float *presence,
                                      const float *params, float *z, float *hsave, int B, int N,
                                      int D, int Din, int Dout, int L, int layer_norm,
                                      const scae_decoder_desc *d, const float *x,
                                      float *tile_sums, float *lse_post, float *lse_prior,
                                      void *stream);

/* scae_render_gmm_sums_bwd_f32 and scae_capsule_likelihood_bwd_f32 as ONE launch: both
 * backward kernels only wait for the loss tail's (stacked_capsule_auto_encoder.py:217-287)
 * and are independent of each other; the capsule likelihood's one-workgroup-per-image kernel
 * rides as the first block range of the reconstruction likelihood's launch.  `k` holds the
 * arguments of scae_capsule_likelihood_bwd_f32 (same names).  Two launches where the shapes do
 * not fit; identical results. */
typedef struct scae_likelihood_bwd_desc {
  const float *vote, *scale, *vote_presence, *dummy_vote, *x, *presence, *posterior;
  const int64_t *winner_idx;
  const float *g_lpp, *g_winner, *g_winner_presence, *g_soft_winner, *g_soft_winner_presence,
      *g_posterior, *g_mixing_log_prob, *g_mixing_logit;
  float *gvote, *gscale, *gvote_presence, *gx, *gpresence, *gdummy_partial;
  int B, O, M;
} scae_likelihood_bwd_desc;
int scae_render_gmm_sums_bwd_likelihood_f32(
    const scae_decoder_desc *d, const float *x, const float *lse_post, const float *lse_prior,
    const float *g_tile_sums, float *g_templates, float *g_alpha_partial, float *g_pose,
    float *g_presence, float *g_bg_image, float *g_scalar_partial,
    const scae_likelihood_bwd_desc *k, void *stream);

/* generic mixture over materialised tensors: distributions.py:34-47.
 *   loc (B,K,C,P), mixing_logits (B,K,Cm,P) with Cm in {1,C}, sigma (1) device
 *   scalar (the Normal's scale), x (B,C,P) -> log_prob (B,C,P). */
int scae_gmm_log_prob_fwd_f32(const float *loc, const float *mixing_logits,
                              const float *sigma, const float *x,
                              float *log_prob, int B, int K, int C, int Cm,
                              int64_t P, void *stream);
/* g_loc (B,K,C,P), g_ml (B,K,Cm,P), g_sigma_partial (B) (caller sums),
 * g_x (B,C,P) nullable. */
int scae_gmm_log_prob_bwd_f32(const float *loc, const float *mixing_logits,
                              const float *sigma, const float *x,
                              const float *g_logprob, float *g_loc,
                              float *g_ml, float *g_sigma_partial, float *g_x,
                              int B, int K, int C, int Cm, int64_t P,
                              void *stream);
/* mean (distributions.py:37-39) and mode (:50-77, one-hot argmax over K of
 * log_softmax(logits) [+ component log-prob at its own mean when `maximum`])
 * -> out (B,C,P).  mode: Cm == 1 with C > 1 and maximum is rejected like the
 * reference (its in-place add cannot broadcast). */
int scae_gmm_mean_f32(const float *loc, const float *mixing_logits, float *out,
                      int B, int K, int C, int Cm, int64_t P, void *stream);
int scae_gmm_mode_f32(const float *loc, const float *mixing_logits,
                      const float *sigma, float *out, int maximum, int B,
                      int K, int C, int Cm, int64_t P, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAE_HIP_H */
