/*
 * disgat_hip.h - C ABI of libdisgat_hip.so: the MI355X (gfx950) implementation of the
 * DISGAT edge-disentangled message-passing hot path.
 *
 * Every entry point is a stateless launcher: plain device pointers and sizes, no
 * framework types, the caller owns every buffer, work is enqueued on the given HIP
 * stream and the call returns without synchronising.  Return value: 0 = launched,
 * >0 = hipError_t from the runtime, <0 = argument outside the kernel envelope
 * (disgat_last_error() gives the text).  The Python host (edgedisentangle_ssl_amd/_lib.py)
 * binds these with ctypes and raises RuntimeError on a non-zero return.
 *
 * What each launcher replaces in the reference (/root/reference, Python/ATen):
 *
 *  disgat_edge_fwd      DisGALayer.forward_sparse for ALL heads of one layer at once:
 *                       per-edge score (layers.py:349-379), sigmoid (layers.py:392),
 *                       utils.sp_softmax (utils.py:192-200) and the weighted neighbour
 *                       aggregation utils.sp_matmul (utils.py:203-207) used by the AT /
 *                       SAGE / GCN branches (layers.py:397-407, 96-110, 38-54).
 *  disgat_edge_combine  second pass for rows that were split across work items (no
 *                       reference counterpart: load balancing for power-law rows).
 *  disgat_aux_score     the auxiliary node-pair scoring of predict_adjs_sparse
 *                       (layers.py:355-360, 368-372, 381-389; models.py:290-330).
 *  disgat_bwd_alpha,    autograd of the above: the reference relies on ATen autograd through
 *  disgat_seg_grad_*    index / cat / mm / scatter_add_ (loss.backward() at pretrainer.py:752, 631,
 *                       836; trainer.py:200).  Gather-only segment passes, see csrc/edge_bwd.hip.
 *  disgat_pair_loss     sigmoid(sum of heads) + utils.adj_mse_loss partial sums
 *                       (pretrainer.py:727-739, 612-627; utils.py:287-298).
 *  disgat_cls_loss      F.log_softmax + F.nll_loss + utils.accuracy on the train and validation splits
 *                       (trainer.py:186-199) and DifHead's NLL against the head index (pretrainer.py:819-832).
 *  disgat_gemm_f16x3,   the dense contractions of the path - torch.mm / nn.Linear at layers.py:350, 363, 376,
 *  disgat_gemm_split,   398, 110, 39, 905 and models.py:538 (ATen fp32 GEMM) - as fp32-accurate GEMMs on the
 *  disgat_split_f16,    16-bit matrix cores (operand splitting), with the bias / additive / ELU (layers.py:508)
 *  disgat_amax,         / leaky-ReLU (layers.py:917, models.py:535) epilogues fused; weight preparation, the
 *  disgat_act_bwd       scale input and the activation's backward.
 *  disgat_pair_sample_* the training-pair samplers of SupEdge / DisEdge (pretrainer.py:683-707, 524-576).
 *  disgat_adam_multi    the per-sub-module torch.optim.Adam steps of a trainer (trainer.py:58-60, 205-206) as one launch.
 *
 * Layouts (all row-major fp32 unless noted; "ld*" = row stride in floats, a multiple of 4,
 * base pointers 16-byte aligned):
 *   items   int32 [n_items][4] = {row, e_begin, e_end, slot}; slot = -1 for a whole row,
 *           else the index of this chunk's partial record (split rows)
 *   col     int32 [E]   CSR column (= the reference's indices[1], "target": the gathered node)
 *   x       [N][ldx]    layer input, F_in valid columns (F_in % 4 == 0 after padding)
 *   Z       [N][H][F_in] attention-weighted neighbour sums  sum_k alpha_k * x[col_k]
 *   edge_e  [H][E]      raw pre-sigmoid scores in CSR (= coalesced, row-major) edge order
 *   den     [N][2][H]   [.][0][h] softmax denominator sum_k exp(sigmoid(e_k)); [.][1][h] the same sum
 *                       over the edges attention-dropout kept, scaled 1/(1-p) (== [0] when p = 0)
 *   part_z / part_den   [n_slots][H][F_in] / [n_slots][2][H] partial records of split rows
 *   drop_p, drop_seed   attention dropout (layers.py:394): edge k, head h is kept iff the top 32 bits
 *                       of splitmix64(seed + (k*H+h)*0x9E3779B97F4A7C15) >= p*2^32; p = 0 disables it
 *   drop_seed_dev       NULL, or a device uint64 added to drop_seed when the kernel runs: a step captured in a HIP
 *                       graph bakes drop_seed in, the graph advances this counter itself between replays
 *   sign_bits           att 3 only, optional (NULL = not recorded): uint32 [E | M][64].  Word l of a row belongs
 *                       to lane l = (head h = l / G, g = l % G), G = 64/H, QN = F_out / (4*G); its bit
 *                       4*(k + 4*(j&1)) + QH-1-(j>>1), QH = max(1, QN/2), is (P[row] + Q[col] > 0) at feature
 *                       h*F_out + (j*G+g)*4 + k (k < 4, j < QN): which slope leaky_relu took.  disgat_seg_grad_sign computes the score's
 *                       backward from it without gathering any operand row.
 *   H must be a power of two (the host pads missing heads with zero weights).
 *
 * att (the reference's --att / att_type):
 *   1  e = s1[row][h] + s2[col][h]                     rowop = s1 [N][H], colop = s2 [N][H]
 *   2  e = <P[row][h][:], x[col][:]>                   rowop = P [N][H*F_in]  (P = x W W^T)
 *   3  e = sum_f a[h][f]*lrelu_0.01(P[row][h][f] + Q[col][h][f])
 *                                                      rowop = P, colop = Q [N][H*F_out], a [H*F_out]
 *      F_out must equal QN*(64/H)*4 with QN in {1,2,4,8} (host pads with zero columns).
 *   4  (not a reference flag value) att 2 as layers.py:362-365 writes it, for inputs wider than att 2's 512-column
 *      register tile: e = <h[row][h][:], h[col][h][:]>, h = x W per head - rowop = colop = h [N][H*F_out] in the
 *      layout and padding of att 3, a = NULL, no nonlinearity.  Accepted by disgat_edge_fwd and disgat_aux_score;
 *      disgat_seg_grad_att3 with a == NULL is its score backward.
 */
#ifndef DISGAT_HIP_H
#define DISGAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* disgat_stream_t; /* hipStream_t */

int disgat_abi_version(void);          /* 10 in this revision; changes with any launcher's argument list */
const char* disgat_build_flags(void);  /* the -D flags of this build ("" = plain): the host refuses a diagnostic build it did not ask for */
const char* disgat_last_error(void);

/* Fused score -> sigmoid -> row softmax -> aggregation for all H heads of one layer.
 * e_in (or NULL): [H][E] partial scores added before the sigmoid - a head wider than one launch's 1024 features is
 * scored in feature slices (disgat_aux_score on the edge list for all but the last slice); edge_e receives the total.
 * Z_hi / Z_lo / z_bound (or all NULL): write the aggregate as the two fp16 planes of disgat_gemm_planes' A operand
 * ([N][H][F_in] halfs each, hi and lo of Z * s with s from *z_bound >= max |Z|) INSTEAD of fp32 Z (which may then be
 * NULL): the per-head projection that consumes Z then splits nothing. */
int disgat_edge_fwd(int att, const int32_t* items, int n_items, const int32_t* col, int64_t E,
                    int N, int H, int F_in, int F_out,
                    const float* x, int ldx,
                    const float* rowop, int ld_row,
                    const float* colop, int ld_col,
                    const float* a,
                    float* Z, float* edge_e, float* den,
                    float* part_z, float* part_den,
                    int sage_div, float drop_p, uint64_t drop_seed, const uint64_t* drop_seed_dev,
                    uint32_t* sign_bits, const float* e_in,
                    uint16_t* Z_hi, uint16_t* Z_lo, const float* z_bound, disgat_stream_t stream);

/* Sums the partial records of split rows (in chunk order: deterministic) and normalises. */
int disgat_edge_combine(const int32_t* split_rows, const int32_t* split_ptr, int n_split,
                        int H, int F_in, const float* part_z, const float* part_den,
                        float* Z, float* den, int sage_div,
                        uint16_t* Z_hi, uint16_t* Z_lo, const float* z_bound, disgat_stream_t stream);

/* Raw scores of M arbitrary node pairs, heads [h_lo, h_hi) only; out is [H][M]. */
int disgat_aux_score(int att, const int64_t* pair_rows, const int64_t* pair_cols, int64_t M,
                     int N, int H, int F_in, int F_out, int h_lo, int h_hi,
                     const float* x, int ldx,
                     const float* rowop, int ld_row,
                     const float* colop, int ld_col,
                     const float* a,
                     float* out, uint32_t* sign_bits, disgat_stream_t stream);

/* Weighted squared-error sums of pred = sigmoid(sum_{h in [h_lo,h_hi)} aux[h][m]) against 0/1 labels:
 * acc[0] = sum over positives, acc[1] = sum over zeros, acc[2] = #positives (3 doubles, WRITTEN - no zeroing needed).
 * A NEGATIVE label marks padding (a fixed-capacity list whose valid length lives on the device, as a step captured
 * in a HIP graph needs): such an entry enters no sum and gets a zero gradient from disgat_pair_loss_bwd.
 * block_partials: scratch for DISGAT_PAIR_LOSS_MAX_BLOCKS x 3 doubles (per-block sums, added in block order by a second
 * tiny launch: the value is run-to-run deterministic).
 * value (optional, 3 doubles) = {loss, neg_w, m}: utils.adj_mse_loss of the whole list (utils.py:287-298) - m = *count (a
 * device double: the valid length of a fixed-capacity list) or M when count is NULL, neg_w = acc[2] / (m^2 - acc[2]),
 * loss = (acc[0] + neg_w acc[1]) / m; loss32 (optional, needs value) = (float)loss.  A caller that must combine several
 * processes' sums first passes value = NULL and finishes on its side. */
#define DISGAT_PAIR_LOSS_MAX_BLOCKS 2048
int disgat_pair_loss(const float* aux, int64_t M, int h_lo, int h_hi, const float* labels, const double* count,
                     double* acc, double* block_partials, double* value, float* loss32, disgat_stream_t stream);

/* Classification loss in one pass over logits [n_rows][n_cls] (row stride ld): per row log_softmax (written to logp when
 * non-NULL, row stride ld_logp), and over the rows of split s = 0, 1 the sums of -logp[label] and of (argmax == label).
 *   row_code    int32 [n_rows]: -1 = the row is in no split, else label + (split << 16); NULL: every row is in split 0
 *               with label (row % label_mod) - DifHead's (node, head) rows
 *   div0, div1  the divisors of split 0 / 1 (the GLOBAL split sizes: on a row shard the caller adds the ranks' results)
 *   loss        float [1]  = NLL sum of split 0 / div0, the value the backward differentiates
 *   res         double [4] = {NLL_0 / div0, correct_0 / div0, NLL_1 / div1, correct_1 / div1}
 *   block_partials  scratch of DISGAT_CLS_LOSS_MAX_BLOCKS x 4 doubles, needed above DISGAT_CLS_LOSS_ONE_BLOCK rows (per-block
 *               sums added in block order by a second tiny launch; up to that size one block does everything in one launch)
 * Sums are doubles taken in a fixed order: run-to-run deterministic. */
#define DISGAT_CLS_LOSS_MAX_BLOCKS 1024
#define DISGAT_CLS_LOSS_ONE_BLOCK 8192
int disgat_cls_loss(const float* logits, int64_t ld, const int32_t* row_code, int label_mod, int64_t n_rows, int n_cls,
                    double div0, double div1, float* logp, int64_t ld_logp, double* block_partials, float* loss,
                    double* res, disgat_stream_t stream);

/* ---- backward -------------------------------------------------------------------------- */

/* Gradient of disgat_cls_loss's `loss` w.r.t. the logits from the saved logp: rows of split 0 get
 * (exp(logp) - onehot(label)) * g[0] / div0, every other row zeros (g: the upstream gradient, one device float). */
int disgat_cls_loss_bwd(const float* logp, int64_t ld_logp, const int32_t* row_code, int label_mod, int64_t n_rows,
                        int n_cls, const float* g, double div0, float* grad_logits, int64_t ld_grad, disgat_stream_t stream);

/* Gradient of disgat_pair_loss's loss w.r.t. the raw scores: g[h][m] = c[labels[m] != 0 ? 0 : 1] * 2 (p - t) p (1 - p) for h
 * in [h_lo, h_hi), 0 for the other of the H rows.  c = coef (2 device floats: upstream gradient x class weight / m) or, with
 * coef == NULL, {gout / m, neg_w gout / m} from the forward's value = {loss, neg_w, m} and the upstream gradient gout (one
 * device float).  g_transposed != 0: g is written as [M][H] (g[m][h]) - the layout in which disgat_seg_grad_sign's
 * column-side pass, which visits the pairs in column order, finds the H gradients of a pair in one 32-byte run. */
int disgat_pair_loss_bwd(const float* aux, int64_t M, int H, int h_lo, int h_hi, const float* labels, const float* coef,
                         const double* value, const float* gout, float* g, int g_transposed, disgat_stream_t stream);

/* Per edge and head: ge_out = ge_in + d(loss)/d(e) through softmax-of-sigmoid given gZ (grad of Z),
 * and beta = alpha*sc, the coefficient of x[col] in Z (used by the transposed pass for grad x).
 * ge_in may be NULL.  Items/col as in disgat_edge_fwd; Z, den, edge_e are the forward's outputs.
 * ge_transposed != 0: ge_out is written as [E][H] (see disgat_pair_loss_bwd); ge_in and beta stay [H][E]. */
int disgat_bwd_alpha(const int32_t* items, int n_items, const int32_t* col, int64_t E, int H, int F_in,
                     const float* x, int ldx, const float* gZ, const float* Z, const float* edge_e,
                     const float* den, const float* ge_in, float* ge_out, int ge_transposed, float* beta, int sage_div,
                     float drop_p, uint64_t drop_seed, const uint64_t* drop_seed_dev, disgat_stream_t stream);

/* Segment gradient of the att-3 score e = sum_f a_f lrelu(keyop[key] + otherop[other]).
 * items = {key, m_begin, m_end, slot} over a list sorted by key; other[m] = gathered node of list
 * position m; perm[m] (or NULL = identity) = column of g holding that position's upstream grad.
 * gkey[key] = sum_m g*a*lrelu'(z) (plain store, or atomic add into host-zeroed rows when slot >= 0);
 * ga_part (or NULL): [n_waves][H*F_out] per-wave partial sums of g*lrelu(z) (sum them on the host).
 * n_waves: multiple of 4; the launch is persistent (grid-stride over items).
 * In all segment launchers an item with key < 0 is padding and does nothing, and so does a negative entry of
 * disgat_seg_combine's split_keys: item tables of FIXED size for steps captured in a HIP graph. */
int disgat_seg_grad_att3(const int32_t* items, int n_items, const int32_t* other, const int32_t* perm,
                         const float* g, int64_t g_stride, int h_lo, int h_hi, int H, int F_out,
                         const float* keyop, int ld_key, const float* otherop, int ld_other, const float* a,
                         float* gkey, int ld_gkey, float* ga_part, int n_waves, float* part, disgat_stream_t stream);

/* The same gradient from the forward's sign record instead of a second operand gather (layout: sign_bits
 * above; row perm[m] of g and sign_bits belongs to list position m, NULL = identity):
 *   u[key] = sum_m g_m * (bit ? 1 : 0.01),  gkey[key] = a (.) u[key]  (atomic add into host-zeroed rows when
 *   slot >= 0),  ga_part[wave] = per-wave partial of sum_key keyop[key] (.) u[key]  (NULL: not wanted).
 * The a-gradient of a list is the sum of the row-side pass (keyop = P) and the column-side pass (keyop = Q),
 * because lrelu(z) = lrelu'(z) * (P + Q).  Reads 64 words + H floats per list position.
 * accumulate = 1 adds into gkey instead (several lists scoring against the same operands share one gradient buffer;
 * columns of heads outside [h_lo, h_hi) are then left untouched).
 * g: head h of list position p at g[h * g_stride + p * g_pos_stride] ([H][M]: strides (M, 1); [M][H]: (1, H)).
 * amax_out (or NULL): a device float the caller zeroed, raised to max |value stored into gkey| over whole keys (split keys:
 * disgat_seg_combine's amax_out) - the scale input of the f16x3 GEMMs that consume gkey, measured without another pass.
 * a == NULL: gkey receives u itself.  With keyop = x W the operand's producer then needs no operand row at all:
 * G = x^T u is the weight-gradient GEMM that runs anyway, grad W = G (.) a (per column), grad a = sum_rows W (.) G,
 * grad x = u (W (.) a)^T. */
int disgat_seg_grad_sign(const int32_t* items, int n_items, const int32_t* perm, const float* g,
                         int64_t g_stride, int64_t g_pos_stride, int h_lo, int h_hi, int H, int F_out,
                         const uint32_t* sign_bits,
                         const float* keyop, int ld_key, const float* a, float* gkey, int ld_gkey,
                         float* ga_part, int n_waves, int accumulate, float* part, float* amax_out,
                         disgat_stream_t stream);

/* col_mode = 0: gkey[key][h][:] = sum_m coef[h][perm(m)] * otherop[other_m][:]          (F floats per row)
 * col_mode = 1: gkey[key][:] (+)= sum_m sum_h coef[h][perm(m)] * otherop[other_m][h][:]  (otherop rows H*F) */
int disgat_seg_grad_hx(int col_mode, const int32_t* items, int n_items, const int32_t* other,
                       const int32_t* perm, const float* coef, int64_t coef_stride, int h_lo, int h_hi, int H,
                       int F, const float* otherop, int ld_other, float* gkey, int ld_gkey, int accumulate,
                       float* part, disgat_stream_t stream);

/* att 1 (e = s1[row][h] + s2[col][h], layers.py:349-353): the gradient of a score operand is the segment sum of the score
 * gradients, gkey[key][h] (+)= sum over the key's list positions m of g[h * g_stride + perm(m) * g_pos_stride] for h in
 * [h_lo, h_hi) (0 for the other heads).  Items / perm / part / accumulate as in the launchers above; ld_gkey = H rounded up to
 * a multiple of 4 (columns H .. ld_gkey-1 are written as zeros).  Fixed summation order: no float atomics. */
int disgat_seg_sum(const int32_t* items, int n_items, const int32_t* perm, const float* g, int64_t g_stride,
                   int64_t g_pos_stride, int h_lo, int h_hi, int H, float* gkey, int ld_gkey, int accumulate,
                   float* part, disgat_stream_t stream);

/* Split keys (hub rows / columns cut into several work items, slot >= 0): when `part` ([n_slots][ld_gkey] floats) is
 * given to the three segment launchers above, every slice stores its partial result in part[slot] instead of adding
 * to gkey with float atomics, and this launcher then forms gkey[key][0:width] (+)= sum of the key's slices in slice
 * order - run-to-run deterministic gradients.  split_keys [n_split], split_ptr [n_split+1] as in disgat_edge_combine.
 * part == NULL in the launchers above keeps the atomic path (gkey rows of split keys must then be zeroed by the host).
 * amax_out (or NULL): raised to max |value stored| as in disgat_seg_grad_sign. */
int disgat_seg_combine(const int32_t* split_keys, const int32_t* split_ptr, int n_split, int width,
                       const float* part, float* gkey, int ld_gkey, int accumulate, float* amax_out,
                       disgat_stream_t stream);

/* Work-item tables of the three segment launchers above for a key-sorted list of FIXED capacity C (a pair list of a step
 * captured in a HIP graph: nothing is read back, every table has a fixed shape, unused tails are padding).
 *   keys [C] int64 (node ids < n_keys), sorted - or sorted through perm [C] int64 (the sort's indices: list position m holds
 *   keys[perm[m]]); perm32 [C] then receives perm as int32 (both NULL for a sorted list).
 *   ptr [n_keys + 1], key_off [2 n_keys]: scratch (ptr[k] = first list position of key k is also the CSR pointer of the list).
 *   items [cap_items][4], cap_items >= n_keys + C / chunk: {key, m_begin, m_end, slot} in key order, a key longer than chunk
 *   cut into ceil(deg / chunk) near-equal slices with consecutive slots; {-1, 0, 0, -1} past the last item.
 *   split_rows [n_split_cap], split_ptr [n_split_cap + 1], n_split_cap >= max(1, C / chunk): disgat_seg_combine's tables
 *   (-1 / total padded).  totals [4] (device): items, split keys, slots in use. */
int disgat_seg_tables(const int64_t* keys, const int64_t* perm, int64_t C, int n_keys, int chunk, int32_t* ptr,
                      int32_t* key_off, int32_t* perm32, int32_t* items, int cap_items, int32_t* split_rows,
                      int32_t* split_ptr, int n_split_cap, int32_t* totals, disgat_stream_t stream);

/* ---- dense contractions --------------------------------------------------------------------- */

/* fp32-accurate GEMM on the bf16 matrix cores (split-bf16, 6 partial products; terms = 3 keeps 3):
 *   C[b] = act(A[b] * B[b] + bias[b] + init[b]),  A[b] = A + b*a_batch_stride: M x K fp32, row stride lda
 *   Bt_planes: [batch][3][N][K] bf16 = hi / mid / lo parts of B^T (k contiguous), prepared by the host
 *   bias [batch][N] or NULL; init (row stride ldi, batch stride) or NULL; C row stride ldc.
 *   act: 0 none, 1 ELU(alpha 1), 2 leaky ReLU(slope).  N % 128 == 0, K % 32 == 0.
 * Replaces the torch.mm / nn.Linear calls of layers.py:350, 363, 376, 398, 110, 39, 905 and
 * models.py:538 (reference: ATen fp32 GEMM). */
int disgat_gemm_split(const float* A, int64_t lda, int64_t a_batch_stride, const uint16_t* Bt_planes,
                      const float* bias, const float* init, int64_t ldi, int64_t init_batch_stride, float* C,
                      int64_t ldc, int64_t c_batch_stride, int M, int N, int K, int batch, int act,
                      float slope, int terms, disgat_stream_t stream);

/* The same contraction with two fp16 planes per operand and 3 partial products (half the matrix-core work):
 * every element of A*s_A and B*s_B (s = power of two placing the operand's max magnitude in [2^13, 2^14)) is
 * hi + lo*2^-11 with fp16 hi, lo, accurate to 2^-23 of the element; hi*hi, hi*lo, lo*hi are accumulated in fp32.
 *   Bt_planes: [batch][2][N][K] fp16 = hi, lo of (B^T * s_B); b_scale: device float holding s_B;
 *   a_amax: device float holding max |A| over the whole (batched) operand (disgat_amax); s_A is derived in-kernel.
 * Other arguments as disgat_gemm_split. */
int disgat_gemm_f16x3(const float* A, int64_t lda, int64_t a_batch_stride, const uint16_t* Bt_planes,
                      const float* a_amax, const float* b_scale, const float* bias, const float* init,
                      int64_t ldi, int64_t init_batch_stride, float* C, int64_t ldc, int64_t c_batch_stride,
                      int M, int N, int K, int batch, int act, float slope, disgat_stream_t stream);

/* Weight gradient dW[b] = A[b]^T G[b] (A [M][Ka], G [M][N] fp32, the reduction runs over the M rows): split-K over
 * `splits` row ranges with the f16x3 scheme, both operands split on the fly.  Writes partials [batch][splits][Ka][N];
 * the caller sums them over `splits` (deterministic).  Ka, N multiples of 128; a_amax / g_amax: device floats holding
 * (upper bounds of) max |A|, max |G|.  Replaces ATen's mm(a.t(), g) in the autograd of nn.Linear / torch.mm. */
int disgat_gemm_f16x3_tn(const float* A, int64_t lda, int64_t a_batch_stride, const float* G, int64_t ldg,
                         int64_t g_batch_stride, const float* a_amax, const float* g_amax, float* partials,
                         int M, int Ka, int N, int batch, int splits, disgat_stream_t stream);

/* Scale bound of a GEMM output x W from a bound on |x|: out[0] = max over (batch, column) of sum_k |W[b][k][n]| (the largest
 * column abs-sum of an arbitrarily strided [batch][K][N] weight of at most 131 072 elements), and, when in_bound is given,
 * out[1] = max(floor_value, in_bound[0] * out[0] * scale) - what the host formulation spent six element-wise / reduce launches
 * per layer on (layers.py has no counterpart: the reference's fp32 GEMMs need no scale). */
int disgat_weight_bound(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                        const float* in_bound, float scale, float floor_value, float* out, disgat_stream_t stream);

/* out[b] = A[b]^T G[b] for a SMALL result (K x N a few 64 x 64 tiles; the nhid = 64 weight gradients of Cora-sized graphs,
 * where the reduction over the M node rows is long and a library GEMM uses a handful of workgroups): fp32 FMAs, the rows cut
 * into `splits` ranges (1..1024) whose partial results - partials [batch][splits][K][N], scratch, NULL when splits == 1 - a
 * second launch adds in range order.  A [M][K] (row stride lda), G [M][N] (ldg); K, N and all strides multiples of 4 floats,
 * bases 16-byte aligned; out [batch][K][N] contiguous.  Replaces the autograd matmuls behind layers.py:350, 363, 376, 398, 110
 * and models.py:538 for those shapes. */
int disgat_wgrad_small(const float* A, int64_t lda, int64_t a_batch_stride, const float* G, int64_t ldg, int64_t g_batch_stride,
                       int M, int K, int N, int batch, int splits, float* partials, float* out, disgat_stream_t stream);

/* *out = max |A[b][m][k]| over batch x M x K (K, lda, batch stride multiples of 4); 0 for an empty operand. */
int disgat_amax(const float* A, int64_t lda, int64_t a_batch_stride, int M, int K, int batch, float* out,
                disgat_stream_t stream);

/* Weight preparation for disgat_gemm_f16x3: W is [batch][K][N] fp32 with arbitrary element strides.  Writes
 * planes [batch][2][N*K] fp16 = hi, lo of (W^T * s) and amax_scale[0] = max |W|, amax_scale[1] = s (pass
 * amax_scale + 1 as b_scale).  Element order inside a plane: K > 256: row-major [N][K] (k contiguous); K <= 256
 * (K % 32 == 0: the shapes disgat_gemm_f16x3 runs on its A-stationary kernel): fragment-major - the 16 x 32 block of
 * (n / 16, k / 32) is stored contiguously in MFMA lane order (lane = 16 * (k % 32 / 8) + n % 16, 8 halfs per lane), so
 * a wave's weight-fragment load is one contiguous KB.  The two launchers share this rule; planes are opaque to callers. */
int disgat_split_f16(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                     uint16_t* planes, float* amax_scale, disgat_stream_t stream);

/* disgat_split_f16 with the row-major [N][K] plane order for every K: the weight operand of disgat_gemm_planes. */
int disgat_split_f16_rm(const float* W, int64_t stride_b, int64_t stride_k, int64_t stride_n, int K, int N, int batch,
                        uint16_t* planes, float* amax_scale, disgat_stream_t stream);

/* The f16x3 contraction with an A operand that arrives ALREADY SPLIT - written as two fp16 planes by the kernel that
 * produced it (disgat_edge_fwd with Z_hi / Z_lo, this launcher's own plane output, or disgat_split_planes):
 *   A_hi, A_lo [batch][M][K] fp16 (row stride lda halfs, batch stride a_batch_stride halfs) = hi, lo of A * s_A with
 *   s_A = the power of two that puts *a_bound in [2^13, 2^14) - producer and consumer read the same device scalar;
 *   Bt_planes from disgat_split_f16_rm, b_scale = its amax_scale + 1; bias / init / act / slope as disgat_gemm_f16x3.
 * Outputs, either or both: C fp32 (row stride ldc); C_hi / C_lo planes of C * s_C (row stride ldp halfs), s_C from
 * *c_bound (an upper bound of max |C|, e.g. a_bound x the largest column abs-sum of B + max |bias|): the A operand of
 * the next GEMM of the chain.  N % 256 == 0, K % 32 == 0, K >= 64; plane rows 16-byte aligned.
 * Replaces, on forwards that record no autograd graph, the same torch.mm / nn.Linear calls as disgat_gemm_f16x3
 * (layers.py:397-399 per-head projection, :905 FuseLayer, models.py:538 DifHead classifier). */
int disgat_gemm_planes(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, int64_t a_batch_stride,
                       const uint16_t* Bt_planes, const float* a_bound, const float* b_scale, const float* bias,
                       const float* init, int64_t ldi, int64_t init_batch_stride, float* C, int64_t ldc,
                       int64_t c_batch_stride, uint16_t* C_hi, uint16_t* C_lo, int64_t ldp, int64_t p_batch_stride,
                       const float* c_bound, int M, int N, int K, int batch, int act, float slope,
                       disgat_stream_t stream);

/* disgat_gemm_planes followed by a skinny second product in the same launch - the DifHead classifier on the head planes
 * (pretrainer.py:819-832: MLP(cat(layer_in, head_out_i)) per head i; models.py:523-543 with cls_layer == 2):
 *   L[(m, b)][0 .. n_out) = act( A[b][m][:] W1[b] + bias + init[m][:] ) W2 + bias2         N == 256, 1 <= n_out <= 16
 * The hidden layer ([M * batch][256]) exists only as accumulator tiles: each becomes the operand of the second f16x3
 * product while it is still in registers.  mid_bound: device scalar >= max |act(.)| (e.g. a_bound x the largest column
 * abs-sum of W1 + max |init| + max |bias|).  W2_frags: W2 * s_W2 as MFMA fragments, [8 groups of 32 hidden columns][2 planes:
 * hi, lo][64 lanes][8 halfs], lane l of group g holding hidden columns 32 g + 8 (l / 16) .. + 7 of output l % 16 (zero past
 * n_out); w2_scale: device scalar s_W2; bias2: 16 floats (zero past n_out) or NULL.  L: fp32, row (m, b) at
 * (m * batch + b) * n_out - the order of the reference's per-head concatenation viewed as [M * nhead, hidden].
 * Replaces nn.Linear -> LeakyReLU -> nn.Linear of models.py:538 on no-graph forwards. */
int disgat_gemm_planes_logits(const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, int64_t a_batch_stride,
                              const uint16_t* Bt_planes, const float* a_bound, const float* b_scale, const float* bias,
                              const float* init, int64_t ldi, int64_t init_batch_stride, const float* mid_bound,
                              const uint16_t* W2_frags, const float* w2_scale, const float* bias2, float* L, int M, int N,
                              int K, int batch, int n_out, int act, float slope, disgat_stream_t stream);

/* The per-head output projection AND the FuseLayer that consumes it as ONE launch (back-to-back GEMM; no-graph forwards):
 *   C = act2( cat_h [ elu( Z_h W1_h + bias1_h ) ] W2 + bias2 )
 * Z_hi / Z_lo: the aggregate as disgat_edge_fwd writes it (row m, head h at m * ldz + h * z_head_stride halfs; hi, lo of
 * Z * s_Z, s_Z from *z_bound).  W_chunks: both weights as one stream of chunk images, [H][N1 / 32][slot]: chunk (h, j) =
 * the disgat_split_f16 fragment blocks (1 KB each) of W1_h's columns 32 j .. 32 j + 31 - [plane][n-tile 2j, 2j+1][k-step] -
 * followed by those of W2's rows h * N1 + 32 j .. + 31 for every column tile - [plane][n-tile] -, where W2 [H * N1][N2] is
 * viewed as [H][N1][N2] and the rows of every group of 32 are permuted first (new position 8 q + e <- row 4 q + e for e < 4,
 * 16 + 4 q + e - 4 for e >= 4: the order a lane of GEMM 1's accumulator tiles holds its columns in); w1_scale / w2_scale:
 * the amax_scale + 1 of the two splits.  mid_bound: device scalar >= max |elu(.)| (the scale of the intermediate, which never
 * leaves the registers).  bias1 [H * N1] / bias2 [N2] or NULL.  act2: 0 none, 2 leaky ReLU.
 * K1 in {64, 128, 256}, N1 % 32 == 0 (64 <= N1 <= 256), N2 in {64, 128, 256}.
 * Replaces layers.py:397-399 / :404-407 (+ F.elu, :508) followed by layers.py:896-921 (residue_type 0, no residue
 * columns) for gnn_type AT / GCN - the pair disgat_gemm_planes ran as two launches through an [M][H * N1] plane buffer. */
int disgat_proj_fuse(const uint16_t* Z_hi, const uint16_t* Z_lo, int64_t ldz, int64_t z_head_stride, const float* z_bound,
                     const uint16_t* W_chunks, const float* w1_scale, const float* bias1, const float* w2_scale,
                     const float* bias2, const float* mid_bound, float* C, int64_t ldc, int M, int H, int K1, int N1, int N2,
                     int act2, float slope, disgat_stream_t stream);

/* fp32 X [batch][M][K] -> planes P_hi, P_lo (hi, lo of X * s, s from *bound >= max |X|); and back:
 * X = (P_hi + P_lo * 2^-11) / s.  K and every stride multiples of 4. */
int disgat_split_planes(const float* X, int64_t ldx, int64_t x_batch_stride, int M, int K, int batch,
                        const float* bound, uint16_t* P_hi, uint16_t* P_lo, int64_t ldp, int64_t p_batch_stride,
                        disgat_stream_t stream);
int disgat_planes_to_f32(const uint16_t* P_hi, const uint16_t* P_lo, int64_t ldp, int64_t p_batch_stride, int M, int K,
                         int batch, const float* bound, float* X, int64_t ldx, int64_t x_batch_stride,
                         disgat_stream_t stream);

/* Diagnostic: with DISGAT_PL_DEBUG & 32 disgat_gemm_planes accumulates s_memtime cycles per loop phase of waves 0 and 4 of
 * every block ([2][8]: wait, barrier, fragment reads + DMA issue, MFMA, epilogue, unit set-up); copies them out
 * (out16 may be NULL) and optionally zeroes them.  Synchronises the device. */
int disgat_debug_stamps(unsigned long long* out16, int reset);
/* The same for disgat_gemm_f16x3's register-stationary kernel (a -DRS_DIAG=1 build, DISGAT_RS_DEBUG & 32; csrc/gemm_rs.hip). */
int disgat_debug_stamps_rs(unsigned long long* out16, int reset);
/* The same for disgat_proj_fuse (a -DBB_DIAG=32 build, ablation bits added at compile time; csrc/gemm_b2b.hip). */
int disgat_debug_stamps_b2b(unsigned long long* out16, int reset);

/* Backward of the epilogue activation from the saved output (n contiguous floats, n % 4 == 0):
 * gin = g * (out > 0 ? 1 : (act == 1 ? out + 1 : slope)); act 1 = ELU, 2 = leaky ReLU.  gin may alias g.
 * amax_out (or NULL): receives max |gin|, the scale input of the GEMMs that consume gin. */
int disgat_act_bwd(const float* g, const float* out, float* gin, int64_t n, int act, float slope,
                   float* amax_out, disgat_stream_t stream);

/* Y[M][N] = X[M][K] W^T + bias for N <= 16 output columns, K = 256 or 512 (W [N][ldw] as nn.Linear keeps it, bias [N] or
 * NULL): the hidden -> nhead layer at the end of the DifHead classifier (models.py:523-543 on pretrainer.py:819-832's
 * N_nodes * nhead rows) - 1 KB read per 32 B written, which a GEMM library runs far below the HBM rate.  fp32 FMAs. */
int disgat_linear_skinny(const float* X, int64_t ldx, int64_t M, int K, const float* W, int64_t ldw,
                         const float* bias, int N, float* Y, int64_t ldy, disgat_stream_t stream);
/* Its weight gradient dW[N][K] = G^T X (G [M][N], ldg): every wave of the launch (n_waves, a multiple of 4) writes one
 * partial [N][K] into partials [n_waves][N][K]; the caller adds them (fixed order: deterministic). */
int disgat_linear_skinny_wgrad(const float* X, int64_t ldx, int64_t M, int K, const float* G, int64_t ldg, int N,
                               float* partials, int n_waves, disgat_stream_t stream);

/* ---- SSL pair sampler ------------------------------------------------------------------------ */

/* SupEdgeTrainer.sample_train / GeneratedEdgeTrainer.sample_train (pretrainer.py:683-707, 524-576) without the dense
 * N x N tensors: the list  nonzero( (rand(N,N) < p) | {a uniform subset of exactly n_sel of the positives} )  in row-major
 * order, with labels = "is a positive", drawn from a counter-based generator whose (seed, step) live on the device.
 *   items    int32 [n_items][8] = {row, col_lo, col_hi, pos_lo, pos_hi, 0, 0, 0}, ordered by (row, col_lo): a partition of
 *            every row's column range [0, n_cols) into intervals; positives [pos_lo, pos_hi) are those of the row whose
 *            column lies in the interval - at most DISGAT_SAMPLE_PCAP of them - and (col_hi - col_lo) * p should not
 *            exceed DISGAT_SAMPLE_RMEAN (an item keeps at most DISGAT_SAMPLE_RCAP random columns; meta[4] counts items
 *            that hit either limit).  One wave per item, items_per_wave (1..16) consecutive items per wave.
 *   pos_col  int32 [n_pos]: columns of the positives, row-major (a CSR column array: sorted inside a row)
 *   p        the Bernoulli probability of the random part (the reference's edge_ratio * 3), in [0, 1]
 *   meta     int64 [8], device: [0] seed, [1] step (the plan advances it), [2] the step the last plan drew with,
 *            [3] length of the last list (clamped to capacity), [4] / [5] event counters: item / list over capacity.
 * disgat_pair_sample_plan  counts (item_count [n_items], block_count / block_off [ceil(n_items / (4 * items_per_wave))]
 *            scratch), leaves the list length in meta[3] and, as a double, in *count_out (NULL: not wanted).
 * disgat_pair_sample_emit  writes idx_out int64 [2][capacity] (rows, then columns) and lab_out [capacity] for the plan that
 *            preceded it on the stream; entries beyond `capacity` are dropped; pad_tail != 0 fills [length, capacity) with
 *            the pair (n_rows - 1, n_cols - 1) and label -1 (the padding disgat_pair_loss skips).
 * A caller that wants a list of exact length reads meta[3] between the two calls and passes it as the capacity. */
#define DISGAT_SAMPLE_PCAP 256
#define DISGAT_SAMPLE_RCAP 256
#define DISGAT_SAMPLE_RMEAN 96
int disgat_pair_sample_plan(const int32_t* items, int n_items, int items_per_wave, const int32_t* pos_col, int64_t n_pos,
                            int64_t n_sel, double p, int64_t capacity, int64_t* meta, int32_t* item_count,
                            int32_t* block_count, int64_t* block_off, double* count_out, disgat_stream_t stream);
int disgat_pair_sample_emit(const int32_t* items, int n_items, int items_per_wave, const int32_t* pos_col, int64_t n_pos,
                            int64_t n_sel, double p, int64_t n_rows, int64_t n_cols, int64_t capacity, int64_t* meta,
                            const int32_t* item_count, const int64_t* block_off, int64_t* idx_out, float* lab_out,
                            int pad_tail, disgat_stream_t stream);

/* ---- optimiser --------------------------------------------------------------------------- */

/* One Adam step for `count` parameter tensors in ONE launch per DISGAT_ADAM_MAX_TENSORS tensors (the table travels in
 * the kernel arguments).  All arrays are HOST arrays of length count; params / grads / exp_avg / exp_avg_sq hold DEVICE
 * pointers to numel[i] contiguous floats.  Per tensor: step_size = lr / (1 - beta1^t), inv_sqrt_bc2 = 1 / sqrt(1 -
 * beta2^t), weight_decay (L2, added to the gradient) - t is that tensor's own step count, kept by the caller.
 *   g += wd*p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= step_size * m / (sqrt(v)*inv_sqrt_bc2 + eps)
 * Replaces the per-sub-module torch.optim.Adam steps of trainer.py:58-60, 205-206 and pretrainer.py:754-756, 633-635,
 * 838-840 (same arithmetic as torch.optim.Adam, amsgrad off). */
#define DISGAT_ADAM_MAX_TENSORS 64
int disgat_adam_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const int64_t* numel, const float* step_size,
                      const float* inv_sqrt_bc2, const float* weight_decay, double beta1, double beta2, float eps,
                      disgat_stream_t stream);

/* The same step with the step counts on the DEVICE: steps[i] (int32, device memory) = t of tensor i, already advanced by
 * the caller - inside a train_step captured in a HIP graph, by an op of the same graph; lr[i] (host, double) is the plain
 * learning rate and the two bias corrections are formed in the kernel, in double like the host forms them above.  A
 * replayed graph would otherwise repeat the corrections of the step it was captured at. */
int disgat_adam_multi_dev(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, const double* lr, const float* weight_decay,
                          const int32_t* steps, double beta1, double beta2, float eps, disgat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DISGAT_HIP_H */
