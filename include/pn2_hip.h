/*
 * pn2_hip.h -- C ABI of libpn2hip.so: the MI355X (gfx950) PointNet++ set-abstraction /
 * feature-propagation hot path.
 *
 * The reference has no FFI layer for this path: the boundary it exposes is the Python module
 * surface of models/pointnet2_utils.py.  Every entry point below replaces one
 * torch-op composition of that file (cited per function) and is what a ctypes / cffi /
 * pybind stub inside the reference's models/pointnet2_utils.py binds to.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller; the library allocates
 *     nothing and is reentrant.  Its only process state are idempotent per-device memos of host-side facts about
 *     its own kernels (dynamic-LDS attribute raised, occupancy) and the PN2_TUNE_* developer knobs, which are read
 *     from the environment once, at the first launch, never per call;
 *   - float tensors are fp32, index tensors int64, row-major contiguous, layouts as in the
 *     reference's free functions: xyz [B,N,3], points [B,N,D], idx [B,S] / [B,S,K];
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     nothing synchronises, nothing allocates: every launcher is hipGraph-capture safe;
 *   - return value: PN2_OK, a negative PN2_ERR_* for invalid arguments (nothing launched),
 *     or a positive hipError_t from the launch;
 *   - data-dependent faults (the reference raises IndexError, pointnet2_utils.py:59) are
 *     counted into the caller's device word `err_count` (may be NULL): the host shim zeroes
 *     it before and reads it after, lazily.
 */
#ifndef PN2_HIP_H
#define PN2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_ABI_VERSION 1

typedef void *pn2_stream_t; /* hipStream_t */

enum {
    PN2_OK = 0,
    PN2_ERR_NULL = -1,        /* required pointer is NULL */
    PN2_ERR_SHAPE = -2,       /* negative / zero / inconsistent size */
    PN2_ERR_UNSUPPORTED = -3, /* size outside what the kernels are built for */
};

int pn2_abi_version(void);
const char *pn2_error_string(int rc);

/* farthest_point_sample(xyz, npoint)                     models/pointnet2_utils.py:63-84
 * start[B]: the indices the reference draws with torch.randint (:75), supplied by the caller.
 * out_idx[B,npoint] int64.  new_xyz (nullable) [B,npoint,3] receives xyz[out_idx] -- the
 * index_points call that always follows (:125) -- for free.
 * Limits: 1 <= N <= 32768.  err_count += 1 per block whose start is outside [0,N). */
int pn2_farthest_point_sample(const float *xyz, int B, int N, int npoint, const int64_t *start,
                              int64_t *out_idx, float *new_xyz, int32_t *err_count, pn2_stream_t stream);

/* square_distance(src, dst) -> out[B,N,M]                models/pointnet2_utils.py:19-40
 * bit-for-bit the reference's CPU result (expansion form).  The other
 * kernels evaluate the same expression in registers and never materialise this matrix. */
int pn2_square_distance(const float *src, const float *dst, int B, int N, int M, float *out,
                        pn2_stream_t stream);

/* query_ball_point(radius, nsample, xyz, new_xyz) fused with the grouping half of
 * sample_and_group                                      models/pointnet2_utils.py:87-107, 127-132
 * idx[B,S,nsample] int64; grouped (nullable) [B,S,nsample,3+D] = [xyz[idx]-new_xyz, points[idx]];
 * points nullable (then D must be 0).  A centroid with no point inside the radius
 * (reference: IndexError at :59) gets idx = N, a zero grouped row block and err_count += 1.
 * ldg = row pitch of grouped in floats (0 = dense 3+D; a larger pitch zero-fills columns [3+D, ldg),
 * used to keep the MLP's rows 16-byte aligned when 3+D is not a multiple of 4).
 * Limits: 1 <= nsample <= 64. */
int pn2_ball_query_group(double radius, int nsample, const float *xyz, const float *new_xyz,
                         const float *points, int B, int N, int S, int D, int64_t *idx, float *grouped,
                         int ldg, int32_t *err_count, pn2_stream_t stream);

/* pn2_ball_query_group through ONE named kernel of the library (parity tests compare every shipped kernel with the
 * oracle; benchmarks compare them with each other): which = 0 the library's choice (= pn2_ball_query_group),
 * 1 cell-pruned (pn2_ball_grid.hip), 2 vector-unit scan (pn2_ball_group.hip).
 * PN2_ERR_UNSUPPORTED (nothing launched) when the named kernel does not take the operands. */
int pn2_ball_query_group_select(int which, double radius, int nsample, const float *xyz, const float *new_xyz,
                                const float *points, int B, int N, int S, int D, int64_t *idx, float *grouped,
                                int ldg, int32_t *err_count, pn2_stream_t stream);

/* ---- planned form of the two entries above (sample_and_group, models/pointnet2_utils.py:110-138) ------------
 * Everything of query_ball_point that depends on the block's geometry only is prepared once per block in a
 * caller-owned workspace `plans` (B * pn2_ball_plan_bytes(N, S, D) bytes, 128-byte aligned; layout in
 * csrc/pn2_ball_bin.h): the points sorted into a uniform grid whose cells are at least radius wide, per centroid the
 * nine runs of that array holding the 27 neighbouring cells, and the packed rows [x, y, z, feats] the grouping
 * gathers from.  pn2_farthest_point_sample_plan = pn2_farthest_point_sample followed, on the same stream, by the
 * launch that writes the sort and the runs (new_xyz required); pn2_ball_pack_rows adds the packed rows (needed when
 * pn2_ball_query_group_planned is asked for `grouped` with the dense pitch 3+D, (3+D) % 4 == 0);
 * pn2_ball_plan = the stand-alone producer of all three for a given new_xyz.  pn2_ball_query_group_planned is
 * pn2_ball_query_group on such a plan: same outputs, bit for bit.  Limits: N <= 8192 (pn2_ball_plan_bytes
 * returns 0 beyond). */
long long pn2_ball_plan_bytes(int N, int S, int D);
int pn2_farthest_point_sample_plan(const float *xyz, int B, int N, int npoint, const int64_t *start, int64_t *out_idx,
                                   float *new_xyz, double radius, int D, void *plans, int32_t *err_count,
                                   pn2_stream_t stream);
int pn2_ball_pack_rows(const float *xyz, const float *points, int B, int N, int S, int D, void *plans, pn2_stream_t stream);
int pn2_ball_plan(double radius, const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int D,
                  void *plans, pn2_stream_t stream);
int pn2_ball_query_group_planned(double radius, int nsample, const void *plans, const float *xyz, const float *new_xyz,
                                 const float *points, int B, int N, int S, int D, int64_t *idx, float *grouped, int ldg,
                                 int32_t *err_count, pn2_stream_t stream);

/* index_points(points, idx) -> out[B,M,C]                models/pointnet2_utils.py:43-60
 * idx is [B,M] (any trailing idx dims flattened into M).  Out-of-range index: zero row,
 * err_count += 1. */
int pn2_index_points(const float *points, const int64_t *idx, int B, int N, int C, int64_t M, float *out,
                     int32_t *err_count, pn2_stream_t stream);

/* autograd of index_points w.r.t. points (scatter-add)    models/pointnet2_utils.py:59
 * grad_points[B,N,D] += grad_out[B,M,Cg][:, :, col0:col0+D] at rows idx.  The caller zeroes
 * grad_points first.  col0/Cg let the grouped tensor's gradient [B,S*K,3+D] be scattered
 * without slicing (col0 = 3). */
int pn2_index_points_backward(const float *grad_out, const int64_t *idx, int B, int N, int D, int64_t M,
                              int Cg, int col0, float *grad_points, pn2_stream_t stream);

/* grouping half of sample_and_group for a given idx      models/pointnet2_utils.py:127-132 */
int pn2_group_points(const float *xyz, const float *new_xyz, const float *points, const int64_t *idx,
                     int B, int N, int S, int K, int D, float *grouped, int ldg, int32_t *err_count,
                     pn2_stream_t stream);

/* three nearest neighbours + inverse-distance weights of PointNetFeaturePropagation
 *                                                       models/pointnet2_utils.py:296-302
 * xyz1[B,N,3] queries, xyz2[B,S,3] sources, S >= 3.  idx3[B,N,3] int64 ascending distance
 * (exact ties: lowest index first -- the reference's unstable sort leaves them unpinned),
 * dist3 (nullable) the expansion-form distances, weight3[B,N,3]. */
int pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx3, float *dist3,
                 float *weight3, pn2_stream_t stream);
/* n <= 8 (queries, sources) pairs of one batch size in one launch (host arrays of device pointers / sizes, read before
 * return; no dist3). */
int pn2_three_nn_many(int n, const float *const *xyz1, const float *const *xyz2, int B, const int *N, const int *S,
                      int64_t *const *idx3, float *const *weight3, pn2_stream_t stream);

/* interpolated[b,i,:] = sum_k points2[b,idx3[b,i,k],:] * weight3[b,i,k]
 *                                                       models/pointnet2_utils.py:303 */
int pn2_three_interpolate(const float *points2, const int64_t *idx3, const float *weight3, int B, int N,
                          int S, int D, float *out, pn2_stream_t stream);

/* autograd of :303 w.r.t. points2: grad_points2[B,S,D] (zeroed by the caller) +=
 * weight3 * grad_out[B,N,D] at rows idx3. */
int pn2_three_interpolate_backward(const float *grad_out, const int64_t *idx3, const float *weight3,
                                   int B, int N, int S, int D, float *grad_points2, pn2_stream_t stream);


/* ---- grouped / pointwise MLP: [Conv 1x1 -> BatchNorm -> ReLU] x n (+ max over nsample) ----------
 *                                   models/pointnet2_utils.py:196-200 (SA), :312-314 (FP)
 * Rows are channel-last: M = B*S*K (SA) or B*N (FP) rows of Ci channels.  All GEMMs are exact-fp32
 * MFMA (v_mfma_f32_32x32x2_f32).  Normalised activations are never stored: a layer writes its raw
 * conv output Z and the next consumer applies max(scale*z+shift, 0) while loading. */

/* out[M,N] = A * W^T + bias, with
 *   prologue 0: A = [x1 (K1 cols, pitch ld1) | x2 (K2 cols, pitch ld2)]           (x2 may be NULL)
 *   prologue 1: A = max(scale[k]*[x1|x2] + shift[k], 0)       previous layer's BatchNorm + ReLU
 *   prologue 2: A = dz(g = x1, z = x2)  [BatchNorm+ReLU backward of an N_prev = K1 channel layer]:
 *               gh = (scale*z+shift > 0) ? g : 0;  dz = scale*(gh - c1 - (z-mean)*invstd*c2);
 *               with argk != NULL, g is the gradient of the max-pooled output [M/pool_k, K1] and
 *               is routed to row argk (autograd of torch.max over nsample, :200).
 * W is [N][K] row-major (w_is_kn = 0) or [K][N] (w_is_kn = 1; the weight itself, for dX = dZ * W).
 * out2 (nullable): columns >= nsplit are written to out2[M][N-nsplit] (pitch ldo2) instead of out --
 * the gradient of a two-source input [x1 | x2] lands in two dense tensors without a slicing copy.
 * stat_partial (nullable) [pn2_mlp_gemm_max_partials(M)][2][N]: per-workgroup column sums of out
 * and out^2 (train-mode batch statistics) -- or, when mask_z != NULL (backward), out is first
 * masked by the ReLU of the layer below (mscale*mask_z+mshift > 0) and the sums are those of
 * out and out*(mask_z-mmean)*minvstd (its dbeta / dgamma); mask_z is accepted with prologue 2 only. */
int pn2_mlp_gemm_max_partials(int M);
int pn2_mlp_gemm(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int prologue,
                 const float *scale, const float *shift, const float *mean, const float *invstd,
                 const float *c1, const float *c2, const unsigned char *argk, int pool_k, const float *w,
                 int ldw, int w_is_kn, const float *bias, float *out, int ldo, float *out2, int ldo2, int nsplit,
                 int M, int N, float *stat_partial, const float *mask_z, int ldm, const float *mscale, const float *mshift,
                 const float *mmean, const float *minvstd, pn2_stream_t stream);

/* partial[P][2][C] -> train-mode BatchNorm coefficients scale = gamma*invstd, shift = beta -
 * mean*scale (biased variance), mean/invstd for backward, and the running-estimate update
 * running = (1-momentum)*running + momentum*batch (unbiased variance), nn.BatchNorm semantics;
 * num_batches_tracked (nullable, int64 device word) is incremented.
 * momentum_dev (nullable): one float in device memory read when the kernel RUNS and used instead of `momentum`,
 * so that a launch replayed from a hipGraph follows the reference loop's per-epoch schedule
 * (localfunctions.py:191-195).  A negative momentum = nn.BatchNorm(momentum=None): the cumulative moving
 * average, factor 1 / num_batches_tracked, with the counter incremented by the caller BEFORE this launch
 * (it is then only read here). */
int pn2_bn_finalize(const float *partial, int P, int C, double count, const float *gamma, const float *beta,
                    float eps, float momentum, const float *momentum_dev, float *running_mean, float *running_var,
                    float *scale, float *shift, float *mean_out, float *invstd_out, long long *num_batches_tracked,
                    pn2_stream_t stream);

/* The forward GEMM of a stack's LAST layer when its rows are max-pooled in groups of 32 (nsample = 32,
 * models/pointnet2_utils.py:200): prologue 0 / 1 of pn2_mlp_gemm, W [N][K], and in the epilogue -- one 32-row
 * accumulator block is one group -- the largest and the smallest z of every (group, column) with the first row that
 * holds it: pool_max / pool_min [M/32][N] floats, pool_amax / pool_amin [M/32][N] bytes.  The pooled activation is
 * then relu(scale*(scale >= 0 ? zmax : zmin) + shift) (rounding is monotone), chosen by pn2_bn_finalize_out once the
 * batch statistics exist; z is written as usual (the backward reads it) but never re-read by the forward.
 * M % 32 == 0.  out = NULL: z is not stored at all (inference: nothing reads it).  PN2_ERR_UNSUPPORTED (nothing launched)
 * when the operands do not allow the pipelined kernels. */
int pn2_mlp_gemm_pool32(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int prologue,
                        const float *scale, const float *shift, const float *w, int ldw, const float *bias, float *out,
                        int ldo, int M, int N, float *stat_partial, float *pool_max, float *pool_min,
                        unsigned char *pool_amax, unsigned char *pool_amin, pn2_stream_t stream);

/* pn2_bn_finalize and the stack's output in ONE launch (models/pointnet2_utils.py:198-200 / :314):
 *   pool_max == NULL: y[rows_out][C] = max(scale*z + shift, 0), z with row pitch ldz;
 *   pool_max != NULL: y / argk [rows_out][C] selected from the extrema of pn2_mlp_gemm_pool32 (argk = 0 where y == 0
 *                     or scale == 0: all 32 rows tie and torch.max reports the first).
 * partial == NULL: scale / shift are inputs (eval mode), nothing is finalized.  C % 4 == 0. */
int pn2_bn_finalize_out(const float *partial, int P, int C, double count, const float *gamma, const float *beta,
                        float eps, float momentum, const float *momentum_dev, float *running_mean, float *running_var,
                        float *scale, float *shift, float *mean_out, float *invstd_out, long long *num_batches_tracked,
                        const float *z, int ldz, const float *pool_max, const float *pool_min,
                        const unsigned char *pool_amax, const unsigned char *pool_amin, long long rows_out, float *y,
                        unsigned char *argk, pn2_stream_t stream);

/* eval-mode coefficients from the running estimates */
int pn2_bn_eval_coeff(int C, const float *gamma, const float *beta, const float *running_mean,
                      const float *running_var, float eps, float *scale, float *shift, pn2_stream_t stream);

/* y = max(scale*z+shift, 0) for [rows_out, C] (pool_k = 0), or max over groups of pool_k
 * consecutive rows of z [rows_out*pool_k, C] with the winning k in argk (nullable).  C % 4 == 0. */
int pn2_bn_relu_out(const float *z, long long rows_out, int C, int pool_k, const float *scale,
                    const float *shift, float *y, unsigned char *argk, pn2_stream_t stream);

/* dW[N][K1+K2] = dz^T * act([x1|x2]), db[N] = column sums of dz (db nullable); dz as in prologue 2
 * of pn2_mlp_gemm (g, z, argk/pool_k, BatchNorm constants of this layer); act = BatchNorm+ReLU of
 * the layer below when ascale/ashift are given.  partial: workspace
 * [pn2_mlp_dw_partials(M, N, K1+K2)][N][K1+K2+1] floats. */
int pn2_mlp_dw_partials(int M, int N, int K);
int pn2_mlp_dw(const float *g, int ldg, const float *z, int ldz, const unsigned char *argk, int pool_k,
               const float *scale, const float *shift, const float *mean, const float *invstd,
               const float *c1, const float *c2, const float *x1, int ld1, int K1, const float *x2, int ld2,
               int K2, const float *ascale, const float *ashift, int M, int N, float *partial, float *dw,
               float *db, pn2_stream_t stream);

/* One-pass backward of a conv/BatchNorm/ReLU layer with N, K <= 128 (N, K % 4 == 0): dz as in prologue 2 of
 * pn2_mlp_gemm from (g, z, argk/pool_k, constants of this layer); gp[M][K] = dz * w[N][K] (nullable), masked by
 * the ReLU of the layer below when ascale/ashift/amean/ainvstd are given (x is then that layer's raw z and
 * stat_partial [P][2][K] receives the column sums of gp and gp*xhat; otherwise x is an activation);
 * dw[N][K] = dz^T * act(x), db[N] (nullable).  dw_partial: workspace [P][N][K+1] floats with
 * P = pn2_mlp_bwd_layer_partials(M, N, K) (0: shape not covered -> use pn2_mlp_gemm + pn2_mlp_dw).
 * With c1_below/c2_below (and optionally dgamma_below/dbeta_below, all [K]) the pn2_bn_bwd_finalize of
 * stat_partial (count = M) runs in the same launch as the slab reduction. */
int pn2_mlp_bwd_layer_partials(int M, int N, int K);
int pn2_mlp_bwd_layer(const float *g, int ldg, const float *z, int ldz, const unsigned char *argk, int pool_k,
                      const float *scale, const float *shift, const float *mean, const float *invstd,
                      const float *c1, const float *c2, const float *w, int ldw, const float *x, int ldx,
                      const float *ascale, const float *ashift, const float *amean, const float *ainvstd,
                      float *gp, int ldgp, float *stat_partial, float *dw_partial, float *dw, float *db,
                      float *dgamma_below, float *dbeta_below, float *c1_below, float *c2_below, int M, int N, int K,
                      pn2_stream_t stream);

/* The two reductions that follow a layer's backward in one launch: dw/db from dw_partial [P][N][K+1] (as the
 * tail of pn2_mlp_dw, which leaves the slabs unreduced when called with dw = NULL) and pn2_bn_bwd_finalize of
 * stat_partial [Ps][2][C] for the layer below. */
int pn2_mlp_bwd_post(const float *dw_partial, int P, int N, int K, float *dw, float *db, const float *stat_partial,
                     int Ps, int C, double count, float *dgamma, float *dbeta, float *c1, float *c2,
                     pn2_stream_t stream);

/* The slab sums of n layers in one launch (16 per launch): dw[i] [N[i]][Kstore[i]] (the first Kstore <= K columns: a
 * first layer whose input rows carry zero pad columns has no gradient entries for them), db[i] [N[i]] (nullable) from
 * partial[i] [P[i]][N[i]][K[i]+1] as pn2_mlp_dw / pn2_mlp_bwd_layer / pn2_head_logits_dropout_backward leave them when
 * called with dw = NULL.  The bottom layers' weight gradients are needed by nobody before the optimizer: a training
 * step sums them all at the end of backward (mlp.deferred_weight_sums).  Arrays on the host, read before return. */
int pn2_mlp_dw_reduce_many(int n, const float *const *partial, const int *P, const int *N, const int *K, const int *Kstore,
                           float *const *dw, float *const *db, pn2_stream_t stream);

/* BatchNorm+ReLU backward statistics of the top layer of a stack: partial
 * [pn2_bn_bwd_reduce_partials(rows)][2][C] sums of gh and gh*xh over rows (rows = M, or the
 * M/pool_k pooled rows with argk).  pn2_bn_bwd_finalize turns partials (from here or from the
 * backward epilogue of pn2_mlp_gemm) into dgamma, dbeta, c1 = dbeta/count, c2 = dgamma/count. */
int pn2_bn_bwd_reduce_partials(long long rows);
int pn2_bn_bwd_reduce(const float *g, int ldg, const float *z, int ldz, long long rows, int C,
                      const unsigned char *argk, int pool_k, const float *scale, const float *shift,
                      const float *mean, const float *invstd, float *partial, pn2_stream_t stream);
int pn2_bn_bwd_finalize(const float *partial, int P, int C, double count, float *dgamma, float *dbeta,
                        float *c1, float *c2, pn2_stream_t stream);

/* dst[r][c] = c < cols_src ? src[r][c] : 0, c < cols_dst, r < rows (pitches lds, ldd): pads the first conv weight of
 * a stack to the 16-byte aligned row width of its input, and slices its gradient back (cols_dst < cols_src). */
int pn2_copy_pad_cols(const float *src, int lds, int cols_src, float *dst, int ldd, int cols_dst, long long rows,
                      pn2_stream_t stream);

/* ---- transposed index tables: atomic-free, order-fixed backward of the gather operators ---------------
 * The autograd of index_points / grouping (models/pointnet2_utils.py:43-60, :127-132) and of the 3-NN
 * interpolation (:296-303) is a scatter-add through idx.  pn2_invert_index turns idx [B][E] (values in
 * [0, Nkeys); others are dropped) into per-key lists: offsets [B][Nkeys+1], entries [B][E] (entry numbers,
 * ascending inside a list, -1 behind the last list).  E <= 24576, Nkeys <= 8192 per batch, else
 * PN2_ERR_UNSUPPORTED (callers keep the scatter-add operators).
 * pn2_gather_sum: out[b][key][c] = sum_{e in list(b,key)} w[b][e] * src[b][e / ediv][col0 + c], c < D
 * (weight nullable = 1; src rows [B][rows_src][lds]).  Grouping: src = d grouped [B][S*K][ldg], col0 = 3,
 * ediv = 1.  Interpolation: src = d out [B][N][D], weight = weight3 [B][N*3], ediv = 3. */
int pn2_invert_index(const int64_t *idx, int B, long long E, int Nkeys, int32_t *offsets, int32_t *entries,
                     pn2_stream_t stream);
/* n <= 8 tables of the same batch size in one launch (host arrays of device pointers / sizes, read before return). */
int pn2_invert_index_many(int n, const int64_t *const *idx, int B, const long long *E, const int *Nkeys,
                          int32_t *const *offsets, int32_t *const *entries, pn2_stream_t stream);
int pn2_gather_sum(const float *src, long long rows_src, int lds, int col0, const int32_t *offsets,
                   const int32_t *entries, const float *weight, long long E, int ediv, int B, int Nkeys, int D,
                   float *out, pn2_stream_t stream);
/* ... + addend[b][key][c] (nullable): the other gradient of the same rows (a skip connection's) joins the sum. */
int pn2_gather_sum_add(const float *src, long long rows_src, int lds, int col0, const int32_t *offsets,
                       const int32_t *entries, const float *weight, long long E, int ediv, int B, int Nkeys, int D,
                       const float *addend, float *out, pn2_stream_t stream);

/* ---- segmentation head tail and loss (caller of the hot path) ------------------------
 * x = conv2(x); x = F.log_softmax(x, dim=1)                   models/pointnet2_sem_seg.py:37-38
 * logp[M][C] = log_softmax(y[M][K] * w[C][K]^T + bias[C]) per row.  K <= 128, K % 4 == 0, C <= 32. */
int pn2_head_logits(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K,
                    int C, pn2_stream_t stream);
/* autograd of the above: from glogp [M][C] -> gy [M][K] (nullable), dw [C][K], db [C] (nullable).
 * partial: workspace [pn2_head_logits_partials(M)][C][K+1] floats. */
int pn2_head_logits_partials(int M);
int pn2_head_logits_backward(const float *glogp, const float *logp, const float *y, int ldy, const float *w,
                             float *gy, int ldgy, float *partial, float *dw, float *db, int M, int K, int C,
                             pn2_stream_t stream);
/* The same two with the dropout of models/pointnet2_sem_seg.py:36 (nn.Dropout(0.5) between relu(bn1(conv1)) and conv2)
 * applied to y on the fly: an element (row, column) is kept with probability 1 - drop_p (kept values scaled by
 * 1/(1-drop_p)), the keep-mask being a counter-based hash of (*drop_seed, row, column) that forward and backward
 * regenerate instead of storing it.  drop_seed: one 64-bit word in device memory, read when the kernel runs (so a
 * captured graph sees a fresh value per replay); NULL or drop_p == 0: no dropout.  gy is the gradient w.r.t. the
 * un-dropped y.  pn2_dropout_mask writes that mask ([M][K] bytes, 1 = kept) for tests. */
int pn2_head_logits_dropout(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K,
                            int C, const unsigned long long *drop_seed, float drop_p, pn2_stream_t stream);
/* The forward with a COUNTED seed: state[0] = base seed, state[1] = calls so far, state[2] = 0 (ticket), all on the device; the
 * kernel hashes its seed from (base, calls), writes it to *seed_out (what the backward takes as drop_seed) and its last
 * workgroup counts the call.  No random-number launch in front of the head, and a captured step draws nothing from torch's
 * generator (a replay of a graph that does is preceded by two fill launches for the generator's state). */
int pn2_head_logits_dropout_counted(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K,
                                    int C, unsigned long long *state, unsigned long long *seed_out, float drop_p,
                                    pn2_stream_t stream);
int pn2_head_logits_dropout_backward(const float *glogp, const float *logp, const float *y, int ldy, const float *w,
                                     float *gy, int ldgy, float *partial, float *dw, float *db, int M, int K, int C,
                                     const unsigned long long *drop_seed, float drop_p, pn2_stream_t stream);
int pn2_dropout_mask(const unsigned long long *drop_seed, float drop_p, long long M, int K, unsigned char *mask,
                     pn2_stream_t stream);
/* F.nll_loss(pred, target, weight=weight)  (reduction 'mean')   models/pointnet2_sem_seg.py:48
 * loss = sum_i -w[t_i] logp[i][t_i] / sum_i w[t_i] over rows with t_i != ignore_index; wsum receives the
 * denominator for the backward.  weight nullable (= ones).  A target outside [0,C): row skipped,
 * err_count += 1 (nullable).  partial: workspace [pn2_nll_loss_partials(M)][2] doubles. */
int pn2_nll_loss_partials(long long M);
int pn2_nll_loss(const float *logp, const int64_t *target, const float *weight, long long M, int C,
                 long long ignore_index, double *partial, float *loss, float *wsum, int32_t *err_count,
                 pn2_stream_t stream);
/* The same in ONE launch: the workgroup that finishes last sums the partials (in index order).  ticket: one device word
 * the caller zero-initialised once and otherwise leaves alone (the kernel returns it to zero); one launch at a time
 * per ticket word. */
int pn2_nll_loss_ticketed(const float *logp, const int64_t *target, const float *weight, long long M, int C,
                          long long ignore_index, double *partial, float *loss, float *wsum, int32_t *err_count,
                          unsigned int *ticket, pn2_stream_t stream);
int pn2_nll_loss_backward(const float *gloss, const int64_t *target, const float *weight, const float *wsum,
                          long long M, int C, long long ignore_index, float *glogp, pn2_stream_t stream);

/* ---- optimiser of the reference loop on one flat buffer ------------------------------------------------
 * torch.optim.Adam(lr, betas, eps, weight_decay) (sem_seg_training.py:576-582) over n fp32 parameters that are
 * views of `param`, gradients packed in `grad` (multiplied by grad_scale first: 1/world after an all-reduce sum).
 * lr [1] and state [3] (step count, 1-beta1^t, sqrt(1-beta2^t); zero-initialised) live on the device, so the two
 * launches can be replayed from a hipGraph. */
int pn2_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const float *lr,
                  float *state, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                  pn2_stream_t stream);

/* The same update reading every parameter tensor's gradient where backward left it: grads[i] (HOST array of n_tensors
 * device pointers; NULL = zero gradient) belongs to elements offsets[i] .. offsets[i+1] of `param` (HOST array of
 * n_tensors + 1 ascending element offsets, offsets[0] = 0).  No packing pass; the pointers travel in the kernel
 * arguments, so a captured launch keeps them.  n_tensors <= 192, fewer than 2^32 elements, else PN2_ERR_UNSUPPORTED. */
int pn2_adam_step_scattered(float *param, int n_tensors, const float *const *grads, const long long *offsets,
                            float *exp_avg, float *exp_avg_sq, const float *lr, float *state, double beta1, double beta2,
                            double eps, double weight_decay, double grad_scale, pn2_stream_t stream);

/* ---- whole-scene inference aggregation --------------------------------------
 * add_vote(vote_label_pool, point_idx, pred_label, weight)            localfunctions.py:339-346
 * vote_pool[P][C] int32 += 1 at (point_idx[m], label[m]) for every m < M whose weight is neither 0
 * nor inf (weight nullable = all ones).  label = arg-max over the C log-probabilities of row m of
 * logp (first maximum wins, the `seg_pred.max(2)[1]` of localfunctions.py:399) or, when logp is
 * NULL, pred_label[m].  Out-of-range point index / label: skipped, err_count += 1. */
int pn2_add_vote(const float *logp, const int64_t *pred_label, const int64_t *point_idx, const float *weight,
                 long long M, int C, long long P, int32_t *vote_pool, int32_t *err_count, pn2_stream_t stream);

/* ---- loop glue on the device --------------------------------------------------------
 * pn2_input_blocks: the input preparation of a training step (localfunctions.py:205-209) in one pass: per block b
 * the xyz columns are rotated about the up axis by angles[b] radians (provider.rotate_point_cloud_z,
 * provider.py:66-84; angles NULL = no rotation) and the batch is laid out for the network.  in: [B][C][N] when
 * channel_first != 0 (what the loop hands to the classifier) else [B][N][C]; pts [B][N][C] channel-last rows;
 * xyz (nullable) [B][N][3].  C >= 3.
 * pn2_seg_metrics: accuracy / IoU bookkeeping of a batch (localfunctions.py:214, 220-223, 271-283) added to
 * int64 device counters [2 + 3*C]: [0] correct, [1] seen, [2+c] label == c, [2+C+c] pred == c && label == c,
 * [2+2C+c] pred == c || label == c; pred = arg-max of the row of logp [M][C] (first maximum wins).  C <= 64. */
/* Sliding-window tiler of whole-scene inference: TestCustomDataset.__getitem__ (sem_seg_testing.py:182-254) on a scene
 * resident on the device, bucketed like pn2_sample_blocks' (order / cell_start over an nx x ny grid of `cell`-sized
 * cells with origin x0, y0).
 * pn2_tile_windows: for each of W closed windows [xmin, xmax] x [ymin, ymax] (windows [W][4] doubles, padding included;
 *   the np.where of :202-203) either the number of points inside (members == NULL: counts [W]) or the points themselves
 *   (members + member_off[w], in a deterministic order: grid rows in order, cell order inside a row).
 * pn2_tile_fill: the blocks of all windows.  Window w owns blocks block_off[w] .. block_off[w+1] (ceil(count / block_points)
 *   of them, :205-206); its slots take the members plus a random top-up of the members (without replacement while the
 *   top-up is at most the population, :209-210), in random order (:212) -- keyed pseudo-random permutations of `seed`,
 *   or, with srcpos [slots], the member position given per slot (the caller replays numpy's choice / shuffle stream:
 *   bit-identical blocks).  data [blocks][block_points][6+E] = [x - cx, y - cy, z, xyz / coord_max, extra] (:214-239,
 *   double arithmetic rounded to float once), labels, labelweights[label] (NULL: 1) and the point indices. */
int pn2_tile_windows(const double *xyz, const int *order, const int *cell_start, double x0, double y0, double cell, int nx,
                     int ny, const double *windows, int W, const long long *member_off, int *counts, int *members,
                     pn2_stream_t stream);
int pn2_tile_fill(const double *xyz, const float *extra, const long long *labels, const float *labelweights, int P, int E,
                  int num_classes, const double *coord_max, const int *members, const long long *member_off,
                  const int *counts, const double *centre, const long long *block_off, int W, long long blocks,
                  int block_points, const int *srcpos, unsigned long long seed, float *data, long long *out_labels,
                  float *out_weight, long long *out_index, pn2_stream_t stream);

/* pn2_sample_blocks: TrainCustomDataset.__getitem__ (sem_seg_training.py:200-259) for B blocks of one scene that
 * lives on the device: xyz [P][3] double; order [P] / cell_start [nx*ny+1] = the points bucketed into a 2-D grid of
 * `cell`-sized cells from (x0, y0), row-major, ascending index inside a cell; extra [E][P] (nullable) the extra
 * feature columns already scaled; labels [P]; coord_max [3] HOST doubles (room_coord_max).  Per block: a random
 * point as centre, re-drawn until the block_size column around it holds more than min_points points; num_point of
 * them, a uniformly random subset in random order (or, when fewer, uniform draws with replacement); feats
 * [B][num_point][6+E] = [x - cx, y - cy, z, xyz / coord_max, extra], out_labels [B][num_point], info [B][4] =
 * (centre index, points in the window, attempts, 1 if no such column was found in 256 attempts: zero block);
 * sel_idx (nullable) [B][num_point] the chosen point indices.
 * All randomness derives from `seed`: same seed, same blocks.  num_point <= 4096. */
int pn2_sample_blocks(const double *xyz, const int *order, const int *cell_start, const float *extra,
                      const long long *labels, double x0, double y0, double cell, int nx, int ny, int P, int E,
                      double block_size, const double *coord_max, int num_point, int min_points, unsigned long long seed,
                      int B, float *feats, long long *out_labels, int *info, int *sel_idx, pn2_stream_t stream);

/* pn2_sample_blocks over SEVERAL scenes in one launch: the reference's loader mixes rooms inside a batch
 * (sem_seg_training.py:184-193).  rooms = device table of nrooms 104-byte descriptors {xyz, order, cell_start, extra,
 * labels (pointers); x0, y0, cell, max_x, max_y, max_z (double); nx, ny, P, pad (int)} -- the per-scene arguments of
 * pn2_sample_blocks; room_of_block [B] int32 = the room each output block is drawn from: a DEVICE array, or with
 * room_ids_on_host != 0 a HOST array that travels in the launch's arguments (B <= 256, nrooms <= 256; no upload in
 * front of the step).  E, block_size, num_point, min_points are common.  Same outputs as pn2_sample_blocks. */
int pn2_sample_blocks_multi(const void *rooms, int nrooms, const int *room_of_block, int room_ids_on_host, int E,
                            double block_size, int num_point, int min_points, unsigned long long seed, int B, float *feats,
                            long long *out_labels, int *info, int *sel_idx, pn2_stream_t stream);
int pn2_input_blocks(const float *in, int channel_first, int B, int N, int C, const float *angles, float *pts, float *xyz,
                     pn2_stream_t stream);
int pn2_seg_metrics(const float *logp, const int64_t *target, long long M, int C, long long *counters, pn2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_HIP_H */
