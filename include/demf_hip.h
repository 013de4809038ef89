/*
 * demf_hip.h — C ABI of libdemf_hip.so: the MI355X (gfx950) kernels behind the
 * DeMF fusion hot path.
 *
 * Every entry point replaces one native operator that the reference
 * (haoy945/DeMF) reaches through mmdet3d.ops / mmcv.ops.  The reference has no
 * native code of its own; each declaration cites the reference call site that
 * consumes the operator (file:line under the reference tree) and the upstream
 * operator it stands in for.
 *
 * Conventions (all entry points):
 *   - All pointers are DEVICE pointers on the current HIP device unless the
 *     name says otherwise.  Tensors are dense, row-major, fp32 / int32 / int64.
 *   - The caller owns every buffer.  The library never allocates, frees or
 *     synchronises; work is enqueued on `stream` (graph-capture safe).
 *   - Buffers documented "accumulated" must arrive zero-filled.
 *   - Return 0 on success; a negative DEMF_E* code otherwise.  A human-readable
 *     message for the calling thread is available from demf_last_error().
 *   - Re-entrant; no global mutable state besides the thread-local error text and a device-side ring
 *     of self-resetting tile counters (256 KB) that the persistent shared-MLP GEMM launches claim
 *     their row tiles from: a launch is handed the next counter set of the ring and leaves it zeroed,
 *     so captured launches replay with the set they were given.  Two launches share a set only if
 *     a multiple of 1024 counter-using launches lies between their enqueues; that matters only if
 *     they also run concurrently on different streams (the path itself is single-stream;
 *     DEMF_STATIC_TILES=1 switches the counters off).
 */
#ifndef DEMF_HIP_H_
#define DEMF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging the HIP headers into the consumer. */
typedef void* demf_stream_t;

#define DEMF_OK 0
#define DEMF_EINVAL (-1)   /* bad argument (sizes, null pointer)            */
#define DEMF_ELAUNCH (-2)  /* hipGetLastError() reported a launch failure   */
#define DEMF_EUNSUPPORTED (-3)

#define DEMF_ABI_VERSION 2

/* Two tiers of entry points.
 *   PUBLIC (unmarked): the operators with an upstream counterpart that the reference reaches through mmdet3d.ops /
 *     mmcv.ops - furthest_point_sample, ball_query, group_points, gather_points, three_nn, three_interpolate,
 *     ms_deform_attn - with upstream's argument lists (+ stream), plus demf_version / demf_last_error.  These are
 *     what a maintainer of the reference binds (INTEGRATION.md sections 1-2) and what stays stable.
 *   DEMF_INTERNAL: everything else - fused layer kernels, layout-specific variants, scratch-size queries, the step
 *     engine's helpers.  They are exported (the Python host layer of this package calls them through ctypes) and
 *     declared here so that tests/test_abi.py can hold header, exports and binding table against each other, but
 *     they are shaped by THIS package's modules (one decoder layer, one stack shape) and may change with them.   */
#define DEMF_INTERNAL

int demf_version(void);
const char* demf_last_error(void);

/* A stream restricted to the listed CUs (invert != 0: to all CUs except them); used to keep the
 * pipelined FPS pre-pass and the main training stream on disjoint CUs.  Host-side helper: `cus`
 * and `out` are HOST pointers; the stream is owned by the caller (hipStreamDestroy).        */
DEMF_INTERNAL int demf_stream_create_cu_masked(const int* cus, int n, int invert, void** out);

/* One wave that spins for `microseconds` of wall-clock time on `stream` and touches no memory: a
 * stand-in of known length for the gradient all-reduce (reference: MMDistributedDataParallel's NCCL
 * all-reduce, /root/reference/train.py:56-63) when the engine's communication / compute overlap is
 * measured on a single GPU (bench.py --allreduce-stub-us).                                        */
DEMF_INTERNAL int demf_spin_us(int microseconds, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * PointNet++ set-abstraction operators
 * ------------------------------------------------------------------ */

/* furthest_point_sample(points_xyz (B,N,3), num_points M) -> idx (B,M) i32.
 * Reference call sites: demf/modeling/heads/class_agnostic_vote_head.py:429-430
 * (direct) and every PointSAModule of the backbone / vote aggregation
 * (configs/demf/demf_votenet.py:48-62,155-162).  Stands in for mmdet3d.ops
 * furthest_point_sample.  idx[b,0] = 0; ties resolve exactly as the upstream
 * block reduction does (see DESIGN.md "canonical arithmetic").
 * `temp` is the (B,N) scratch the upstream ABI carries; it may be NULL for 64 <= N <= 24576.
 * When it is given (>= B ints) and N <= 4M <= 4096, a dependency-free check first tests whether
 * the cloud already is in FPS order (as every SA level after the first is) and, if so, answers
 * 0..M-1 without running the M-round chain; ties fall back to the chain, results are identical. */
int demf_fps_f32(int B, int N, int M, const float* xyz, float* temp, int* idx,
                 demf_stream_t stream);
/* The same with the scratch size stated (floats): with >= B * (M + 2) floats the ordered-input check runs spread
 * over the chip (two launches, one workgroup per 256 points) instead of on one compute unit per scene - 162 -> ~25 us
 * at 2 048 -> 1 024, on the serial pre-pass chain.  Identical results.                                   */
DEMF_INTERNAL int demf_fps_ws_f32(int B, int N, int M, const float* xyz, float* temp, long long temp_floats, int* idx,
                    demf_stream_t stream);

/* ball_query(min_radius, max_radius, sample_num, xyz (B,N,3), center (B,M,3))
 * -> idx (B,M,nsample) i32.  Reference: QueryAndGroup inside build_sa_module
 * (class_agnostic_vote_head.py:383,455; config radii demf_votenet.py:51-53,158).
 * First `nsample` hits in index order; first hit pre-fills all slots; a centre
 * with no hit yields zeros.                                                  */
int demf_ball_query_f32(int B, int N, int M, float min_radius, float max_radius,
                        int nsample, const float* center_xyz, const float* xyz,
                        int* idx, demf_stream_t stream);
/* The same result (min_radius = 0) for large clouds, N <= 32768, through a hashed uniform grid built on
 * the fly (two launches): every centre only meets the points of the 2x2x2 cells of edge 2r its ball can
 * reach instead of all N; the hit set and the index order are the scan's (csrc/ball_query.hip).
 * Workspace (device): ws_start = *start_ints ints, ws_cells = *cell_floats floats (16-byte aligned), the
 * sizes from demf_ball_query_grid_ws.                                                              */
DEMF_INTERNAL int demf_ball_query_grid_ws(int B, int N, long long* start_ints, long long* cell_floats);
DEMF_INTERNAL int demf_ball_query_grid_f32(int B, int N, int M, float max_radius, int nsample,
                             const float* center_xyz, const float* xyz, int* idx,
                             int* ws_start, float* ws_cells, demf_stream_t stream);

/* grouping_operation: features (B,C,N), idx (B,M,ns) -> out (B,C,M,ns).
 * bwd: grad_out (B,C,M,ns) -> grad_features (B,C,N), accumulated.            */
int demf_group_points_fwd(int B, int C, int N, int M, int ns, const float* features,
                          const int* idx, float* out, demf_stream_t stream);
int demf_group_points_bwd(int B, int C, int N, int M, int ns, const float* grad_out,
                          const int* idx, float* grad_features, demf_stream_t stream);

/* gather_points: features (B,C,N), idx (B,M) -> out (B,C,M).
 * bwd: grad_out (B,C,M) -> grad_features (B,C,N), accumulated.               */
int demf_gather_points_fwd(int B, int C, int N, int M, const float* features,
                           const int* idx, float* out, demf_stream_t stream);
int demf_gather_points_bwd(int B, int C, int N, int M, const float* grad_out,
                           const int* idx, float* grad_features, demf_stream_t stream);

/* three_nn(target (B,n,3), source (B,m,3)) -> dist2 (B,n,3) SQUARED distances
 * (the Python wrapper applies sqrt, as upstream does), idx (B,n,3) i32.
 * Reference: PointFPModule of the backbone (demf_votenet.py:56).             */
int demf_three_nn_f32(int B, int n, int m, const float* target, const float* source,
                      float* dist2, int* idx, demf_stream_t stream);
/* the same search + what PointFPModule.forward derives from it (mmdet3d PointFPModule: dist_recip =
 * 1 / (dist + 1e-8), weight = dist_recip / sum(dist_recip)): dist (B,n,3) = sqrt of the squared
 * distances, idx (B,n,3), weight (B,n,3) - one launch instead of the search + 6 element-wise ones. */
DEMF_INTERNAL int demf_three_nn_weights_f32(int B, int n, int m, const float* target, const float* source,
                              float* dist, int* idx, float* weight, demf_stream_t stream);

/* three_interpolate: features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n).
 * bwd: grad_out (B,C,n) -> grad_features (B,C,m), accumulated.               */
int demf_three_interpolate_fwd(int B, int C, int m, int n, const float* features,
                               const int* idx, const float* weight, float* out,
                               demf_stream_t stream);
int demf_three_interpolate_bwd(int B, int C, int n, int m, const float* grad_out,
                               const int* idx, const float* weight,
                               float* grad_features, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Channels-last (point-major) fused variants used by the MI355X modules.
 * Same arithmetic as the operators above; layout chosen so that a gathered
 * neighbour is one contiguous C*4-byte row.
 * ------------------------------------------------------------------ */

/* QueryAndGroup fused (ball-query indices -> grouped MLP input), one row per
 * neighbour:  out[b,m,s, xyz_col:xyz_col+3]  = (xyz[b,idx]-center[b,m]) / radius
 * (no division when normalize_xyz == 0), out[b,m,s, feat_col:feat_col+C] =
 * feat[b,idx,:]; every other column of the ldo-wide row is written as zero.
 * xyz (B,N,3), center (B,M,3), feat (B,N,C) or NULL (C=0), idx (B,M,ns),
 * out (B,M,ns,ldo).  The two column ranges must not overlap.                 */
DEMF_INTERNAL int demf_group_concat_cl_fwd(int B, int N, int M, int ns, int C, int ldo,
                             int xyz_col, int feat_col, float radius,
                             int normalize_xyz, const float* xyz,
                             const float* center, const float* feat, const int* idx,
                             float* out, demf_stream_t stream);
/* bwd: grad_out (B,M,ns,ldo) -> grad_feat (B,N,C), grad_xyz (B,N,3), grad_center
 * (B,M,3), all accumulated.  grad_feat may be NULL; grad_xyz/grad_center may both be
 * NULL (backbone levels: raw coordinates carry no gradient; the vote aggregation of
 * class_agnostic_vote_head.py:455 needs them - vote_points are learned).        */
DEMF_INTERNAL int demf_group_concat_cl_bwd(int B, int N, int M, int ns, int C, int ldo,
                             int xyz_col, int feat_col, float radius,
                             int normalize_xyz, const float* grad_out, const int* idx,
                             float* grad_feat, float* grad_xyz, float* grad_center,
                             demf_stream_t stream);

/* rows gather: feat (B,N,C), idx (B,M) -> out (B,M,C); bwd accumulated.      */
DEMF_INTERNAL int demf_gather_rows_cl_fwd(int B, int N, int M, int C, const float* feat,
                            const int* idx, float* out, demf_stream_t stream);
DEMF_INTERNAL int demf_gather_rows_cl_bwd(int B, int N, int M, int C, const float* grad_out,
                            const int* idx, float* grad_feat, demf_stream_t stream);

/* three_nn + inverse-distance weights + interpolate fused, channels-last:
 * feat (B,m,C), idx/weight (B,n,3) -> out (B,n,ldo) columns [col0, col0+C).  */
DEMF_INTERNAL int demf_three_interpolate_cl_fwd(int B, int m, int n, int C, int ldo, int col0,
                                  const float* feat, const int* idx,
                                  const float* weight, float* out,
                                  demf_stream_t stream);
DEMF_INTERNAL int demf_three_interpolate_cl_bwd(int B, int m, int n, int C, int ldo, int col0,
                                  const float* grad_out, const int* idx,
                                  const float* weight, float* grad_feat,
                                  demf_stream_t stream);
/* PointFPModule.forward's interpolate + torch.cat([interpolated, target_feats]) in one launch:
 * out (B,n,C+Cs) = [ three_interpolate(feat (B,m,C)) | skip (B,n,Cs) ].  Backward: the interpolated
 * columns through demf_three_interpolate_cl_bwd(ldo = C+Cs, col0 = 0); the skip gradient is the
 * column range [C, C+Cs) of the incoming gradient itself.                                        */
DEMF_INTERNAL int demf_three_interpolate_cat_cl_fwd(int B, int m, int n, int C, int Cs, const float* feat,
                                      const int* idx, const float* weight, const float* skip,
                                      float* out, demf_stream_t stream);

/* max over the ns neighbours: x (R, ns, C) -> out (R, C), arg (R, C) i32
 * (first maximum wins, as torch max_pool2d does).  bwd scatters to x grad.   */
DEMF_INTERNAL int demf_maxpool_ns_fwd(int R, int ns, int C, const float* x, float* out, int* arg,
                        demf_stream_t stream);
DEMF_INTERNAL int demf_maxpool_ns_bwd(int R, int ns, int C, const float* grad_out, const int* arg,
                        float* grad_x /* (R,ns,C), fully written */,
                        demf_stream_t stream);

/* Inverse neighbour lists of a ball-query result: idx (B,E) int32 with values in [0,N), E = M*ns
 * -> CSR by source point: off (B,N+1), rows (B,E) = entry positions e = m*ns+s, ascending within a
 * list.  N <= 16384.  Coordinate-only (QueryAndGroup's idx, class_agnostic_vote_head.py:383). */
DEMF_INTERNAL int demf_invert_index(int B, int N, int E, const int* idx, int* off, int* rows, demf_stream_t stream);
/* The same with B * E ints of workspace: for E >= 32 768 entries per scene (SA1: the lists do not fit one workgroup's
 * LDS) the inversion runs spread over the chip - histogram, scan, fill, rank (five launches) - instead of one
 * workgroup per scene with a global insertion sort (102 -> ~35 us).  Identical output.                     */
DEMF_INTERNAL int demf_invert_index_ws(int B, int N, int E, const int* idx, int* off, int* rows, int* workspace,
                         demf_stream_t stream);

/* sa_indices of the backbone (mmdet3d PointNet2SASSG.forward as used by demf/modeling: every level's samples as
 * indices into the INPUT cloud): out[0] (B,N) = arange(N), out[l] (B, samples[l-1]) = out[l-1] gathered by the
 * level's FPS indices idx[l-1] (B, samples[l-1]) int32; int64 outputs as the reference returns them.  nlev <= 8. */
DEMF_INTERNAL int demf_sa_index_chain(int B, int N, int nlev, const int* const* idx, const int* samples, int64_t* const* out,
                        demf_stream_t stream);
/* points (rows, 3 + C) -> xyz (rows, 3) | feat (rows, C): `points[..., :3]`, `points[..., 3:]` of the backbone
 * input as contiguous tensors, one launch. */
DEMF_INTERNAL int demf_split_points(long long rows, int C, const float* points, float* xyz, float* feat, demf_stream_t stream);

/* grad_feat (B,N,C) of demf_group_concat_cl_fwd through the inverse lists: every source point sums
 * the grad_out rows (B,E,ldo)[.., feat_col:feat_col+C] that gathered it.  No atomics; grad_feat is
 * fully written (need not arrive zeroed).  C % 4 == 0.  Same result as demf_group_concat_cl_bwd's
 * grad_feat up to summation order. */
DEMF_INTERNAL int demf_group_concat_cl_bwd_gather(int B, int N, int E, int C, int ldo, int feat_col,
                                    const float* grad_out, const int* off, const int* rows,
                                    float* grad_feat, demf_stream_t stream);

/* First 1x1 convolution of a set-abstraction level without the grouped tensor (PointSAModule built by
 * build_sa_module, class_agnostic_vote_head.py:383,455; QueryAndGroup with use_xyz / normalize_xyz).
 * The grouped row is [(xyz_j - centre)/radius | feat_j]; with U (B*N, C1) = feat . Wf^T computed per
 * SOURCE point and Wx (3, C1) the xyz columns of the weight (k-major),
 *   Y[b,m,s,:] = U[b, idx[b,m,s], :] + rel(b,m,s) . Wx          (rows (B*M*ns, C1), fully written)
 * stats (2*C1 fp64: column sum | sum of squares, accumulated, as demf_mlp_gemm_fwd) or NULL.
 * C1 in {64,128,256}. */
DEMF_INTERNAL int demf_group_first_fwd(int B, int N, int M, int ns, int C1, float radius, int normalize_xyz,
                         const float* xyz, const float* center, const int* idx, const float* U,
                         const float* Wx, int w_ld /* 0: Wx is the (3, C1) copy; > 0: Wx is the layer's
                         weight (C1 x w_ld row-major) itself, columns 0..2 read in place */,
                         float* Y, double* stats,
                         /* train-mode BatchNorm bookkeeping of the layer (demf_bn_finalize's arguments) done by the
                          * launch's last workgroup - or, without a counter set, by a demf_bn_finalize launch behind it;
                          * scale_shift == NULL: none, the sums stay in stats */
                         const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* scale_shift, float* mean_invstd,
                         const float* conv_bias, demf_stream_t stream);
/* Backward of the above through the inverse lists of demf_invert_index: with dY the BN-backward
 * transform of (G, Y) by vec6 (demf_bn_bwd_vectors; same formula as demf_mlp_gemm_bwd_dx),
 *   dU[b,j,:]  = sum of dY over the rows that gathered point j        (fully written, no atomics)
 *   dWx[k,:]  += sum over all rows of rel_k . dY                      (3*C1 fp32, arrives zeroed)
 * and, when the coordinates carry a gradient (the vote aggregation), d rel = dY . Wx^T per row:
 *   dxyz[b,j,:] = + sum over the rows of point j / radius,  dcenter[b,m,:] -= sum over s / radius */
DEMF_INTERNAL int demf_group_first_bwd(int B, int N, int M, int ns, int C1, float radius, int normalize_xyz,
                         const float* xyz, const float* center, const float* G, const float* Y,
                         const float* U /* the forward's U (B*N, C1): y is formed again from it instead of being read
                         (half the bytes; needs Wx), or NULL: y read from Y */,
                         const float* vec6, const int* inv_off, const int* inv_rows, float* dU,
                         float* dWx, int dw_ld /* 0: (3, C1) layout; > 0: dWx[c*dw_ld + k], i.e. columns
                         0..2 of the (C1 x dw_ld) weight gradient */,
                         const float* Wx /* needed with dxyz or U */, int w_ld /* as in the forward */,
                         float* dxyz /* (B,N,3) fully written, or NULL */,
                         float* dcenter /* (B,M,3) accumulated: arrives zeroed, or NULL */,
                         double* wacc /* 8*3*C1 doubles, zero on entry and left zeroed: the workgroups' partial dWx are
                         summed there (contiguous atomics) and the last workgroup adds the totals to dWx; or NULL:
                         every workgroup adds to dWx itself */,
                         demf_stream_t stream);

/* One pyramid level (B,C,HW) channel-major -> rows [row0,row0+HW) of the channels-last token buffer
 * (B,S,C): the flatten + transpose + concat of prepare_decoder_inputs
 * (class_agnostic_vote_head.py:570-591) as a tiled transpose.  mask (B,S) bytes or NULL: tokens
 * with a non-zero mask byte (image padding) are written as zeros.                               */
DEMF_INTERNAL int demf_nchw_to_tokens(int B, int C, int HW, int S, int row0, const float* src,
                        const unsigned char* mask, float* dst, demf_stream_t stream);
/* Every level of the pyramid in one launch: srcs[l] (B,C,hws[l]) channel-major -> consecutive row
 * ranges of dst (B,S,C), S = sum hws (nlev <= 8; srcs / hws are HOST arrays of device pointers / sizes). */
DEMF_INTERNAL int demf_pyramid_to_tokens(int B, int C, int S, int nlev, const float* const* srcs, const int* hws,
                           const unsigned char* mask, float* dst, demf_stream_t stream);
/* The same transposes into bf16 token rows (2-byte elements, round to nearest even; C % 4 == 0): what
 * demf_msda_{fwd,bwd}_bf16 gather from - BASELINE configs[3], half the bytes of the 152 MB token tensor. */
DEMF_INTERNAL int demf_pyramid_to_tokens_bf16(int B, int C, int S, int nlev, const float* const* srcs, const int* hws,
                                const unsigned char* mask, uint16_t* dst, demf_stream_t stream);

/* out (N) += column sums of x (R,N; row stride ld).  out arrives zeroed.  The bias gradient of the
 * path's linear layers (mmcv FFN / MultiheadAttention / MultiScaleDeformableAttention projections,
 * transformer.py:73; conv_cls / conv_reg, class_agnostic_vote_head.py:398) - replaces
 * at::sum's two-stage semaphore reduction, which mis-reduces inside hipGraph replays. */
DEMF_INTERNAL int demf_colsum_f32(int R, int N, int ld, const float* x, float* out, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Fused shared-MLP (1x1 conv + train-mode BatchNorm + ReLU [+ max over
 * neighbours]) on point-major rows: the dense half of every PointSAModule
 * (class_agnostic_vote_head.py:383; demf_votenet.py:48-62).  fp32 MFMA.
 * act(y) = max(0, y*scale + shift) with scale = gamma*invstd, shift = beta - mean*scale.
 * ------------------------------------------------------------------ */

/* Y (R,N) = pro(X) (R,K; row stride ldx) @ Wt (N,K)^T.  pro = act of the PREVIOUS layer when
 * pro_scale_shift ([scale(K)|shift(K)]) is non-NULL, identity otherwise.  When stats is
 * non-NULL the per-column sum and sum of squares of Y are ACCUMULATED into stats[0:N],
 * stats[N:2N] (fp64).  K%4==0, ldx%4==0, N%32==0, N<=256.                        */
DEMF_INTERNAL int demf_mlp_gemm_fwd(int R, int K, int N, int ldx, const float* X,
                      const float* pro_scale_shift, const float* Wt, float* Y,
                      double* stats, demf_stream_t stream);

/* demf_mlp_gemm_fwd with the max-pool of the PointSAModule fused into the epilogue: per group of
 * ns consecutive rows and column, pmax/pmin (R/ns,N) = max/min of the raw output and amax/amin
 * their row offsets (first on ties).  max_s relu(sc*y_s+sh) = relu(sc*(sc>0 ? pmax : pmin)+sh), so
 * the pooled activation follows from these once the batch statistics are known
 * (demf_pool_select) without re-reading Y.  ns 16, 32 or 64; DEMF_EUNSUPPORTED otherwise.                                                                   */
DEMF_INTERNAL int demf_mlp_gemm_fwd_pool(int R, int K, int N, int ldx, const float* X,
                           const float* pro_scale_shift, const float* Wt, float* Y, double* stats,
                           int ns, float* pmax, float* pmin, int* amax, int* amin,
                           demf_stream_t stream);

/* demf_mlp_gemm_fwd / demf_mlp_gemm_fwd_pool followed by demf_bn_finalize(N, count = R, stats, ...)
 * as ONE launch: the last workgroup of the GEMM turns the column sums into scale_shift,
 * mean_invstd and the running statistics (torch.nn.BatchNorm semantics, see demf_bn_finalize) and
 * leaves `stats` zeroed.  The following layer's launch reads scale_shift as its prologue.
 * demf_mlp_gemm_fwd_pool_bn with pmin == amin == NULL: the sign of the BN scale is the sign of gamma,
 * so only the extremum the consumer will select is reduced - pmax / amax then hold, per group and
 * column, the maximum of y for gamma >= 0 and the minimum otherwise (pass them to demf_pool_select
 * as both the max and the min operands).                                                            */
DEMF_INTERNAL int demf_mlp_gemm_fwd_bn(int R, int K, int N, int ldx, const float* X, const float* pro_scale_shift,
                         const float* Wt, float* Y, double* stats, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* scale_shift,
                         float* mean_invstd, const float* conv_bias, demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_gemm_fwd_pool_bn(int R, int K, int N, int ldx, const float* X,
                              const float* pro_scale_shift, const float* Wt, float* Y, double* stats,
                              int ns, float* pmax, float* pmin, int* amax, int* amin,
                              const float* gamma, const float* beta, float eps, float momentum,
                              float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* scale_shift,
                              float* mean_invstd, const float* conv_bias, demf_stream_t stream);

/* out (Rp,C) = relu(scale*y*+shift), arg = row offset of y* (see demf_mlp_gemm_fwd_pool);
 * yraw (Rp,C) or NULL = y* itself, which spares demf_bn_bwd_reduce its gather from Y. */
DEMF_INTERNAL int demf_pool_select(int Rp, int C, const float* pmax, const float* pmin, const int* amax,
                     const int* amin, const float* scale_shift, float* out, int* arg, float* yraw,
                     demf_stream_t stream);

/* demf_pool_select behind the NO-STORE pooled forward (demf_mlp_gemm_fwd_pool_bn_st with store_flags = 4:
 * the raw (R x N) output of the pooled last layer is not written; zero-scale channels arrive with slot 0
 * and its raw value in pmax / amax, so yraw is complete).                                              */
DEMF_INTERNAL int demf_pool_select_slot0(int Rp, int C, const float* pmax, const int* amax, const float* scale_shift,
                           float* out, int* arg, float* yraw, demf_stream_t stream);

/* stats (2N fp64) over `count` rows -> scale_shift (2N), mean_invstd (2N); updates the
 * running statistics (momentum, unbiased variance) and increments *num_batches_tracked when
 * they are non-NULL (BatchNorm's train-mode bookkeeping).  The consumed accumulator is left
 * ZEROED, so a persistent stats buffer needs no per-step clearing.  conv_bias (N) or NULL: the bias
 * of the convolution in front of the BatchNorm - it cancels in the normalised output and never
 * reaches the GEMM; it only shifts the batch mean that enters the running mean.              */
DEMF_INTERNAL int demf_bn_finalize(int N, long long count, double* stats, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, long long* num_batches_tracked, float* scale_shift,
                     float* mean_invstd, const float* conv_bias, demf_stream_t stream);

/* y = x / ||x||_2 over rows of C channels (VoteModule norm_feats, class_agnostic_vote_head.py:413-414 ->
 * mmdet3d VoteModule.forward); norm (R) kept for the backward dx = (dy - y (y . dy)) / norm.  */
DEMF_INTERNAL int demf_l2norm_rows_fwd(int R, int C, const float* x, float* y, float* norm, demf_stream_t stream);
DEMF_INTERNAL int demf_l2norm_rows_bwd(int R, int C, const float* y, const float* norm, const float* dy, float* dx,
                         demf_stream_t stream);

/* VoteModule tail (mmdet3d VoteModule.forward with vote_per_seed = 1, with_res_feat, norm_feats; built at
 * class_agnostic_vote_head.py:382) as one pass each way.  votes (R, 3 + C) = conv_out's rows:
 *   vote_xyz = seed_xyz + votes[:, :3];  vote_feats = l2-normalised rows of (rows + votes[:, 3:]).
 * Backward writes d_votes (R, 3 + C) completely and d_rows (R, C); d_vote_feats / d_vote_xyz may be
 * null (= zero).  C in {64, 128, 256, 512, 1024}.                                                  */
DEMF_INTERNAL int demf_vote_combine_fwd(int R, int C, const float* rows, const float* votes, const float* seed_xyz,
                          float* vote_xyz, float* vote_feats, float* norm, demf_stream_t stream);
DEMF_INTERNAL int demf_vote_combine_bwd(int R, int C, const float* vote_feats, const float* norm,
                          const float* d_vote_feats, const float* d_vote_xyz, float* d_votes,
                          float* d_rows, demf_stream_t stream);

/* out (R,C) = max over s of act(Y (R,ns,C)); arg = first maximising s.           */
DEMF_INTERNAL int demf_bnrelu_maxpool_fwd(int R, int ns, int C, const float* Y, const float* scale_shift,
                            float* out, int* arg, demf_stream_t stream);

/* BatchNorm backward reductions of one layer: g12[0:N] += sum dZ, g12[N:2N] += sum dZ*xhat
 * with dZ = dA * [act'(Y)].  Upstream gradient dA is either dense G (R,N) or, for the pooled
 * last layer, sparse: dP (R/ns,N) routed to slot arg (R/ns,N); yraw (R/ns,N) or NULL = Y at
 * those slots (from demf_pool_select), read instead of gathering it.               */
DEMF_INTERNAL int demf_bn_bwd_reduce(int R, int N, int ns, const float* G, const float* dP, const int* arg,
                       const float* Y, const float* yraw, const float* scale_shift,
                       const float* mean_invstd, double* g12, demf_stream_t stream);

/* g12 -> the five per-channel vectors the backward GEMM prologues consume (vec6: 5N floats:
 * scale, shift, gi = gamma*invstd, a, b with dY = gi*dZ + a*y + b) + dgamma, dbeta.  The
 * consumed g12 is left ZEROED.                                                            */
DEMF_INTERNAL int demf_bn_bwd_vectors(int N, long long count, double* g12, const float* gamma,
                        const float* scale_shift, const float* mean_invstd, float* vec6,
                        float* dgamma, float* dbeta, demf_stream_t stream);

/* dX (R,K; row stride ldo) = dY (R,N) @ W (N,K) with dY = BN/ReLU backward of (dA, Y)
 * formed on the fly.  Wtt = W^T stored (K,N) row-major.  N%4==0.                  */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dx(int R, int N, int K, int ldo, const float* G, const float* dP,
                         const int* arg, int ns, const float* Y, const float* vec6,
                         const float* Wtt, float* dX, demf_stream_t stream);
/* Same, reading the layer's weight W (N x K row-major, K % 4 == 0) itself instead of a transposed
 * copy made per step (the B slab is transposed on its way into LDS). */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dx_w(int R, int N, int K, int ldo, const float* G, const float* dP,
                         const int* arg, int ns, const float* Y, const float* vec6,
                         const float* W, float* dX, demf_stream_t stream);
/* demf_mlp_gemm_bwd_dx_w that also adds layer l-1's BN-backward sums - exactly what
 * demf_bn_bwd_reduce(R, K, G = dX, Yprev, ...) would add to g12_prev (2K fp64) - taken from the
 * output tiles on their way out, so that pass over dX and Yprev disappears.  K % 4 == 0. */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dx_red(int R, int N, int K, int ldo, const float* G, const float* dP,
                             const int* arg, int ns, const float* Y, const float* vec6,
                             const float* W, float* dX, const float* Yprev,
                             const float* scale_shift_prev, const float* mean_invstd_prev,
                             double* g12_prev, demf_stream_t stream);

/* Backward of a shared-MLP stack whose FIRST layer has a 4-float input row and needs no input
 * gradient (SA1: [height | rel_xyz]).  The dx GEMM of layer 1 (dY1 from (G, Y1, vec6); W1 (N x K0))
 * does not store its (R x K0) output - the gradient of layer 0's activation - but takes the raw
 * sums of layer 0's whole backward from it: sums (10*K0 + 4 fp64, accumulated, arrives zeroed) =
 * g1 | g2 | P = dZ0^T X0 (K0 x 4) | Q = Y0^T X0 (K0 x 4) | colsum(X0).  K0 <= 64, K0 % 4 == 0.
 * demf_mlp_first_finish then forms dW0 (K0 x 4) = gi*P + a*Q + b*cx^T, dgamma0 = g2, dbeta0 = g1
 * (the BN-backward identities of demf_bn_bwd_vectors) and leaves the sums zeroed. */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dx_first(int R, int N, int K0, const float* G, const float* Y1,
                               const float* vec6, const float* W1, const float* X0,
                               const float* Y0, const float* scale_shift0,
                               const float* mean_invstd0, double* sums, demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_first_finish(int N0, long long count, double* sums, const float* gamma0,
                          const float* mean_invstd0, float* dW0, float* dgamma0, float* dbeta0,
                          demf_stream_t stream);

/* demf_mlp_gemm_bwd_dx_red + layer l-1's backward vectors (demf_bn_bwd_vectors on g12_prev) formed by
 * the launch's last workgroup: the ~5 us vectors launch behind every layer's backward disappears. */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dx_red_v(int R, int N, int K, int ldo, const float* G, const float* dP,
                               const int* arg, int ns, const float* Y, const float* vec6,
                               const float* W, float* dX, const float* Yprev,
                               const float* scale_shift_prev, const float* mean_invstd_prev,
                               double* g12_prev, const float* gamma_prev, float* vec6_prev,
                               float* dgamma_prev, float* dbeta_prev, demf_stream_t stream);
/* demf_bn_bwd_reduce (sparse form: the pooled last layer of a stack) + demf_bn_bwd_vectors in ONE launch.
 * y_bf16: Y holds bf16 rows (demf_mlp_gemm_fwd_pool_bn_st).                                              */
DEMF_INTERNAL int demf_bn_bwd_reduce_vectors(int R, int N, int ns, const float* dP, const int* arg, const float* Y,
                               const float* yraw, const float* scale_shift, const float* mean_invstd,
                               double* g12, const float* gamma, float* vec6, float* dgamma, float* dbeta,
                               int y_bf16, demf_stream_t stream);

/* demf_mlp_gemm_fwd_bn / demf_mlp_gemm_fwd_pool_bn with the rows stored as bf16 in HBM (bf16 compute
 * mode, BASELINE configs[3]: "bf16 training step"): store_flags bit 0 = X holds bf16, bit 1 = Y is
 * written as bf16; ldx counts elements.  Statistics and the pooled extremum are taken from the fp32
 * accumulators.  Weight-resident forward only (K = 64, N = 64 / 128, R >= 16384; the forms built:
 * flags 2 for N = 64, flags 3 pooled for N = 128): DEMF_EUNSUPPORTED otherwise, callers keep fp32 rows. */
DEMF_INTERNAL int demf_mlp_gemm_fwd_bn_st(int R, int K, int N, int ldx, const void* X, const float* pro_scale_shift,
                            const float* Wt, void* Y, double* stats, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float* scale_shift,
                            float* mean_invstd, const float* conv_bias, int store_flags,
                            demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_gemm_fwd_pool_bn_st(int R, int K, int N, int ldx, const void* X,
                                 const float* pro_scale_shift, const float* Wt, void* Y, double* stats,
                                 int ns, float* pmax, int* amax, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var,
                                 long long* num_batches_tracked, float* scale_shift, float* mean_invstd,
                                 const float* conv_bias, int store_flags, demf_stream_t stream);

/* One pass over a layer's saved output for its whole backward (csrc/mlp_bwd.hip): what
 * demf_mlp_gemm_bwd_dx_red + demf_mlp_gemm_bwd_dw do in two (or, with first_sums != NULL,
 * demf_mlp_gemm_bwd_dx_first + demf_mlp_gemm_bwd_dw: dX is then not stored and X0 (R x 4) / first_sums
 * take the place of dX / g12_prev).  The dY tile is rebuilt and split into bf16 planes once and feeds
 * both contractions; W stays in registers.  Yprev (R x K) = pre-BN output of layer l-1, row stride K;
 * dX row stride K; dW (N x K) accumulated (arrives zeroed).  Compute modes 1 (bf16) and 2 (fp32 as
 * three bf16 terms) only; shapes (N,K) in {(128,64), (128,128), (64,64), (256,128)}, ns % 4 == 0 when sparse
 * (G == NULL); anything else returns DEMF_EINVAL and callers use the two-launch path.  With gamma_prev !=
 * NULL (and not first_sums) the launch's last workgroup also forms layer l-1's backward vectors - what
 * demf_bn_bwd_vectors(K, R, g12_prev, gamma_prev, ...) would: vec6_prev (5K), dgamma_prev, dbeta_prev,
 * g12_prev left zeroed.  store_flags != 0 (bf16 compute mode, BASELINE configs[3]): rows stored as bf16 -
 * bit 0: Yprev, bit 1: Y, G and dX (the pointers then address 2-byte elements); built for the two forms
 * of SA1's stack ((128,64) sparse with 3, (64,64) first_sums with 2), DEMF_EUNSUPPORTED otherwise.  Replaces the
 * autograd backward of Conv2d -> BatchNorm2d -> ReLU in mmdet3d's PointSAModule stacks
 * (configs/demf/demf_votenet.py:48-62; class_agnostic_vote_head.py:383). */
DEMF_INTERNAL int demf_mlp_bwd_fused(int R, int N, int K, const float* G, const float* dP, const int* arg, int ns,
                       const float* Y, const float* vec6, const float* W, const float* Yprev,
                       const float* scale_shift_prev, const float* mean_invstd_prev, float* dX,
                       float* dW, double* g12_prev, const float* X0, double* first_sums,
                       const float* gamma_prev, float* vec6_prev, float* dgamma_prev, float* dbeta_prev,
                       int store_flags, demf_stream_t stream);
/* The same pass over columns c0 .. c0 + Kc of a Ktot-channel layer l-1 (Kc = 64 / 128, (N, Kc) one of the
 * shapes above; dense row storage): Yprev, dX (R x Ktot), W, dW (N x Ktot), scale_shift_prev / mean_invstd_prev
 * (2 Ktot), g12_prev (2 Ktot) and the vector outputs are the WHOLE layer's; one call per column chunk. */
DEMF_INTERNAL int demf_mlp_bwd_fused_cols(int R, int N, int Ktot, int c0, int Kc, const float* G, const float* dP,
                            const int* arg, int ns, const float* Y, const float* vec6, const float* W,
                            const float* Yprev, const float* scale_shift_prev,
                            const float* mean_invstd_prev, float* dX, float* dW, double* g12_prev,
                            const float* gamma_prev, float* vec6_prev, float* dgamma_prev,
                            float* dbeta_prev, demf_stream_t stream);

/* SA1-shaped stacks (4-float grouped rows -> N0 <= 64 channels -> ...): the first layer WITHOUT its (R x N0)
 * output.  y = x.W0^T is linear in the 16-byte row, so
 *   demf_mlp_first_stats      the layer's train-mode BN statistics + bookkeeping from the 14 second moments of the
 *                             rows (one pass over 16 MB instead of a GEMM writing 268 MB); ``moments``: >= 14
 *                             doubles of a zeroed accumulator, left zeroed;
 *   demf_mlp_gemm_fwd_bn_x4   the second layer (64 -> 64): demf_mlp_gemm_fwd_bn with its BN + ReLU prologue
 *                             fed by rows rebuilt from X4 and W0 (R >= 16384, compute modes 1 / 2);
 *   demf_mlp_bwd_fused_x4     the second layer's one-pass backward with the FIRST epilogue
 *                             (demf_mlp_bwd_fused, first_sums != NULL), likewise rebuilding layer 0's output.
 * mmdet3d PointSAModule's shared MLP at SA1 (configs/demf/demf_votenet.py:48-62).                          */
DEMF_INTERNAL int demf_mlp_first_stats(int R, int N0, const float* X, const float* W0, double* moments, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* scale_shift,
                         float* mean_invstd, const float* conv_bias, demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_gemm_fwd_bn_x4(int R, int N, const float* X4, const float* W0, const float* prev_scale_shift,
                            const float* Wt, float* Y, double* stats, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var,
                            long long* num_batches_tracked, float* scale_shift, float* mean_invstd,
                            const float* conv_bias, demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_bwd_fused_x4(int R, int N, int K, const float* G, const float* Y, const float* vec6,
                          const float* W, const float* X0, const float* W0, const float* scale_shift_prev,
                          const float* mean_invstd_prev, float* dW, double* first_sums, demf_stream_t stream);

/* Backward of a POOLED last layer without its (R x N) output (SA1: 537 MB neither written by the forward
 * nor read here).  With y = A.W^T and dY = gi*dZ + a*y + b (train-mode BN), A = act(Y_{L-1}):
 *     dA = (gi*dZ).W + A.(W^T diag(a) W) + b^T W,    dW = (gi*dZ)^T.A + diag(a) W (A^T A) + b (x) colsum(A)
 * - functions of A and of the sparse pooled gradient (dP at the selected rows ``arg``, gated by the ReLU on
 * ``yraw`` = the raw output at the selected row, from demf_pool_select_slot0) alone.
 * N = 128, K = 64, ns = 64, R % 64 == 0, compute modes 1 / 2.  dX (R x K), dW (N x K) written;
 * g12_prev (2K doubles, zeroed accumulator) accumulated and, with gamma_prev, turned into layer L-1's
 * backward vectors by the last workgroup.  workspace: *floats of demf_mlp_bwd_pool_ws(R, &floats), scratch.
 * Replaces demf_mlp_bwd_fused for that layer
 * (mmdet3d PointSAModule's shared MLP, demf/modeling/heads/class_agnostic_vote_head.py:383,
 * configs/demf/demf_votenet.py:48-62).                                                                  */
DEMF_INTERNAL int demf_mlp_bwd_pool_ws(int R, long long* floats);
DEMF_INTERNAL int demf_mlp_bwd_pool(int R, int N, int K, int ns, const float* dP, const int* arg, const float* yraw,
                      const float* vec6, const float* W, const float* Yprev,
                      const float* scale_shift_prev, const float* mean_invstd_prev, float* dX, float* dW,
                      double* g12_prev, const float* gamma_prev, float* vec6_prev, float* dgamma_prev,
                      float* dbeta_prev, float* workspace, demf_stream_t stream);

/* dW (N,K) += dY^T @ A_prev, A_prev = act_prev(Xprev (R,K; stride ldx)) or Xprev itself
 * when prev_scale_shift is NULL (first layer).  dW accumulated (fp32 atomics).        */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dw(int R, int N, int K, int ldx, const float* G, const float* dP,
                         const int* arg, int ns, const float* Y, const float* vec6,
                         const float* Xprev, const float* prev_scale_shift, float* dW,
                         demf_stream_t stream);
/* Same with an explicit row stride of dW (>= K): the (N x K) result lands inside a wider weight
 * gradient (the feature columns 3.. of a set-abstraction level's first layer).                  */
DEMF_INTERNAL int demf_mlp_gemm_bwd_dw_ld(int R, int N, int K, int ldx, const float* G, const float* dP,
                            const int* arg, int ns, const float* Y, const float* vec6,
                            const float* Xprev, const float* prev_scale_shift, float* dW, int lddw,
                            demf_stream_t stream);

/* Several demf_mlp_gemm_bwd_dw_ld products in as few launches as their kernel variants allow (one per variant,
 * twelve jobs at a time).  Nothing in a backward depends on a layer's weight gradient, so a caller may queue the
 * jobs of many layers and issue them together once the operands exist; fields as the arguments of
 * demf_mlp_gemm_bwd_dw_ld. */
typedef struct demf_dw_job {
  int R, N, K, ldx;
  const float* G;
  const float* dP;
  const int* arg;
  int ns;
  const float* Y;
  const float* vec6;
  const float* Xprev;
  const float* prev_scale_shift;
  float* dW;
  int lddw;
} demf_dw_job;
DEMF_INTERNAL int demf_mlp_gemm_bwd_dw_group(int n, const demf_dw_job* jobs, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Fused head losses: DeMFVoteHead._loss (class_agnostic_vote_head.py:622-712)
 * over the raw conv-head rows  cls (R,12) = [objectness 2 | semantic 10],
 * reg (R,30) = [centre offset 3 | size 3 | dir class 12 | dir residual 12]
 * (split_pred of coder.py:196-240 folded in); base_xyz (R,3) = aggregated points.
 * hyper12 (host floats) = objectness class weights (2), loss weights of objectness /
 * dir_class / dir_res / size / centre / semantic / iou (7), SmoothL1 betas of dir_res /
 * size / centre (3)   [configs/demf/demf_votenet.py:116-141].
 * out7 (accumulated) = the seven reduction='sum' losses in that order.
 * ------------------------------------------------------------------ */
DEMF_INTERNAL int demf_head_loss_fwd(int R, int num_dir_bins, int num_classes, const float* hyper12,
                       const float* cls, const float* reg, const float* base_xyz,
                       const float* center_t, const float* size_t_, const int64_t* dir_class_t,
                       const float* dir_res_t, const int64_t* sem_t, const int64_t* obj_t,
                       const float* obj_w, const float* box_w, float* out7,
                       demf_stream_t stream);
/* gradients wrt cls, reg, base_xyz given the seven upstream scalars grad_out7 (device).  */
DEMF_INTERNAL int demf_head_loss_bwd(int R, int num_dir_bins, int num_classes, const float* hyper12,
                       const float* cls, const float* reg, const float* base_xyz,
                       const float* center_t, const float* size_t_, const int64_t* dir_class_t,
                       const float* dir_res_t, const int64_t* sem_t, const int64_t* obj_t,
                       const float* obj_w, const float* box_w, const float* grad_out7,
                       float* grad_cls, float* grad_reg, float* grad_base,
                       demf_stream_t stream);

/* VoteModule.get_loss (called at class_agnostic_vote_head.py:641-644): chamfer-L1 of each
 * seed's vote against its gt_per_seed target votes, min over targets, weighted by
 * mask/(mask_sum+1e-6)*dst_weight.  grad_out == NULL: forward (out accumulated);
 * otherwise backward: grad_vote (B,S,3) written.                                  */
DEMF_INTERNAL int demf_vote_loss(int B, int S, int N, int gt_per_seed, float dst_weight,
                   const float* seed_points, const float* vote_points,
                   const int64_t* seed_indices, const int64_t* vote_target_masks,
                   const float* vote_targets, const float* mask_sum, const float* grad_out,
                   float* out, float* grad_vote, demf_stream_t stream);
/* forward of the same with the denominator counted in the kernel: mask_sum = sum over the seeds of
 * vote_target_masks[b, seed_indices[b, s]] (torch.gather(...).sum() of VoteModule.get_loss), stored to
 * mask_sum_out[0] for the backward call above; out[0] accumulated.                              */
DEMF_INTERNAL int demf_vote_loss_fwd(int B, int S, int N, int gt_per_seed, float dst_weight,
                       const float* seed_points, const float* vote_points,
                       const int64_t* seed_indices, const int64_t* vote_target_masks,
                       const float* vote_targets, float* mask_sum_out, float* out,
                       demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * DeMF fusion: multi-scale deformable attention core
 * ------------------------------------------------------------------ */

/* MultiScaleDeformableAttnFunction.forward: reference call site
 * demf/modeling/layers/transformer.py:73 -> mmcv MultiScaleDeformableAttention
 * (config demf_votenet.py:79-85: H=8, L=4, P=2, Dh=32).
 * value (B,S,H,Dh), spatial_shapes (L,2) i64 (h,w), level_start_index (L) i64,
 * sampling_loc (B,Q,H,L,P,2) (x,y in [0,1]), attn_weight (B,Q,H,L,P)
 * -> out (B,Q,H*Dh).  Bilinear, align_corners=False, zero padding.  Dh must be
 * a multiple of 4 and <= 256.                                                */
int demf_msda_fwd_f32(int B, int S, int H, int Dh, int L, int Q, int P,
                      const float* value, const int64_t* spatial_shapes,
                      const int64_t* level_start_index, const float* sampling_loc,
                      const float* attn_weight, float* out, demf_stream_t stream);

/* backward: grad_out (B,Q,H*Dh) -> grad_value (B,S,H,Dh) accumulated (NULL: not wanted, the
 * scatter is skipped), grad_sampling_loc (B,Q,H,L,P,2) and grad_attn_weight (B,Q,H,L,P) written. */
int demf_msda_bwd_f32(int B, int S, int H, int Dh, int L, int Q, int P,
                      const float* value, const int64_t* spatial_shapes,
                      const int64_t* level_start_index, const float* sampling_loc,
                      const float* attn_weight, const float* grad_out,
                      float* grad_value, float* grad_sampling_loc,
                      float* grad_attn_weight, demf_stream_t stream);

/* The same two entry points over bf16 VALUE rows (B,S,H,Dh) of 2-byte elements (the upper halves of the fp32
 * bit patterns): the gather - the dominant traffic of the operator - moves half the bytes; sampling locations,
 * weights, every accumulation, the output and all gradients (grad_value too: fp32, same layout) stay fp32.
 * BASELINE.json configs[3]: a bf16 image-token buffer halves the 152 MB token tensor. */
DEMF_INTERNAL int demf_msda_fwd_bf16(int B, int S, int H, int Dh, int L, int Q, int P,
                       const uint16_t* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, float* out, demf_stream_t stream);
DEMF_INTERNAL int demf_msda_bwd_bf16(int B, int S, int H, int Dh, int L, int Q, int P,
                       const uint16_t* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, const float* grad_out,
                       float* grad_value, float* grad_sampling_loc,
                       float* grad_attn_weight, demf_stream_t stream);

/* ---- the frozen image stream's encoder layers --------------------------------------------- *
 * Long-row linear layer: C (R,N; ldc) = epi((A [+ A2 on output columns < a2_cols]) . W^T + bias), R ~ 150 000
 * token rows, K % 32 == 0, N % 128 == 0.  W arrives PRE-SPLIT as bf16 planes (planes, N, K): planes = 3 is
 * fp32-grade arithmetic (w = h + m + l exactly, six products per term pair on the bf16 matrix cores, as
 * compute mode 2 of the demf_mlp_gemm_* kernels), planes = 1 the bf16 compute mode; the A rows are fp32 and are
 * split inside the kernel.  mode 0: bias; rows with row_mask[r] != 0 are zeroed in columns >= mask_col0 (the
 * padding mask of value_proj); mode 1: bias + ReLU; mode 2 (N == 256): LayerNorm(resid + A.W^T + bias) * gamma +
 * beta.  Replaces, per encoder layer of demf/modeling/layers/deform_detr_encoder.py:68-154, six library GEMMs
 * and the elementwise launches between them (query + pos, masked_fill, ReLU, residual adds, two LayerNorms).  */
DEMF_INTERNAL int demf_rows_gemm_f32(int R, int N, int K, const float* A, long long lda, const float* A2, int a2_cols,
                       int a2_op /* 0: A + A2 on those columns, 1: A2 INSTEAD of A (a pre-added operand) */,
                       const void* w_planes, int planes, const float* bias, int mode,
                       const unsigned char* row_mask, int mask_col0, const float* resid, long long ldr,
                       const float* gamma, const float* beta, float eps, float* C, long long ldc,
                       demf_stream_t stream);
/* y = LayerNorm(resid + x) * gamma + beta over rows of C = 256 channels (gamma == beta == NULL: y = resid + x) and
 * ypos = y + pos; y or ypos may be NULL.  The FFN tail of an encoder layer and the `query + query_pos` of the next
 * layer's attention (mmcv MultiScaleDeformableAttention.forward) in one pass.                                    */
DEMF_INTERNAL int demf_rows_ln_pos_f32(int R, int C, const float* x, const float* resid, const float* gamma, const float* beta,
                         float eps, const float* pos, float* y, float* ypos, demf_stream_t stream);
/* Multi-scale deformable attention forward with RAW inputs: offsets (H*L*P*2 columns from off_col0) and attention
 * logits (H*L*P columns from lgt_col0) of row b*Q+q of `raw` (row stride ldraw), reference points ref (B,Q,L,2),
 * value rows of pitch vpitch floats (columns h*Dh.. of the same projection output).  The softmax over a head's
 * L*P logits and loc = ref + offset / (W_l, H_l) - separate elementwise launches in mmcv's
 * MultiScaleDeformableAttention.forward - happen on the way in.  Dh = 32, L = 4, P in {2, 4}. -> out (B,Q,H*Dh) */
DEMF_INTERNAL int demf_msda_fwd_raw_f32(int B, int S, int H, int Dh, int L, int Q, int P, const float* value,
                          long long vpitch, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* raw, long long ldraw, int off_col0, int lgt_col0, const float* ref,
                          float* out, demf_stream_t stream);
/* The same for H = 8, Dh = 32, L = 4 with the value rows of levels first_staged_level (2 or 3) .. 3 of one (scene, head)
 * resident in LDS: staged_tokens = S - level_start_index[first_staged_level], known to the caller on the host (the
 * library never reads a device array back); staged_tokens * 128 B + 8 KB * P / 2 ... must fit 160 KB or the call is
 * refused.  Results equal demf_msda_fwd_raw_f32's up to the summation order of a query's samples (Q = S rows). */
DEMF_INTERNAL int demf_msda_fwd_raw_head_f32(int B, int S, int Q, int P, const float* value, long long vpitch,
                               const int64_t* spatial_shapes, const int64_t* level_start_index, const float* raw,
                               long long ldraw, int off_col0, int lgt_col0, const float* ref, float* out,
                               int first_staged_level, int staged_tokens, demf_stream_t stream);

/* Per-point vote targets of DeMFVoteHead.get_targets_single (class_agnostic_vote_head.py:828-858),
 * batched: points (B,N,point_stride>=3), gt_boxes (B,G,7) = (x,y,z_bottom,dx,dy,dz,yaw) padded to
 * G <= 64 with valid (B,G) bytes, cos/sin of -yaw (B,G) -> vote_targets (B,N,9) = votes to the
 * gravity centres of the first | second-or-first | last-or-first containing box, zeros outside
 * every box; vote_target_masks (B,N) int64 = inside any box.                                  */
DEMF_INTERNAL int demf_vote_targets(int B, int N, int point_stride, int G, const float* points,
                      const float* gt_boxes, const float* cos_neg_yaw, const float* sin_neg_yaw,
                      const unsigned char* valid, float* vote_targets, int64_t* vote_target_masks,
                      demf_stream_t stream);

/* Per-proposal targets of DeMFVoteHead.get_targets_single (class_agnostic_vote_head.py:877-934),
 * batched over scenes: nearest valid ground-truth centre (first minimum of the squared distance),
 * the gathered centre / size / direction / class targets, the distance targets in the box frame
 * (rotation by -yaw when with_rot) and objectness label + mask (pos/neg distance thresholds).
 * gt_dir_class / gt_dir_res = bbox_coder.angle2class(yaw) per box; res_scale = pi/num_dir_bins.  */
DEMF_INTERNAL int demf_proposal_targets(int B, int Q, int G, int with_rot, float pos_thr, float neg_thr,
                          float res_scale, const float* aggregated_points, const float* gt_boxes,
                          const float* cos_neg_yaw, const float* sin_neg_yaw,
                          const int64_t* gt_dir_class, const float* gt_dir_res,
                          const int64_t* gt_labels, const unsigned char* valid,
                          float* center_targets, float* size_targets, int64_t* dir_class_targets,
                          float* dir_res_targets, float* dir_targets, int64_t* mask_targets,
                          float* distance_targets, int64_t* objectness_targets,
                          float* objectness_masks, demf_stream_t stream);

/* Position-embedding input of a fusion decoder layer from a prediction head's raw regression rows
 * (reg_rows (R, nreg >= 6) point-major, base_xyz (R,3)): out8 (R,8) = [base + reg[0:3] | reg[3:6] | 0 0] -
 * the reference's torch.cat([center, size], -1).detach() (class_agnostic_vote_head.py:497-498), padded
 * to the 8 columns the first GEMM of the position embedding stages.                                  */
DEMF_INTERNAL int demf_query_pos_rows(int R, int nreg, const float* reg_rows, const float* base_xyz, float* out8,
                        demf_stream_t stream);

/* Tail of DeMFVoteHead.loss (class_agnostic_vote_head.py:604-612: the losses of the num_fusion_layers + 1
 * decode results are averaged, the training loop sums the dict): vecs[0..n) are the (7,) per-decode-layer
 * loss vectors, vote the (1,) vote loss or NULL -> out8 = [mean of the vectors (7) | their sum + vote].
 * n <= 4.  demf_loss_total_bwd: g8 = gradient of out8 -> gvecs (n,7) = (g8[i] + g8[7]) / n, gvote = g8[7]. */
DEMF_INTERNAL int demf_loss_total(int n, const float* const* vecs, const float* vote, float* out8, demf_stream_t stream);
DEMF_INTERNAL int demf_loss_total_bwd(int n, const float* g8, float* gvecs, float* gvote, demf_stream_t stream);

/* Per-scene ground-truth lists -> the static-shape padded form the target kernels read: gt_padded
 * (B,G,7) fp32, labels_padded (B,G) int64 with -1 on padding slots, valid (B,G) u8 or NULL.  An empty
 * scene gets the reference's single all-zero fake box with label 0
 * (demf/modeling/heads/class_agnostic_vote_head.py:766-773).  counts / box_dims / boxes / labels are
 * HOST arrays of length B (<= 32): scene b's device pointers to its (counts[b], box_dims[b] >= 7) fp32
 * box rows and (counts[b]) int64 labels; they travel by value in the kernel arguments, so the call
 * issues one launch and no host -> device copy.                                                      */
DEMF_INTERNAL int demf_pad_gt(int B, int G, const int* counts, const int* box_dims, const void* const* boxes,
                const void* const* labels, float* gt_padded, int64_t* labels_padded,
                unsigned char* valid, demf_stream_t stream);

/* Everything the two target kernels need that depends on the padded ground truth alone (label -1 =
 * padding slot), one launch: cos/sin(-yaw), PartialBinBasedBBoxCoder.angle2class(yaw) in torch's
 * fp32 remainder / floor-divide semantics (class_agnostic_vote_head.py:877-883 via
 * bbox_coder.encode), valid = label >= 0, labels clamped at 0, gravity centres (B,G,3).           */
DEMF_INTERNAL int demf_gt_prep(int B, int G, int num_dir_bins, const float* gt_boxes, const int64_t* labels_padded,
                 float* cos_neg_yaw, float* sin_neg_yaw, int64_t* gt_dir_class, float* gt_dir_res,
                 unsigned char* valid, int64_t* labels_clamped, float* gravity_center,
                 demf_stream_t stream);

/* objectness_weights = masks / (sum + 1e-6), box_loss_weights = objectness / (sum + 1e-6) over all
 * R = B*Q proposals (class_agnostic_vote_head.py:797-816).                                          */
DEMF_INTERNAL int demf_target_weights(int R, const float* objectness_masks, const int64_t* objectness_targets,
                        float* objectness_weights, float* box_loss_weights, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Test-time post-processing: DeMFVoteHead.get_bboxes (class_agnostic_vote_head.py:714-754) +
 * the inherited mmdet3d VoteHead.multiclass_nms_single / aligned_3d_nms.
 * ------------------------------------------------------------------ */

/* boxes7 (B,K,7) = decode() output (gravity centre, size, yaw) with cos/sin(yaw) (B,K) ->
 * boxes_bottom (B,K,7) upstream's bottom-centre form, extent6 (B,K,6) = min|max over the 8 rotated
 * corners, count (B,K) = scene points inside the box (points (B,N,point_stride>=3)).           */
DEMF_INTERNAL int demf_box_extent_count(int B, int N, int point_stride, int K, const float* points,
                          const float* boxes7, const float* cos_yaw, const float* sin_yaw,
                          float* boxes_bottom, float* extent6, int* count, demf_stream_t stream);

/* aligned_3d_nms per scene over the boxes with valid != 0 (K <= 1024): descending score order, a
 * kept box removes same-class boxes with IoU > iou_thr.  keep (B,K) bytes.                     */
DEMF_INTERNAL int demf_aligned_nms(int B, int K, float iou_thr, const float* extent6, const float* scores,
                     const int64_t* classes, const unsigned char* valid, unsigned char* keep,
                     demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Optimizer step on flat buffers: torch.optim.AdamW + clip_grad_norm_ as the reference's
 * runner applies them (configs/_base_/schedules/schedule_3x.py:6-7: AdamW lr 0.008, wd 0.01,
 * grad_clip max_norm 10; configs/demf/demf_votenet.py:16-24: 'decoder' lr_mult 0.05).  One call
 * per parameter group.  g_eff = grad * grad_scale * min(1, max_norm / (norm*grad_scale + 1e-6))
 * with norm read from the device scalar grad_norm (NULL: no clipping); grad itself is not
 * modified.  step is the 1-based step count (bias correction).
 * ------------------------------------------------------------------ */
DEMF_INTERNAL int demf_adamw_f32(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   const float* grad_norm, float max_norm, float grad_scale, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, demf_stream_t stream);

/* The same update with its state ON THE DEVICE, so that norm + clip + AdamW are nodes of the step's
 * hipGraph (nothing the host passes changes between replays).  `opt_state`: 64 zero-initialised bytes
 *   { double sumsq; int64 t; uint32 ticket; float lr_factor; ... }
 * sumsq = squared L2 norm of the gradients of ALL groups, accumulated by demf_multi_copy_sumsq (the
 * gradient pack of a captured step) or demf_sumsq_f32 and cleared by demf_adamw_state_f32; t = completed
 * steps (the launch uses t + 1 for the bias corrections and publishes it from its last workgroup);
 * lr_factor multiplies every group's learning rate (the reference's step schedule,
 * configs/_base_/schedules/schedule_3x.py:7-9).  Up to 4 parameter groups (segments [start, start + n) of
 * the flat buffers, HOST arrays) in ONE launch; max_norm <= 0: no clipping.                            */
DEMF_INTERNAL int demf_multi_copy_sumsq(int n, const void* table, int blocks_per_segment, void* opt_state,
                          demf_stream_t stream);
DEMF_INTERNAL int demf_sumsq_f32(long long n, const float* x, void* opt_state, demf_stream_t stream);
DEMF_INTERNAL int demf_adamw_state_f32(int nseg, const long long* seg_start, const long long* seg_n, const float* seg_lr,
                         const float* seg_weight_decay, float* param, const float* grad, float* exp_avg,
                         float* exp_avg_sq, void* opt_state, float max_norm, float grad_scale, float beta1,
                         float beta2, float eps, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Dense blocks of the DeMF fusion decoder layer (csrc/dense.hip)
 * Reference: demf/modeling/layers/transformer.py:55-80 -> mmcv DetrTransformerDecoderLayer
 * (nn.MultiheadAttention, MultiScaleDeformableAttention, FFN, 3 x LayerNorm; cfg
 * configs/demf/demf_votenet.py:71-91).  Upstream reaches cuBLAS / ATen for these; here they are
 * one strided GEMM with fused prologue / epilogue and a few row kernels.
 * ------------------------------------------------------------------ */

#define DEMF_GEMM_RELU 1      /* v = max(v, 0) after the bias                                  */
#define DEMF_GEMM_DROPOUT 2   /* v *= keep / (1 - drop_p), counter-based mask (rng, op_id)      */
#define DEMF_GEMM_GATE 4      /* v = gate[m,n] != 0 ? v * gate_scale : 0  (ReLU+dropout bwd)    */
#define DEMF_GEMM_ACCUM 8     /* C += v (split-K always accumulates, atomically, into zeroed C) */
#define DEMF_GEMM_ROWBIAS 16  /* bias term = bias[n] * rowscale[m]                              */
#define DEMF_GEMM_ACCUM2 32   /* C2 += v instead of C2 = v                                      */
#define DEMF_GEMM_FP32 64     /* keep this launch on the fp32 MFMA path in bf16 compute mode    */

/* C[m,n] (+)= epi( alpha * sum_k (A[m,k] + A2[m,k]) * (B[n,k] + B2[n,k]) + bias[n] ), all operands
 * fp32 and addressed by element strides: A[m,k] at A + m*sam + k*sak, B[n,k] at B + n*sbn + k*sbk,
 * C[m,n] at C + m*scm + n.  Batched over z = zo*zdiv + zi in [0,batch): operand offsets
 * zo*s?b + zi*s?b2.  A2 (same strides as A) is added for blocks whose first column < a2_cols,
 * B2 (same strides as B) for blocks whose first row < b2_rows.  splitk > 1 splits K over blocks.
 * C2 (same strides as C): optional second destination of the same values.
 * HOST struct; every pointer inside is a device pointer.                                      */
typedef struct demf_gemm_desc {
  int M, N, K, batch, zdiv, splitk;
  const float* A; long long sam, sak, sab, sab2;
  const float* A2; int a2_cols;
  const float* B; long long sbn, sbk, sbb, sbb2;
  const float* B2; int b2_rows;
  float* C; long long scm, scb, scb2;
  float* C2;
  const float* bias; long long sbias_b;          /* bias[n + z*sbias_b]                  */
  const float* rowscale; long long srs_m, srs_b; /* rowscale[m*srs_m + z*srs_b]          */
  float alpha;
  int flags;
  const float* gate; long long sgm, sgb;         /* gate[z*sgb + m*sgm + n]              */
  float gate_scale;
  float drop_p;
  const void* rng;                                /* 2 x uint64: seed, step counter       */
  int op_id;
  float* asum;   /* optional, accumulated: asum[z*M + m] += sum_k A[m,k] - the bias gradient when this
                  * launch is a weight gradient dW = dY^T.X (A = dY, reduction-strided, small-tile path) */
} demf_gemm_desc;

DEMF_INTERNAL int demf_gemm_f32(const demf_gemm_desc* desc, demf_stream_t stream);
/* n GEMMs in as few launches as possible: consecutive descriptors of the weight-gradient form (both
 * operands reduction-strided, few tiles) share ONE launch (their descriptors travel by value in the
 * kernel arguments); anything else runs as its own demf_gemm_f32.  The weight / bias gradients of the
 * linear layers of mmcv's DetrTransformerDecoderLayer (configs/demf/demf_votenet.py:71-91) depend on
 * nothing but saved tensors, so a decoder layer issues them as one group at the end of its backward. */
DEMF_INTERNAL int demf_gemm_group_f32(const demf_gemm_desc* descs, int n, demf_stream_t stream);

/* Compute dtype of the dense MFMA kernels (demf_mlp_gemm_*, demf_gemm_f32): 0 = fp32 MFMA (the
 * reference's precision, class_agnostic_vote_head.py:384 fp16_enabled=False), 1 = bf16 MFMA with
 * fp32 accumulation (BASELINE.json configs[3]): operands are rounded to bf16 on their way into LDS,
 * after the fp32 prologues; stored tensors, BN statistics, indices and losses stay fp32;
 * 2 = fp32 emulated on the bf16 MFMA in the demf_mlp_gemm_* kernels: each fp32 operand is split
 * exactly into three bf16 terms (3 x 8 significand bits) and the six products of weight >= 2^-16
 * are accumulated in fp32 (dropped terms <= 2^-23 relative); demf_gemm_f32 stays on the fp32 MFMA.
 * demf_set_compute_dtype sets the PROCESS DEFAULT; a caller that runs several models / threads in different
 * modes passes the mode with the call instead (demf_ctx below).  demf_get_compute_dtype: the mode a dense
 * call made by this thread right now would run in.                                                 */
DEMF_INTERNAL int demf_set_compute_dtype(int mode);
/* Mode 2 only: kernels that have the form take every fp32 operand as TWO fp16 terms (x = h + l, three products on
 * v_mfma_f32_32x32x16_f16; gradient operands scaled by a power of two per slab) instead of three bf16 terms (six
 * products): half the matrix work, ~2^-22 relative instead of 2^-24 - the size of fp32's own accumulation noise.
 * Process default: on (off with DEMF_F16_TERMS=0 in the environment).  csrc/common.h. */
DEMF_INTERNAL int demf_set_f16_terms(int on);
DEMF_INTERNAL int demf_get_compute_dtype(void);

/* The compute mode as a PARAMETER.  compute_mode: 0 / 1 / 2 as above, -1 = the process default; reserved
 * must be zero.  Two ways to hand it over:
 *   - demf_*_ctx(ctx, ...) variants of the generic dense entry points: the context applies to that call;
 *   - demf_ctx_push(ctx) ... demf_ctx_pop(): every dense call THIS THREAD makes in between (all demf_mlp_*,
 *     demf_gemm_*, demf_attn_core_*, demf_linear_* entry points) - a per-thread stack of depth 8, so two host
 *     threads in different modes never see each other's setting and nothing process-global is written. */
typedef struct demf_ctx {
  int compute_mode;
  int reserved[7];
} demf_ctx;
DEMF_INTERNAL int demf_ctx_push(const demf_ctx* ctx);
DEMF_INTERNAL int demf_ctx_pop(void);
DEMF_INTERNAL int demf_gemm_f32_ctx(const demf_ctx* ctx, const demf_gemm_desc* desc, demf_stream_t stream);
DEMF_INTERNAL int demf_gemm_group_f32_ctx(const demf_ctx* ctx, const demf_gemm_desc* descs, int n, demf_stream_t stream);
DEMF_INTERNAL int demf_mlp_gemm_fwd_ctx(const demf_ctx* ctx, int R, int K, int N, int ldx, const float* X,
                          const float* pro_scale_shift, const float* Wt, float* Y, double* stats,
                          demf_stream_t stream);

/* Self-attention core of the fusion decoder layer (nn.MultiheadAttention inside mmcv's
 * DetrTransformerDecoderLayer; demf/modeling/layers/transformer.py:55-80, configs/demf/demf_votenet.py:71-91):
 * out[b, q, h] = dropout(softmax(scale * q_h k_h^T)) v_h for the packed projections qkv (B*Q, 3*H*Dh) =
 * [q | k | v], one launch, the (B*H, Q, Q) score / probability tensors never stored.  stats (B*H*Q, 2) = row
 * maximum of the scaled scores and 1 / sum of exponentials, all the backward needs besides qkv, out and dout.
 * Dropout element ((b*H + h)*Q + q)*Q + k of stream (rng, op_id), as demf_softmax_dropout_fwd draws it.
 * prob / prob_dropped: optional (B*H, Q, Q) outputs for tests (both or neither).
 * Built for the reference's head shape Q = 256, Dh = 32: DEMF_EUNSUPPORTED otherwise (callers keep the
 * demf_gemm_f32 + demf_softmax_dropout_* form).  Compute mode 1 rounds the operands to bf16 where those
 * launches did; modes 0 / 2 are fp32 FMAs. */
DEMF_INTERNAL int demf_attn_core_fwd(int B, int H, int Q, int Dh, const float* qkv, float scale, float p, const void* rng,
                       int op_id, float* out, float* stats, float* prob, float* prob_dropped,
                       demf_stream_t stream);
/* dqkv (B*Q, 3*H*Dh) = gradients of [q | k | v] for dout (B*Q, H*Dh): probabilities are recomputed from
 * stats bit for bit, D = dout . out per query; per (scene, head) four query blocks produce dq and four key
 * blocks dk, dv - no atomics, every element written once. */
DEMF_INTERNAL int demf_attn_core_bwd(int B, int H, int Q, int Dh, const float* qkv, const float* out, const float* dout,
                       const float* stats, float scale, float p, const void* rng, int op_id, float* dqkv,
                       demf_stream_t stream);

/* s = identity + dropout(x) ; y = LayerNorm(s) over rows of C channels (C in 64..1024, power of two
 * multiples of 64).  s_out may alias x; stats (R,2) = mean, rstd.  nn.LayerNorm + the
 * `identity + dropout(out)` tails of mmcv MultiheadAttention / MultiScaleDeformableAttention / FFN.
 * s_out and stats may be NULL (forward-only callers). */
DEMF_INTERNAL int demf_add_dropout_ln_fwd(int R, int C, const float* x, const float* identity, const float* gamma,
                            const float* beta, float eps, float p, const void* rng, int op_id,
                            float* s_out, float* y, float* stats, demf_stream_t stream);
/* backward of the above for dy (+ dy2): ds_out (=/+= per ds_accum) the gradient of s (residual
 * path), dx_out = ds * keep/(1-p) the gradient of x, dgamma / dbeta ACCUMULATED (atomics).        */
DEMF_INTERNAL int demf_add_dropout_ln_bwd(int R, int C, const float* dy, const float* dy2, const float* s,
                            const float* stats, const float* gamma, float p, const void* rng, int op_id,
                            float* ds_out, int ds_accum, float* dx_out, float* dgamma, float* dbeta,
                            demf_stream_t stream);
/* prob = softmax(scores) over rows of S keys, out = dropout(prob): nn.MultiheadAttention's core. */
DEMF_INTERNAL int demf_softmax_dropout_fwd(int R, int S, const float* scores, float p, const void* rng, int op_id,
                             float* prob, float* out, demf_stream_t stream);
/* in place: dio (gradient of `out`) -> gradient of `scores`.                                     */
DEMF_INTERNAL int demf_softmax_dropout_bwd(int R, int S, const float* prob, float p, const void* rng, int op_id,
                             float* dio, demf_stream_t stream);
/* Sampling locations + attention weights of the fusion attention from the raw projection
 * raw (R, H*L*P*3) = [offsets (H,L,P,2) | logits (H,L*P)] and the query points pts (R,3):
 * DeMFVoteHead.get_reference_points (class_agnostic_vote_head.py:524-547; M (B,4,4) and
 * ab (B,4) = (au,bu,av,bv) composed on the host), x valid ratios (transformer.py:62-68),
 * + offsets / (W_l,H_l), softmax over L*P (mmcv MultiScaleDeformableAttention.forward).
 * -> loc (R,H,L,P,2), w (R,H,L,P), uvw (R,4) kept for the backward.  R = B*Q, L*P <= 16.       */
DEMF_INTERNAL int demf_msda_prep_fwd(int R, int Q, int H, int L, int P, const float* pts, const float* M,
                       const float* ab, const float* valid_ratios, const int64_t* shapes,
                       const float* raw, float* loc, float* w, float* uvw, demf_stream_t stream);
/* backward: (dloc + dloc2, dw + dw2) -> draw (R, H*L*P*3), dpts (R,3) or NULL.                   */
DEMF_INTERNAL int demf_msda_prep_bwd(int R, int Q, int H, int L, int P, const float* pts, const float* M,
                       const float* ab, const float* valid_ratios, const int64_t* shapes,
                       const float* w, const float* uvw, const float* dloc, const float* dloc2,
                       const float* dw, const float* dw2, float* draw, float* dpts,
                       demf_stream_t stream);
/* rng[1] += 1: one per training step (a captured graph then draws fresh masks at every replay). */
DEMF_INTERNAL int demf_rng_advance(void* rng, demf_stream_t stream);
/* rng[1] += 1 and snapshot[0..1] = (seed, new step): the forward of a node that carries dropout
 * (nn.Dropout inside mmcv's DetrTransformerDecoderLayer, configs/demf/demf_votenet.py:78,84,87) draws
 * its own step and saves the pair for its backward, as torch saves the mask.                         */
DEMF_INTERNAL int demf_rng_next(void* rng, void* snapshot, demf_stream_t stream);
/* out[i] = keep(i) / (1-p): the mask the fused kernels apply for (rng, op_id) - test hook.       */
DEMF_INTERNAL int demf_dropout_mask(long long n, float p, const void* rng, int op_id, float* out,
                      demf_stream_t stream);

/* n device-to-device copies (or zero fills where src == 0) in one launch.  `table` is a DEVICE
 * array of 3*n int64: n source addresses (0 = fill with zeros), n destination addresses, n lengths in
 * 4-byte words.  Replaces the per-tensor copy / memset launches of a training step: the refresh of
 * the step's static index buffers from the pipelined pre-pass, the zeroing of accumulated outputs. */
DEMF_INTERNAL int demf_multi_copy(int n, const void* table, int blocks_per_segment, demf_stream_t stream);
/* x[0..n) = 0 as a kernel launch (the step's zero arena: one fill per step).                         */
DEMF_INTERNAL int demf_zero_f32(long long n, float* x, demf_stream_t stream);

/* ------------------------------------------------------------------ *
 * Frozen image stream: convolutions on channels-last (NHWC) activations (csrc/conv.hip)
 * Replaces the library convolutions under mmdet ResNet-50 / ChannelMapper as the reference runs them in
 * DeMFVoteNet.extract_img_feat (/root/reference/demf/modeling/detectors/demfnet.py:124-132; configuration
 * /root/reference/configs/deformdetr/imvotenet_image.py:3-20).
 * ------------------------------------------------------------------ */

/* y (B,Ho,Wo,Cout) = [relu]( conv(x (B,H,W,Cin), w) + bias [+ resid (B,Ho,Wo,Cout)] ), implicit GEMM.
 * w_planes: (planes, Cout, KH*KW*Cin) bf16 - the (Cout,Cin,KH,KW) weight permuted to (Cout,KH,KW,Cin) (a frozen
 * BatchNorm folded in by the caller) and split into `planes` bf16 terms (1: bf16 arithmetic; 3: fp32-grade, see
 * demf_split_planes).  Cin % 32 == 0, Cout % 64 == 0.
 * ksplit > 1: the reduction is split over ksplit workgroup rows that ADD their partial tiles into y, which
 * must arrive zeroed (few output pixels x a long reduction: the neck's 3x3 level); bias, resid, relu unused. */
DEMF_INTERNAL int demf_conv_nhwc_f32(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       const float* x, const void* w_planes, int planes, const float* bias,
                       const float* resid, int relu, int ksplit, float* y, demf_stream_t stream);
/* ResNet's 7x7 stride-2 pad-3 stem on an image stored (B,H,W,4) (zero fourth channel; demf_nchw3_to_nhwc4_f32).
 * w_planes: (planes, Cout, 7*32) with element [kh*32 + kw*4 + c] = w[cout][c][kh][kw] (zeros at kw == 7, c == 3). */
DEMF_INTERNAL int demf_conv_stem7_nhwc4_f32(int B, int H, int W, int Cout, const float* x4, const void* w_planes, int planes,
                              const float* bias, int relu, float* y, demf_stream_t stream);
/* 3x3 stride-2 pad-1 max-pool, (B,H,W,C) -> (B,(H+1)/2,(W+1)/2,C), C % 4 == 0. */
DEMF_INTERNAL int demf_maxpool3x3s2_nhwc_f32(int B, int H, int W, int C, const float* x, float* y, demf_stream_t stream);
/* (B,3,H,W) -> (B,H,W,4), fourth channel zero. */
DEMF_INTERNAL int demf_nchw3_to_nhwc4_f32(int B, int H, int W, const float* x, float* y, demf_stream_t stream);
/* GroupNorm(G groups) over (B,HW,C = 256) channels-last rows; image b's rows are written at y + b * y_batch_stride
 * (floats): the levels of the pyramid land directly in the encoder's (B,S,256) token buffer.  `sums`: 2*B*G doubles
 * of scratch.                                                                                       */
DEMF_INTERNAL int demf_groupnorm_nhwc_f32(int B, int HW, int C, int G, float eps, const float* x, const float* gamma,
                            const float* beta, double* sums, float* y, long long y_batch_stride,
                            demf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEMF_HIP_H_ */
