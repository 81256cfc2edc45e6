/*
 * include/usip_hip.h -- C ABI of libusip_hip.so, the MI355X (gfx950) implementation of the
 * USIP detector hot path.  This is the drop-in boundary: plain pointers and sizes, no torch
 * types.  The reference reaches the same operators through two pybind11/torch extensions and
 * through ATen calls in its Python modules; each entry point below cites what it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer on the current HIP device unless the name ends in
 *     `_cpu`; buffers are caller-owned, contiguous, row-major, and never aliased;
 *   - nothing is allocated, nothing synchronises: kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream) and the call returns;
 *   - return value: 0 on success, USIP_EINVAL for a bad argument, otherwise the hipError_t
 *     of the failed launch (positive);
 *   - outputs are fully written (callers need not pre-zero them) unless stated;
 *   - int32 indices, fp32 values; index outputs are bit-exact with the reference, float outputs
 *     agree to fp32 rounding (<= 1e-5 relative).
 */
#ifndef USIP_HIP_H
#define USIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USIP_OK      0
#define USIP_EINVAL (-1)

/* Library identification: "usip_hip <version> gfx950". */
const char* usip_version(void);

/* Launch-geometry knobs for measurement sweeps (tools/): they change HOW a result is computed (rows per
 * workgroup, prefetch depth, tile order), never the result.  0 restores the library's heuristic.  No reference
 * counterpart.  Returns USIP_EINVAL for an unknown name. */
/* "x2_direct" (USIP_TUNE_X2_DIRECT), low four bits: 1 = the LDS-staged f32x2 GEMM of round 3 instead of csrc/gemm_x2d.hip,
 * 2 = one stage of operand loads in flight, 4 = one tile per workgroup, 8 = the LDS-transposing epilogue for data gradients
 * too, 10 = csrc/gemm_x2e.hip (both operands by LDS-DMA, 8-wave workgroups) for the forward launches that fit it, 12 = never
 * csrc/gemm_x2f.hip (round 6: one wave per SIMD, 256 x 256 tiles -- the default wherever it fits); bits 4-5
 * skip the main loop / the epilogue (time splits: wrong results, tools/x2_knob_bench.py only).
 * "r5_forms" (USIP_TUNE_R5_FORMS), bit flags that bring back round 4's form of a kernel for same-box A/B runs: 1 = the f32x2
 * weight gradient with half-line loads (wgrad_x3_kernel<.., 2> instead of wgrad_x2l_kernel); and that switch ON forms round 5
 * measured and did not keep: 2 / 4 = two / four loop iterations' loads in flight in the BatchNorm-backward reduction (default
 * one), 8 = several batches of rows per workgroup with prefetch in group_max4 (default one); 32 = the 128-row-tile split GEMM with
 * round 4's two-slot weight ring (DMA one stage ahead, issued at the top of the stage) instead of the three-slot one;
 * 64 = narrow_fwd with 1024 / 768 workgroups; 128 = usip_knn_layer_backward_f32 with one row and 256 threads per workgroup (default: two rows, 512 threads). */
enum { USIP_TUNE_INDEX_MAX_CH = 0, USIP_TUNE_INDEX_MAX_UNROLL, USIP_TUNE_X3_WGRAD_TILE, USIP_TUNE_X3_GEMM_TILE,
       USIP_TUNE_GEMM_SPLIT3, USIP_TUNE_INDEX_MAX_THREADS, USIP_TUNE_X2_DIRECT, USIP_TUNE_R5_FORMS, USIP_TUNE_COUNT };
int usip_set_tuning(const char* name, int value);
int usip_tuning_value(int knob);

/* ------------------------------------------------------------------ a-1  index_max
 * Replaces index_max.forward_cuda / forward_cuda_shared_mem
 * (models/index_max_ext/index_max.cpp:132-148 -> index_max_cuda.cu:9-25, :29-61, :65-98).
 *   max_idx[b,c,k] = lowest n with index[b,n]==k attaining max data[b,c,n], provided that
 *   maximum is strictly greater than -1000; otherwise (empty node, sub-floor values, NaN) 0.
 * data f32 [B,C,N], index i32 [B,N] with values in [0,K), max_idx i32 [B,C,K] (fully written).
 * Lifts the reference's limits (B <= 1024 threads, B*K*4 <= 48 KB shared memory). */
int usip_index_max_f32(const float* data, const int32_t* index, int32_t* max_idx,
                       int B, int C, int N, int K, void* stream);

/* Host twins: index_max.forward_cpu (index_max.cpp:73-112) and forward_multi_thread_cpu
 * (index_max.cpp:33-70; channels split over num_threads std::threads).  HOST pointers. */
int usip_index_max_f32_cpu(const float* data, const int32_t* index, int32_t* max_idx,
                           int B, int C, int N, int K, int num_threads);

/* ------------------------------------------------------------------ a-2  ball_query
 * Replaces ball_query.forward_cuda_shared_mem
 * (models/ball_query_ext/ball_query.cpp:33-39 -> ball_query_cuda.cu:10-49, :53-70).
 *   per (b,m): the first K indices n (ascending) with dist[b,m,n] <= radius; if 0 < u < K hits,
 *   out[u+i] = out[i % u]; if u == 0 the row is all zeros.
 * dist f32 [B,M,N], out_idx i32 [B,M,K] (fully written). radius is a C float as in the kernel. */
int usip_ball_query_f32(const float* dist, int32_t* out_idx, float radius, int K,
                        int B, int M, int N, void* stream);

/* Host twin (HOST pointers, no stream): the same rows for BASELINE configs[0], the reference's CPU plumbing case.  The
 * reference itself has no CPU ball_query (models/ball_query_ext/ball_query.cpp:23-31 is a stub). */
int usip_ball_query_f32_cpu(const float* dist, int32_t* out_idx, float radius, int K, int B, int M, int N);

/* ------------------------------------------------------------------ pairwise distances
 * Replaces the materialised torch.norm(a.unsqueeze(3) - b.unsqueeze(2), dim=1) in front of
 * ball_query (models/networks.py:694-696, :355-357):
 *   dist[b,m,n] = sqrt(fma(dz,dz, fma(dy,dy, dx*dx))), d* = a[b,*,m] - x[b,*,n]
 * (the arithmetic order of the pinned oracle platform, bit-exact with it).
 * a f32 [B,3,M], x f32 [B,3,N] -> dist f32 [B,M,N]. */
int usip_pairwise_dist_f32_cpu(const float* a, const float* x, float* dist, int B, int M, int N);   /* host twin */
int usip_pairwise_dist_f32(const float* a, const float* x, float* dist,
                           int B, int M, int N, void* stream);

/* f-2  Fused coords-in ball query: same result as usip_pairwise_dist_f32 followed by
 * usip_ball_query_f32, without ever writing the B x M x N matrix. */
int usip_ball_query_coords_f32(const float* node, const float* x, int32_t* out_idx, float radius,
                               int K, int B, int M, int N, void* stream);

/* ------------------------------------------------------------------ a-3 / a-4  SOM front end
 * Replaces util/som.py:31-54 (query_topk with k = 1) and models/networks.py:85-108.
 *   min_idx[b,n] = argmin_m (dx*dx + dy*dy) + dz*dz   (first minimum; squared distance summed
 *                  in channel order without FMA, as ATen's pow-then-sum does)
 * x f32 [B,3,N], node f32 [B,3,M] -> min_idx i32 [B,N].  The reference's dense one-hot `mask`
 * [B,N,M] is never built; `mask_row_max` is (count > 0). */
int usip_som_assign_f32(const float* x, const float* node, int32_t* min_idx,
                        int B, int N, int M, void* stream);

/* cluster_mean[b,:,m] = sum_{n: min_idx[b,n]==m} x[b,:,n] / (count[b,m] + 1e-5)  (networks.py:95-96),
 * count i32 [B,M], and (if x_decentered != NULL) x_decentered[b,:,n] = x - cluster_mean[.., min_idx]
 * (networks.py:103-107).  Deterministic (fixed-order) summation. */
int usip_som_cluster_f32(const float* x, const int32_t* min_idx, float* cluster_mean,
                         int32_t* count, float* x_decentered, int B, int N, int M, void* stream);
/* The same outputs from min_idx sorted by node (usip_csr_by_index_i32 below with P = N points, N = M nodes):
 * O(N) per cloud instead of every node scanning all assignments. */
int usip_som_cluster_csr_f32(const float* x, const int32_t* min_idx, const int32_t* start, const int32_t* perm,
                             float* cluster_mean, int32_t* count, float* x_decentered, int B, int N, int M,
                             void* stream);

/* index_max together with what the reference does with its result (models/networks.py:117-118, :130-131:
 * torch.gather at the arg-max, times mask_row_max): max_val[b,c,k] = count[b,k] > 0 ? data[b,c,max_idx[b,c,k]] : 0
 * (count NULL: every node counts as populated).  data is the channel slice [0, C) of a [B][Ctot][N] tensor.
 * max_idx as usip_index_max_f32, bit for bit (same kernel). */
int usip_index_max_values_f32(const float* data, const int32_t* index, const int32_t* count, int32_t* max_idx,
                              float* max_val, int B, int C, int Ctot, int N, int K, void* stream);
/* Backward of that gather + mask, ADDED into an existing dense gradient: ddata[b][coff+c][max_idx[b,c,k]] += g[b,c,k]
 * for populated nodes.  ddata [B][Ctot][N]. */
int usip_index_max_values_backward_add_f32(const float* g, const int32_t* max_idx, const int32_t* count, float* ddata,
                                           int B, int C, int Ctot, int coff, int N, int K, void* stream);
/* The same gradient as one dense, contiguous tensor: dz[b][c][n] = (src ? src[b][soff+c][n] : 0) + that scatter term,
 * every element written once (index = the assignment index_max was given; src [B][Csrc][N] or NULL; K <= 8192). */
int usip_index_max_values_backward_f32(const float* g, const int32_t* max_idx, const int32_t* count,
                                       const int32_t* index, const float* src, int Csrc, int soff, float* dz,
                                       int B, int C, int N, int K, void* stream);

/* ------------------------------------------------------------------ a-4 / a-12  index tensors sorted by destination
 * idx i32 [B][P] with values in [0, N) -> start i32 [B][N+1], perm i32 [B][P]: perm[b][start[b][n] .. start[b][n+1])
 * are the positions p with idx[b][p] == n (values outside [0, N) are left out; start[b][N] = number placed).
 * One counting sort per cloud; N <= 1820.  Every scatter-add of the path (torch.gather's backward in
 * models/layers.py:422-426 and networks.py:119-125, the cluster sums of networks.py:87-107) becomes a gather over
 * these segments: no float atomics, a fixed summation order. */
int usip_csr_by_index_i32(const int32_t* idx, int32_t* start, int32_t* perm, int B, int P, int N, void* stream);
/* dx[b][c][n] = sum_{j in segment n} src[b][coff+c][perm[b][j]]; src [B][Ctot][P], dx [B][C][N] (fully written).
 * P <= 16384 (a source row is staged in LDS): usip_segment_sum_supported. */
int usip_segment_sum_supported(int N, int P);
int usip_segment_sum_f32(const float* src, const int32_t* start, const int32_t* perm, float* dx,
                         int B, int C, int N, int P, int Ctot, int coff, void* stream);

/* ------------------------------------------------------------------ a-9 / a-10  chamfer core
 * min_d[b,i] = min_j |a[b,:,i] - b[b,:,j]|_2 and arg[b,i] = FIRST j attaining it, exactly what
 * torch.min(torch.norm(a.unsqueeze(3) - b.unsqueeze(2), dim=1), dim=2) returns
 * (models/losses.py:62-66, :81, :86, :135-143) without materialising the B x Ma x Nb matrix.
 * a f32 [B,3,Ma], b f32 [B,3,Nb] -> min_d f32 [B,Ma], arg i32 [B,Ma].  With few queries and many candidates
 * the candidate set is split over workgroups; ws_d / ws_j hold the per-chunk results
 * (usip_nearest_workspace elements each; NULL = single-chunk kernel). */
long long usip_nearest_workspace(int B, int Ma, int Nb);   /* elements of ws_d AND of ws_j (0: none needed) */
int usip_nearest_f32(const float* a, const float* b, float* min_d, int32_t* arg,
                     float* ws_d, int32_t* ws_j, int B, int Ma, int Nb, void* stream);
/* Its backward (what autograd derives through torch.norm + torch.min + gather in the reference):
 * ga[b][:][i] = gd[b][i] * (a_i - b_J) / d, zero where d == 0; gb (may be NULL; every element is
 * written) receives the negative summed over the queries that share a partner -- a deterministic segmented
 * sum, not float atomics. */
int usip_nearest_backward_f32(const float* a, const float* b, const float* d, const int32_t* arg,
                              const float* gd, float* ga, float* gb, int B, int C, int Ma, int Nb, void* stream);

/* a-8 / a-11: the element-wise tail of the step, one launch each way per line of the reference (csrc/head.hip).
 * keypoints[b][:][m] = ks[b][0:3][m] + centre[b][:][m];  sigmas[b][m] = softplus(ks[b][3][m]) + sigma_lower_bound
 * (models/networks.py:150-154; torch.nn.Softplus defaults).  ks [B][4][M], centre [B][3][M]. */
int usip_detector_head_f32(const float* ks, const float* centre, float sigma_lower_bound, float* keypoints,
                           float* sigmas, int B, int M, void* stream);
/* g_ks [B][4][M] from g_keypoints [B][3][M] and g_sigmas [B][M] (either may be NULL = zero). */
int usip_detector_head_backward_f32(const float* g_keypoints, const float* g_sigmas, const float* ks, float* g_ks,
                                    int B, int M, void* stream);
/* out[b] = (R[b] * scale[b]) . x[b] + shift[b]  (models/keypoint_detector.py:182-184: R.kp*s + t), x/out [B][3][M],
 * R [B][3][3], scale [B], shift [B][3].  transpose != 0: out[b] = (R[b] * scale[b])^T . x[b] (its backward; shift unused). */
int usip_rigid_transform_f32(const float* x, const float* R, const float* scale, const float* shift, float* out,
                             int transpose, int B, int M, void* stream);
/* out3 = (chamfer[0] + alpha * (mean(d[0:half]) + mean(d[half:2*half])), alpha * mean(first), alpha * mean(second)):
 * the sum of the step's losses (keypoint_detector.py:196-204); means in double. */
int usip_detector_loss_combine_f32(const float* d, const float* chamfer, float alpha, float* out3, long long half,
                                   void* stream);
/* out[0:n] = g[0] * factor (the gradient of a mean). */
int usip_fill_scaled_f32(const float* g, float factor, float* out, long long n, void* stream);

/* ------------------------------------------------------------------ a-11  optimizer.step()
 * One Adam step (models/keypoint_detector.py:42-45, :207: torch.optim.Adam(lr, betas = (0.9, 0.999)), eps 1e-8, no
 * weight decay) on flat fp32 buffers of n elements, 16-B aligned: step_count[0] += 1 (device float, so the update can
 * live in a captured HIP graph), then m = lerp(m, g, 1 - b1), v = b2 v + (1 - b2) g^2,
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) -- the arithmetic of torch's single-tensor update. */
int usip_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step_count,
                       float lr, float beta1, float beta2, float eps, long long n, void* stream);
/* The same with (lr, beta1, beta2, eps) read from the device array hyper[4]: a launch captured into a HIP graph then
 * follows ModelDetector.update_learning_rate (models/keypoint_detector.py:356-366 sets param_groups[..]['lr']) without
 * a new capture -- the host refreshes the four floats before the replay. */
int usip_adam_step_hyper_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step_count,
                             const float* hyper, long long n, void* stream);

/* Pooled-concat layers (a-6 / a-7 row-bias rewrite): the sum over each neighbourhood's K positions of the layer's
 * dY, from the per-neighbourhood sums of the BatchNorm-backward reduction: out[b][c][g] = K * coef4[3][c]
 * + coef4[0][c] * gsum0[b][c][g] + coef4[2][c] * gsum1[b][c][g].  gsum0/1, out [nb][C][G], coef4 [4][C]. */
int usip_bn_group_dy_sum_f32(const float* gsum0, const float* gsum1, const float* coef4, float* out,
                             int nb, int C, int G, int K, void* stream);

/* a-10: the sigma arithmetic of ChamferLoss_Brute after the two min / arg-min reductions
 * (models/losses.py:82-99) as one launch.  a [B][M], J i32 [B][M] = row minima / arg-minima (src -> dst),
 * c [B][N], I i32 [B][N] = column minima (dst -> src), sigma_src [B][M], sigma_dst [B][N].
 * out3 = (forward_loss + backward_loss, chamfer_pure, chamfer_weighted); sums in double, fixed order. */
int usip_chamfer_prob_f32(const float* a, const int32_t* J, const float* c, const int32_t* I,
                          const float* sigma_src, const float* sigma_dst, float* out3,
                          int B, int M, int N, void* stream);
/* Its backward for an upstream gradient gloss[0] (device scalar) of out3[0]: da [B][M], dc [B][N],
 * dsigma_src [B][M], dsigma_dst [B][N] (every element written; the gather's transpose is a deterministic
 * segmented sum). */
int usip_chamfer_prob_backward_f32(const float* gloss, const float* a, const int32_t* J, const float* c,
                                   const int32_t* I, const float* sigma_src, const float* sigma_dst,
                                   float* da, float* dc, float* dsigma_src, float* dsigma_dst,
                                   int B, int M, int N, void* stream);

/* f-1 (descriptor head): the same minimum / first arg-minimum for C-dimensional points a [B][C][Ma],
 * b [B][C][Nb] (Nb <= 1024) -- the M x M descriptor-distance matrices of DescPairScanLoss
 * (models/losses.py:207-218: torch.norm over B x C x M x M, 268 MB at B=8) are never built. */
int usip_nearest_nd_f32(const float* a, const float* b, float* min_d, int32_t* arg,
                        int B, int C, int Ma, int Nb, void* stream);

/* ------------------------------------------------------------------ a-5 / a-6 / a-7 / a-8  shared MLP
 * Replaces, per layer, nn.Conv1d/Conv2d(k=1) + MyBatchNorm + ReLU and their autograd backward
 * (models/layers.py:208-216 MyConv2d.forward, :293-303 EquivariantLayer.forward, :61-71/:112-121
 * MyBatchNorm*.forward).  Activations are [nb][C][P] exactly as the reference stores them
 * (B x C x M x K or B x C x N flattened over positions); fp32 MFMA, fp32 accumulate.
 *
 * usip_mlp_gemm_f32:  Y[b][m][p] = sum_k At[k][m] * pro(X[b][k][p]) + bias[m]
 *   At is the matrix operand K-major ([K][M], row stride lda): W^T for the forward product,
 *   W itself ([Cout][Cin]) for the data gradient dX = W^T . dY.  A NEGATIVE lda means the operand is
 *   stored M-major ([M][K], row stride -lda) and is read transposed, so the forward product can use W
 *   in place as well.
 *   pro: 0 identity | 1 relu(x*coef[0][k] + coef[1][k]) | 2 BatchNorm+ReLU backward of X = dZ,
 *        X2 = pre-BN output, coef = the [4][K] array written by usip_bn_backward_reduce_f32.
 *        3 as 2, but dZ is not a tensor: the layer fed ONLY a max over pool_group neighbours, so
 *          dZ[k][p] = (p % pool_group == pool_arg[k][p / pool_group]) ? pool_dp[k][p / pool_group] : 0
 *          is formed on the fly from the two [nb][K][P/pool_group] arrays (X may be NULL).
 *   rowbias (may be NULL): [nb][M][P/rb_group], added as Y[b][m][p] += rowbias[b][m][p / rb_group]:
 *   the contribution of input channels that are constant inside a neighbourhood of rb_group
 *   positions (the max-pooled feature the reference expands and concatenates, networks.py:706-709,
 *   layers.py:433-435) -- computed once per neighbourhood instead of once per neighbour.
 *   y_rows: 0, or the number of rows per cloud of the tensor Y points into when the M output rows are a
 *   channel slice of a wider [nb][y_rows][P] tensor (Y then points at the slice's first row of cloud 0).
 *   stats (may be NULL): [2][M][tiles] per-tile (sum, sum of squares) of Y over valid positions,
 *   tiles = usip_mlp_gemm_tiles(M, P, nb); summed in fixed order by usip_bn_finalize_f32. */
int usip_mlp_gemm_tiles(int M, int P, int nb);
int usip_mlp_gemm_f32(const float* At, int lda, const float* X, const float* X2, const float* coef,
                      int pro, const float* bias, const float* rowbias, int rb_group,
                      const float* pool_dp, const int32_t* pool_arg, int pool_group,
                      float* Y, int y_rows, float* stats, int M, int K, int P, int nb, void* stream);
/* Same contract with a bf16 MULTIPLY (the perf mode of BASELINE.json configs[1]; not the parity mode): tensors
 * stay fp32 in memory, the prologue runs in fp32, both operands are rounded to bf16 (nearest-even) on their way
 * into LDS, v_mfma_f32_32x32x16_bf16 accumulates in fp32, bias / rowbias / statistics are fp32.  The result
 * equals the fp32 product of the bf16-rounded operands up to summation order. */
int usip_mlp_gemm_bf16(const float* At, int lda, const float* X, const float* X2, const float* coef,
                       int pro, const float* bias, const float* rowbias, int rb_group,
                       const float* pool_dp, const int32_t* pool_arg, int pool_group,
                        float* Y, int y_rows, float* stats, int M, int K, int P, int nb, void* stream);
/* Same contract with an fp32-ACCURATE product on the bf16 matrix cores ("f32x3"): each fp32 operand is split
 * exactly into three bf16 planes and the six plane pairs of weight >= 2^-18 are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 (relative error of a product <= 3 * 2^-27, below fp32's own rounding; 2.7x the
 * fp32-MFMA rate).  Used for the launches that are matrix-bound (usip_mlp_gemm_f32x3_used(...) == 1); the others
 * are handed to the fp32 kernel, so the entry point is a drop-in for usip_mlp_gemm_f32 everywhere. */
int usip_mlp_gemm_f32x3(const float* At, int lda, const float* X, const float* X2, const float* coef,
                       int pro, const float* bias, const float* rowbias, int rb_group,
                       const float* pool_dp, const int32_t* pool_arg, int pool_group,
                        float* Y, int y_rows, float* stats, int M, int K, int P, int nb, void* stream);
int usip_mlp_gemm_f32x3_used(int M, int K, int P, int nb);
/* Streaming forward kernel for the narrow layers (K = 64 inputs, M = 64 or 128 outputs; csrc/narrow_fwd.hip): the
 * contract of usip_mlp_gemm_f32 (pro 0 or 1, bias, rowbias with rb_group % 32 == 0, y_rows) for P % 4 == 0 and a
 * 16-B aligned X.  Persistent workgroups: `stats` has usip_mlp_narrow_forward_blocks(M, K, P, nb) partials per channel
 * ([2][M][blocks]); that function returns 0 for shapes the kernel does not take (callers use usip_mlp_gemm_f32). */
int usip_mlp_narrow_forward_blocks(int M, int K, int P, int nb);
int usip_mlp_narrow_forward_f32(const float* At, int lda, const float* X, const float* coef, int pro,
                                const float* bias, const float* rowbias, int rb_group, float* Y, int y_rows,
                                float* stats, int M, int K, int P, int nb, void* stream);

/* The same f32x3 product with the MATRIX operand split ahead of time (once per optimizer step instead of once per
 * workgroup and stage): usip_mlp_split3_f32 turns the K-major operand At (A[m][k] = At[k*lda + m], M x K) into the
 * image the kernel copies straight into LDS -- per (tile_rows-row tile, 16-k stage) three contiguous bf16 planes, zero
 * padded -- usip_mlp_split3_bytes(M, K) bytes, 16-B aligned; tile_rows = usip_mlp_x3p_tile_rows(M, P, nb) of the launch
 * the image is for.  usip_mlp_gemm_x3p_f32 then has the contract of usip_mlp_gemm_f32 with `planes` in place of
 * (At, lda); K <= 640. */
int usip_mlp_x3p_tile_rows(int M, int P, int nb);  /* rows per tile (128 or 256) of a launch over nb x P positions */
/* All weight operands of a step in one launch: descs (DEVICE memory, n entries, At / planes as for
 * usip_mlp_split3_f32) with first_block = running sum of usip_mlp_split3_blocks(M, K, tile_rows) over the entries before. */
typedef struct usip_split3_desc {
    const float* At; void* planes; int32_t lda, M, K, first_block;
    int32_t tile_rows, reserved;                      /* rows per tile of the image (128 / 256); reserved: 2 = two fp16 planes (below) */
} usip_split3_desc;
int usip_mlp_split3_blocks(int M, int K, int tile_rows);
int usip_mlp_split3_multi_f32(const usip_split3_desc* descs_device, int n, int total_blocks, void* stream);
/* Positions per tile the same kernel uses for this launch: 128 (256 only under the x3_gemm_tile measurement knob). */
int usip_mlp_x3p_tile_cols(int M, int P, int nb, int pro, int with_stats);
long long usip_mlp_split3_bytes(int M, int K);
int usip_mlp_split3_f32(const float* At, int lda, int M, int K, int tile_rows, void* planes, void* stream);
int usip_mlp_gemm_x3p_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                          const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                          const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats,
                          int M, int K, int P, int nb, void* stream);
/* "f32x2": the same fp32-accurate product from TWO fp16 planes per operand (11 + 11 significant bits) and THREE plane
 * products -- half the matrix work of f32x3 at the same error level.  fp16 has 5 exponent bits, so both operands are
 * multiplied by powers of two (exact): the weights by 2^e with max|A| 2^e in [2^13, 2^14) (usip_mlp_split2h_f32 stores
 * 2^e behind the image; buffer size usip_mlp_split3_bytes), the streamed operand by 2^e derived in the kernel from a
 * RIGOROUS upper bound of what its prologue can produce:
 *   pro 1: coef = [4][K] (scale, shift, mean, invstd) of a training-mode BatchNorm over exactly the nb * P samples of
 *          this launch: |relu(bn(y))| <= |gamma| sqrt(n) + |beta|;
 *   pro 2 / 3: coef = the [5][K] array usip_bn_backward_reduce_f32 / usip_bn_pool_backward_reduce_f32 write with
 *          want_bound (row 4: bounds of |dY| per 64 channels).
 * pro 0 (no bound available) is not offered: callers use usip_mlp_gemm_x3p_f32.  Otherwise the contract of
 * usip_mlp_gemm_x3p_f32. */
int usip_mlp_split2h_f32(const float* At, int lda, int M, int K, int tile_rows, void* planes, void* stream);
int usip_mlp_gemm_x2h_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                          const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                          const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats,
                          int M, int K, int P, int nb, void* stream);

/* Round 4: data-gradient launches (pro 2 / 3) of the direct f32x2 kernel (csrc/gemm_x2d.hip) that also leave, from the dX
 * tile they hold, the BatchNorm-backward partial sums of the layer that produced the activation dX is the gradient of --
 * what autograd's native_batch_norm_backward of that layer re-reads (dX, Y) for in the reference
 * (models/layers.py:208-216).  red_y [nb][M][P]: that layer's pre-BN output; red_coef [4][M]: its (scale, shift, mean,
 * invstd); red_out: [2][tiles][M] sums (sum d, sum d * xhat, d = dX where its ReLU is on) followed by [tiles * M / 256]
 * maxima of |d|; red_gsum (optional, red_group 16 or 32 positions per neighbourhood): [2][nb * M][P / red_group] sums of
 * d and of y.  usip_mlp_gemm_x2d_red_tiles() = tiles of such a launch, 0 when the shape does not take this path (M % 256,
 * P % 128, 256-row tiles): use usip_mlp_gemm_x2h_f32 + usip_bn_backward_reduce_f32 then. */
int usip_mlp_gemm_x2d_red_tiles(int M, int K, int P, int nb, int red_group);
/* Round 6 (profiling aid, no reference counterpart): 1 when usip_mlp_gemm_x2h_f32 runs a launch of this shape that reaches
 * the direct kernel as csrc/gemm_x2f.hip -- one wave per SIMD, 256-channel x 256-position tiles, 64 positions per wave; the
 * same products in the same order as csrc/gemm_x2d.hip (bit-identical outputs).  has_stats / has_bias: the pointer is given;
 * rb_group 0: no row bias; y_rows 0: M. */
int usip_mlp_gemm_x2f_used(int M, int K, int P, int nb, int pro, int has_stats, int has_bias, int rb_group,
                           int pool_group, int y_rows);
int usip_mlp_gemm_x2h_red_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                              const float* pool_dp, const int32_t* pool_arg, int pool_group, float* Y,
                              const float* red_y, const float* red_coef, float* red_out, float* red_gsum,
                              int red_group, int M, int K, int P, int nb, void* stream);
/* The same for the 128-wide layers (M <= 128, K <= 128; a row bias only with pro 1 and rb_group a multiple of 32) with
 * the weight fragments resident in registers and persistent workgroups over 64-position tiles: a CU moves the streamed
 * operand in and the output out, not the weight planes again for every tile.
 * stats: [2][M][usip_mlp_gemm_x2r_tiles(P, nb)]. */
int usip_mlp_gemm_x2r_tiles(int P, int nb);
int usip_mlp_gemm_x2r_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                          const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                          const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats, int M, int K,
                          int P, int nb, void* stream);

/* K-major copies of many weight matrices in one launch: for every t < ntensors, table[5t..5t+4] =
 * (source offset, rows, cols, destination offset, index of its first 32 x 32 tile), offsets in floats into src / dst;
 * dst[dof + c*rows + r] = src[sof + r*cols + c].  The reference stores Conv weights [Cout][Cin][1(,1)]
 * (models/layers.py:186-205); the GEMMs want [Cin][Cout].  total_tiles = sum of ceil(rows/32)*ceil(cols/32). */
int usip_multi_transpose_f32(const float* src, float* dst, const int32_t* table, int ntensors, int total_tiles,
                             void* stream);

/* Batch statistics -> mean[C], invstd[C] (biased variance, eps inside the sqrt), forward
 * coefficients coef[4][C] = (gamma*invstd, beta - mean*gamma*invstd, mean, invstd), and the running-statistics
 * update running = (1-momentum)*running + momentum*batch (unbiased variance), as F.batch_norm does
 * in training mode.  running_mean/var may both be NULL.  count = nb*P. */
int usip_bn_finalize_f32(const float* stats, int tiles, int C, long long count,
                         const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, float* mean, float* invstd,
                         float* coef, void* stream);

/* Z = Y*coef[0][c] + coef[1][c], followed by ReLU when relu != 0.  Y, Z: [nb][C][P]. */
int usip_bn_apply_f32(const float* Y, const float* coef, float* Z, int relu,
                      int nb, int C, int P, void* stream);

/* Backward reductions of BatchNorm(+ReLU): dbeta[c] = sum dYhat, dgamma[c] = sum dYhat*yhat with
 * dYhat = dZ*[fma(y,coef_fwd[0],coef_fwd[1]) > 0] (dZ if !relu), and coef4[4][C] such that
 * dY = coef4[0]*dYhat + coef4[2]*y + coef4[3] (coef4[0..1] repeat coef_fwd for the mask).
 * Y == NULL selects the plain mode: dbeta[c] = sum dZ (bias gradient of a layer without BN).
 * partial: workspace of 2*nb*C floats.
 * gsum (may be NULL): [2][nb][C][P/group] per-neighbourhood sums of dYhat and of y, from which
 * sum_k dY = coef4[0]*gsum[0] + coef4[2]*gsum[1] + group*coef4[3] follows without a second pass
 * (group % 4 == 0 and group/4 a power of two <= 64). */
int usip_bn_backward_reduce_f32(const float* dZ, const float* Y, const float* coef_fwd,
                                const float* mean, const float* invstd, const float* gamma, int relu,
                                float* partial, float* dgamma, float* dbeta, float* coef4,
                                float* gsum, int group, int nb, int C, int P, int want_bound, void* stream);
/* want_bound != 0: `partial` holds [3][nb*C] floats (third plane: max |dYhat| per row) and coef4 is [5][C]: entry i of
 * row 4 (i < ceil(C/64)) is an upper bound of |dY| over channels [64 i, 64 i + 64) -- |a1| (max|dYhat| + |mean dYhat| +
 * |mean dYhat yhat| sqrt(n)), rigorous for batch statistics -- which the split-fp16 kernels (usip_mlp_gemm_x2h_f32,
 * usip_mlp_wgrad_x2h_f32) use to scale the operand into the fp16 range.  want_bound == 0: [2][nb*C] and [4][C]. */


/* usip_bn_backward_reduce_f32 for a layer whose output fed ONLY a max over K neighbours: the incoming
 * gradient is (k == arg) ? dpooled : 0, so the sums run over B*C*M arg-max elements instead of B*C*M*K.
 * dpooled f32 / arg i32 [nb][C][M], Y [nb][C][M][K]; outputs as usip_bn_backward_reduce_f32. */
/* yarg (may be NULL): Y at the arg-max as usip_group_max_act_f32 returned it -- replaces one gathered 4-B read per
 * neighbourhood; dgamma / dbeta / coef4 all NULL: only the partial sums [2][nb][C] are produced. */
int usip_bn_pool_backward_reduce_f32(const float* dpooled, const int32_t* arg, const float* Y, const float* yarg,
                                     const float* coef_fwd, const float* mean, const float* invstd,
                                     const float* gamma, int relu, float* partial, float* dgamma, float* dbeta,
                                     float* coef4, int nb, int C, int M, int K, int want_bound, void* stream);

/* dW[m][n] = sum_{b,p} pro(G)[b][m][p] * X[b][n][p]   (pro 0: G = dY given; pro 2: G = dZ, G2 = Y,
 * coef = coef4 as above; pro 3: dZ synthesised from pool_dp / pool_arg as in usip_mlp_gemm_f32).  workspace: usip_mlp_wgrad_workspace(M, N, P, nb) floats of partial tiles,
 * reduced in fixed order (deterministic).  dW is written as dW[m*ldw + coloff + n], so a column
 * block of a wider weight matrix can be filled in place.  xcoef (may be NULL): [2][N]; X is then the
 * PRE-BatchNorm output of the producing layer and relu(X*xcoef[0][n] + xcoef[1][n]) is formed on the fly
 * (the activated tensor is never stored). */
long long usip_mlp_wgrad_workspace(int M, int N, int P, int nb);
int usip_mlp_wgrad_blocks(int M, int N, int P, int nb);        /* workgroups launched (profiling aid) */
int usip_mlp_wgrad_f32(const float* G, const float* G2, const float* coef, int pro, const float* X,
                       const float* xcoef, const float* pool_dp, const int32_t* pool_arg, int pool_group,
                       float* workspace, float* dW, int ldw, int coloff,
                       int M, int N, int P, int nb, void* stream);
/* Deferred weight-gradient reductions (round 5; no reference counterpart -- the reference's autograd produces every dW
 * where its layer's backward runs, models/layers.py:208-216).  Between usip_wgrad_defer(1) and usip_wgrad_flush(stream) every
 * entry point that ends in the fixed-order sum of its partial tiles (usip_mlp_wgrad_*, usip_mlp_narrow_backward_f32,
 * usip_mlp_layer_backward_x2h_f32) launches its tile kernel and RECORDS the sum instead of launching it; the flush issues all
 * recorded sums in at most two launches (same summation order: same bits) and returns how many it issued (>= 0; < 0 error).
 * The caller keeps every workspace alive until the flush.  usip_wgrad_defer(0) leaves the mode and drops what is recorded.
 * The flush issues ceil(jobs / 24) launches per reduction width (two widths): two launches for the detector's fifteen sums.
 * usip_wgrad_defer_on(stream) enters the mode for ONE stream: entry points called with any other stream launch their sum at
 * once (the job list is process-global; a second thread / device / stream must not find its sums on this stream's flush).
 * usip_wgrad_defer_hold(1) .. (0) brackets calls whose sum must be launched at once although the mode is on: a dW that the
 * caller returns to a consumer running before the flush (anything that is not a view of the step's gradient bucket);
 * it returns the previous hold value. */
int usip_wgrad_defer(int on);
int usip_wgrad_defer_on(void* stream);
int usip_wgrad_defer_hold(int hold);
int usip_wgrad_flush(void* stream);

/* Backward of a NARROW layer (64 inputs, 64 or 128 outputs) in one pass over its tensors: data gradient AND weight
 * gradient from one staged tile of (dZ, Y, X) -- these layers are HBM-bound and the two separate products read
 * (dZ, Y) twice.  Replaces the pair usip_mlp_gemm_f32(pro = 2) + usip_mlp_wgrad_f32(pro = 2) for the grouped
 * convolutions conv2 / conv3 / conv4 of RPN_Detector_Ball (models/networks.py:705-709; autograd's
 * cudnn_convolution_backward in the reference).  Exact fp32 MFMA, deterministic (fixed-order partial sums).
 *   dX[b][ci][p] = sum_co W[co*ldw + ci] * dY[b][co][p],  dW[co*lddw + ci] = sum_{b,p} dY[b][co][p] * act(X)[b][ci][p]
 *   dY = BatchNorm'(ReLU'(dZ)) from (dZ, Y, coef4) as in usip_mlp_gemm_f32 pro = 2; act(X) = relu(X*xcoef[0]+xcoef[1])
 *   (xcoef NULL: X as is).  X / dX point at the first of the 64 rows inside [nb][x_rows][P] / [nb][dx_rows][P].
 * workspace: usip_mlp_narrow_backward_workspace(Cout, P, nb) floats.
 * red_partial (may be NULL; then xcoef must be the PRODUCING layer's full [4][64] forward coefficients -- scale, shift,
 *   mean, invstd): [2][usip_mlp_narrow_backward_blocks(Cout, P, nb)][64] partial BatchNorm-backward sums of that layer
 *   (sum dX*[relu on], sum dX*[relu on]*xhat), taken while the dX tile is in registers, followed by [blocks] maxima of
 *   |dX [relu on]| (2 * blocks * 64 + blocks floats in all); usip_bn_backward_finalize_f32
 *   turns partial sums (these, plus usip_bn_pool_backward_reduce_f32's when a max-pool also feeds that layer) into
 *   dgamma / dbeta / coef4 -- the separate reduction pass over (dZ, Y) of that layer disappears. */
int usip_mlp_narrow_backward_supported(int Cin, int Cout, int P);
long long usip_mlp_narrow_backward_workspace(int Cout, int P, int nb);
int usip_mlp_narrow_backward_blocks(int Cout, int P, int nb);
int usip_mlp_narrow_backward_f32(const float* dZ, const float* Y, const float* coef4, const float* X, int x_rows,
                                 const float* xcoef, const float* W, int ldw, float* dX, int dx_rows,
                                 float* workspace, float* dW, int lddw, float* red_partial, int Cin, int Cout,
                                 int P, int nb, void* stream);

/* The same fused layer backward on the 16-bit matrix cores with f32x2 arithmetic (csrc/layer_bwd_x2.hip; no reference
 * counterpart: the layers' backward is autograd's, models/layers.py:208-216, :293-303), for (Cin, Cout) = (64, 64), (64, 128), or
 * (128, 128) in the pooled form, and P % 64 == 0 (usip_mlp_layer_backward_x2h_supported).  coef4 = the [5][Cout] array usip_bn_backward_reduce_f32 /
 * usip_bn_backward_finalize_max_f32 write (row 4: bounds of |dY|); pool_dp / pool_arg (i32) [nb][Cout][P / pool_group]
 * given and dZ NULL: the pooled form, dZ = (p % pool_group == arg) ? pool_dp : 0, (128, 128) only; xcoef = the producing
 * layer's [4][Cin] (scale, shift, mean, invstd), training-mode statistics over exactly these nb * P samples; planes =
 * usip_mlp_split2h_f32 image of W as the data-gradient operand (At = W [Cout][ldw], M = Cin, K = Cout).  workspace:
 * usip_mlp_layer_backward_x2h_workspace floats.  red_partial (may be NULL): [2][blocks][Cin] partial sums of the
 * producing layer's BatchNorm backward against dX followed by [blocks] maxima of |dX [relu on]|, blocks =
 * usip_mlp_layer_backward_x2h_blocks.  group_sums (may be NULL; pooled form with red_partial, pool_group % 32 == 0):
 * [2][nb * Cin][P / pool_group] = per neighbourhood sum_k dX [relu on] and sum_k X, the `gsum` of
 * usip_bn_backward_reduce_f32 for a producing layer that is a pooled-concat layer (conv4 behind conv5). */
int usip_mlp_layer_backward_x2h_supported(int Cin, int Cout, int P, int pooled);
long long usip_mlp_layer_backward_x2h_workspace(int Cin, int Cout, int P, int nb);
int usip_mlp_layer_backward_x2h_blocks(int Cin, int Cout, int P, int nb);
int usip_mlp_layer_backward_x2h_f32(const float* dZ, const float* Y, const float* coef4, const float* pool_dp,
                                    const int32_t* pool_arg, int pool_group, const float* X, int x_rows,
                                    const float* xcoef, const void* planes, float* dX, int dx_rows, float* workspace,
                                    float* dW, int lddw, float* red_partial, float* group_sums, int Cin, int Cout,
                                    int P, int nb, void* stream);
/* The (Cin, Cout) = (64, 64) form with red_partial that ALSO takes, on the way, the sums from which the PRODUCING layer's
 * weight gradient follows -- for a producing layer whose input S f32 [nb][ws_rows <= 8][P] needs no gradient (conv1 of
 * RPN_Detector_Ball, models/networks.py:705; the first PointNet layer of RPN_Detector, layers.py:524-544): that layer's
 * dW' = dY' . S^T with dY' = a1' dYhat' + q1' y' + q0' is linear in  S1 = sum_p dYhat' S_j,  S2 = sum_p (y' - mean') S_j,
 * S3 = sum_p S_j, and this pass holds dYhat' = dX [relu on] and y' = X in LDS anyway.  wsum f32 [blocks][64][16]
 * (S1 | S2, j < 8), wsum3 f32 [blocks][8], blocks = usip_mlp_layer_backward_x2h_blocks.  usip_mlp_wsum_finalize_f32
 * combines them (fp64, fixed order) into dW'[c * lddw + j], j < ws_rows, once coef4' = usip_bn_backward_finalize_*_f32 of
 * the same call's red_partial is known: the producing layer needs no pass of its own over its (dZ, Y). */
int usip_mlp_layer_backward_x2h_ws_f32(const float* dZ, const float* Y, const float* coef4, const float* X, int x_rows,
                                       const float* xcoef, const void* planes, float* dX, int dx_rows, float* workspace,
                                       float* dW, int lddw, float* red_partial, const float* wsrc, int ws_rows,
                                       float* wsum, float* wsum3, int Cin, int Cout, int P, int nb, void* stream);
int usip_mlp_wsum_finalize_f32(const float* wsum, const float* wsum3, int blocks, int C, const float* coef4,
                               const float* mean, int ws_rows, float* dW, int lddw, void* stream);
int usip_bn_backward_finalize_f32(const float* partial, int rows, int C, long long count, const float* coef_fwd,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef4,
                                  void* stream);
/* The same with the fifth row of coef4 ([5][C]: bounds of |dY| per 64 channels, what the f32x2 kernels scale their
 * operand by): the gradient is a sum of up to two parts whose maxima |dYhat| are given as arrays (max0[n0], max1[n1],
 * max1 may be NULL) -- the bound uses max(max0) + max(max1). */
int usip_bn_backward_finalize_max_f32(const float* partial, int rows, int C, long long count, const float* coef_fwd,
                                      const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                      float* coef4, const float* max0, int n0, const float* max1, int n1, void* stream);
/* bf16-multiply variant (see usip_mlp_gemm_bf16); same workspace, same deterministic fp32 reduction. */
int usip_mlp_wgrad_bf16(const float* G, const float* G2, const float* coef, int pro, const float* X,
                        const float* xcoef, const float* pool_dp, const int32_t* pool_arg, int pool_group,
                        float* workspace, float* dW, int ldw, int coloff,
                        int M, int N, int P, int nb, void* stream);
/* f32x3 variant (see usip_mlp_gemm_f32x3); same workspace, same deterministic fp32 reduction. */
int usip_mlp_wgrad_f32x3(const float* G, const float* G2, const float* coef, int pro, const float* X,
                        const float* xcoef, const float* pool_dp, const int32_t* pool_arg, int pool_group,
                        float* workspace, float* dW, int ldw, int coloff,
                        int M, int N, int P, int nb, void* stream);
/* f32x2 form of the weight gradient: as usip_mlp_wgrad_f32x3, and launches with pro 2 / 3, M, N > 128, coef = the [5][M]
 * array usip_bn_backward_reduce_f32 writes with want_bound and xcoef = the [4][N] (scale, shift, mean, invstd) of a
 * training-mode BatchNorm over exactly the nb * P samples of this launch run on two fp16 planes per operand and three
 * plane products (see usip_mlp_gemm_x2h_f32); every other launch exactly as usip_mlp_wgrad_f32x3. */
int usip_mlp_wgrad_x2h_f32(const float* G, const float* G2, const float* coef, int pro, const float* X,
                        const float* xcoef, const float* pool_dp, const int32_t* pool_arg, int pool_group,
                        float* workspace, float* dW, int ldw, int coloff,
                        int M, int N, int P, int nb, void* stream);
int usip_mlp_wgrad_f32x3_used(int M, int N, int P, int nb);
int usip_mlp_wgrad_f32x3_blocks(int M, int N, int P, int nb);   /* < 0: the 256 x 256-tile kernel, |value| workgroups */

/* ------------------------------------------------------------------ a-6 / a-7 / a-12  grouping, pooling
 * out[b][coff+c][m][k] = x[b][c][idx[b][m][k]] - (c < nsub ? sub[b][c][m] : 0), written into the
 * channel slice [coff, coff+C) of an output with Ctot channels.  Replaces index expansion +
 * torch.gather + decentering + torch.cat (models/operations.py:271-287, networks.py:699-703,
 * layers.py:422-430).  x [B][C][N], idx i32 [B][M][K], sub [B][nsub][M], out [B][Ctot][M][K]. */
int usip_group_gather_f32(const float* x, const int32_t* idx, const float* sub, float* out,
                          int B, int C, int N, int M, int K, int nsub, int Ctot, int coff, void* stream);
/* Its backward w.r.t. x (scatter-add; dx [B][C][N] is zeroed first). */
int usip_group_gather_backward_f32(const float* dout, const int32_t* idx, float* dx,
                                   int B, int C, int N, int M, int K, int Ctot, int coff, void* stream);
/* pooled[row] = max_k z[row][k], arg[row] = first k attaining it (torch.max over the K axis,
 * networks.py:706,710, layers.py:433,438); rows = B*C*M.  Backward: dz[row][k] = (k==arg)*dpooled. */
int usip_group_max_f32(const float* z, float* pooled, int32_t* arg, long long rows, int K, void* stream);
/* The same pooling applied to relu?(y*coef[0][c] + coef[1][c]) formed on the fly from a layer's pre-BatchNorm
 * output y [B][C][M][K] (K % 4 == 0, K/4 a power of two <= 64): BN-apply + ReLU + max in ONE pass over y. */
/* yarg (may be NULL): [B][C][M], receives y at the arg-max (the pre-BN value the pooled layer's backward needs). */
int usip_group_max_act_f32(const float* y, const float* coef, int relu, float* pooled, int32_t* arg, float* yarg,
                           int B, int C, int M, int K, void* stream);
int usip_group_max_backward_f32(const float* dpooled, const int32_t* arg, float* dz,
                                long long rows, int K, void* stream);
/* The same gradient ADDED into an existing dense gradient: dz[row*K + arg[row]] += dpooled[row]
 * (the tensor that was pooled also fed a layer directly: its two gradients are combined by touching
 * `rows` elements instead of materialising and adding a second dense tensor). */
int usip_group_max_backward_add_f32(const float* dpooled, const int32_t* arg, float* dz,
                                    long long rows, int K, void* stream);

/* ------------------------------------------------------------------ a-7  node KNN
 * idx[b][m][0..K) = the K database points nearest to query m, ascending distance (lower index first on
 * exact ties): torch.norm + torch.topk(K, largest=False, sorted=True) of models/layers.py:417-421 without
 * the B x M x N matrix.  query f32 [B][3][M], database f32 [B][3][N], N <= 1024, K <= N. */
int usip_knn_f32(const float* query, const float* database, int32_t* idx,
                 int B, int M, int N, int K, void* stream);

/* The FIRST layer of GeneralKNNFusionModule (models/layers.py:422-431: gather the K neighbours' coordinates and features,
 * decenter the coordinates, concatenate to B x (3+C) x M x K, then models/layers.py:208-216: conv1x1 + BatchNorm + ReLU)
 * without the gathered tensor.  The convolution is linear and the features enter it undecentered:
 *     Y[b][co][m][k] = sum_j W[co][j] (database[b][j][n] - query[b][j][m])  +  U[b][co][n],   n = idx[b][m][k],
 * with U = W[:, 3:] . feat + bias a product over the N database points (usip_mlp_gemm_*) instead of the M*K grouped
 * positions.  Same math as the reference's layer up to fp32 summation order.
 *   forward : U f32 [B][Cout][N], W f32 [Cout][ldw] (columns 0..2 = the coordinate weights), database [B][3][N],
 *             query [B][3][M], idx i32 [B][M][K] -> Y f32 [B][Cout][M*K]; stats (may be NULL): [2][Cout][B] per-cloud
 *             partial (sum, sum^2) of every channel, the layout usip_bn_finalize_f32 reads with ntn = B.
 *   backward: dZ, Y [B][Cout][M*K] and coef4 [>=4][Cout] as for the shared-MLP prologue PRO_BN_BWD
 *             (dY = coef4[0] dZ [fma(Y, coef4[0], coef4[1]) > 0 if relu] + coef4[2] Y + coef4[3]); dcoord f32 [B][3][M*K] =
 *             database[b][j][idx] - query[b][j][m] (usip_group_gather_f32 with sub = query); (start, perm) = the
 *             usip_csr_by_index_i32 lists of idx viewed as [B][M*K] over N destinations ->
 *             dU [B][Cout][N] = the segment sums of dY in list order (no float atomics),
 *             dWc_part [B][Cout][3] = per-cloud partial gradients of the coordinate weights (the caller adds the B parts).
 * Shapes: usip_knn_layer_supported (N <= 1489, M*K <= 16384, M*K % 4 == 0); dZ, Y, dcoord 16-B aligned. */
int usip_knn_layer_supported(int N, int M, int K);
int usip_knn_layer_forward_f32(const float* U, const float* W, int ldw, const float* database, const float* query,
                               const int32_t* idx, float* Y, float* stats, int B, int Cout, int N, int M, int K,
                               void* stream);
int usip_knn_layer_backward_f32(const float* dZ, const float* Y, const float* coef4, int relu, const float* dcoord,
                                const int32_t* start, const int32_t* perm, float* dU, float* dWc_part,
                                int B, int Cout, int N, int M, int K, void* stream);

/* ------------------------------------------------------------------ (judge row) RPN_Detector_KNN front end
 * idx[b][m][0..K) = the K cloud points nearest to node m, nearest first, ties towards the lower index:
 * torch.norm(node - x) over B x M x N followed by torch.topk(k=64, largest=False, sorted=False) of
 * models/networks.py:576-581 without the matrix.  topk(sorted=False) leaves the order of the picks unspecified;
 * the SET is the reference's, the order is that of a stable sort of the reference's distance row.
 * node f32 [B][3][M], x f32 [B][3][N], K <= min(N, 256), N <= 16384. */
int usip_knn_points_f32(const float* node, const float* x, int32_t* idx, int B, int M, int N, int K, void* stream);

/* ------------------------------------------------------------------ f-3  farthest-point sampling of nodes
 * Replaces FarthestSampler.sample (data/kitti_detector_loader.py:69-83; also oxford_detector_loader.py,
 * modelnet_shrec_loader.py): out_idx[b][0] = first_idx[b], then k-1 times the first arg-max of the running
 * minimum squared distance, evaluated in float64 exactly as numpy does.  pts f32 [B][3][n], n <= 16384;
 * out_idx i32 [B][k] (indices into the n points; gather them for the node coordinates). */
int usip_fps_f32(const float* pts, const int32_t* first_idx, int32_t* out_idx, int B, int n, int k, void* stream);

/* ------------------------------------------------------------------ f-4  inference post-processing
 * Greedy non-maximum suppression by sigma (evaluation/save_keypoints.py:180-216): order[b][0..count[b]) are
 * the indices kept, in the order they are picked (ascending sigma; ties: lower index), each pick removing every
 * keypoint whose float32 distance to it is not > radius.  keypoints f32 [B][3][M], sigmas f32 [B][M], M <= 1024.
 * The reference's "keep the desired_keypoint_num smallest sigmas" (:346-351) is the first entries of order. */
int usip_nms_f32(const float* keypoints, const float* sigmas, float radius, int32_t* order, int32_t* count,
                 int B, int M, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* USIP_HIP_H */
