/*
 * mval_hip.h -- C ABI of libmval_hip.so: the MI355X (gfx950) implementation of the
 * data-parallel hot path of facebookresearch/multi_view_active_learning.
 *
 * The reference has NO native layer: its boundary is a Python call surface.
 * Each entry point below therefore cites the reference *Python call site* it replaces
 * (paths relative to the reference root), and multi_view_active_learning_amd/_lib.py is the
 * ctypes binding a maintainer would use (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer borrowed from the
 *     caller (who keeps it alive until the stream has drained); nothing is allocated
 *     inside a launcher except via caller-provided workspaces;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it and
 *     graph-capturable (no synchronisation, no allocation inside);
 *   - return 0 on success, negative on error; mval_last_error() gives the message of the
 *     last failing call on the calling thread;
 *   - tensors are dense row-major with the shapes given in brackets.
 */
#ifndef MVAL_HIP_H
#define MVAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int mval_version(void);
const char* mval_last_error(void);

/* ------------------------------------------------------------------------------------
 * Keypoint decode
 * ---------------------------------------------------------------------------------- */

/* utils/evaluation.py:13-30 get_scaled_pred_corrdinates -- hard arg-max per (frame, view,
 * joint) map, first index on ties (NaN counts as maximum, like torch.argmax), then
 *   x = (idx % split_width) * stride ; y = (idx / split_width) * stride.
 * The reference passes shape[2] (= hh) as split_width for BOTH (its non-square quirk,
 * SURVEY A.2); pass wh for the geometrically correct split.  Invalid joints -> (0, 0).
 *   heatmaps [B,V,J,hh,wh] f32 ; valid [B,J] u8 or NULL ; kp2d [B,V,J,2] i64 (x,y). */
int mval_argmax_decode(const float* heatmaps, const uint8_t* valid, int64_t* kp2d,
                       int B, int V, int J, int hh, int wh, int stride, int split_width, void* stream);
/* The same key-points from the arg-max keys of mval_net_forward_keys (no read of the heat-maps):
 *   keys [B*V][MVAL_ARGMAX_SLOTS][J] u64 ; valid [B,J] u8 or NULL ; kp2d [B,V,J,2] i64 (x,y). */
int mval_argmax_from_keys(const uint64_t* keys, const uint8_t* valid, int64_t* kp2d, int B, int V, int J,
                          int stride, int split_width, void* stream);

/* utils/triangulation.py:191-200 (kornia.spatial_soft_argmax2d(hm, normalized_coordinates=False)
 * * stride): softmax over hh*wh, expectation of the pixel grid; kp2d [n_maps,2] f32 (x,y)*scale. */
int mval_soft_argmax(const float* heatmaps, float* kp2d, int64_t n_maps, int hh, int wh, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Triangulation
 * ---------------------------------------------------------------------------------- */

/* utils/triangulation.py:209-233 + :260-338 _triangulate_ransac + :341-368 _triangulate_dlt +
 * :371-384 _calc_reprojection_error_matrix, batched over (frame, joint), float64, SVD-free
 * (Givens QR of the 2n x 4 DLT system, then one-sided Jacobi on the 4x4 R factor).
 * Deterministic pair order (V <= 11), first strictly larger inlier set wins, the sampled
 * pair is always an inlier, final solve on the sorted inlier views.
 *   kp2d [B,V,J,2] i64 (kp_is_f32 = 0) or f32 (kp_is_f32 = 1) ; proj [B,V,3,4] f64 ;
 *   valid [B,J] u8 or NULL ;
 *   kp3d [B,J,3] f64 (0 for invalid joints) ; joint_err [B,J] f64 mean inlier reprojection
 *   error ; joint_inliers [B,J] i32 ; metric [B] f64 = numpy-order mean of joint_err over
 *   valid joints (NaN if none) ; inlier_count [B] i32 = min over valid joints (-1 if none:
 *   the reference raises ValueError there, the host wrapper does the same). */
int mval_triangulate_ransac(const void* kp2d, int kp_is_f32, const double* proj, const uint8_t* valid,
                            double* kp3d, double* joint_err, int32_t* joint_inliers, double* metric,
                            int32_t* inlier_count, int B, int V, int J, double eps, void* stream);

/* utils/triangulation.py:236-257 _compute_xe: sum over (view, joint) of
 * mean_px (hm - exp(-|grid - kp|^2 / (2 sigma^2)))^2 with kp the reprojection of kp3d
 * (input-pixel units on the heat-map grid, as the reference does).  out [B] f64 ;
 * ws [B*V*J] f64 scratch. */
int mval_reprojection_xe(const double* kp3d, const double* proj, const float* heatmaps, double* out, double* ws,
                         int B, int V, int J, int hh, int wh, double sigma, void* stream);

/* ------------------------------------------------------------------------------------
 * Uncertainty scorers (strategy.py:1149-1215)
 * ---------------------------------------------------------------------------------- */
enum { MVAL_SCORE_HP = 0, MVAL_SCORE_MPE = 1, MVAL_SCORE_BSB = 2 };
enum { MVAL_REDUCE_AVG_F64 = 0, MVAL_REDUCE_AVG_F32 = 1, MVAL_REDUCE_STD_F64 = 2, MVAL_REDUCE_STD_F32 = 3 };

/* Per-map statistic, one heat-map read:
 *   HP  (strategy.py:1185-1187)  1 - max(row_softmax(map))           (row-wise softmax: SURVEY A.9)
 *   MPE (strategy.py:1168-1175)  entropy of softmax over the local peaks
 *        (skimage.feature.peak_local_max(map, min_distance=2): 5x5 maxima strictly above
 *        map.min(), 2-px border excluded, sorted by descending value -- equal values in row-major
 *        order, where the library's order is numpy's unstable argsort --, plateau spacing; equal to
 *        scikit-image 0.18.3 on every map without tied candidates: tests/golden/peaks_skimage.npz)
 *   BSB (strategy.py:1202-1208)  |p0 - p1| of the two highest local peaks of row_softmax(map)
 *   heatmaps [n_maps,hh,wh] f32 ; stat [n_maps] f32 ; n_peaks [n_maps] i32 (MPE/BSB: peaks after the spacing pass, any number
 *   of candidates -- maps with more than 2048 are redone by a second pass; 0 for HP).  BSB with fewer than two peaks: NaN
 *   (the reference raises IndexError). */
int mval_score_maps(int kind, const float* heatmaps, float* stat, int32_t* n_peaks,
                    int64_t n_maps, int hh, int wh, void* stream);

/* The same statistic AND the hard arg-max key-point of every map (utils/evaluation.py:13-30, as mval_argmax_decode)
 * from ONE staged read of each heat-map: what a scoring pass of strategy.py:1027-1090 needs per map.
 *   heatmaps [B,V,J,hh,wh] f32 ; valid [B,J] u8 or NULL (invalid joints -> key-point (0, 0); their statistic is
 *   still written, mval_score_reduce skips it) ; stat, n_peaks [B*V*J] ; kp2d [B,V,J,2] i64 (x, y). */
int mval_score_decode_maps(int kind, const float* heatmaps, const uint8_t* valid, float* stat, int32_t* n_peaks,
                           int64_t* kp2d, int B, int V, int J, int hh, int wh, int stride, int split_width, void* stream);

/* AVG / STD over the valid (view, joint) maps of each frame in the reference's python /
 * numpy evaluation order and precision (strategy.py:1151-1155,1188-1193,1210-1215;
 * SURVEY A.8).  per_map [B,V,J] f32 ; valid [B,J] u8 or NULL ; out [B] f64. */
int mval_score_reduce(const float* per_map, const uint8_t* valid, double* out, int B, int V, int J,
                      int mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Loss / metric
 * ---------------------------------------------------------------------------------- */

/* pose_estimators/loss.py:14-20: out = sum(valid ? (h-g)^2 : 0) / denom.
 *   h,g [lead,hw] f32 ; valid [lead] u8 or NULL ; out [1] f32 ; ws >= 2048 doubles. */
int mval_masked_mse_fwd(const float* h, const float* g, const uint8_t* valid, float* out, double* ws,
                        int64_t lead, int64_t hw, double denom, void* stream);
/* d loss / d h = grad_out * 2 (h-g) valid / denom.  grad_out [1] f32 ; grad_h [lead,hw] f32. */
int mval_masked_mse_bwd(const float* h, const float* g, const uint8_t* valid, const float* grad_out,
                        float* grad_h, int64_t lead, int64_t hw, double denom, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-view input pipeline (dataset/dataset.py:158-220 prepare_single_view, pixel work only)
 * ------------------------------------------------------------------------------------------- */
typedef struct mval_view_desc {
  const uint8_t* img;               /* decoded RGB image [h0][w0][3] (device) */
  int32_t h0, w0;
  int32_t left, top, right, bottom; /* square / scaled box (utils/triangulation.py:96-134); may leave the image */
  int64_t tmp_off;                  /* byte offset of this view's [bottom-top][in_w][3] slab in the workspace's temp part (slabs back to back:
                                     * multiples of in_w * 3; the vertical pass reads 4 bytes at a time where the slab is 4-byte aligned) */
} mval_view_desc;
/* ws >= mval_prepare_views_workspace_bytes(n_views, sum of the views' crop heights, in_w, in_h). */
size_t mval_prepare_views_workspace_bytes(int n_views, int64_t total_crop_rows, int in_w, int in_h);
/* BGR flip + zero-filled crop + PIL LANCZOS resize (Pillow's 8-bit fixed-point algorithm, bit-exact) +
 * ImageNet normalisation: out [n_views][3][in_h][in_w] f32.  views: DEVICE array. */
int mval_prepare_views(const mval_view_desc* views, int n_views, int max_crop_h, int max_crop_w, int in_w, int in_h,
                       float* out, void* ws, void* stream);
/* Gaussian ground-truth heat-maps (dataset.py:198-207): pt [n][2] f64 (x, y in heat-map pixels) ->
 * out [n][h][w] f32 = (float) exp(-((x - px)^2 + (y - py)^2) / (2 sigma^2)) evaluated in float64. */
int mval_gt_heatmaps(const double* pt, int64_t n, double sigma, int h, int w, float* out, void* stream);

/* utils/evaluation.py:198-208 compute_mkpe over S samples: pred [S,J,3] f32, gt [S,gt_rows,J]
 * f32 (rows 0..2 used), valid [S,J] f32 ; out [1] f32 = mean_j (sum_s d_sj / sum_s valid_sj) ;
 * per_sample [S] f32 = the same metric evaluated on each sample alone (strategy.py:1134). */
int mval_mkpe(const float* pred, const float* gt, const float* valid, float* out, float* per_sample,
              int64_t S, int J, int gt_rows, void* stream);
/* 3-D PCK / PCKh counters (utils/evaluation.py:150-195 compute_3d_pckh / compute_3d_pck, called from
 * strategy.py:638-649).  pred [S,J,3], gt [S,gt_rows,J] (rows 0..2 = x,y,z), valid [S,J] (mode 0) f32;
 * thresholds [T] f64 (device).  mode 0 = PCK in mm over valid joints, mode 1 = PCKh (threshold x the
 * distance between gt joints 0 and 1, every joint).  hits [T,J] and counts [J] are int64; the fraction
 * hits/counts is formed by the caller.  float32 distances with torch's operation order, so the counts
 * equal the reference's. */
int mval_pck3d(const float* pred, const float* gt, const float* valid, const double* thresholds, int T, int mode,
               long long* hits, long long* counts, int64_t S, int J, int gt_rows, void* stream);

/* ------------------------------------------------------------------------------------
 * Core-set selection (utils/coreset.py:35-95)
 * ---------------------------------------------------------------------------------- */

/* utils/coreset.py:35-47: pose [n,J,rows] f64 ([joint][coord], rows >= 3) ->
 * feat [n,3J] f64 = [x_0-x_r .. , y_0-y_r .. , z_0-z_r ..]. */
int mval_coreset_features(const double* pose, double* feat, int64_t n, int J, int rows, int root, void* stream);

/* strategy.py:981-989 (cluster-balanced pseudo-labelling): label [n] i32 = index of the nearest of K
 * cluster centres [K,D] f64 for every feature row feat [n,D] f64 (first minimum, as KMeans.predict). */
int mval_nearest_center(const double* feat, const double* centers, int64_t n, int D, int K, int* label, void* stream);

size_t mval_kcenter_workspace_bytes(int64_t n_obs, int D);

/* utils/coreset.py:49-95 greedy k-center on feat [n_obs,D] f64 with sklearn's expanded
 * Euclidean form  sqrt(max(0, (-2 x.c + |x|^2) + |c|^2)).
 *   labeled [n_labeled] i64 row indices (may be NULL/0) ;
 *   have_min_dist != 0: min_dist [n_obs] already holds the running minimum (continuation) ;
 *   row_norms [n_obs] f64 scratch ; picks [n_select] i64 ; ws >= mval_kcenter_workspace_bytes.
 * First maximum wins ties; nothing is masked (duplicates possible, as in the reference). */
int mval_kcenter_select(const double* feat, int64_t n_obs, int D, const int64_t* labeled, int64_t n_labeled,
                        int n_select, int have_min_dist, double* row_norms, double* min_dist,
                        int64_t* picks, void* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * Heat-map network forward (pose_estimators/hrnet.py:468-501, pose_resnet.py:139-153)
 * ---------------------------------------------------------------------------------- */

/* One fused operator:  out = act(((bn(conv(in)) + res1) + res2))  [nearest-upsampled by 2^up].
 * Activations are NHWC f32 (NCHW for the network input / heat-map output).  Offsets are in
 * floats from the workspace base; -1 = absent.  Weights are in the packed fragment order
 * written by mval_pack_conv_weights; scale/shift are the folded eval-mode BatchNorm
 * (y = x * scale + shift, torch's batch_norm inference formula) or (1, bias). */
enum { MVAL_OP_CONV = 0, MVAL_OP_MAXPOOL = 1, MVAL_OP_DECONV = 2, MVAL_OP_BLOCK = 3, MVAL_OP_TO_P2 = 4, MVAL_OP_BNECK = 5, MVAL_OP_STEM_P2 = 6, MVAL_OP_FUSE_UP = 7 };
enum { MVAL_ALGO_DIRECT = 0, MVAL_ALGO_MFMA = 1, MVAL_ALGO_MFMA_BF3 = 2, MVAL_ALGO_MFMA_H2 = 3, MVAL_ALGO_MFMA_P2 = 4 };
enum { MVAL_PACK_HWIO = 0, MVAL_PACK_MFMA16 = 1, MVAL_PACK_MFMA16_BF3 = 2, MVAL_PACK_MFMA16_H2 = 3 };

typedef struct mval_op {
  int32_t kind;      /* MVAL_OP_*: conv, maxpool (k, stride, pad), transposed conv k4 s2 p1 */
  int32_t algo;      /* MVAL_ALGO_*: which kernel family (and weight packing) the op uses */
  int32_t k, stride, pad;
  int32_t cin, cout;
  int32_t hin, win, hout, wout; /* hout/wout BEFORE the fused upsample */
  int32_t up;        /* log2 upsample factor applied on store (res1/res2/out are at hout<<up) */
  int32_t relu;
  int32_t in_nchw, out_nchw;
  int64_t in_off, out_off, res1_off, res2_off;
  int64_t w_off, scale_off, shift_off; /* floats from the params base */
  int32_t phase, lane; /* scheduling hints: ops of one phase on different lanes are independent
                          (HRNet branches / fuse outputs) and run on separate HIP streams; all
                          lanes join at a phase change.  0/0 = plain in-order execution. */
  /* Activation scales of the fp16-split kernels (MVAL_ALGO_MFMA_H2).  Float offsets into the workspace of
   * n_images rows of MVAL_AMAX_ROW dwords, one row per image of an activation tensor (per image, so that a frame's
   * heat-maps do not depend on the rest of the batch): row[0] = number of partial maxima that follow, row[1..] =
   * bits of max |x| over the part of the image each producing wave wrote (plain stores, no atomics, nothing to
   * zero between forwards); the maximum of the partials is the image's max |x|.  0 = none.  out_amax_off: the op
   * writes the rows of its output (any algo); in_amax_off: the rows of the op's input, required by
   * MVAL_ALGO_MFMA_H2 (single-op callers fill them with mval_amax). */
  int64_t in_amax_off, out_amax_off;
  /* MVAL_OP_BLOCK (hrnet.py:19-52, a whole BasicBlock in one launch, csrc/conv_block.hip):
   *   out = relu(bn2(conv3x3(relu(bn1(conv3x3(in))))) + in),  cin = cout in {32, 48, 64}, k 3, stride 1, pad 1,
   * algo MVAL_ALGO_MFMA_H2.  w_off / scale_off / shift_off are conv1 + bn1, these three conv2 + bn2 (both
   * weights packed MVAL_PACK_MFMA16_H2); res1_off must equal in_off (or be -1). */
  int64_t w2_off, scale2_off, shift2_off;
  /* MVAL_ALGO_MFMA_P2 (csrc/conv_p2.h): the op reads AND writes "P2" activations -- each fp32 value kept as the pair
   * of fp16 planes the fp16-split MFMA consumes, halves [n][plane h,l][C/8][H][W][8] at in_off / res1_off / res2_off /
   * out_off (4 bytes per element, so the float offsets and sizes are those of the fp32 tensor), the per-image rows at
   * in_amax_off / res1_amax_off / res2_amax_off / out_amax_off carrying 2^-s in their last dword.  out_nchw != 0:
   * the output is fp32 NCHW instead (the heat-map layer).  bound_off: params offset of two floats
   * [A = max_c |scale_c| * sum |w_c|, B = max_c |shift_c|], the output's magnitude bound (bound2_off: conv2 of a
   * MVAL_OP_BLOCK).  Weights are packed MVAL_PACK_MFMA16_H2. */
  int64_t bound_off, bound2_off, res1_amax_off, res2_amax_off;
  /* MVAL_OP_TO_P2: format change at the head of a P2 plan -- the fp32 NHWC tensor at in_off ([n][hin][win][cin], its
   * rows [count, partials ...] at in_amax_off, as every NHWC producer keeps them) becomes P2 planes at out_off with
   * their rows at out_amax_off. */
  /* MVAL_OP_BNECK (hrnet.py:75-95, a whole Bottleneck with 64 planes in one launch, csrc/conv_bneck_p2.hip; algo
   * MVAL_ALGO_MFMA_P2 only):
   *   out = relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(in)))))))) + res1),  cin in {64, 256}, cout = 256, stride 1.
   * w_off / scale_off / shift_off / bound_off are conv1 + bn1 (cin -> 64), the *2 fields conv2 + bn2 (64 -> 64, 3x3),
   * these four conv3 + bn3 (64 -> 256); res1_off / res1_amax_off: the 256-channel residual (the block's input or its
   * downsample branch), required. */
  int64_t w3_off, scale3_off, shift3_off, bound3_off;
  /* MVAL_OP_STEM_P2 (hrnet.py:303-310, the two stride-2 stem convs in one launch, csrc/conv_stem_p2.hip; algo
   * MVAL_ALGO_MFMA_P2 only):  out = relu(bn2(conv3x3 s2(relu(bn1(conv3x3 s2(image))))))  from the fp32 NCHW network input
   * (in_off = -1, cin = 3, hin x win multiples of 4) to 64 channels of P2 planes at out_off (hout = hin / 4) with rows at
   * out_amax_off.  w_off: conv1's weights packed MVAL_PACK_HWIO, scale_off / shift_off / bound_off: bn1 and its bound;
   * the *2 fields: conv2 (packed MVAL_PACK_MFMA16_H2) + bn2.  in_amax_off: n_images P2 rows the launch fills with the
   * images' max |x| (a small pass over the input first). */
  /* MVAL_OP_FUSE_UP (hrnet.py:424-447, the up-sampling terms of one fuse-layer output in one launch,
   * csrc/conv_fuse_up_p2.hip; algo MVAL_ALGO_MFMA_P2 only):
   *   out = act(((res1 + up(bn(conv1x1(t_0)))) + up(bn(conv1x1(t_1)))) [+ up(bn(conv1x1(t_2)))])   (left to right, fp32)
   * cout in {32, 64} channels at hout x wout (= hin x win here); res1_off / res1_amax_off: the partial sum so far (required);
   * n_terms in {2, 3}; term j: P2 input of t_cin[j] channels at (hout >> t_up[j]) x (wout >> t_up[j]) at t_in_off[j] with rows at
   * t_in_amax_off[j], 1x1 weights packed MVAL_PACK_MFMA16_H2 at t_w_off[j], folded BN at t_scale_off[j] / t_shift_off[j], bound
   * [A, B] at t_bound_off[j]; relu: the activation of the sum. */
  int32_t n_terms, t_cin[3], t_up[3];
  int32_t reserved0; /* (round 4's multi-conv launch marker; must be 0) */
  int64_t t_in_off[3], t_in_amax_off[3], t_w_off[3], t_scale_off[3], t_shift_off[3], t_bound_off[3];
} mval_op;

/* Weight packing.  MVAL_PACK_HWIO: [k*k][cin][cout] (direct kernels, deconv);
 * MVAL_PACK_MFMA16: v_mfma_f32_16x16x4_f32 B-fragment order
 * [k*k][cin/16][cout/16][lane 0..63][4] with lane = (cin_quad << 4) | cout_lane, cin and
 * cout zero-padded to multiples of 16;
 * MVAL_PACK_MFMA16_BF3: the exact three-way bf16 split of every weight in
 * v_mfma_f32_16x16x32_bf16 B-fragment order [k*k][cin/32][cout/16][plane h,m,l][lane][8 bf16]
 * (conv_mfma_split.hip);
 * MVAL_PACK_MFMA16_H2: the two-way fp16 split of every weight times a power of two chosen so that max |w| lands in
 * [2^13, 2^14), same fragment order with two planes [k*k][cin/32][cout/16][plane h,l][lane][8 fp16], followed by
 * a 4-float trailer whose first float is the inverse of that power of two (the conv epilogue multiplies by it).
 * `transposed`: 0 = Conv2d weight [cout,cin,k,k]; 1 = ConvTranspose2d
 * weight [cin,cout,k,k] as the direct kernel reads it; 2 = a [cin,cout,k,k] tensor with the taps
 * flipped: the data-gradient form of a Conv2d weight (pass cout' = cin, cin' = cout) and the
 * forward form of a ConvTranspose2d weight on the MFMA kernels, which run MVAL_OP_DECONV as a
 * stride-1 conv over the zero-dilated input (pass cout, cin as stored).  A ConvTranspose2d's data
 * gradient is a plain conv with its weight read as [cout' = cin][cin' = cout]: transposed = 0. */
size_t mval_packed_weight_floats(int pack, int cout, int cin, int k);
int mval_pack_conv_weights(int pack, int transposed, const float* w, float* packed, int cout, int cin, int k,
                           void* stream);
/* Many MVAL_PACK_MFMA16_BF3 packings in ONE launch (a training plan re-packs every conv weight after each optimizer
 * step, strategy.py:478-484).  jobs_dev: n_jobs descriptors IN DEVICE MEMORY (w / packed are device pointers, mode =
 * the `transposed` argument above); first_block_dev[j] = 256-thread blocks of the jobs before j, a job has
 * (k * k * ceil(cin / 32) * ceil(cout / 16) * 512 + 255) / 256 blocks; total_blocks = their sum. */
typedef struct mval_pack_job {
  const float* w;
  void* packed;
  int mode, cout, cin, k;
} mval_pack_job;
int mval_pack_bf3_jobs(const mval_pack_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, void* stream);
/* The same for a mix of both splits: a job with bit 8 of `mode` set (mode | 0x100) is packed MVAL_PACK_MFMA16_H2
 * (its max |w| is taken first, in the same call: three launches in all), the others MVAL_PACK_MFMA16_BF3; a job's block
 * count is the same for both. */
int mval_pack_split_jobs(const mval_pack_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, void* stream);
/* scale = gamma / sqrt(var + eps) ; shift = beta - mean * scale  (all [c] f32). */
int mval_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                 float* scale, float* shift, int c, void* stream);

/* Writes the max-magnitude rows (see mval_op; [n_images][MVAL_AMAX_ROW] dwords) of a tensor of n_images images of
 * per_image consecutive floats each: for tensors that did not come out of an op of the plan. */
#define MVAL_AMAX_ROW 4096
#define MVAL_P2_ROW 512 /* dwords per (tensor, image) row of a P2 activation: 256 partial-maximum slots ... 2^-s in the last one */
int mval_amax(const float* x, int64_t per_image, int n_images, uint32_t* rows, void* stream);
/* fp32 NHWC [n][H][W][C] (C % 8 == 0) whose rows_in ([count, partials ...], as every NHWC producer keeps them) hold
 * its per-image max |x| -> P2 planes (csrc/conv_p2.h) and their rows ([P2 partial slots ... 2^-s]; zero-initialised
 * by the caller once).  And back (tests, network boundaries). */
int mval_nhwc_to_p2(const float* x, const uint32_t* rows_in, void* planes, uint32_t* rows, int n_images, int H, int W, int C,
                    void* stream);
int mval_p2_to_nhwc(const void* planes, const uint32_t* rows, float* out, int n_images, int H, int W, int C, void* stream);

/* 1 when the MFMA kernel family has a configuration for this op geometry (the plan builder
 * asks before choosing MVAL_ALGO_MFMA / MVAL_PACK_MFMA16), else 0. */
int mval_op_mfma_supported(const mval_op* op, int n_images);
/* Same question for a given MVAL_ALGO_* (MFMA, MFMA_BF3 or MFMA_H2). */
int mval_op_algo_supported(const mval_op* op, int n_images, int algo);

int mval_op_launch(const mval_op* op, int n_images, float* workspace, const float* params,
                   const float* net_input, float* net_output, void* stream);

void* mval_net_create(const mval_op* ops, int n_ops);
void mval_net_destroy(void* net);
/* Branch concurrency of mval_net_forward: -1 = decided by the MVAL_STREAMS environment variable (default:
 * on), 0 = every op on the caller's stream, 1 = fork / join over private streams.  Both forms can be
 * captured into a hipGraph (the fork / join uses events recorded on the capturing stream).  The side streams and
 * events are per device and shared by all nets: ONE forward (or training pass) in flight per device at a time. */
int mval_net_set_multi_stream(void* net, int mode);
int mval_net_forward(void* net, int n_images, float* workspace, const float* params,
                     const float* input_nchw, float* output_nchw, void* stream);
/* Decode from the heat-map layer's epilogue (hrnet.py:344-350,500 / pose_resnet.py final_layer -> utils/evaluation.py:13-30):
 * the same forward, and the kernel that stores the NCHW heat-maps also keeps 64-bit arg-max keys of what it stores in
 * argmax_keys [n_images][MVAL_ARGMAX_SLOTS][joints] (device; zeroed here first): every wave leaves the best key of its
 * part of a map in its own slot of the map (plain stores; more partials per map than slots -- maps above ~8 000
 * pixels -- fold with atomicMax).  key: high word = the stored value as an order-preserving unsigned (NaN highest,
 * -0 == +0), low word = ~flat index, so the largest key is torch.argmax's choice, first index on ties.
 * mval_argmax_from_keys reduces them and yields the key-points mval_argmax_decode would compute from a second read of
 * the maps (bit-equal).  mval_net_keeps_argmax_keys: 1 when the plan's heat-map layer runs on a kernel that does this
 * (the MFMA families; not the generic direct kernels), else mval_net_forward_keys fails. */
#define MVAL_ARGMAX_SLOTS 128
int mval_net_keeps_argmax_keys(void* net);
int mval_net_forward_keys(void* net, int n_images, float* workspace, const float* params,
                          const float* input_nchw, float* output_nchw, uint64_t* argmax_keys, void* stream);
/* Same, with a hipEvent recorded on `stream` around every op; blocks until the stream has
 * drained and writes the elapsed milliseconds of each op to ms_per_op [n_ops] (HOST pointer).
 * Measurement only (not graph-capturable). */
int mval_net_forward_timed(void* net, int n_images, float* workspace, const float* params,
                           const float* input_nchw, float* output_nchw, void* stream, float* ms_per_op);
/* Algorithmic FLOPs (2*MAC) of one op for n_images. */
double mval_op_flops(const mval_op* op, int n_images);

/* ------------------------------------------------------------------------------------
 * Training step of the heat-map network (strategy.py:460-487: forward in train mode,
 * ``batch_loss.backward()``); the optimizer stays in PyTorch (Adam, strategy.py:405).
 * ---------------------------------------------------------------------------------- */

/* Train-mode BatchNorm statistics of z [M, C] (NHWC rows): mean, invstd = 1/sqrt(biased var +
 * eps); running stats updated in place with `momentum` and the unbiased variance (torch).
 * ws >= 512 * C * 2 doubles. */
int mval_bn_batch_stats(const float* z, int64_t M, int C, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, double* ws, void* stream);
/* The same statistics from per-(channel, conv workgroup) partials part[C][tiles][2] (float64 sum, sum of squares) that the
 * training forward's conv epilogues keep while they store z (mval_train_forward): no second pass over z. */
int mval_bn_finalize_stats(const double* part, int tiles, int64_t M, int C, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, void* stream);
/* out = act(((z*alpha + (beta - mean*alpha)) nearest-upsampled 2^up) + res1 + res2), alpha = invstd*gamma. */
int mval_bn_apply_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* res1, const float* res2, float* out, int N, int H, int W, int C, int up, int relu,
                      void* stream);
/* The same, and the max |out| of the tensor left in amax_row ([count, partial maxima ...], >= 513 dwords; NULL = as
 * above): the activation scale of the fp16-split convs that read `out` in training (ONE row per tensor there). */
int mval_bn_apply_fwd_amax(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* res1, const float* res2, float* out, int N, int H, int W, int C, int up, int relu,
                           uint32_t* amax_row, void* stream);
/* The same, and (relu_mask != NULL) the bits (out > 0) of every float4 of `out` in one byte each: relu_mask[i / 4] for the float4
 * that starts at element i (N*Ho*Wo*C/4 bytes) -- what mval_bn_bwd_fused_mask needs of `out`. */
int mval_bn_apply_fwd_mask(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* res1, const float* res2, float* out, int N, int H, int W, int C, int up, int relu,
                           uint32_t* amax_row, uint8_t* relu_mask, void* stream);
/* The same, and the output ALSO as P2 planes (csrc/conv_p2.h: [n][plane][C/8][Ho][Wo][8] fp16 pairs, rows p2_rows[n][MVAL_P2_ROW]) for
 * the P2 convs of the training forward; the tensor's scale comes from the a-priori bound |bn(z)| <= |gamma| sqrt(M - 1) + |beta|
 * (Samuelson) plus the residuals' exact maxima (res*_row: their magnitude rows, required with a residual).  out may be NULL. */
int mval_bn_apply_fwd_p2(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         const float* res1, const float* res2, float* out, void* p2_planes, uint32_t* p2_rows, int N, int H, int W,
                         int C, int up, int relu, uint32_t* amax_row, uint8_t* relu_mask, const uint32_t* res1_row,
                         const uint32_t* res2_row, void* stream);
/* The same with residuals that exist as P2 planes only (same shape as the output; res*_p2_rows: their rows, 2^-s in the scale slot):
 * res1 / res1_p2 are alternatives (both NULL: no residual); the magnitude rows res*_row stay required for the output's scale. */
int mval_bn_apply_fwd_p2_res(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const float* res1, const float* res2, float* out, void* p2_planes, uint32_t* p2_rows, int N, int H, int W,
                             int C, int up, int relu, uint32_t* amax_row, uint8_t* relu_mask, const uint32_t* res1_row,
                             const uint32_t* res2_row, const void* res1_p2, const uint32_t* res1_p2_rows, const void* res2_p2,
                             const uint32_t* res2_p2_rows, void* stream);
/* Backward of the above: masks gout by (out > 0) when relu, adds it into gres1/gres2 (stores it
 * instead where `overwrite` bit 0 / bit 1 is set: the first writer of a gradient slot), window-sums
 * it to the conv resolution, then (has_bn) dgamma/dbeta and dz = gamma*invstd*(g - dbeta/M -
 * xhat*dgamma/M) written to gz; without BN gz is the masked/window-summed gradient and dbeta its
 * per-channel sum (bias gradient).  ws >= 512*C*2 doubles, sums >= 2*C floats. */
int mval_bn_bwd(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                const float* gamma, float* gres1, float* gres2, float* gz, float* dgamma, float* dbeta, double* ws,
                float* sums, int N, int H, int W, int C, int up, int relu, int has_bn, int overwrite, void* stream);
/* The same, and max |gz| left in gz_amax_row (has_bn only; the scale of the fp16-split data-gradient conv). */
int mval_bn_bwd_amax(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                     const float* gamma, float* gres1, float* gres2, float* gz, float* dgamma, float* dbeta, double* ws,
                     float* sums, int N, int H, int W, int C, int up, int relu, int has_bn, int overwrite,
                     uint32_t* gz_amax_row, void* stream);
/* Backward of out = act(bn(z) + res1 + res2) without upsample, BatchNorm present, C % 4 == 0: the results of
 * mval_bn_bwd_amax with one tensor write and one to two tensor reads fewer -- the reduction pass does not write the
 * masked gradient (the apply pass re-reads it from the residual slot this op wrote first, or from gout with the mask
 * re-derived), and a ReLU without residuals takes its mask from z (`out` may be NULL then; beta is needed for it). */
int mval_bn_bwd_fused(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float* gres1, float* gres2, float* gz, float* dgamma,
                      float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu, int overwrite,
                      uint32_t* gz_amax_row, void* stream);
/* The same with the ReLU mask of an op WITH residuals taken from the bytes mval_bn_apply_fwd_mask kept (relu_mask != NULL: `out`
 * is not read; a sixteenth of its bytes). */
int mval_bn_bwd_fused_mask(const float* gout, const float* out, const uint8_t* relu_mask, const float* z, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float* gres1, float* gres2, float* gz,
                           float* dgamma, float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu, int overwrite,
                           uint32_t* gz_amax_row, void* stream);
/* mval_bn_bwd_fused_mask, and (dz_planes != NULL) dz ALSO as P2 planes [n][plane][C/8][H][W][8] + rows dz_rows[n][MVAL_P2_ROW] for the
 * data-gradient conv on the P2 kernels; gmax_ws >= 512 floats, bound_slot one dword of scratch; gz may be NULL then. */
int mval_bn_bwd_fused_p2(const float* gout, const float* out, const uint8_t* relu_mask, const float* z, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, float* gres1, float* gres2, float* gz,
                         float* dgamma, float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu, int overwrite,
                         uint32_t* gz_amax_row, void* dz_planes, uint32_t* dz_rows, float* gmax_ws, uint32_t* bound_slot, void* stream);
/* Weight gradient dw [cout][cin][k][k] of a conv: x NHWC (NCHW when x_nchw), dz NHWC.
 * ws >= mval_conv_wgrad_workspace_floats(cin, cout, k) floats. */
size_t mval_conv_wgrad_workspace_floats(int cin, int cout, int k);
int mval_conv_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin,
                    int Hout, int Wout, int Cout, int k, int stride, int pad, int x_nchw, void* stream);
/* The same; with both magnitude rows ([count, partials], one row per tensor) the split kernels use the fp16x2 form. */
int mval_conv_wgrad_scaled(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int k, int stride, int pad, int x_nchw,
                           const uint32_t* x_amax_row, const uint32_t* dz_amax_row, void* stream);
int mval_slab_reduce(const float* slabs, int S, int64_t n, float* out, int accumulate, void* stream);
/* Backward of MaxPool2d(k, stride, pad) (pose_resnet.py:35 under autograd): gin (+)= gout routed to the
 * first maximum of each window in ATen's scan order (NaN wins); x / gin NHWC [N,Hin,Win,C], gout
 * [N,Hout,Wout,C], C % 4 == 0; accumulate = 0 stores. */
int mval_maxpool_bwd(const float* gout, const float* x, float* gin, int N, int Hin, int Win, int C, int Hout,
                     int Wout, int k, int stride, int pad, int accumulate, void* stream);
/* Data gradient of a conv (geometry given in FORWARD terms: x [N,hin,win,cin] -> z
 * [N,hout,wout,cout], k/stride/pad): dx (+)= conv(dz zero-dilated by the stride, flipped W^T).
 * w_packed: mval_pack_conv_weights(pack, transposed = 2, w, ..., cout' = cin, cin' = cout, k);
 * ones / zeros: >= cin floats of 1.0 / 0.0; algo = MVAL_ALGO_MFMA needs cout % 16 == 0,
 * MVAL_ALGO_MFMA_BF3 cout % 32 == 0 (or 48) and k = 3, or k = 1 with stride 1. */
int mval_conv_dgrad(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                    int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int k,
                    int stride, int pad, int algo, void* stream);
/* Data gradient of Conv2d(k3, s2, p1) with even hin / win as four 2x2 stride-1 convs over dz, one per parity of dx, in ONE
 * launch (16 tap-pixels per dz pixel instead of the zero-dilated form's 36; the fp16 split applies).  algo: MVAL_ALGO_MFMA_BF3
 * or _H2 (then dz_amax_row); w_packed: mval_pack_conv_weights(pack, transposed = 4, w, ..., cout' = cin, cin' = cout, k = 4)
 * from the conv's own [cout][cin][3][3] weight.  _supported: 1 when a kernel exists for the shape. */
/* 1 when the split weight-gradient kernel that can read x as P2 planes covers the conv (3x3 stride 1 / 2, wide 1x1; cin % 8 == 0). */
int mval_conv_wgrad_p2_covers(int cin, int cout, int k, int stride);
/* 1 when the split weight-gradient kernel covers the conv at all (it can then read dz as P2 planes when cout % 8 == 0). */
int mval_conv_wgrad_split_covers(int cin, int cout, int k, int stride);
/* 1 when the training forward's P2 conv can apply its producer's BatchNorm + ReLU while staging (3x3 stride 1; mval_train_op.zin_rel). */
int mval_conv_p2_inz_supported(int cin, int cout, int h, int w, int n);
/* 1 when a 3x3 stride-1 data gradient (cin = the conv's cout, cout = its cin, h x w = its input map) can keep those sums. */
int mval_conv_p2_bsum_supported(int cin, int cout, int h, int w, int n);
int mval_conv_dgrad_parity_supported(int N, int hin, int win, int cin, int hout, int wout, int cout, int algo);
int mval_conv_dgrad_parity(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                           int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int algo,
                           const uint32_t* dz_amax_row, void* stream);
/* The same with algo = MVAL_ALGO_MFMA_H2 allowed (stride 1): dz_amax_row = the magnitude row of dz (one row for the
 * tensor), w_packed in MVAL_PACK_MFMA16_H2 form (with its trailer). */
int mval_conv_dgrad_scaled(const float* dz, const float* w_packed, const float* ones, const float* zeros, float* dx,
                           int accumulate, int N, int hin, int win, int cin, int hout, int wout, int cout, int k,
                           int stride, int pad, int algo, const uint32_t* dz_amax_row, void* stream);

/* One operator of the training graph: the forward geometry / arena offsets (`op`, as in
 * inference, weights packed for the forward kernel at op.w_off; op.shift_off = bias for a conv
 * without BatchNorm) plus what backward needs.  Offsets are floats into `arena` (activations,
 * no reuse), `garena` (activation gradients, same layout) and `params`.  garena is NOT zero-filled:
 * `first_touch` marks the gradient slots this op writes FIRST in backward order (bit 0 data
 * gradient, bit 1 res1, bit 2 res2): those are stored, later writers accumulate; the caller
 * zero-fills only slots that no op writes and places the loss gradient in the last op's gout.
 * op.kind may be MVAL_OP_CONV, MVAL_OP_MAXPOOL (no parameters) or MVAL_OP_DECONV (k4 s2 p1 with
 * BatchNorm, MVAL_ALGO_MFMA; wd_off = its weight packed as a Conv2d weight, transposed = 0). */
typedef struct mval_train_op {
  mval_op op;
  int64_t z_off;      /* raw conv output at conv resolution (arena); unused when has_bn == 0 */
  int64_t gin_off, gout_off, gres1_off, gres2_off; /* garena; -1 = no gradient needed */
  int64_t wd_off;     /* params: weights packed for the data-gradient conv (-1: none) */
  int32_t has_bn, dgrad_algo, first_touch;
  int32_t dgrad_form; /* 0: plain (stride 2: dz read zero-dilated); 1: Conv2d(k3, s2, p1) on even sizes as four 2x2 parity convs
                       * (mval_conv_dgrad_parity; wd_off packed with transposed = 4, k = 4) */
  float* gamma; float* beta; float* running_mean; float* running_var; /* device pointers */
  float* mean; float* invstd;            /* saved batch statistics [cout] */
  float* dweight; float* dgamma; float* dbeta; /* gradient outputs (dbeta = bias grad w/o BN) */
  /* fp16-split kernels in training (op.algo / dgrad_algo == MVAL_ALGO_MFMA_H2): magnitude rows ([count, partials],
   * ONE row per tensor, >= 513 dwords each, float offsets into `arena`; 0 = none).  op.in_amax_off = the row of the
   * op's INPUT activation (written by its producer's out_amax_off); out_amax_off = where this op's BatchNorm apply
   * leaves max |out|; gz_amax_off = the row of the op's dz scratch (written by its BatchNorm backward, read by its
   * data-gradient conv). */
  int64_t out_amax_off, gz_amax_off;
  /* > 0: float offset into `arena` of N*hout*wout*cout/4 BYTES -- the forward apply keeps (out > 0) of every float4 as four
   * bits of one byte there and the backward takes the ReLU mask from it instead of reading `out` (ops with relu, a residual
   * and no upsample; 0 = none). */
  int64_t mask_off;
  /* Round 4: the training forward's convs on the P2 kernels (csrc/conv_p2.hip, raw fp32 NHWC z out + batch-statistics partials).
   * fwd_p2 != 0: this op's conv reads its input as P2 planes at in_p2_off with rows at in_p2_rows_off (float offsets into `arena`;
   * weights packed MVAL_PACK_MFMA16_H2 at op.w_off).  out_p2_off > 0: this op's BatchNorm apply ALSO writes its output as P2 planes
   * there (rows at out_p2_rows_off, n_images * MVAL_P2_ROW dwords, zero-initialised once by the caller); res1_amax_off /
   * res2_amax_off: the magnitude rows ([count, partials]) of its residuals, needed for the P2 scale (0: none / no residual). */
  int32_t fwd_p2;
  int32_t p2_flags; /* bit 0: this op's weight gradient reads its input from the P2 planes (mval_conv_wgrad_p2_covers); bit 1: this op's
                     * apply writes ONLY the P2 planes (every consumer of its output reads those: no fp32 NHWC copy);
                     * bit 6: this op's BatchNorm backward is round 3's pair (masked copy + in-place dz; reads `out`: not with bits 1 - 3);
                     * bit 7: this op's batch statistics come from the separate pass over z, not from its conv's epilogue partials
                     * (bits 6 / 7 are the caller's A/B switches: the library reads no environment variable for them) */
  int64_t in_p2_off, in_p2_rows_off, out_p2_off, out_p2_rows_off, res1_amax_off, res2_amax_off;
  /* p2_flags bit 3: with bit 2 -- the weight gradient reads dz from the planes as well, so the BatchNorm backward writes no fp32 dz.
   * p2_flags bit 2: the op's data gradient runs on the P2 kernels (stride 1): its BatchNorm backward ALSO writes dz as P2 planes into the
   * scratch at gz_p2_off (float offset into `arena`: planes of the largest dz, then n_images * MVAL_P2_ROW row dwords (zeroed once by the
   * caller), then 512 floats + 64 dwords of reduction scratch), and mval_conv_p2 reads them with the data-gradient packing at wd_off. */
  int64_t gz_p2_off, gz_p2_rows_off;
  /* p2_flags bit 4 / bit 5: the forward apply reads res1 / res2 from that activation's P2 planes at res1_p2_off / res2_p2_off (rows at
   * res*_p2_rows_off) -- the residual's producer then writes no fp32 copy when its other readers take the planes too. */
  int64_t res1_p2_off, res1_p2_rows_off, res2_p2_off, res2_p2_rows_off;
  /* Round 6: BatchNorm apply inside the consumer.  z_out != 0: this op's forward apply is NOT run -- its output (ReLU, no residual, no
   * upsample, planes only) has exactly one reader, the 3x3 stride-1 P2 conv `zin_rel` ops away, whose forward staging and whose weight
   * gradient's staging compute relu(BatchNorm(z)) from this op's raw z, batch statistics and affine parameters on the way into LDS
   * (csrc/conv_p2.h P2Args::in_z, conv_wgrad_bf3.hip XZ: the same arithmetic, scale and split as the apply pass -- bit-identical operands).
   * zin_rel != 0 (the reader): ops[i + zin_rel] is that producer (zin_rel < 0: it precedes the reader in the list). */
  int32_t zin_rel, z_out;  /* p2_flags bit 12 (MVAL_TRAIN_BSUM, with zin_rel == -1): this op's data gradient -- the ONLY writer of its
                            * producer's output gradient -- also keeps the BatchNorm backward reduction of what it writes (P2Args::bs_z), and the
                            * producer's backward skips that pass */
} mval_train_op;
#define MVAL_TRAIN_BSUM 4096
/* p2_flags bit 13 on every op of a backward call (the library looks at the call's last op): the weight gradients' slab reductions of the call run
 * as ONE launch per 64 ops when its lanes have joined; wsf is then one arena of n_lanes * wsf_floats_per_lane floats that must hold the sum of
 * mval_conv_wgrad_workspace_floats (rounded up to 64) over the call's ops. */
#define MVAL_TRAIN_WGRAD_DEFER 8192

/* ones_off / zeros_off: params offsets of >= max(cout) floats of 1.0 / 0.0.
 * ws: ws_doubles >= 512*maxC*2 doubles.  With room for cout * (conv workgroups) * 2 doubles of an op (about
 * n_images*hout*wout*cout / 16 for the 32-pixel tiles, less for larger ones) that op's batch statistics come from its
 * conv epilogue; ops whose partials do not fit run the separate statistics pass. */
int mval_train_forward(const mval_train_op* ops, int n_ops, int n_images, float* arena, const float* params,
                       int64_t ones_off, int64_t zeros_off, const float* input_nchw, float* output_nchw,
                       double* ws, int64_t ws_doubles, float momentum, float eps, void* stream);
/* gz: scratch >= max over ops of N*hout*wout*cout floats; wsf: >= max wgrad workspace;
 * sums: >= 2*maxC floats.  The gradient w.r.t. the network output must already be in garena at
 * the last op's gout_off (NHWC). */
/* Measurement only (profiles/r06 item 2b): wsf of the next backward calls holds k regions of region_floats per lane, walked op by op. */
int mval_train_slab_rotation(int k, int64_t region_floats);
int mval_train_backward(const mval_train_op* ops, int n_ops, int n_images, float* arena, float* garena,
                        const float* params, int64_t ones_off, int64_t zeros_off, const float* input_nchw,
                        float* gz, float* wsf, double* ws, float* sums, void* stream);
/* Round 5: the same two passes with the independent lanes of a phase (mval_op.phase / .lane: HRNet branches, hrnet.py:199-287) on
 * separate HIP streams, joined on `stream` at every phase change and before returning.  An op runs on side stream op.lane (1 ..
 * n_lanes - 1 <= 3) in the forward if p2_flags carries MVAL_TRAIN_LANE_FWD, in the backward if it carries MVAL_TRAIN_LANE_BWD -- the
 * caller sets the backward bit only for phases in which every gradient slot (gin / gres1 / gres2) has all its writers on ONE lane, so
 * the first-touch / accumulate order of every slot is the serial one and the results are bit-identical to mval_train_forward /
 * mval_train_backward.  Every scratch buffer holds n_lanes slices (lane l at base + l * its per-lane size); the per-op dz plane
 * scratch (gz_p2_off ...) must be distinct per lane as well.  One pass per device at a time (the side streams are per device). */
#define MVAL_TRAIN_LANE_FWD 256
#define MVAL_TRAIN_LANE_BWD 512
/* With MVAL_TRAIN_LANE_BWD on the ops of a phase whose lanes DO share gradient slots (a fuse layer's chains, a transition), EVERY op
 * of that phase -- lane 0 included -- carries MVAL_TRAIN_LANE_ORD: each kernel that writes a gradient slot then waits for the slot's
 * previous writer of the phase (list order) through an event, so the slot's store / accumulate order stays the one-stream order. */
#define MVAL_TRAIN_LANE_ORD 1024
/* MVAL_TRAIN_LANE_FREE on every op of a backward call: the lanes do not join at phase changes; they fork once and join once per call,
 * and every op also waits for the last writer of the gradient slot it READS (its own output's gradient).  Same results. */
#define MVAL_TRAIN_LANE_FREE 2048
int mval_train_forward_lanes(const mval_train_op* ops, int n_ops, int n_images, float* arena, const float* params,
                             int64_t ones_off, int64_t zeros_off, const float* input_nchw, float* output_nchw,
                             double* ws, int64_t ws_doubles_per_lane, int n_lanes, float momentum, float eps, void* stream);
int mval_train_backward_lanes(const mval_train_op* ops, int n_ops, int n_images, float* arena, float* garena,
                              const float* params, int64_t ones_off, int64_t zeros_off, const float* input_nchw,
                              float* gz, float* wsf, double* ws, float* sums, int n_lanes, int64_t gz_floats_per_lane,
                              int64_t wsf_floats_per_lane, int64_t ws_doubles_per_lane, int64_t sums_floats_per_lane, void* stream);

/* ---- optimizer step (replaces torch.optim.Adam.step, /root/reference/strategy.py:405-407 and :479) -------------------
 * Adam over MANY tensors in one launch, arithmetic = torch/optim/adam.py _single_tensor_adam in float32:
 *   g = grad (+ weight_decay * p);  m += (1 - beta1) (g - m);  v = v beta2 + (1 - beta2) g g;
 *   p -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) evaluated by the caller in double (as
 * python does) and passed as float32 (as torch casts python scalars).  jobs_dev: n_jobs descriptors IN DEVICE MEMORY (device
 * pointers to `count` contiguous float32 each; 16-byte aligned pointers take the vector path); first_block_dev[j] = blocks of
 * the jobs before j, a job has ceil(count / mval_adam_block_elems()) blocks; total_blocks = their sum.  In place. */
typedef struct mval_adam_job {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t count;
} mval_adam_job;
int mval_adam_block_elems(void);
int mval_adam_step(const mval_adam_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, float one_minus_beta1,
                   float beta2, float one_minus_beta2, float eps, float weight_decay, float step_size, float bias_correction2_sqrt,
                   void* stream);

/* Bound-slack probe of the P2 training tensors (reference has no counterpart: a guard of this implementation's number format;
 * /root/reference/strategy.py:460-487 is the step it watches).  mval_p2_plane_stats measures one P2 tensor ([n][plane][C/8][HW][8] fp16 planes,
 * rows of MVAL_P2_ROW dwords per image) into out4 (device, zeroed by the caller): [0] float bits of 2^-s of image 0, [1] float bits of
 * max |h + l| in scaled units (the a-priori bound sits in [2^13, 2^14) there), [2] number of non-zero values below 2^-3 scaled (fewer than
 * 22 significand bits kept), [3] number of non-zero values.  mval_train_p2_probe(dev_out, n_ops): while dev_out != NULL every
 * mval_train_forward / mval_train_backward call measures the P2 planes it writes -- op i's output planes into dev_out[i][0][4], its dz
 * planes into dev_out[i][1][4] (i = position in the forward's op list; backward calls on a sub-range pass their base through
 * mval_train_timing_base) -- NULL / 0 switches it off. */
int mval_p2_plane_stats(const void* planes, const uint32_t* rows, int n_images, int channels, int hw, uint32_t* out4, void* stream);
int mval_train_p2_probe(uint32_t* dev_out, int n_ops);

/* Measurement only (bench.py, training workload): with a non-NULL HOST array of 6 floats every later
 * mval_train_forward / mval_train_backward call brackets its launches with hipEvents and ADDS the elapsed
 * milliseconds per kernel family -- [conv forward, BN statistics, BN apply, BN backward, weight gradient, data
 * gradient] -- synchronising the stream at the end of the call; NULL switches it off again. */
int mval_train_timing(float* ms_per_family);
/* The same plus a per-operator breakdown into a HOST array [n_ops][6]; mval_train_timing_base(first_op) tells a
 * mval_train_backward call on a sub-range of the op list where that range starts. */
int mval_train_timing_ops(float* ms_per_family, float* ms_per_op_family, int n_ops);
int mval_train_timing_base(int first_op);

#ifdef __cplusplus
}
#endif
#endif /* MVAL_HIP_H */
