/* C-ABI of libphnet_hip.so: the MI355X (gfx950) kernels behind PHNet's per-clip hot path.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary"): plain pointers and sizes, no torch types.
 * The only native FFI the reference has on this path is the pybind11 module nms_impl
 * (libs/ops/csrc/nms.cpp:44-61 -> nms_kernel.cu:147-192); phnet_lane_nms replaces it.  Every other entry
 * point replaces an ATen/cuDNN/cuBLAS call the reference's Python makes (file:line given per function) and
 * follows the same conventions:
 *   - all pointers are DEVICE pointers (hipMalloc / torch.cuda memory), fp32 unless stated otherwise;
 *   - outputs and workspaces are CALLER-allocated; nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); kernels are enqueued, not awaited;
 *   - return value: 0 = enqueued, <0 = error (no work enqueued unless stated):
 *       PHNET_ERR_ARG (-1) bad shape / null pointer / unsupported size,
 *       PHNET_ERR_WORKSPACE (-2) workspace too small, PHNET_ERR_LAUNCH (-3) HIP launch error;
 *   - re-entrant and thread-safe (no global state).
 * Activation layout is NHWC; convolution weights are OHWI ([Co][R][S][Ci]) = the memory order of a
 * torch.channels_last OIHW parameter, so reference checkpoints load without a re-layout pass.
 */
#ifndef PHNET_HIP_H
#define PHNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PHNET_OK 0
#define PHNET_ERR_ARG (-1)
#define PHNET_ERR_WORKSPACE (-2)
#define PHNET_ERR_LAUNCH (-3)
#define PHNET_ABI_VERSION 1

int phnet_abi_version(void);

/* ---- lane NMS: replaces nms_impl.nms_forward (libs/ops/csrc/nms.cpp:44-57, nms_kernel.cu:26-192) ----
 * rows   [frames][k_max][5+n_offsets]  (cls0, cls1, start_y, start_x_px, length_strips, x_0.. px)
 * scores [frames][k_max];  counts [frames] int32 valid rows per frame, or NULL (= k_max for every frame)
 * keep [frames][k_max] int64 (first num entries valid, rest 0), num_to_keep [frames] int64,
 * parent [frames][k_max] int64 (1-based group id, 0 = none).  Score ties are broken by lower row index.
 * Limits: k_max <= ~900 (single-workgroup design), n_offsets <= 250. */
int phnet_lane_nms(const float* rows, const float* scores, const int32_t* counts, int64_t frames,
                   int64_t k_max, int32_t n_offsets, float thresh, int64_t top_k,
                   int64_t* keep, int64_t* num_to_keep, int64_t* parent, void* stream);

/* ---- fused eval decode: replaces DetNetV2.get_lanes up to the kept rows (libs/models/Router4OL.py:441-471: softmax +
 * confidence mask + boolean-mask compaction + NMS rows + libs.ops.nms + keep[:num] + gather + length rounding) in one
 * launch without host synchronisation.  lines [frames][N][6+S], N <= 256, top_k <= 64.  keep_mask u8 [frames][N];
 * num i64 [frames]; keep_c i64 [frames][top_k] (indices into the compacted candidate list, NMS order, -1 padded = the
 * reference's `keep`); anchors / anchors_sorted i64 [frames][top_k] (anchor indices, NMS order / ascending);
 * kept_rows [frames][top_k][6+S] (column 5 rounded to strips; zero rows beyond num). */
int phnet_lane_decode(const float* lines, int64_t frames, int32_t N, int32_t n_offsets, float conf_thresh,
                      float nms_thresh, int64_t top_k, float img_w, uint8_t* keep_mask, int64_t* num,
                      int64_t* keep_c, int64_t* anchors, int64_t* anchors_sorted, float* kept_rows, void* stream);

/* ---- lane-anchor ROI pooling: replaces F.grid_sample(..., align_corners=True) + permutes
 * (libs/models/Router4OL.py:132-150, 269-272) and its backward (ATen grid_sampler_2d_backward).
 * fmap [B][h][w][64]; xs [B][N][P] = priors_on_featmap (un-flipped); ys [P] = prior_feat_ys; out [B][N][P][64]. */
/* out_cp (optional): the same samples laid out [B][N][64][P] for the routing gate. */
int phnet_roi_pool_fwd(const float* fmap, const float* xs, const float* ys, float* out, float* out_cp,
                       int32_t B, int32_t N, int32_t P, int32_t h, int32_t w, int32_t C, void* stream);
/* dmap [B][h][w][64] is accumulated into (may be NULL); dxs [B][N][P] is overwritten (may be NULL). */
int phnet_roi_pool_bwd(const float* dout, const float* fmap, const float* xs, const float* ys,
                       float* dmap, float* dxs,
                       int32_t B, int32_t N, int32_t P, int32_t h, int32_t w, int32_t C, void* stream);

/* ---- convolution / linear on fp32 MFMA: replaces F.conv2d (libs/models/resnet.py:79-95,293-307;
 * libs/models/fpn.py:109-163) and F.linear (Router4OL.py:308-392, utils/dynamic_head.py:31-59, Router.py:72-81).
 * x [N][Hi][Wi][Ci], w [Co][R][S][Ci], bias [Co] or NULL, y [N][Ho][Wo][Co]; Ci%4==0, Co%4==0.
 * workspace optional (split-K partial sums; recommended: >= 16*N*Ho*Wo*Co*4 bytes for small grids). */
int phnet_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                     int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                     int32_t stride, int32_t pad, int32_t relu, void* workspace, uint64_t ws_bytes, void* stream);
/* forward with a fused epilogue: y = relu?(conv + bias + addend) (addend shaped like y: the residual branch of an eval-mode
 * BatchNorm block whose scale / shift were folded into w / bias), and / or stats = per-channel (sum | sum of squares) partials
 * of y for a following training-mode BatchNorm: phnet_conv2d_stats_blocks() rows of 2*Co floats, consumed by
 * phnet_bn_finalize_partials (no separate statistics pass over y; needs bias == addend == NULL, relu == 0, Co = 2^k <= 1024). */
uint64_t phnet_conv2d_stats_blocks(int64_t M, int32_t Co, int32_t K, uint64_t ws_bytes);
int phnet_conv2d_fwd_fused(const float* x, const float* w, const float* bias, const float* addend, float* y, float* stats,
                           int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                           int32_t stride, int32_t pad, int32_t relu, void* workspace, uint64_t ws_bytes, void* stream);
/* dx = dgrad(dy, w) (+ addend): addend is optional, shaped like dx, and may alias dx. */
int phnet_conv2d_dgrad(const float* dy, const float* w, const float* addend, float* dx,
                       int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                       int32_t stride, int32_t pad, void* workspace, uint64_t ws_bytes, void* stream);
/* ---- 3x3 / stride 1 / pad 1 convolutions on PACKED weights (csrc/conv3p.hip): the trunk / FPN 3x3 layers of
 * libs/models/resnet.py:79-95 and fpn.py:156-160, forward and data gradient.  phnet_conv3p_pack splits a weight
 * [Co][3][3][Ci] once into its three bf16 terms, laid out in MFMA fragment order (dgrad = 0: the forward's operand;
 * 1: the data gradient's - flipped taps, transposed channels); phnet_conv3p_fwd then computes
 *   y = conv3x3(x, w) (+ bias) (+ addend) (ReLU)          on the dgrad = 0 packing (Ca = Ci, Nn = Co), or
 *   dx = conv3x3_dgrad(dy, w) (+ addend)                   on the dgrad = 1 packing (x = dy, Ca = Co, Nn = Ci),
 * same arithmetic and epilogues as phnet_conv2d_fwd_fused (stats: rows of 2*Nn floats, phnet_conv3p_stats_blocks of them).
 * Needs Ca % 16 == 0, Nn % 64 == 0 (phnet_conv3p_applies). */
int phnet_conv3p_applies(int64_t M, int32_t Ca, int32_t Nn);
uint64_t phnet_conv3p_packed_bytes(int32_t Co, int32_t Ci);
int phnet_conv3p_pack(const float* w, void* packed, int32_t Co, int32_t Ci, int32_t dgrad, void* stream);
/* all 3x3 weights of a model in one launch: jobs = DEVICE array of njobs records {const float* w; void* dst; int32 Co, Ci, dgrad, 0;
 * int64 first} sorted by `first` = running sum of the jobs' item counts 9*(Ca/16)*(Nn/32)*64; total = the sum of all counts */
int phnet_conv3p_pack_jobs(const void* jobs, int32_t njobs, int64_t total, void* stream);
uint64_t phnet_conv3p_stats_blocks(int64_t M, int32_t Ca, int32_t Nn, uint64_t ws_bytes);
int phnet_conv3p_splits(int64_t M, int32_t Ca, int32_t Nn, uint64_t ws_bytes);
int phnet_conv3p_fwd(const float* x, const void* packed, const float* bias, const float* addend, float* y, float* stats,
                     int32_t N, int32_t H, int32_t W, int32_t Ca, int32_t Nn, int32_t relu,
                     void* workspace, uint64_t ws_bytes, void* stream);
/* host-side query (no device work; bm/bn/splits/k_tile are HOST pointers): tile, split-K factor and K-tile depth the
 * two calls above use, i.e. the template arguments of the conv_igemm_kernel<BM, BN, DGRAD, BKT, UNI> they launch
 * (UNI = uniform-tap variant: 64x64 tile and A-side channel count % BKT == 0). */
int phnet_conv2d_plan(int64_t M, int32_t Co, int32_t K, uint64_t ws_bytes, int32_t* bm, int32_t* bn, int32_t* splits,
                      int32_t* k_tile);
uint64_t phnet_conv2d_wgrad_workspace(int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co,
                                      int32_t R, int32_t S, int32_t stride, int32_t pad);
/* dbias (optional, [Co]) = sum of dy over all pixels = the bias gradient, produced by the same launch. */
int phnet_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                       int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                       int32_t stride, int32_t pad, int32_t accumulate, void* workspace, uint64_t ws_bytes, void* stream);
/* Backward of a Linear layer over few rows (the M = 240 layers of the lane head) as ONE launch: dx [M][K] = dy w and
 * dw [N][K] (+)= dy^T x, dbias [N] (+)= column sums of dy (dbias optional).  dy [M][N], x [M][K], w [N][K] row-major.
 * phnet_linear_bwd_fusable tells whether a shape qualifies (M <= 256, N <= 640, few tiles); otherwise use
 * phnet_conv2d_dgrad + phnet_conv2d_wgrad. */
int phnet_linear_bwd_fusable(int64_t M, int64_t K, int64_t N);
int phnet_linear_bwd(const float* dy, const float* x, const float* w, const float* relu_y, float* dx, float* dw, float* dbias,
                     int32_t M, int32_t K, int32_t N, int32_t accumulate, void* stream);   /* relu_y (optional, [M][N]): saved
                     output of a layer that ended in a ReLU; dy is masked by relu_y > 0 while it is staged */
/* stem helpers: NCHW 3-channel frames -> NHWC padded to 4 channels; innermost-dimension pad/truncate. */
int phnet_nchw3_to_nhwc4(const float* x, float* y, int32_t N, int32_t H, int32_t W, void* stream);
int phnet_pad_channels(const float* src, float* dst, int64_t rows, int32_t cs, int32_t cd, void* stream);

/* ---- one-shot all-reduce of small messages over peer-mapped buffers (csrc/ipc_allreduce.hip): the SyncBatchNorm statistic
 * exchanges of trainOL.py:141 (<= 8 KB, 72 per step, on the critical path).  Each rank allocates an exchange buffer
 * (phnet_ipc_alloc, phnet_ipc_buffer_bytes), hands its 64-byte handle to the peers over any host channel, maps theirs
 * (phnet_ipc_open_handle) and passes the table of mapped pointers (a DEVICE array, entry `rank` = the local buffer) to
 * phnet_oneshot_allreduce: in place, SUM, contributions added in rank order (bit-identical on every rank), one launch on
 * `stream`, capturable.  ctrl: DEVICE {uint32 sequence, uint32 error}, sequence initialised to 1 on every rank. */
uint64_t phnet_ipc_buffer_bytes(int32_t world, uint64_t max_bytes);
int phnet_ipc_alloc(uint64_t bytes, void** ptr);
int phnet_ipc_free(void* ptr);
int phnet_ipc_get_handle(void* ptr, void* handle64);
int phnet_ipc_open_handle(const void* handle64, void** ptr);
int phnet_ipc_close_handle(void* ptr);
int phnet_oneshot_allreduce(void* data, int32_t count, int32_t dtype, const void* peers, int32_t rank, int32_t world,
                            int32_t cap, void* ctrl, void* stream);

/* ---- BatchNorm2d / ReLU / residual: replaces F.batch_norm + relu + add (libs/models/resnet.py:79-95, 293-297) ---- */
uint64_t phnet_channel_partials_size(int64_t M, int32_t C);   /* floats needed in `partial` */
int phnet_bn_fwd_stats(const float* x, int64_t M, int32_t C, float eps, float momentum,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float* save_mean, float* save_invstd, float* scale, float* shift,
                       float* partial, int32_t training, void* stream);
/* the finalize half of phnet_bn_fwd_stats on partials somebody else produced (phnet_conv2d_fwd_fused): partial [nblk][2C] */
int phnet_bn_finalize_partials(const float* partial, int64_t nblk, int64_t M, int32_t C, float eps, float momentum,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float* save_mean, float* save_invstd, float* scale, float* shift, void* stream);
int phnet_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                   int64_t M, int32_t C, int32_t relu, void* stream);
int phnet_bn_bwd(const float* dy, const float* x, const float* y, const float* save_mean,
                 const float* save_invstd, const float* gamma, float* dx, float* dres,
                 float* dgamma, float* dbeta, float* partial, float* c1, float* c2,
                 int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate, int32_t param_accumulate, void* stream);

/* split form for SyncBatchNorm (trainOL.py:141): local sums[2][C] = (sum g*xhat, sum g) -> caller all-reduces -> apply. */
int phnet_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                        float* sums, float* partial, float* c1_scratch, float* c2_scratch,
                        int64_t M, int32_t C, int32_t relu, void* stream);
int phnet_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                       const float* gamma, const float* c1, const float* c2, float* dx, float* dres,
                       int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate, void* stream);

/* device-resident SyncBatchNorm (trainOL.py:141): no host read of the element count, so the step stays sync-free.
 * fwd: phnet_bn_local_sums -> all-reduce(SUM) sums[2C+1] (fp64: sum x, sum x^2, count) -> phnet_bn_finalize_sums -> phnet_bn_apply
 * bwd: phnet_bn_bwd_reduce -> all-reduce(SUM) sums[2][C] -> phnet_bn_bwd_apply_sums (count = &sums_fwd[2C]) */
int phnet_bn_local_sums(const float* x, int64_t M, int32_t C, float* partial, double* sums, void* stream);
/* the same sums from the per-block partials of the convolution's epilogue (phnet_conv2d_fwd_fused stats): no pass over x */
int phnet_bn_partials_to_sums(const float* partial, int64_t nblk, int64_t M, int32_t C, double* sums, void* stream);
int phnet_bn_finalize_sums(const double* sums, int32_t C, float eps, float momentum, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                           float* scale, float* shift, void* stream);
int phnet_bn_bwd_apply_sums(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                            const float* gamma, const float* sums, const double* count, float* c1, float* c2,
                            float* dx, float* dres, int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate, void* stream);

/* ---- input pre-processing of a clip: libs/dataset/openlane/datasetOL.py:40-52 (crop top rows, optional
 * flip), transforms.py:150-156 (iaa.Resize = cv2 INTER_CUBIC on uint8), datasetOL.py:63-75,11-17 (ToTensor, Normalize, stack).
 * frames u8 [T][H0][W0][3] RGB on the device; out f32 NCHW [T][3][oh][ow] (layout 0) or NHWC4 [T][oh][ow][4] (layout 1);
 * out_u8 optional [T][oh][ow][3]; xi/xc [ow][4], yi/yc [oh][4]: clamped source indices (int32) and 11-bit taps (int16, each saturate_cast<short>(c * 2048): sum 2047..2049)
 * of the half-pixel a = -0.75 cubic kernel, built once per geometry by the caller; mean3 / std3 are HOST pointers. ---- */
int phnet_preprocess_u8(const uint8_t* frames, float* out, uint8_t* out_u8,
                        const int32_t* xi, const int16_t* xc, const int32_t* yi, const int16_t* yc,
                        int32_t T, int32_t H0, int32_t W0, int32_t crop_top, int32_t out_h, int32_t out_w, int32_t flip,
                        int32_t layout, const float* mean3_host, const float* std3_host, void* stream);

/* ---- lane IoU of the CULane-style evaluator: replaces LaneCompare::get_lane_similarity's two
 * cv::Mat canvases + cv::line + cv::sum (evaluation/culane/src/lane_compare.cpp:11-57).  A lane is a bit mask
 * [height][ceil(width/32)] uint32 in device memory.  phnet_lane_raster ORs thick segments into masks the caller zeroed: segs
 * [n_segs][5] int32 = (x0, y0, x1, y1, lane) with end points inside +-8192 (the cvRound'ed points of the interpolated lane);
 * a pixel is set iff its centre lies within lane_width/2 of the segment (exact integer test; parity against OpenCV's scan
 * conversion of the same segment unpinned - see oracle/culane_cpu.py).  phnet_lane_mask_stats adds the set bits of every mask to
 * area[n_lanes] and of masks[pairs[p][0]] & masks[pairs[p][1]] to inter[n_pairs] (int64, zeroed by the caller).
 * height, width <= 4096; lane_width <= 256; n_lanes + n_pairs <= 65535. ---- */
int phnet_lane_raster(const int32_t* segs, int64_t n_segs, uint32_t* masks, int32_t n_lanes, int32_t height, int32_t width,
                      int32_t lane_width, void* stream);
int phnet_lane_mask_stats(const uint32_t* masks, int32_t n_lanes, int32_t height, int32_t width, const int32_t* pairs,
                          int32_t n_pairs, int64_t* area, int64_t* inter, void* stream);

/* ---- MaxPool2d(3,2,1): libs/models/resnet.py:217,297 ---- */
int phnet_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t Hi, int32_t Wi, int32_t C, void* stream);
int phnet_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t Hi, int32_t Wi, int32_t C, void* stream);

/* ---- FPN top-down add: laterals[i-1] += F.interpolate(laterals[i], size=..., mode='nearest') (libs/models/fpn.py:127-141) ---- */
int phnet_upsample_add(float* fine, const float* coarse, int32_t N, int32_t H, int32_t W, int32_t h, int32_t w, int32_t C, void* stream);
int phnet_upsample_add_bwd(const float* dfine, float* dcoarse, int32_t N, int32_t H, int32_t W, int32_t h, int32_t w, int32_t C, void* stream);

/* ---- LayerNorm (+residual)(+ReLU): replaces F.layer_norm at Router.py:72-81, utils/dynamic_head.py:42-58,
 * utils/transformer.py:275-298.  rows x L, affine w/b [L]; y = relu?(LN(x)*w + b (+res)). ---- */
int phnet_layernorm_fwd(const float* x, const float* w, const float* b, const float* res, float* y,
                        float* mean, float* rstd, int64_t rows, int32_t L, float eps, int32_t relu, void* stream);
uint64_t phnet_layernorm_bwd_workspace(int64_t rows, int32_t L);
int phnet_layernorm_bwd(const float* dy, const float* x, const float* y, const float* w,
                        const float* mean, const float* rstd, float* dx, float* dres, float* dw, float* db,
                        int64_t rows, int32_t L, int32_t relu, int32_t param_accumulate,
                        void* workspace, uint64_t ws_bytes, void* stream);

/* ---- routing-gate depth-wise 3x3: replaces Conv2d(N,N,3,padding=1,groups=N) (libs/models/Router.py:55-61).
 * x,y [N][C][P] planes (one per anchor), w [N][3][3], bias [N] or NULL; flip=1 computes the data gradient. ---- */
int phnet_dwconv3x3(const float* x, const float* w, const float* bias, float* y,
                    int32_t N, int32_t C, int32_t P, int32_t flip, void* stream);
int phnet_dwconv3x3_wgrad(const float* dy, const float* x, float* dw, float* db,
                          int32_t N, int32_t C, int32_t P, int32_t accumulate, void* stream);

/* ---- label assignment: replaces dynamic_assign.assign + scipy linear_sum_assignment on a .cpu() copy
 * (libs/utils/dynamic_assign.py:128-190) with one device launch.  pred [N][6+S], tgt [L][6+S] (col 1 == 1 = valid),
 * N <= 256, L <= 4.  rows_by_col [L] i64 (anchor matched to label j, -1 = none), rows_sorted [L] i64 (matched anchors
 * ascending, -1 padded), n_valid (optional) i32, cost (optional) [N][L] f32 (the solver's matrix, +inf = invalid). ---- */
int phnet_lane_assign(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S,
                      float img_w, float img_h, int64_t* rows_by_col, int64_t* rows_sorted,
                      int32_t* n_valid, float* cost, void* stream);
/* One-to-many assignment: libs/utils/dynamic_assign.py:292-357 `assignOne2Many` (focal-cost alpha 0.5; label j wants
 * k_j = max(1, int(sum of its 4 largest line IoUs)) anchors; rounds of the exact matching, each keeps the pairs at the
 * positions whose k is still positive and retires their anchors) in one launch.  rows / cols [16] i64 (anchor, label row),
 * the reference's pair order, -1 padded; n_pairs (optional) i32. */
int phnet_lane_assign_one2many(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S,
                               float img_w, float img_h, int64_t* rows, int64_t* cols, int32_t* n_pairs, void* stream);
/* phnet_lane_assign + phnet_memory_tokens on its result in one launch (the pair the training schedule issues after every branch-B
 * pass): feat [N][E] this frame's tokens, tokens [L+1][E] / valid u8 [L+1] the memory entry they leave (Router4OL.py:563-584). */
int phnet_lane_assign_tokens(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                             int64_t* rows_by_col, int64_t* rows_sorted, const float* feat, int32_t E,
                             float* tokens, uint8_t* valid, void* stream);

/* ---- fused per-frame criterion: replaces Criterion4OL.loss4OneStep (libs/utils/loss4OLV3.py:34-82,100-123: assignment,
 * focal, smooth-L1, LaneIoU, gate-weighted combination) AND its autograd backward with two launches.
 * pred / gate / dpred are HOST arrays of device pointers: pred[6] = branch A stages 0..2 then branch B stages 0..2,
 * each [N][6+S]; gate[3] each [N]; dpred[6] each [N][6+S].  tgt [L][6+S], L <= 4, N <= 256.
 * Outputs: loss [1]; dpred, dgate [3][N] = d loss / d input for unit upstream gradient; rows_by_col / rows_sorted
 * [6][L] int64 matched anchors (-1 padded).  Scratch: focal [6][N], scalars [12]. ---- */
int phnet_frame_loss(const float* const* pred, const float* const* gate, const float* tgt,
                     int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                     float cls_w, float reg_w, float iou_w,
                     float liou_half_width, float liou_img_h, float liou_img_w,
                     float* loss, float* const* dpred, float* dgate,
                     int64_t* rows_by_col, int64_t* rows_sorted, float* focal, float* scalars, void* stream);
/* The T frames of a clip in the SAME two launches (the criterion is called frame by frame, trainOLV3.py:150-171, and nothing
 * flows from one frame's loss to the next): pred [T*6] / gate [T*3] / dpred [T*6] host arrays of device pointers, frame-major;
 * tgt [T][L][6+S]; loss [T]; dgate [T][3][N]; rows_by_col / rows_sorted [T][6][L]; scratch focal [T][6][N], scalars [T][12].
 * T <= 8.  phnet_frame_loss is the T = 1 case. */
int phnet_clip_loss(const float* const* pred, const float* const* gate, const float* tgt, int32_t T,
                    int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                    float cls_w, float reg_w, float iou_w,
                    float liou_half_width, float liou_img_h, float liou_img_w,
                    float* loss, float* const* dpred, float* dgate,
                    int64_t* rows_by_col, int64_t* rows_sorted, float* focal, float* scalars, void* stream);
/* the reference's other two criteria, fused the same way (csrc/loss_variants.hip; two launches per frame):
 * variant 1 = libs/utils/loss4OL.py:88-232 (per-pair terms summed by position, placed on the last stage's anchors),
 * variant 2 = libs/utils/loss4OLV2.py:12-186 (one-to-many assignment, up to 16 pairs per branch and stage).
 * pair_rows / pair_cols [6][16] int64 (-1 padded), rows_sorted [6][L] int64 (variant 1), scratch 6*N + 192 floats. */
int phnet_frame_loss_variant(int32_t variant, const float* const* pred, const float* const* gate, const float* tgt,
                             int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                             float cls_w, float reg_w, float iou_w,
                             float* loss, float* const* dpred, float* dgate,
                             int64_t* pair_rows, int64_t* pair_cols, int64_t* rows_sorted, float* scratch, void* stream);

/* ---- lane prior update: replaces the tanh/tan/repeat/cat chain of forward_first/second (Router4OL.py:328-345) and its
 * backward.  priors [N][6+S]; head [N][HW] = (cls 2 | reg 4 | offsets S | zero pad); ys [S] = prior_ys. ---- */
int phnet_lane_update_fwd(const float* priors, const float* head, const float* ys, float* preds, float* lines,
                          int32_t N, int32_t S, int32_t HW, float img_w, float img_h, void* stream);
int phnet_lane_update_bwd(const float* dpreds, const float* dlines, const float* lines, const float* head,
                          const float* ys, float* dhead, float* dpriors,
                          int32_t N, int32_t S, int32_t HW, float img_w, float img_h, void* stream);

/* ---- ReLU backward through the saved output (fused-ReLU epilogues of the linears) ---- */
int phnet_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* ---- fused depth-wise stack of the routing gate: pre_norm + 4 x relu(LN(dw(relu(LN(dw(x))))) + x)
 * (libs/models/Router.py:72-75) in one launch forward, one backward (+ one reduce).  x/out [N][C][P], C*P <= 3072.
 * params / grads: HOST arrays of 34 device pointers: pre_norm.weight, pre_norm.bias, then per block b = 0..3:
 * conv1.weight [N][9], conv1.bias [N], ln1.weight [C*P], ln1.bias, conv2.weight, conv2.bias, ln2.weight, ln2.bias.
 * saved: phnet_gate_stack_saved_floats() floats written by fwd (NULL = inference), read by bwd.
 * N planes may cover several frames: plane n belongs to anchor n % anchors (the per-anchor filters are [anchors][9], N % anchors
 * == 0); the filter gradients of such a batch go through per-plane partials and one more small reduce launch. ---- */
uint64_t phnet_gate_stack_saved_floats(int32_t N, int32_t C, int32_t P);
uint64_t phnet_gate_stack_bwd_workspace(int32_t N, int32_t C, int32_t P);
int phnet_gate_stack_fwd(const float* x, const float* const* params, float* out, float* saved,
                         int32_t N, int32_t anchors, int32_t C, int32_t P, float eps, void* stream);
int phnet_gate_stack_bwd(const float* gout, const float* x, const float* out, const float* const* params,
                         const float* saved, float* const* grads, int32_t N, int32_t anchors, int32_t C, int32_t P, float eps,
                         int32_t accumulate, void* workspace, uint64_t ws_bytes, void* stream);
/* the wave-per-plane forms of the two calls above (csrc/gate_wave.hip; C = 64, P = 36: what the model runs) - phnet_gate_stack_fwd /
 * _bwd dispatch to them where phnet_gate_wave_applies; `saved` then holds only the four block inputs [4][N][C*P] */
int phnet_gate_wave_applies(int32_t C, int32_t P);
int phnet_gate_wave_partial_planes(int32_t N);
int phnet_gate_wave_fwd(const float* x, const float* const* params, float* out, float* saved,
                        int32_t N, int32_t anchors, float eps, void* stream);
int phnet_gate_wave_bwd(const float* gout, const float* x, const float* out, const float* const* params, const float* saved,
                        float* const* grads, int32_t N, int32_t anchors, float eps, int32_t accumulate,
                        void* workspace, void* stream);

/* ---- fused attention core (heads of width 16, Lq/Lk <= 256): replaces the scale/bmm/mask/softmax/dropout/bmm chain inside
 * nn.MultiheadAttention (libs/models/utils/transformer.py:275-298) and its backward.  q/k/v/o and the gradients are
 * addressed with row strides (floats, multiples of 4, 16-byte aligned bases), heads packed along the row.  key_valid u8[Lk] optional; keep u8[H][Lq][Lk]
 * optional explicit dropout keep-mask (kept weights scaled by keep_scale); with keep == NULL, rng_state != NULL and
 * drop_p > 0 the mask is drawn in the kernel instead: element (h, q, k) of dropout site rng_call is kept iff a splitmix64
 * hash of (*rng_state, rng_call, element index) clears drop_p * 2^32, scale 1/(1-drop_p); the backward recomputes the
 * same bits (F.dropout inside nn.MultiheadAttention).  *rng_state is a device counter the caller bumps once per step.
 * lse [H][Lq].  B clips per launch: clip b owns rows [b*Lq, (b+1)*Lq) of q/o/dq and [b*Lk, (b+1)*Lk) of k/v/dk/dv; key_valid
 * [B][Lk], keep [B][H][Lq][Lk], lse [B][H][Lq]. ---- */
int phnet_attention_fwd(const float* q, const float* k, const float* v, const uint8_t* key_valid, const uint8_t* keep,
                        float* o, float* lse, int32_t B, int32_t Lq, int32_t Lk, int32_t H, int32_t E,
                        int64_t sq, int64_t sk, int64_t sv, int64_t so, float keep_scale,
                        const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream);
int phnet_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                        const float* lse, const uint8_t* key_valid, const uint8_t* keep,
                        float* dq, float* dk, float* dv, int32_t B, int32_t Lq, int32_t Lk, int32_t H, int32_t E,
                        int64_t sq, int64_t sk, int64_t sv, int64_t so, int64_t sdq, int64_t sdk, int64_t sdv,
                        float keep_scale, const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream);

/* ---- transformer glue: `tgt + dropout(x)` and `dropout(gelu(x))` (libs/models/utils/transformer.py:275-298) as one launch
 * each, dropout bits from the same counter-based generator as the attention kernels (site id rng_call; drop_p = 0 or
 * rng_state = NULL: no dropout).  phnet_dropout_add with res = NULL is plain dropout, which is also its own backward. ---- */
int phnet_dropout_add(const float* x, const float* res, float* y, int64_t n,
                      const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream);
int phnet_gelu_dropout_fwd(const float* x, float* y, int64_t n, const uint64_t* rng_state, uint64_t rng_call, float drop_p,
                           void* stream);
int phnet_gelu_dropout_bwd(const float* dy, const float* x, float* dx, int64_t n, const uint64_t* rng_state,
                           uint64_t rng_call, float drop_p, void* stream);

/* ---- small fused pieces of the per-frame loop ----
 * memory tokens (Router4OL.py:563-584, _tokens): rows i64[L] = positive anchors ascending, -1 padded; tokens [L+1][E] = their
 *   features, then the mean of all other anchors; valid u8[L+1].
 * gate tail (Router.py:76-80): gate = sigmoid(relu(h . w + b)) and its backward from the saved output.
 * stage hand-over (Router4OL.py:298-302): blended priors and their sampled x positions. ---- */
int phnet_memory_tokens(const float* feat, const int64_t* rows, float* tokens, uint8_t* valid,
                        int32_t B, int32_t N, int32_t E, int32_t L, void* stream);      /* B clips: leading dimension of all four */
int phnet_gate_tail_fwd(const float* h, const float* w, const float* b, float* out, int32_t N, int32_t K, void* stream);
int phnet_gate_tail_bwd(const float* dout, const float* out, const float* h, const float* w, float* dh, float* dw, float* db,
                        int32_t N, int32_t K, int32_t accumulate, void* stream);
int phnet_blend_priors(const float* gate, const float* a, const float* b, const int64_t* idx, float* priors, float* on_map,
                       int32_t N, int32_t W, int32_t P, void* stream);

/* ---- Router4OLV2 model family (what testOLV3.py imports; inference only - its training path cannot run as shipped) ----
 * phnet_gate_v2_fwd: AdaptiveRouter4LaneV2.forward (libs/models/Router.py:83-132) in one launch: Conv1d(k3, pad 1, no bias)
 *   + BatchNorm1d + ReLU, Conv1d(k1) + BatchNorm1d + ReLU, Flatten, Linear(C2*P -> P), mean over the P outputs, sigmoid.
 *   x [M][C][P]; w1 [C1][C][3]; s1 / t1 [C1] BatchNorm folded to conv * s + t; w2 [C2][C1]; s2 / t2 [C2]; wl [P][C2*P];
 *   bl [P]; out [M].
 * phnet_dyn_bmm_ln_relu_fwd_any: the two per-anchor products of DynamicConvV2 (libs/models/utils/dynamic_head.py:94-104) at
 *   run-time shapes (per-level widths 64/32/16, 24/48/96 sample points), forward only: y[n] = relu(LayerNorm_J(x[n] @ w[n])).
 * phnet_route_lines: RouterOL.forward, eval (libs/models/Router4OLV2.py:508-511): d = mean over the S stage gates, then
 *   hard != 0: out = d >= 0.5 ? b : a (torch.where);  hard == 0: out = b * d + a * (1 - d) (Router4OL.py:538-541).
 *   gates [S][M]; a (branch A), b (branch B), out [M][W].
 * The per-level ROI pooling is phnet_roi_pool_fwd with C < 64, the 8 x 32 attention phnet_attention_fwd with E = 32 H. */
int phnet_gate_v2_fwd(const float* x, const float* w1, const float* s1, const float* t1, const float* w2,
                      const float* s2, const float* t2, const float* wl, const float* bl, float* out,
                      int32_t M, int32_t C, int32_t P, int32_t C1, int32_t C2, void* stream);
int phnet_dyn_bmm_ln_relu_fwd_any(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                                  int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream);
int phnet_route_lines(const float* gates, const float* a, const float* b, float* out,
                      int32_t S, int32_t M, int32_t W, int32_t hard, void* stream);

/* ---- row-local chain of a pre-norm transformer layer in one launch, forward only (libs/models/utils/transformer.py:275-298
 * between the attention cores): [v = in @ Wa^T + ba; t = resid + dropout(v)] | t = in;  h = LayerNorm(t; ln1);
 * [f = dropout(gelu(h @ W1^T + b1)); t += dropout(f @ W2^T + b2); h = LayerNorm(t; ln2)];  [y = h @ Wg^T + bg].
 * in / resid / t_out / h_out [R][E], y_out [R][NG]; E = 128 (FF 256) or 256 (FF 512); optional parts are NULL.  Dropout sites as
 * in phnet_dropout_add (same generator, same element indices as the unfused kernels: a recomputation through those sees the
 * masks drawn here). ---- */
int phnet_rowchain_fwd(const float* in, const float* resid, const float* Wa, const float* ba, const float* ln1w, const float* ln1b,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* ln2w, const float* ln2b,
                       const float* Wg, const float* bg, float* t_out, float* h_out, float* y_out,
                       int32_t R, int32_t E, int32_t FF, int32_t NG, float eps,
                       const uint64_t* rng_state, uint64_t call_a, uint64_t call_f, uint64_t call_3, float drop_p, void* stream);

/* ---- tower weights of a lane-head branch (libs/models/Router4OL.py:68-99, 308-326: T towers of two Linear + ReLU layers and one
 * Linear head each) assembled for the 3-GEMM chain, and their gradients scattered back: one launch each instead of the ~12
 * torch.cat / block_diag launches per branch and clip and the ~40 of their autograd backward.  params / grads: HOST arrays of 6*T
 * device pointers (per tower: layer-1 weight [C][C], bias [C], layer-2 weight, bias, head weight [o_t][C], head bias [o_t]).
 * The assembled tensors live in ONE buffer w1 [TC][C] | b1 [TC] | w2 [TC][TC] | b2 [TC] | wh [HW][TC] | bh [HW]
 * (phnet_tower_layout: its size, the six offsets, HW = sum o_t rounded up to a multiple of 4). ---- */
uint64_t phnet_tower_layout(int32_t T, int32_t C, const int32_t* head_out, int64_t* offsets, int32_t* hw);
int phnet_assemble_towers(const float* const* params, int32_t T, int32_t C, const int32_t* head_out, float* dst, void* stream);
int phnet_scatter_tower_grads(const float* src, float* const* grads, int32_t T, int32_t C, const int32_t* head_out,
                              int32_t accumulate, void* stream);

/* ---- towers + heads + lane prior update of a branch in one launch, forward only (libs/models/Router4OL.py:308-345): per tower
 * relu(x W1^T + b1) -> relu(. W2^T + b2) -> head; the heads concatenated are (cls 2 | reg 4 | offsets S); then the prior update of
 * phnet_lane_update_fwd.  params: HOST array of 6*T device pointers, used where they are (no assembled copies).  C = 64 or 128. ---- */
int phnet_tower_chain_fwd(const float* x, const float* const* params, int32_t T, int32_t C, const int32_t* head_out,
                          const float* priors, const float* ys, float* preds, float* lines, int32_t R, int32_t S,
                          float img_w, float img_h, void* stream);

/* ---- optimizer: one AdamW step (torch.optim.AdamW semantics, libs/utils/optimizer.py:33-35) over flat parameter / gradient /
 * moment arrays; elements [0, n_decay) get decoupled weight decay.  n % 4 == 0.  step: device int64, 1-based, already
 * incremented by the caller for this step.  lr_dev (optional): DEVICE pointer to the learning rate; when non-NULL it
 * replaces `lr`, so that a hipGraph-captured step follows an LR schedule (the host rewrites the scalar between replays). ---- */
int phnet_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, const int64_t* step,
                     float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* residual + dropout + LayerNorm of the pre-norm decoder layers in one launch: t = res + dropout(x), h = LN_L(t)*w + b
 * (L <= 256), and its backward (dt = gradient arriving on the residual stream, may be NULL): dres, dx, dw, db. */
int phnet_dropout_add_ln_fwd(const float* x, const float* res, const float* w, const float* b, float* t, float* h,
                             float* mean, float* rstd, int64_t rows, int32_t L, float eps,
                             const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream);
int phnet_dropout_add_ln_bwd(const float* dh, const float* dt, const float* t, const float* w, const float* mean,
                             const float* rstd, float* dres, float* dx, float* dw, float* db,
                             int64_t rows, int32_t L, int32_t param_accumulate,
                             const uint64_t* rng_state, uint64_t rng_call, float drop_p,
                             void* workspace, uint64_t ws_bytes, void* stream);

/* ---- per-anchor dynamic convolution: y[n] = relu(LayerNorm_J(x[n] @ w[n]) * gamma + beta), replacing torch.bmm + norm1/norm2
 * + ReLU in libs/models/utils/dynamic_head.py:40-51 (and their backward).  x [N][P][K], w [N][K][J] (generated per anchor),
 * y [N][P][J], stats [N][P][2] = (mean, rstd) (NULL = inference).  J <= 128; J and K divide 256.  Backward: dx optional,
 * dw [N][K][J], dgamma/dbeta [J] overwritten or accumulated; workspace N*2*J floats. ---- */
int phnet_dyn_bmm_ln_relu_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                              float* stats, int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream);
int phnet_dyn_bmm_ln_relu_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats,
                              const float* gamma, float* dx, float* dw, float* dgamma, float* dbeta,
                              int32_t N, int32_t P, int32_t K, int32_t J, float eps, int32_t param_accumulate,
                              void* workspace, uint64_t ws_bytes, void* stream);
/* the matrix-pipe forms (csrc/dyn_mfma.hip: one wavefront per anchor, v_mfma_f32_16x16x4_f32, no LDS); the two calls above
 * dispatch to them where phnet_dyn_mfma_applies (P <= 36, (K, J) = (64, 128) or (128, 64)) */
int phnet_dyn_mfma_applies(int32_t P, int32_t K, int32_t J);
int phnet_dyn_mfma_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y, float* stats,
                       int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream);
int phnet_dyn_mfma_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats, const float* gamma,
                       float* dx, float* dw, float* lnpart, int32_t N, int32_t P, int32_t K, int32_t J, void* stream);

/* ---- bias gradients: column sums of [M][C] ---- */
uint64_t phnet_colsum_workspace(int64_t M, int32_t C);
int phnet_colsum(const float* a, float* out, int64_t M, int32_t C, int32_t accumulate, void* workspace, uint64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
