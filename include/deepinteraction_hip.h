/* deepinteraction_hip.h - C ABI of libdeepinteraction_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the DeepInteraction interaction hot path
 * (SURVEY.md 8(b)).  It replaces, for the reference at /root/reference
 * (paths relative to projects/mmdet3d_plugin/):
 *
 *   - the pybind11 module `locatt_ops.localattention`
 *       models/utils/ops/locatt_ops/localAttention.h:11-40, localAttention.cpp:61-73
 *   - the torch / OpenCV call sequences inside
 *       models/utils/encoder_utils.py:127-135 (LocalContextAttentionBlock.forward)
 *       models/utils/encoder_utils.py:142-199 (BEVWarp.forward + ip_basic fill_in_multiscale)
 *       models/utils/encoder_utils.py:257-320 (MMRI_I2P.forward / group_attn)
 *       models/utils/decoder_utils.py:660-761, 788-841 (RoI blocks: projection, rects, ROIAlignV2)
 *       models/utils/decoder_utils.py:96-103 (200 x 32400 cross attention)
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch types.  All pointers are
 *     DEVICE pointers unless the parameter name ends in `_host`.
 *   - the CALLER owns every buffer (outputs and scratch); nothing is allocated,
 *     freed or synchronised inside; every launch goes to `stream` (a hipStream_t).
 *   - re-entrant and callable from any host thread (autograd worker threads).
 *   - return 0 on success, a negative DI_ERR_* code on failure; no exceptions
 *     cross the ABI; `di_last_error()` returns a thread-local message.
 *   - feature maps are CHANNELS-LAST: (n, H, W, C) with C contiguous
 *     (the physical layout of a torch tensor of logical shape (n,C,H,W) and
 *     memory_format=torch.channels_last).  C must be a multiple of 8, <= 128.
 *   - `dtype`: DI_F32 or DI_F16 = element type of the feature maps.  Arithmetic
 *     accumulates in fp32 in both cases.  Geometry (points, matrices, depth,
 *     window weights of the unfused ops) is always float32.
 */
#ifndef DEEPINTERACTION_HIP_H
#define DEEPINTERACTION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DI_F32 = 0, DI_F16 = 1, DI_F16_HL = 2 /* outputs only: split pair [hi | lo] fp16, x = hi + lo / 2048 */ };
enum {
  DI_OK = 0,
  DI_ERR_ARG = -1,      /* unsupported shape / window / dtype */
  DI_ERR_LAUNCH = -2,   /* hipGetLastError() after a launch    */
  DI_ERR_LDS = -3       /* tile does not fit the 160 KiB LDS   */
};

int di_abi_version(void);
/* One-launch device-to-device copy of a large 16-byte-aligned buffer (the per-sample `load()` into the captured input arena;
 * plumbing of bench.py's step protocol, not a kernel of the reference path). */
int di_copy_d2d(void *dst, const void *src, long long bytes, void *stream);
const char *di_last_error(void);
/* Measurement plumbing: number of nodes of a captured hipGraph_t (HOST handle), < 0 on error.  bench.py reports
 * it as `graph_nodes` of the captured forward. */
long long di_graph_node_count(void *graph_host);

/* ---------------------------------------------------------------- local-window attention
 * Fused forward of LocalContextAttentionBlock.forward (encoder_utils.py:132-134):
 *   w = softmax_k( <q[p], k[p+off_k]> * scale ),  out[p] = sum_k w_k * v[p+off_k]
 * window slot k <-> (dy,dx) = (k / kW - kH/2, k % kW - kW/2) (kernels.cuh:22-27);
 * out-of-image slots score 0 AND stay in the softmax (kernels.cuh:28-39), their
 * value contribution is 0 (kernels.cuh:71-75).  `scale` = 1/sqrt(C) in the reference.
 * Supported windows: kH,kW odd in {3,5,7,9}. */
int di_local_attn_fwd(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                      int C, int kH, int kW, float scale, int dtype, void *stream);
/* Same op with an explicit kernel choice.  DI_LA_AUTO picks, for fp16 / C=128 / 9x9: on large maps the ring generation
 * (local_attn_ring.hip: one workgroup per CU, a 144 KB ring of 128-byte halo rows filled by producer wavefronts with LDS-DMA,
 * eight consumer wavefronts synchronised by LDS flag words, no workgroup barrier), on small maps the persistent
 * software-pipelined register-staged kernel (local_attn_mfma2.hip); both compute the same row-pair 16x16x32 MFMA tiles and are
 * bit-identical.  Everything else runs the generic LDS-tiled VALU kernel (any C, fp32 or fp16, windows 3..9).  The other codes
 * select one implementation (tests, measurements).  Five further matrix-core generations (one tile per workgroup; producer /
 * consumer waves with direct-to-LDS loads; vertical streaming with an LDS halo ring; LDS-DMA in 16 KB units at three
 * workgroups per CU; the same with producer / consumer wavefronts) were measured in rounds 1-3, lost and are gone. */
enum { DI_LA_AUTO = 0, DI_LA_VALU = 1,
       DI_LA_MFMA = 3 /* + configuration: 3 = 16x8 tiles, 4 = 8x8 tiles, 5 = 16x4 tiles, 6 = 8x16 tiles, 7 = timestamps */,
       DI_LA_RING = 24 /* + 0 = 16x8 query tiles (AUTO on large maps), + 1 = 8x16 tiles, + 2..5 = measurement variants
                          (shallower LDS read-ahead, the compiler's schedule, four producer wavefronts, plain instead of sc0 DMA loads) */ };
int di_local_attn_fwd_ex(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                         int C, int kH, int kW, float scale, int dtype, int variant, void *stream);
/* TRAINING form of the same attention, mixed precision (fp16 maps, C = 128, 9 x 9; float32 accumulation): replaces the chain
 * similarFunction -> F.softmax -> weightingFunction and ITS BACKWARD (encoder_utils.py:36-81,132-134; similar.cu:43-92,
 * weighting.cu:44-122), whose (n, H, W, 81) float32 weight tensor never exists here.
 * fwd: out as di_local_attn_fwd, plus lse[n, H, W] = log2 sum_k exp2(scale * log2(e) * <q, k>) over the 81 slots (zero-padded
 *      keys take part with logit 0, as in the reference).
 * bwd: recomputes the probabilities from lse on the matrix cores; dsum[n, H, W] is scratch (receives <grad_out, out> per
 *      query).  Three launches: the row dot product, a query-centred pass (grad_q) and a key-centred pass (grad_k, grad_v) -
 *      the window relation is symmetric, so the key-centred pass gathers instead of scattering (no atomics).
 * All maps channels-last [n, H, W, 128] fp16. */
int di_local_attn_train_fwd(const void *q, const void *k, const void *v, void *out, float *lse, int n, int H, int W,
                            float scale, void *stream);
int di_local_attn_train_bwd(const void *q, const void *k, const void *v, const void *out, const void *grad_out,
                            const float *lse, float *dsum, void *grad_q, void *grad_k, void *grad_v, int n, int H, int W,
                            float scale, void *stream);
/* The DI_LA_RING kernel bounds every flag spin.  A spin that gives up never hangs the device and never hands back plausible
 * numbers: a producer wavefront that gave up issues no further loads, a consumer wavefront that gave up stores NaN for that
 * query tile and every later tile of its workgroup, and a device counter is bumped.  Returns the number of spins that gave up
 * since the library was loaded (0 on a healthy run; synchronises `stream`), -1 on a HIP error.  Host code checks it where it
 * synchronises anyway (`GraphedHotPath.check_health()`, bench.py); environment DI_RING_DBG=32 injects the fault (tests). */
int di_local_attn_ring_timeouts(void *stream);
/* The same without a synchronisation: the counter is copied into the caller's PINNED host word (uint32) behind everything
 * queued on `stream`; the value is monotonic, the caller looks at it whenever it likes (`GraphedHotPath.__call__` checks the
 * word a previous replay filled and raises). */
int di_local_attn_ring_timeouts_async(void *host_word, void *stream);
/* Measurement plumbing (bench.py's live roofline): the duration of ONE kernel as its dispatch reports it.  di_timed_begin
 * creates two events and arms the calling thread: the next di_local_attn_fwd* (matrix-core kernels) or
 * di_ms_deform_attn_hm_fwd launch issued by this thread binds them to its dispatch (hipExtLaunchKernelGGL: the kernel's own
 * begin / end time stamps - what rocprofv3 reports; events recorded on the stream around a launch add the 2-3 us of their own
 * packets).  di_timed_consumed: 1 when a launch took them (disarms the thread either way).  di_timed_elapsed_us waits for the
 * stop event (`recorded` = the value of di_timed_consumed), writes microseconds and destroys both events. */
int di_timed_begin(void **start_ev, void **stop_ev);
int di_timed_consumed(void);
int di_timed_elapsed_us(void *start_ev, void *stop_ev, int recorded, float *us);
/* Measurement (environment DI_RING_DBG & 16): the phase time stamps of workgroup 0's wavefronts of the last DI_LA_RING launch,
 * 16 x 128 uint64 (tag << 56 | shader clock; entry 127 of a wave = its count) copied to `host_out`. */
int di_local_attn_ring_stamps(void *host_out, void *stream);

/* The five entry points of locatt_ops (localAttention.h:11-40), channels-last features,
 * float32 window tensors of shape (n,H,W,kH*kW):
 *   similar_forward(x_ori,x_loc)            -> di_locatt_similar_fwd
 *   similar_backward(x,grad,is_ori)         -> di_locatt_similar_bwd
 *   weighting_forward(x_ori,x_weight)       -> di_locatt_weighting_fwd
 *   weighting_backward_ori(x_weight,grad)   -> di_locatt_weighting_bwd_ori
 *   weighting_backward_weight(x_ori,grad)   -> di_locatt_weighting_bwd_weight */
int di_locatt_similar_fwd(const void *x_ori, const void *x_loc, float *out_w, int n, int H, int W,
                          int C, int kH, int kW, int dtype, void *stream);
int di_locatt_similar_bwd(const void *x, const float *grad_w, void *grad_in, int n, int H, int W,
                          int C, int kH, int kW, int is_ori, int dtype, void *stream);
int di_locatt_weighting_fwd(const void *x_ori, const float *x_weight, void *out, int n, int H, int W,
                            int C, int kH, int kW, int dtype, void *stream);
int di_locatt_weighting_bwd_ori(const float *x_weight, const void *grad_out, void *grad_ori, int n,
                                int H, int W, int C, int kH, int kW, int dtype, void *stream);
int di_locatt_weighting_bwd_weight(const void *x_ori, const void *grad_out, float *grad_w, int n,
                                   int H, int W, int C, int kH, int kW, int dtype, void *stream);

/* ---------------------------------------------------------------- fused 1x1-convolution chains
 * The Conv1x1 + BN (+ ReLU) projections of LocalContextAttentionBlock (encoder_utils.py:92-117) and the
 * out_proj / integration pairs of the encoder layer (deepinteraction_encoder.py:13-19, 26-32), with
 * BatchNorm folded by the caller, as ONE pass over the pixels of channels-last fp16 maps (C = 128):
 *     h = act1(W1 . [x1 ; x2] + b1)        k1 = 128 (x2 = NULL) or 256
 *     y = act2(W2 . [h  ; x3] + b2)        k2 = 0 (no second link: y = h), 128 (x3 = NULL) or 256
 * w1 (128,k1), w2 (128,k2) row-major fp16 (natural column order); b1, b2 (128) float32; relu* 0/1.
 * x*, y: (n_pixels, 128) fp16.  fp32 accumulation; the hidden map h never touches memory. */
int di_pointwise_chain_fwd(const void *x1, const void *x2, const void *x3, const void *w1, const float *b1,
                           const void *w2, const float *b2, void *y, long long n_pixels, int k1, int k2,
                           int relu1, int relu2, void *stream);
/* The same with a bias on link 1 that applies to marked pixels only:  h = act1(W1 . [x1 ; x2] + b1 + mask[pixel] * bm)
 * (mask (n_pixels) fp16, bm (128) float32; both NULL = di_pointwise_chain_fwd).  Used to fold the output projection of
 * the pillar attention into the encoder layer's out_proj: I2P_feat = valid * (W_ov ctx + b_ov) (encoder_utils.py:
 * 314-319) enters P_out_proj as W1a (W_ov ctx) + valid * (W1a b_ov). */
int di_pointwise_chain_masked_fwd(const void *x1, const void *x2, const void *x3, const void *w1, const float *b1,
                                  const void *w2, const float *b2, void *y, long long n_pixels, int k1, int k2,
                                  int relu1, int relu2, const void *mask, const float *bm, void *stream);

/* ---------------------------------------------------------------- image -> BEV pillar attention
 * MMRI_I2P.forward for ONE sample (encoder_utils.py:270-319), with the single-head
 * attention folded algebraically: the caller passes
 *   qfold[y,x,:] = Wk^T (Wq bev[y,x,:] + bq) / sqrt(C)    (Hb,Wb,C)
 * and receives ctx[y,x,:] = sum_j softmax_j(<qfold, s_j>) s_j over the valid keys
 * s_j = bilinear(img[cam_j], uv_j) of the pillar at cell (y,x) (slot = point*6+cam,
 * :298,:309-310; points >= num_points masked, :303-307) and valid[y,x] = 1 where the
 * pillar has at least one valid key (:314); cells without a pillar or without a valid key are 0 (:259, :314-315).
 *
 * Two calls.  di_i2p_build_keys is the geometry pass (projection of the T*n_views (point, camera) slots of every pillar,
 * mask, compaction) - it depends on the points and the metas only, so ONE call per sample serves every encoder layer.
 * It fills `key_table` (di_i2p_key_table_bytes bytes, device memory owned by the caller): per BEV cell the number of
 * valid keys, the pillar id and the compacted sampling coordinates.
 *   pillars (P,T,D) float32 xyz in the first 3 of D; coors (P,4) int32 [b,z,y,x] (unique cells; rows with
 *   num_points <= 0 are padding and ignored); proj (n_views,4,4) float32 row-major lidar2img; aug_rev 12 floats =
 *   A(3x3),t(3) of the REVERSE augmentation flow as `p' = p @ A + t` (apply_3d_transformation, reverse=True, :280);
 *   ori_H/ori_W = img_metas['input_shape'] (:288-290). */
long long di_i2p_key_table_bytes(int Hb, int Wb, int T, int n_views);
int di_i2p_build_keys(const float *pillars, const int32_t *coors, const int32_t *num_points, const float *proj,
                      const float *aug_rev, void *key_table, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                      int Wb, float ori_H, float ori_W, void *stream);
/* The attention pass: writes EVERY cell of ctx (Hb,Wb,C) and valid (Hb,Wb) (no zero fill needed).  dropout_p > 0 is the
 * training form: attention dropout on the probabilities (nn.MultiheadAttention(dropout=0.1), encoder_utils.py:223-224):
 * key (pillar, slot) is dropped with probability dropout_p, decided by a counter-based hash of (seed, pillar, slot) that
 * the backward regenerates; dropped keys stay in the softmax denominator.
 * cell_order: Hb*Wb int32, a permutation of the cells (or NULL = row-major): the order in which they are walked - the
 * host passes the cells sorted by azimuth so that each XCD gathers from one sector's image columns (results do not
 * depend on it). */
int di_i2p_attn_fwd(const void *img, const void *qfold, const void *key_table, const int32_t *cell_order, void *ctx,
                    void *valid, int T, int n_views, int Hi, int Wi, int Hb, int Wb, int C, float dropout_p,
                    unsigned long long seed, int dtype, void *stream);
/* The same, also writing `mass` (Hb*Wb, the maps' element type; may be NULL = di_i2p_attn_fwd): the kept probability mass
 * sum_j d_j p_j of every cell under attention dropout (1 for a non-empty cell without dropout, 0 for an empty one).
 * nn.MultiheadAttention (encoder_utils.py:257-320 -> torch MHA dropout on the probabilities) adds the value bias to every
 * key's value, so the folded bias Wo.bv enters the output scaled by this mass. */
int di_i2p_attn_fwd_mass(const void *img, const void *qfold, const void *key_table, const int32_t *cell_order, void *ctx,
                         void *valid, void *mass, int T, int n_views, int Hi, int Wi, int Hb, int Wb, int C, float dropout_p,
                         unsigned long long seed, int dtype, void *stream);
/* The attention pass ON THE MATRIX CORES (round 6; csrc/i2p_dense.hip) for fp16 maps of C = 128 channels without attention
 * dropout - the benched inference configuration; same inputs, same outputs (every cell of ctx / valid written) as
 * di_i2p_attn_fwd within fp16 round-off (the probabilities times the bilinear weights enter the value product as fp16).
 * It reads the keys from a DENSE STREAM: di_i2p_compact_keys (once per sample, after di_i2p_build_keys; shared by all
 * encoder layers) packs the keys of every group of 8 consecutive cells of the walk order `cell_order` (NULL = row-major;
 * the SAME order must be passed to both calls) one behind the other, with explicit corner offsets, into `dense_table`
 * (di_i2p_dense_bytes bytes, caller-owned; P = the number of pillars the key table was built from, an upper bound is fine). */
long long di_i2p_dense_bytes(int Hb, int Wb, int T, int n_views, int P);
int di_i2p_compact_keys(const void *key_table, const int32_t *cell_order, void *dense_table, int T, int n_views, int Wi, int Hb,
                        int Wb, void *stream);
int di_i2p_attn_dense_fwd(const void *img, const void *qfold, const void *key_table, const void *dense_table,
                          const int32_t *cell_order, void *ctx, void *valid, int n_views, int Hi, int Wi, int Hb, int Wb,
                          void *stream);
/* Backward of the above: grad_ctx (Hb,Wb,C) -> grad_img (n_views,Hi,Wi,C) and grad_qfold (Hb,Wb,C), both
 * float32, zero-filled by the caller (grad_img is accumulated with atomics).  The sampling coordinates carry
 * no gradient (points and metas are data). */
int di_i2p_attn_bwd(const void *img, const void *qfold, const void *grad_ctx, const float *pillars,
                    const int32_t *coors, const int32_t *num_points, const float *proj, const float *aug_rev,
                    float *grad_img, float *grad_qfold, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                    int Wb, int C, float ori_H, float ori_W, float dropout_p, unsigned long long seed, int dtype,
                    void *stream);
/* Training under hipGraph replay: a captured launch has its `seed` argument baked in.  With a device word registered here
 * (NULL to clear) every di_i2p_attn_fwd* / di_i2p_attn_bwd* launch ADDS *dev_ptr to its seed when it RUNS; the caller rewrites
 * the word before each replay (forward and backward of one step then see the same value).  Process-wide state: register it
 * for the duration of the capture only and clear it right after (`train_step.GraphedTrainer` does) - the pointer a launch
 * read while it was captured stays baked into that graph, so the word must outlive the graph, and launches issued after the
 * clear (another trainer, inference) are not affected. */
int di_i2p_set_seed_ptr(const void *dev_ptr);
/* The same with the gradient of the kept mass (Hb*Wb, the maps' element type; NULL = di_i2p_attn_bwd). */
int di_i2p_attn_bwd_mass(const void *img, const void *qfold, const void *grad_ctx, const void *grad_mass, const float *pillars,
                         const int32_t *coors, const int32_t *num_points, const float *proj, const float *aug_rev,
                         float *grad_img, float *grad_qfold, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                         int Wb, int C, float ori_H, float ori_W, float dropout_p, unsigned long long seed, int dtype,
                         void *stream);

/* ---------------------------------------------------------------- BEV -> image gather
 * BEVWarp.forward (encoder_utils.py:142-199) in three steps per sample.
 * (1) project all raw points and scatter camera-z into the feature-resolution sparse
 *     depth maps (:155-174).  INT scatter; duplicate pixels: HIGHEST POINT INDEX WINS
 *     (deterministic form of the reference's last-writer-wins).  `packed` is
 *     (n_views,Hi,Wi) uint64 scratch, zero-filled by the caller; `depth` receives float32. */
int di_depth_scatter(const float *pts, int n_pts, int pt_stride, const float *proj,
                     const float *aug_rev, unsigned long long *packed, float *depth, int n_views,
                     int Hi, int Wi, float ori_H, float ori_W, void *stream);
/* (2) ip_basic fill_in_multiscale(extrapolate=False, blur='bilateral') per view
 *     (depth_map_utils.py:134-287) on the GPU.  scratch: 3*n_views*Hi*Wi + 2*n_views*ceil(Hi*Wi/256) floats;
 *     iscratch: 2*n_views*Wi int32 (first valid row per column of two intermediate maps). */
int di_depth_complete(const float *sparse, float *dense, float *scratch, int32_t *iscratch,
                      int n_views, int Hi, int Wi, void *stream);
/* (3) un-project every feature pixel through its depth, mask by pc_range, bilinear-gather
 *     the BEV map (:183-196).  xs (Wi), ys (Hi): the linspace pixel grids of :183-184;
 *     img2lidar (n_views,4,4) row-major; aug_fwd: forward flow as p' = p @ A + t (:189);
 *     pc_range 6 floats (:190).  out (n_views,Hi,Wi,C). */
int di_bevwarp_gather_fwd(const void *bev, const float *depth, const float *img2lidar,
                          const float *aug_fwd, const float *xs, const float *ys,
                          const float *pc_range, void *out, int n_views, int Hi, int Wi, int Hb,
                          int Wb, int C, int dtype, void *stream);

/* Backward of (3): grad_out (n_views,Hi,Wi,C) -> grad_bev (Hb,Wb,C) float32, zero-filled by the caller,
 * accumulated with atomics (the autograd of F.grid_sample at encoder_utils.py:195 w.r.t. its input). */
int di_bevwarp_gather_bwd(const void *grad_out, const float *depth, const float *img2lidar, const float *aug_fwd,
                          const float *xs, const float *ys, const float *pc_range, float *grad_bev, int n_views,
                          int Hi, int Wi, int Hb, int Wb, int C, int dtype, void *stream);

/* ---------------------------------------------------------------- MMPI decoder
 * (1) query initialisation heat map (deepinteraction_decoder.py:225-238):
 *     out = h * (h == local_max(h)),  h = (sigmoid(a) + sigmoid(b)) / 2, float32 (B,Cc,H,W).
 *     a, b: the two dense heat-map logits, (B,Cc,H,W) CONTIGUOUS (NCHW, Cc is 10).  Classes
 *     whose bit is set in k1_class_mask use kernel 1 (:232-237); the others use `nms_kernel`
 *     with the outermost pad ring forced to 0 (local_max is filled in the interior only). */
/* Top-k of non-negative float32 scores per row (the proposal pick of deepinteraction_decoder.py:242, where the
 * reference arg-sorts all 324 000 scores): indices (B,k) int64 in the order "value descending, lower index first on
 * ties" (deterministic; the reference leaves ties unspecified), optionally the values (may be NULL).  Radix select on
 * a composite (value, index) key + a k-element sort (4 launches); k <= 1024, N <= 2^20; `workspace` =
 * di_topk_workspace_bytes(B, N, k) bytes of device memory, 64-byte aligned.  No host synchronisation. */
long long di_topk_workspace_bytes(int B, int N, int k);
int di_topk_fwd(const float *scores, long long *out_idx, float *out_val, void *workspace, int B, int N, int k,
                void *stream);
int di_heatmap_nms(const void *a, const void *b, float *out, int B, int num_classes, int H, int W,
                   int nms_kernel, unsigned k1_class_mask, int dtype, void *stream);

/* (2) per-query geometry for the RoI blocks (decoder_utils.py:666-738, :804-819;
 *     TransFusionBBoxCoder.decode, transfusion_bbox_coder.py:57-70; LiDARInstance3DBoxes.corners).
 *     center (B,2,Q) in BEV cells, height (B,1,Q), dim (B,3,Q) log-size, rot (B,2,Q) sin/cos,
 *     all float32.  proj (B,V,4,4), aug_rev (B,12), per_sample (B,6) = [input_w, input_h,
 *     flip, orig_w, crop_x, crop_y].  Outputs: on_img (B,V,Q) int32 (centre strictly inside),
 *     rect_img (B,V,Q,4) and rect_bev (B,Q,4) xyxy (input pixels / BEV cells).  Either output
 *     group may be NULL.  dim_scale: 1 for the image block, 2 for the point block (:807). */
int di_query_geometry(const float *center, const float *height, const float *dim, const float *rot,
                      const float *proj, const float *aug_rev, const float *per_sample, int32_t *on_img,
                      float *rect_img, float *rect_bev, int B, int Q, int n_views, float cell, float pc_x0,
                      float pc_y0, float bev_cell, float dim_scale, void *stream);
/* The same on column windows of wider tensors: the inputs are (B,k,ld) with this stage's Q queries at the given
 * pointers (the decoder writes every stage into the concatenated (B,k,L*Q) outputs, deepinteraction_decoder.py:304). */
int di_query_geometry_ld(const float *center, const float *height, const float *dim, const float *rot,
                         const float *proj, const float *aug_rev, const float *per_sample, int32_t *on_img,
                         float *rect_img, float *rect_bev, int B, int Q, int ld, int n_views, float cell, float pc_x0,
                         float pc_y0, float bev_cell, float dim_scale, void *stream);

/* (3) detectron2 ROIAlign(output 7x7, sampling_ratio 2, aligned=True) (decoder_utils.py:641-646,
 *     739-741, 769-774, 822-823).  feat (N,H,W,C) channels-last; rois (R,5) float32 =
 *     [map index, x0, y0, x1, y1]; out (R,49,C) (bin-major: the (49,q,128) operand of
 *     DynamicConv without the reference's flatten/permute). */
int di_roi_align_fwd(const void *feat, const float *rois, void *out, int R, int N, int H, int W, int C,
                     float spatial_scale, int dtype, void *stream);
/* ... with the output type chosen separately: out_dtype = dtype, DI_F32, or DI_F16_HL = (R,49,2C) fp16 rows [hi C | lo C]
 * with float32 accuracy (fp16 map, C = 128: the matrix operand of di_dynconv_fwd). */
int di_roi_align_x_fwd(const void *feat, const float *rois, void *out, int R, int N, int H, int W, int C,
                       float spatial_scale, int dtype, int out_dtype, void *stream);

/* Backward of (3) w.r.t. the feature maps: grad_out (R,49,C) -> grad_feat (N,H,W,C) float32, zero-filled by
 * the caller, accumulated with atomics (the boxes carry no gradient: they come from detached predictions,
 * decoder_utils.py:672-679). */
int di_roi_align_bwd(const void *grad_out, const float *rois, float *grad_feat, int R, int N, int H, int W, int C,
                     float spatial_scale, int dtype, void *stream);

/* (4) multi-head attention core of the decoder layer's 200 x 32400 cross attention
 *     (decoder_utils.py:101-103, :471-485): out = softmax(q k^T * scale) v per head, head_dim 16.
 *     q (B,Q,E), kv (B,S,2E) = [K | V] (already projected), out (B,Q,E);
 *     scratch: di_mha_decode_scratch_floats(B,Q,S,num_heads) floats. */
int di_mha_decode_scratch_floats(int B, int Q, int S, int num_heads);
int di_mha_decode_fwd(const void *q, const void *kv, void *out, float *scratch, int B, int Q, int S,
                      int num_heads, int head_dim, float scale, int dtype, void *stream);

/* (4x) the same cross attention with float32-accurate logits (round 3; the logits of this attention reach |s| ~ 500,
 *     so K and q in fp16 alone break the 1e-3 contract - DESIGN.md "Numerics"):
 *   di_kv_project_fwd: K = Wk x + kbias, V = Wv x + vbias for the B*S tokens x (fp16, 128 channels) of the BEV map
 *     (decoder_utils.py:98-100 with `key + key_pos` folded: kbias = Wk kpe + bk is constant, float32 (S,128));
 *     the float32 weight (256,128) (rows [K ; V]) arrives as hi + lo / 2048 fp16 in MFMA fragment order, tile pairs'
 *     rows permuted for 16-B stores (`ops.pack_kv_weight`);
 *     out (B*S, 384) fp16 = [Khi | Klo | V] with K = Khi + Klo / 2048.
 *   di_mha_decode_x_fwd: q (B,Q,128) float32, UNSCALED; kx from di_kv_project_fwd; 8 heads x 16; writes one partial
 *     soft-max state [m (exp2 domain), l, O[16]] per (sample, head, query, key range) to scratch
 *     (B*8*Q*di_mha_decode_x_ranges(B,Q,S)*18 floats); with `out` (B,Q,128) float32 a second launch merges them (one
 *     wavefront per (sample, head, query)), with out == NULL a DI_TOK_COMBINE step of di_token_program does. */
int di_kv_project_fwd(const void *x, const void *w_packed, const float *kbias, const float *vbias, void *out, int B, int S,
                      void *stream);
int di_mha_decode_x_ranges(int B, int Q, int S);
int di_mha_decode_x_fwd(const float *q, const void *kx, float *scratch, float *out, int B, int Q, int S, float scale,
                        void *stream);

/* ---------------------------------------------------------------- DeepInteraction++ operators (row a20)
 * di_ms_deform_attn_fwd: the core of mmcv-full 1.3.18 `MultiScaleDeformableAttention` (CUDA op `ms_deform_attn`,
 *   reached from necks/fusion_transformerv4.py:170-178 and :238), 8 heads x 16 channels, 4 points, 1 or 2 levels,
 *   with `softmax` over the L*P logits and `loc = ref + off / (W_l, H_l)` fused in.
 *   value (bs, sum H_l W_l, 128); offsets: bs*nq rows of (8, L, 4, 2), `off_row_stride` elements apart; logits:
 *   rows of (8, L*4), `logit_row_stride` apart (both may point into one packed GEMM output); ref (1 or bs, nq, L, 2)
 *   float32 in [0,1] (x, y), `ref_shared` = 1 when its batch dim is 1; out (bs, nq, 128).  `level_hw` is a HOST
 *   array of 2*n_levels ints [H_0, W_0, H_1, W_1].
 * di_grid_gather_fwd: out[g, n, :] = bilinear(feat[g / grids_per_feat], grid[g, n]) (+ add[n, :]); zeros padding,
 *   align_corners=False; feat (Bf,H,W,C) channels-last, grid (n_grids, n_points, 2) float32 in [-1,1]; `add` may
 *   be NULL.  The polar ray queries, fusion_transformerv4.py:574-575.
 * di_polar_bev_sample_fwd: fusion_transformerv4.py:581-640 over all cameras: polar (B,V,Wp,R,C) per-camera polar
 *   maps stored ray-major (image column, radius, channel), bev (B,Hb,Wb,C) residual, proj (B,V,4,4) lidar2img,
 *   aug_rev (B,12) [A row-major | t], cam_xy (B,V,2) camera centres, params = [pc_range(6), input_H, input_W, radius_min, n_radius] (device, float32);
 *   out (B,Hb,Wb,C) = mean over seeing cameras of the sampled polar map + bev.
 * di_mha_small_fwd: softmax(q k^T * scale) v per head for n_seq short sequences; q (n_seq,Tq,*), k/v (n_seq,S,*)
 *   with the given row strides (elements) so packed projections can be passed in place; head_dim 16.
 * di_add_layernorm_fwd: out = LayerNorm(x + res) * gamma + beta over the last dim C (res may be NULL): the
 *   `norm(x + dropout(branch))` post-norm steps of the ++ transformer layers (fusion_transformerv4.py:179-181,196-198;
 *   torch nn.TransformerDecoderLayer inside MMRI_I2P_Polar), statistics in float32. */
int di_add_layernorm_fwd(const void *x, const void *res, const void *gamma, const void *beta, void *out,
                         long long n_tokens, int C, float eps, int dtype, void *stream);
int di_ms_deform_attn_fwd(const void *value, const void *offsets, int off_row_stride, const void *logits,
                          int logit_row_stride, const float *ref, int ref_shared, void *out, int bs, int nq,
                          int n_levels, int n_points, const int32_t *level_hw, int dtype, void *stream);
/* The same over a HEAD-MAJOR fp16 value map (bs, 8, sum H_l W_l, 16) - inference form, round 5.  The channels-last kernel
 * is bound by the texture addresser (one 32-byte piece per clock: a head's 16 channels of one texel); head-major, the two
 * corners of a footprint row are 64 contiguous bytes fetched by four lanes as one piece.  The value projection writes that
 * layout directly:
 *   di_pointwise_chain_hm_fwd: y_hm[(b * 8 + h) * T + t][16] = (x @ w^T + b)[b * T + t][16 h .. 16 h + 15]  over n_tokens =
 *     bs * T fp16 tokens of 128 channels (w (128,128) fp16, b (128) float32; the arithmetic of di_pointwise_chain_fwd);
 *   di_pointwise_multi_warp_hm_fwd: di_pointwise_multi_warp_fwd with a HOST array `hm_tokens` (per chain: tokens per map
 *     for a head-major output, 0 for channels-last) - the ++ P2I block's value projection gathers the warped BEV map
 *     itself AND writes head-major (necks/fusion_transformerv4.py:228-240). */
int di_ms_deform_attn_hm_fwd(const void *value_hm, const void *offsets, int off_row_stride, const void *logits,
                             int logit_row_stride, const float *ref, int ref_shared, void *out, int bs, int nq,
                             int n_levels, int n_points, const int32_t *level_hw, void *stream);
int di_pointwise_chain_hm_fwd(const void *x, const void *w, const float *b, void *y_hm, long long n_tokens,
                              int tokens_per_map, int relu, void *stream);
int di_pointwise_multi_warp_hm_fwd(const void *bev, const float *depth, const float *img2lidar, const float *aug_fwd,
                                   const float *xs, const float *ys, const float *pc_range, int n_views, int Hi, int Wi,
                                   int Hb, int Wb, int n_chains, const void *const *image_host, void *const *y_host,
                                   const int *relu1_host, const int *relu2_host, const int *two_links_host,
                                   const int *hm_tokens_host, void *stream);
int di_grid_gather_fwd(const void *feat, const float *grid, const void *add, void *out, int n_grids, int n_points,
                       int grids_per_feat, int H, int W, int C, int dtype, void *stream);
int di_polar_bev_sample_fwd(const void *polar, const void *bev, const float *proj, const float *aug_rev,
                            const float *cam_xy, const float *params, void *out, int B, int V, int R, int Wp, int Hb,
                            int Wb, int C, int dtype, void *stream);
int di_mha_small_fwd(const void *q, int q_row_stride, const void *k, const void *v, int kv_row_stride, void *out,
                     int out_row_stride, int n_seq, int Tq, int S, int num_heads, int head_dim, float scale, int dtype,
                     void *stream);

/* The "self" branch of the ++ head's V2 RoI blocks (models/utils/decoder_utils.py:970-990 image, :1086-1089 point) in one
 * launch, float32: as published, every query of a group receives the self-branch feature of the group's FIRST query, so the
 * branch works on one token per (sample, view) [image = 1: that view's first query attends to the view's queries - q / k rows
 * `qk` (B*Q, 256) and transposed values `vt` (B, 128, Qp) of the block's packed projection -, output projection `wo`, `bo`,
 * residual with `x`, norm1] or one per sample [image = 0: `x` = norm1(x + attention), query 0], then the self FFN
 * (sw1 (hidden,128), sw2 (128,hidden), ReLU) + residual + LayerNorm (snw, snb) times self_scale[0]; out (B*Q, 128) receives,
 * per query, the feature of its last view (`view` int8, -1 -> view 0; `member` uint8 bit v = seen by view v). */
int di_v2_self_feature(const float *qk, const float *vt, const float *x, const signed char *view,
                       const unsigned char *member, const float *wo, const float *bo, const float *n1w, const float *n1b,
                       float eps1, float scale, const float *sw1, const float *sb1, const float *sw2, const float *sb2,
                       const float *snw, const float *snb, float eps_s, const float *self_scale, float *out, int B, int Q,
                       int Qp, int V, int hidden, int image, void *stream);

/* Backward of the ++ samplers (training).  Gradient maps are float32, zero-filled by the caller, accumulated with
 * atomics; sampling geometry carries no gradient (detached in the reference too).
 * di_ms_deform_attn_bwd: grad_out (bs,nq,128) -> grad_value (bs,S,128) float32 and grad_proj: bs*nq rows of
 *   [d offsets (8,L,4,2) | d logits (8,L*4)] (through the fused softmax), `grad_proj_row_stride` elements apart,
 *   same dtype as the inputs.  mmcv's `ms_deform_attn_backward`.
 * di_grid_gather_bwd: grad_feat (Bf,H,W,C) float32 += bilinear weights * grad_out (n_grids,n_points,C).
 * di_polar_bev_sample_bwd: grad_polar (B,V,Wp,R,C) float32 += weights / n_seeing_cameras * grad_out (B,Hb,Wb,C);
 *   the residual's gradient is grad_out itself. */
int di_ms_deform_attn_bwd(const void *value, const void *offsets, int off_row_stride, const void *logits,
                          int logit_row_stride, const float *ref, int ref_shared, const void *grad_out,
                          float *grad_value, void *grad_proj, int grad_proj_row_stride, int bs, int nq, int n_levels,
                          int n_points, const int32_t *level_hw, int dtype, void *stream);
int di_grid_gather_bwd(const float *grid, const void *grad_out, float *grad_feat, int n_grids, int n_points,
                       int grids_per_feat, int H, int W, int C, int dtype, void *stream);
int di_polar_bev_sample_bwd(const void *grad_out, const float *proj, const float *aug_rev, const float *cam_xy,
                            const float *params, float *grad_polar, int B, int V, int R, int Wp, int Hb, int Wb,
                            int C, int dtype, void *stream);

/* Several 128-channel chains over ONE input map in one launch (the query / key / value projections of a
 * LocalContextAttentionBlock and of the P2I block read the same map, encoder_utils.py:92-117,127-131): chain c is
 * y[c] = act2(W2[c] . act1(W1[c] . x + b1[c]) + b2[c]) (second link optional).  x (n_pixels,128) fp16 is read once.
 * `image`, `y`, `relu1`, `relu2`, `two_links` are HOST arrays of n_chains <= 4 entries; image[c] is the DEVICE pointer
 * of the chain's 66 560-byte LDS image, prepared once by the caller: W1 then W2 (zeros when absent), each 128 rows x
 * 256 B with 16-B chunk c of row r at position c ^ (r & 15), W2's columns k-permuted (chunk 4kk+g = columns
 * 32kk+4g..+3 | 32kk+16+4g..+3), the rows (and bias entries) of the chain's LAST link permuted so that image row
 * 16nb+4g+r is output channel 32(nb/2)+8g+4(nb%2)+r, then b1, b2 as 128 float32 each.  Same arithmetic as
 * di_pointwise_chain_fwd. */
int di_pointwise_multi_fwd(const void *x, int n_chains, const void *const *image_host, void *const *y_host,
                           const int *relu1_host, const int *relu2_host, const int *two_links_host, long long n_pixels,
                           void *stream);
/* The same chains over the BEV map WARPED onto the image maps (MMRI_P2I: BEVWarp, encoder_utils.py:185-196, then the key /
 * value projections of its LocalContextAttentionBlock, :127-131) without the warped map in memory: input pixel p of the
 * (n_views, Hi, Wi) maps is the bilinear sample of bev (Hb, Wb, 128) fp16 where the pixel's completed depth un-projects to -
 * the arithmetic and fp16 rounding of di_bevwarp_gather_fwd (whose arguments these are), so the outputs equal
 * di_pointwise_multi_fwd on that kernel's output bit for bit.  n_chains <= 2: both weight images stay resident in LDS and
 * the pixel groups go through the chains a pair at a time. */
int di_pointwise_multi_warp_fwd(const void *bev, const float *depth, const float *img2lidar, const float *aug_fwd,
                                const float *xs, const float *ys, const float *pc_range, int n_views, int Hi, int Wi, int Hb,
                                int Wb, int n_chains, const void *const *image_host, void *const *y_host,
                                const int *relu1_host, const int *relu2_host, const int *two_links_host, void *stream);

/* Transformer FFN + post-norm of the DeepInteraction++ layers (mmcv FFN followed by LayerNorm: necks/fusion_transformerv4.py
 * operation_order (..., 'ffn', 'norm')) in one pass over the tokens:
 *     y = LayerNorm(x + W2 . relu(W1 . x + b1) + b2)         x, y (n_tokens, 128) fp16; hidden width 128 * n_chunks
 * image: HOST array of n_chunks <= 8 DEVICE pointers, chunk c = the chain image (layout as di_pointwise_multi_fwd) of the
 * two-link chain (W1[128c : 128c+128, :], b1[128c : ..], W2[:, 128c : 128c+128], b2 in chunk 0 / zeros elsewhere);
 * ln_w, ln_b (128) fp16.  The (n_tokens, hidden) activation never exists in memory; fp32 accumulation throughout.
 * residual != NULL selects the single-link form  y = LayerNorm(residual + W . x + b)  (output projection of an attention
 * block + its post-norm): n_chunks = 1, image[0] = the one-link chain image of (W, b). */
int di_ffn_ln_fwd(const void *x, int n_chunks, const void *const *image, const void *residual, const void *ln_w,
                  const void *ln_b, float eps, void *y, long long n_tokens, void *stream);
/* The same with a second output: presum (n_tokens, 128) fp16 = the sum BEFORE the normalisation (NULL = di_ffn_ln_fwd).  A
 * DeepInteraction++ layer keeps the un-normalised self-attention output (`self_feat`, necks/fusion_transformerv4.py:187-190)
 * next to its LayerNorm: output projection + residual + both results are then one launch. */
int di_ffn_ln_fwd_ex(const void *x, int n_chunks, const void *const *image, const void *residual, const void *ln_w,
                     const void *ln_b, float eps, void *y, void *presum, long long n_tokens, void *stream);

/* ---------------------------------------------------------------- 3x3 convolutions (stride 1, pad 1), implicit GEMM
 * The shared convolutions of the MMRI encoder (necks/deepinteraction_encoder.py:45-62) and the heat-map heads of the
 * decoder (dense_heads/deepinteraction_decoder.py:96-119) on fp16 channels-last maps:
 *   y[p][n] = act(sum_{ky,kx,c} w[n][ky*3+kx][c] * x[p + (ky-1, kx-1)][c] + bias[n])
 * x (n,H,W,Cin) fp16, Cin % 32 == 0; w_packed (Cout_pad, 9, Cin) fp16 = the torch weight (Cout,Cin,3,3) permuted to
 * (Cout,3,3,Cin), rows zero-padded to 16 when Cout <= 16; bias float32 (Cout) (a following BatchNorm folded in by the
 * caller); Cout == 128 or Cout <= 16; y (n,H,W,Cout) fp16, or (n,Cout,H,W) when out_nchw
 * (1: fp16, 2: float32 - the heat-map logits that feed the NMS comparison and the top-Q pick).
 * w_staged (optional, Cout == 128): the same weights as (Cin/32, 3 ky, 3 kx, 128, 32) fp16 with the 128 rows
 * permuted (row 16nb+4g+r = output channel 32(nb/2)+8g+4(nb%2)+r, for 16-B stores) and, inside every (chunk, ky) tile of
 * 384 rows x 64 B, the 16-B slot s of row R stored at slot (s + 2 * (R >> 2)) & 3 - the kernel's conflict-free LDS order,
 * so that a tile moves by LDS-DMA / linear copies.  With it the kernels that stage the weights through LDS run (16/12/20-row
 * tiles with an LDS-DMA weight ring for maps of >= 12 rows, the fast path); NULL selects the weights-from-L2 kernel.
 * `ops.pack_conv3x3` builds all three tensors from a torch Conv2d (+ BatchNorm2d). */
int di_conv3x3_fwd(const void *x, const void *w_packed, const void *w_staged, const float *bias, void *y, int n, int H,
                   int W, int Cin, int Cout, int relu, int out_nchw, void *stream);

/* ---------------------------------------------------------------- token-level kernels of the MMPI decoder (float32)
 * Inference form of the reference's decoder layer / RoI blocks / prediction heads on the B*Q query tokens
 * (models/utils/decoder_utils.py:35-113, 498-581, 584-629, 632-841; dense_heads/deepinteraction_decoder.py:242-313).
 * The token state, the RoI features, the generated DynamicConv parameters and every weight of this path carry FLOAT32
 * accuracy: with fp16 anywhere on this path the box outputs leave the 1e-3 contract (DESIGN.md "Numerics"); only the
 * feature MAPS the tokens gather from are fp16.  Matrix operands travel as SPLIT fp16 pairs, x = hi + lo / 2048
 * (hi = fp16(x), lo = fp16((x - hi) * 2048)): a product is three fp16 MFMAs with float32 accumulation, 2^-22 relative.
 * Weights are split once on the host (`*_hi`, `*_lo`, (N, K) row-major as `nn.Linear.weight`); token matrices are
 * float32, row-major with an explicit row stride (`ld*`, in elements), split in registers.
 *
 * di_token_program: ONE launch runs a short program on every group of 16 consecutive tokens of a sample (grid
 *   ceil(Q/16) x B x roles; the rows live in three LDS buffers of 16 x 512 floats, `src` / `dst` / `aux` name them).
 *   ROLES spread one token group's weights over several CUs: the workgroup of role r executes the steps whose
 *   [role_lo, role_hi] contains r; a LINEAR step takes the weight tiles r*rt.. / K-chunks r*rc.. (an N-split - packed
 *   projections, one prediction head per role - or a hidden-dimension split whose partial sums the next launch adds),
 *   LOAD / STORE / LOAD_PARTS add r*roff to p0, HEADS with a = 1 evaluates head r only (hidden at columns 0..63).
 *     DI_TOK_LOAD        dst[:, a:a+K] = p0[m, :K] (+ p1[m, :K])                       rows of ld0 (ld1) floats
 *     DI_TOK_LOAD_PARTS  dst[:, :128]  = sum_{s<a} p0[(s*b + m)*128 : +128] + p1       split-K partial sums (b = B*Q)
 *     DI_TOK_ATTN        dst[:, :128]  = softmax(q k^T) v per head (8 heads x 16) among the Q tokens of the sample, from
 *                        rows p0 = [q | k] (ld0 floats) and the TRANSPOSED values p3 = V^T (B, 128, b) float32 (b = Q
 *                        rounded up to 16), f = scale * log2(e); optional visibility p1 = member (uint8), p2 = view
 *                        (int8): key k is visible to query q when bit view[q] of member[k] is set or view[q] < 0
 *                        (ImageRCNNBlock's per-view attention, :745)
 *     DI_TOK_COMBINE     dst[:, :128]  = merge of the a key-range states of di_mha_decode_x_fwd (p0 = scratch)
 *     DI_TOK_LINEAR      dst[:, :N]    = act_a(src[:, :K] . W^T + p1), W (hi + lo / 2048) packed in MFMA fragment order
 *                        (blocks (N/16, K/128) of 8 x 1 KiB: `ops.pack_linear`); a: 0 none, 1 ReLU, 2 GELU(erf);
 *                        K multiple of 128, N of 16, both <= 512
 *     DI_TOK_ROWOP       dst[:, :128]  = mask_p2(relu_{b&1}(LayerNorm_{p0,p1,eps=f}(src + buf[aux])))   each part optional
 *     DI_TOK_STORE       p0[m, :N]     = src[:, a:a+N];  b = 1: transposed, p0[(sample*N + c)*ld0 + q] (the V^T above);
 *                        b = 2: split, p0 fp16 rows [hi N | lo N] (the token operand of di_token_wide)
 *     DI_TOK_HEADS       second layers of the prediction heads on the hidden rows in src (first layers: a LINEAR step
 *                        with the BatchNorm-folded, stacked (nheads*64, K) weight), `center += query_pos`, the
 *                        on-the-image merge with the first stage (`keep`), written at column col0 of the
 *                        (B, cls_h, ldo) float32 outputs; pos_out = the new centres (B,Q,2)   (:498-581, head :265-311)
 * DynamicConv (:608-624) as three kernels whose operands are laid out for one contiguous KiB per wave-level load
 * (layouts: csrc/token32.hip, host side `decoder_fused._dyn_layout` / `ops.pack_linear` / `ops.pack_ksteps`):
 * di_token_wide: params (M, 65536) fp16 = the generator Linear 128 -> 2*128*128 of the tokens x, given SPLIT as x_hl (M, 256)
 *   fp16 = [hi 128 | lo 128] (a DI_TOK_STORE step with b = 2 writes that), weight stationary,
 *   written as hi / lo fragments in the order di_dynconv_fwd reads them; w_packed = the generator's weight, rows
 *   permuted to that order, in MFMA fragment order; bias (32768) in value order.
 * di_dynconv_fwd: F2 = relu(LN2(relu(LN1(roi . p1)) . p2)) per RoI (:617-622); roi_hl (R,49,256) fp16 = [hi | lo]
 *   (di_roi_align_x_fwd with DI_F16_HL); F2 leaves as
 *   f2p (196, R, 64) fp16 = [k-step of the flattened (49*128) feature][RoI][hi 32 | lo 32].
 * di_token_splitk: partial sums of out_layer (:624) over 14 K slices into workspace (slices, M, 128) float32 from f2p
 *   and the weight in k-step order; a DI_TOK_LOAD_PARTS step of the next program sums them.
 * di_roi_select: image block (on != null): last valid view per query, membership bits, RoIs, keep mask, float view
 *   id (:681-759 bookkeeping); point block (on == null): rois = (b, BEV rect).
 * di_query_init: query features = BEV token (fp16 map) + class encoding, positions, learned positional embedding
 *   (BatchNorm folded) and labels of the top-Q proposals (deepinteraction_decoder.py:242-253); float32 out. */
enum { DI_TOK_LOAD = 1, DI_TOK_LOAD_PARTS = 2, DI_TOK_ATTN = 3, DI_TOK_COMBINE = 4, DI_TOK_LINEAR = 5,
       DI_TOK_ROWOP = 6, DI_TOK_STORE = 7, DI_TOK_HEADS = 8 };
#define DI_TOK_MAX_STEPS 20
#define DI_TOK_MAX_HEADS 8
typedef struct di_tok_step {
  int kind, src, dst, aux;
  int K, N, a, b;
  float f;
  int role_lo, role_hi;     /* the step runs in the workgroups whose role (grid z) lies in [role_lo, role_hi] */
  int rt, rc, nch;          /* LINEAR: tiles / K-chunks added per role to the packed weight's block index; chunks of the weight (0: K/128) */
  const void *p0, *p1, *p2, *p3;
  long long ld0, ld1;
  long long roff;           /* LOAD / LOAD_PARTS / STORE: elements added to p0 per role */
} di_tok_step;
typedef struct di_tok_heads {
  const float *w2, *b2, *qpos;     /* stacked second layers (sum cls, 64), their biases, query positions (B,Q,2) */
  const unsigned char *keep;       /* (B,Q) or NULL */
  float *pos_out;                  /* (B,Q,2) or NULL */
  float *out[DI_TOK_MAX_HEADS];    /* (B, cls_h, ldo) each */
  const float *first[DI_TOK_MAX_HEADS];   /* (B, cls_h, Q) of the first stage (needed with keep) */
  int cls[DI_TOK_MAX_HEADS];
  int nheads, center_head, ldo, col0;
  const float *qpos2;              /* ABI 2 - DeepInteraction++ look-forward (deepinteractionplusplus_decoder.py:291-294): with */
  float *pos2_out;                 /* pos2_out (B,Q,2) != NULL, pos2_out = raw centre offset + qpos2 (B,Q,2); else both NULL */
} di_tok_heads;
int di_token_program(const di_tok_step *steps_host, int nsteps, const di_tok_heads *heads_host, int B, int Q,
                     void *stream);
/* profiling aid: the same launch; workgroup (0,0) writes the shader clock at kernel start and after every step's
 * barrier to stamps[0 .. nsteps] (device memory). */
int di_token_program_timed(const di_tok_step *steps_host, int nsteps, const di_tok_heads *heads_host, int B, int Q,
                           unsigned long long *stamps, void *stream);
int di_token_wide(const void *x_hl, int ldx, const void *w_packed, const float *bias, void *params, int M, void *stream);
long long di_token_splitk_workspace_bytes(int M, int K);
int di_token_splitk(const void *f2p, const void *w_packed, float *workspace, int M, int K, int *nslices_host, void *stream);
int di_dynconv_fwd(const void *roi_hl, const void *params, const float *n1w, const float *n1b, const float *n2w,
                   const float *n2b, void *f2p, int R, float eps, void *stream);
int di_roi_select(const int *on, const float *rect, float *rois, void *view, void *member, void *keep, float *on_img,
                  int B, int V, int Q, void *stream);
int di_query_init(const void *bev, const long long *top, const float *ce_w, const float *ce_b, const float *w1,
                  const float *b1, const float *w2, const float *b2, float *feat, float *pe, float *pos,
                  long long *labels, int B, int Q, int Hb, int Wb, int ncls, void *stream);

/* ---------------------------------------------------------------- pillar / voxel producer
 * Hard voxelisation (spconv PointToVoxel as wrapped by models/updated_modules/sparse_voxelize.py:9-70), three
 * kernels around two key sorts the caller runs (any stable 64-bit sort):
 *   di_voxel_keys     keys[i] = voxel_id << 32 | i, INT64_MAX for points outside `geo` = [range(6), vsize(3)]
 *   (sort keys)       -> sorted_keys
 *   di_voxel_heads    head[i] = 1 at the first point of every voxel; first_key[i] = first point index << 32 | i
 *   (seg_id = cumsum(head) - 1; sort first_key) -> sorted_first
 *   di_voxel_scatter  voxels (max_voxels, max_points, n_feat) float32, coords (max_voxels, 3) int32 [z,y,x],
 *                     num_points (max_voxels,) int32, all zero-filled by the caller; slot_of_seg / head_of_seg:
 *                     n_pts int32 of scratch.  Voxels in the order of their first point, the first max_points
 *                     points of a voxel in point order (the CPU semantics; spconv's GPU order is unspecified). */
int di_voxel_keys(const float *pts, int n_pts, int pt_stride, const float *geo, int gx, int gy, int gz,
                  long long *keys, void *stream);
int di_voxel_heads(const long long *sorted_keys, int n_pts, int32_t *head, long long *first_key, void *stream);
int di_voxel_scatter(const float *pts, int n_pts, int pt_stride, int n_feat, const long long *sorted_keys,
                     const long long *sorted_first, const int32_t *seg_id, int32_t *slot_of_seg, int32_t *head_of_seg,
                     int gx, int gy, int max_points, int max_voxels, float *voxels, int32_t *coords,
                     int32_t *num_points, void *stream);

/* Pairwise 3-D IoU of LiDAR boxes, the IoU3DCost / max_overlaps input of the Hungarian assigner of the head loss:
 * `BboxOverlaps3D(coordinate='lidar')(boxes1, boxes2)` as called at core/bbox/assigners/hungarian_assigner.py:127 (mmdet3d 0.17.1
 * iou3d semantics: rotated BEV intersection x height overlap / union, a positive yaw turns the box clockwise in the BEV plane).
 * boxes1 [n, stride1], boxes2 [m, stride2] float32 rows (x, y, z_bottom, dx, dy, dz, yaw, ...), out [n, m] float32. */
int di_iou3d_lidar(const float *boxes1, int n, int stride1, const float *boxes2, int m, int stride2, float *out, void *stream);

/* Training-mode BatchNorm2d (+ ReLU) of the necks' ConvBNReLU blocks (encoder_utils.py:11-34 in train() mode) on a
 * channels-last map seen as (npix, C), C a multiple of 8 and <= 256, fp16 or float32; statistics float32.
 * fwd: y = relu?(gamma * (x - mean) / sqrt(var + eps) + beta) with the batch statistics (biased variance); `saved` [4 C] =
 *      [mean | rstd | gamma | beta] for the backward; running_mean / running_var (momentum update, unbiased variance) and
 *      num_batches (+= 1) are updated in place when given; gamma / beta may be NULL (affine=False).
 * bwd: grad_x (the ReLU mask is recomputed from x), grad_gamma / grad_beta (may be NULL).
 * Three launches per direction: partial sums per workgroup, a fixed-order finalisation (bit-reproducible), the apply pass.
 * `workspace`: di_bn_workspace_floats(C) floats, not shared between launches in flight. */
int di_bn_workspace_floats(int C);
int di_bn_train_fwd(const void *x, long long npix, int C, int dtype, const float *gamma, const float *beta, float eps,
                    float momentum, float *running_mean, float *running_var, long long *num_batches, int relu, void *y,
                    float *saved, float *workspace, void *stream);
int di_bn_train_bwd(const void *x, const void *grad_y, long long npix, int C, int dtype, const float *saved, int relu,
                    void *grad_x, float *grad_gamma, float *grad_beta, float *workspace, void *stream);

/* Weight (and bias) gradient of a 1x1 convolution / pixel-wise Linear of the training step (the backward of every
 * kernel_size-1 `conv` of encoder_utils.py:11-34 ConvBNReLU, of the K/V projection of decoder_utils.py:91-95), float32:
 *     grad_w[co][ci] = sum_p grad_y[p][co] * x[p][ci],   grad_b[co] = sum_p grad_y[p][co]   (grad_b may be NULL)
 * x (npix, Cin), grad_y (npix, Cout) row-major and dense, Cin and Cout multiples of 128.  Float32 matrix cores
 * (v_mfma_f32_16x16x4_f32), pixel slabs summed in a fixed order by a second launch: bit-reproducible.
 * `workspace`: di_wgrad_workspace_floats(npix, Cin, Cout) floats (-1 + di_last_error() on unsupported shapes). */
long long di_wgrad_workspace_floats(long long npix, int Cin, int Cout);
int di_wgrad_f32(const float *x, const float *grad_y, long long npix, int Cin, int Cout, float *grad_w, float *grad_b,
                 float *workspace, void *stream);

/* Sparse 3-D convolutions of the frozen LiDAR middle encoder (csrc/sparse_conv.hip; mmdet3d 0.17.1 `SparseEncoder` over spconv
 * `SubMConv3d` / `SparseConv3d`, called by models/detectors/deepinteraction.py:127 with Fusion_0075_refactor.py:160-171 -
 * spconv is a CUDA-only third-party dependency: these replace its rulebook + gather-GEMM-scatter ops).  A level of the sparse
 * tensor is its SORTED list of linear voxel keys ((b * D + z) * H + y) * W + x (int32: B * D * H * W < 2^31) and a feature matrix
 * with one row per key.  `geo16` (host memory) = [B, inD, inH, inW, outD, outH, outW, kD, kH, kW, sD, sH, sW, pD, pH, pW].
 *   di_sparse_mark      SparseConv3d output set: occ[out key] = 1 (bytes, B*outD*outH*outW, zeroed by the caller) wherever an
 *                       active input voxel lies in the window; the caller's non-zero scan of `occ` is the sorted output key list.
 *   di_sparse_nbr       the rulebook as a neighbour table nbr (K = kD*kH*kW, M_out): row of `in_keys` holding the voxel at
 *                       out * stride - pad + offset (kernel index order kd, kh, kw), or -1.  Submanifold layers pass
 *                       out_keys = in_keys, stride 1, pad (k - 1) / 2 and share the table among the layers of a resolution.
 *                       `rowstart`: the input level's row table from di_sparse_rowstart (the searches then stay inside one row's
 *                       slice of the list), or NULL (whole-list searches).
 *   di_sparse_rowstart  rowstart[r] (B * inD * inH + 1 int32) = first position of `in_keys` with key >= r * inW: once per level,
 *                       shared by the level's submanifold table and the strided table that leaves it.
 *   di_sparse_conv_fwd  out[m, :] = act(sum_o feats[nbr[o, m], :] . W[o] + bias (+ residual[m, :])), fp16 rows, float32
 *                       accumulation on the matrix cores.  feats (M_in + 1, cin), cin a multiple of 8, ROW M_in ALL ZERO (what a
 *                       missing neighbour reads); out (M_out + 1, cout): the kernel writes the zero row M_out, so an output is the
 *                       next layer's input as it is.  `wfrag` = the (K, cin_pad,
 *                       cout) weights, cin zero-padded to cin_pad (a multiple of 32), in MFMA operand order
 *                       [K][cin_pad / 32][cout / 16][lane = 16 g + i][8]: element e = W[o][32 kk + 8 g + e][16 mt + i]
 *                       (ops.sparse_weight_fragments); bias (cout) float32 or NULL, residual (>= M_out rows of cout) or NULL, relu 0 / 1.
 *                       Shapes: (cin_pad, cout) in {32} x {16, 32, 64}, {64} x {64, 128}, {128} x {128}; K <= 27. */
int di_sparse_mark(const int32_t *in_keys, int M_in, const int32_t *geo16, void *occ, void *stream);
int di_sparse_rowstart(const int32_t *in_keys, int M_in, const int32_t *geo16, int32_t *rowstart, void *stream);
int di_sparse_nbr(const int32_t *in_keys, const int32_t *out_keys, int M_in, int M_out, const int32_t *geo16,
                  const int32_t *rowstart, int32_t *nbr, void *stream);
int di_sparse_conv_fwd(const void *feats, const int32_t *nbr, const void *wfrag, const float *bias, const void *residual,
                       void *out, int M_in, int M_out, int K, int cin, int cin_pad, int cout, int relu, void *stream);

/* Epilogue of a LIBRARY convolution of the frozen backbones, in place on a channels-last fp16 map (csrc/epilogue.hip):
 *   y[p][c] = act(y[p][c] + bias[c] (+ residual[p][c])),  y (npix, C), C a multiple of 8, bias float32, residual NULL or (npix, C).
 * One pass where torch's conv2d-with-bias + add_ + relu_ are three (img_backbone / img_neck of Fusion_0075_refactor.py:120-145,
 * called by detectors/deepinteraction.py:100-118). */
int di_bias_act_inplace(void *y, const float *bias, const void *residual, long long npix, int C, int relu, void *stream);
/* FPN top-down step in place (mmdet FPN.forward: `laterals[i - 1] += F.interpolate(laterals[i], size=..., mode='nearest')`, the
 * img_neck of Fusion_0075_refactor.py:137-145): lo (n, Hl, Wl, C) += nearest(hi (n, Hh, Wh, C)), channels-last fp16, C % 8 == 0. */
int di_upsample_add_inplace(void *lo, const void *hi, int n, int Hl, int Wl, int Hh, int Wh, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPINTERACTION_HIP_H */
