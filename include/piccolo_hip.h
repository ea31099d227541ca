/* piccolo_hip.h — C ABI of the MI355X (gfx950) implementation of PICCOLO's sampling-loss hot path.
 *
 * The reference (82magnolia/piccolo) is pure Python/PyTorch and has no FFI layer; its boundary for this path is
 * the Python surface omniloc.py / utils.py.  This header is what a ctypes binding of that surface calls
 * (see INTEGRATION.md for the stub).  Each entry point cites the reference code it replaces
 * (file:line relative to the reference checkout).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host; all buffers are caller-owned, the device entry
 *    points never allocate, free or synchronise (graph-capture safe); `stream` is a hipStream_t passed as void*;
 *    the dataset text reader (pcl_cloud_txt_*) is host-only code and takes host pointers;
 *  - every function returns 0 on success, a hipError_t value (> 0) for a runtime failure, or one of the
 *    negative PCL_E* codes below for a bad argument;
 *  - fp32 throughout, like the reference (localize.py:159-170); poses are (t[3], yaw, pitch, roll) with
 *    p = R (x - t), R = RZ(yaw) RY(pitch) RX(roll)  (utils.py:425-453).
 */
#ifndef PICCOLO_HIP_H
#define PICCOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCL_ABI_VERSION 9 /* 9: pcl_gd_hyper.fuse, pcl_loss_depth_workspace_bytes takes the occluder stride, pcl_trim_order + the `order` argument of pcl_trim_loss[_images], the library reads no environment variable; 8: PCL_PANO_U8V / pcl_pano_pack_u8v; 7: depth mask on its own grid (pcl_gd_hyper.depth_h / depth_w, pcl_depth_default, pcl_sampling_loss_depth), refresh rule and pcl_gd_depth_refresh_counts removed, pcl_gd_step_from_grads; 6: pcl_select_poses, pcl_gd_set_pano_groups, pcl_gd_winner; 2: fp16-level texels, colour preprocessing, histograms, dataset text reader; 3: backward of the stand-alone ops; 4: pcl_hist_trim_workspace_bytes_n; 5: pcl_source_hash, pcl_timer_calibrate, pcl_trim_*, pcl_gd_plan */

#define PCL_EINVAL (-1)   /* bad size / null pointer / unsupported argument */
#define PCL_EWORKSPACE (-2) /* workspace too small */

/* largest cloud of the loss / refinement / ordering entry points: the six planes of the packed cloud are addressed through one
 * 32-bit buffer descriptor (6 x 4 B x n < 4 GiB); a larger n is answered with PCL_EINVAL before anything is enqueued */
#define PCL_MAX_POINTS ((int64_t)1 << 27)

/* number of floats per pose in result blocks: loss, count, dL/dt[3], dL/dyaw, dL/dpitch, dL/droll */
#define PCL_RESULT_STRIDE 8

int pcl_abi_version(void);
const char *pcl_error_string(int code);
/* Which sources the loss kernel of THIS binary was compiled from: the first 16 hex digits of sha256(csrc/pcl_loss.hip +
 * csrc/pcl_sample_device.h + csrc/pcl_gd_device.h + csrc/pcl_device.h), stamped in by piccolo_amd/build.py ("unstamped" for a build that did not pass -DPCL_SOURCE_HASH).
 * Measurement aid: counter-derived figures (VALU instructions per point-pose, profiles/roofs.json) carry the hash of the
 * library they were collected from, and bench.py reports them only for a library with the same hash. */
const char *pcl_source_hash(void);
/* The same over EVERY source of the library (all .hip and .h files of csrc, this header): what the per-kernel roofs of the pipeline kernels
 * (trim, bin / resolve, z pass: profiles/pipeline_roofs.json) are stamped with. */
const char *pcl_library_hash(void);

/* ---- data layout in HBM ------------------------------------------------------------------------------------
 * cloud : 6 planes (x, y, z, -r, -g, -b) of pcl_cloud_stride(n) floats each (colours negated: the loss needs c - rgb) — SoA so that a wavefront's 64 lanes
 *         read 256 contiguous bytes per plane.  Built once per point cloud from the reference's row-major
 *         (N,3) xyz and rgb tensors (localize.py:159-164).  `order` (nullable, int64[n]) gathers
 *         point order[i] into slot i: passing a space-filling-curve order makes consecutive lanes hit
 *         neighbouring texels (the loss is a sum over points, so the order does not change the result beyond
 *         fp32 summation rounding).
 * pano  : the query image (H,W,3) float (localize.py:167-170) repacked as (H+2, W+2) texels with a one-texel zero
 *         border, so grid_sample's zero padding (utils.py:98) needs no bounds test.  Three texel formats:
 *           PCL_PANO_F32 : RGBA float4, 16 B/texel — any float image;
 *           PCL_PANO_U8  : RGBA8, 4 B/texel — for images whose every value is exactly k/255 in fp32, which is what
 *                          the reference always feeds (cv2.imread -> uint8 -> .float() / 255., localize.py:167-170,
 *                          211-213; color_mod also re-quantises to uint8, color_utils.py:48-50).  A 2x2 bilinear
 *                          footprint is then two 8-byte loads and the whole panorama is 4x smaller, so the gathers
 *                          stay in L2.  The kernel interpolates the integer levels and scales by 1/255 once.
 *           PCL_PANO_F16 : RGBA half4, 8 B/texel — the same k/255 images with the levels 0..255 held as fp16 (exact).
 *                          A footprint is two 16-byte loads; the tap differences are exact fp16 and the lerps read
 *                          fp16 operands directly (v_fma_mix_f32), so no tap is converted: same results as PCL_PANO_U8
 *                          bit for bit, loss kernel 4-5 % faster, twice the texture bytes (still L2/MALL resident).
 *         pcl_pano_pack_u8 / _f16 set *not_exact (device int, caller zeroes it) if some value is NOT exactly k/255:
 *         the caller must then fall back to PCL_PANO_F32.
 */
#define PCL_PANO_F32 0
#define PCL_PANO_U8 1
#define PCL_PANO_F16 2
/* PCL_PANO_U8P: RGBA8 with the ROWS INTERLEAVED IN PAIRS — element (x, k) = 8 bytes = texel (x, 2k), texel (x, 2k + 1) of the bordered
 * image, element rows of W + 2 elements.  A 2 x 2 footprint that starts on an even row is ONE 16-byte access, on an odd row two:
 * 1.5 texture accesses per sample on average instead of 2, with the memory of PCL_PANO_U8.  Only the forward-only trim launch takes
 * it (pcl_trim_loss / pcl_trim_loss_images: bound by the texture unit's line rate); pcl_pano_pack_u8p packs it. */
#define PCL_PANO_U8P 3
/* PCL_PANO_U8V: RGBA8 in VERTICAL PAIRS — element (x, y) = 8 bytes = texel (x, y), texel (x, y + 1) of the bordered image, H + 2 element
 * rows of W + 2 elements.  Every 2 x 2 footprint is ONE 16-byte access, for twice the texture bytes of PCL_PANO_U8.  Only the trim
 * launch takes it (dense clouds, where that launch is bound by texture accesses and VALU issue alike); pcl_pano_pack_u8v packs it. */
#define PCL_PANO_U8V 4
int64_t pcl_cloud_stride(int64_t n);
size_t pcl_cloud_bytes(int64_t n);
int pcl_cloud_pack(const float *xyz, const float *rgb, const int64_t *order, int64_t n, float *cloud, void *stream);
/* The Morton order in one call, entirely on the device: bounding box, 63-bit keys, stable radix sort of (key, index);
 * order[i] = index of the point for packed slot i.  workspace: pcl_cloud_order_workspace_bytes(n). */
size_t pcl_cloud_order_workspace_bytes(int64_t n);
int pcl_cloud_order(const float *xyz, int64_t n, int64_t *order, void *workspace, size_t workspace_bytes, void *stream);
/* 63-bit Morton keys of xyz quantised to 21 bits per axis inside [lo, hi] (host arrays of 3); sort them to get `order`. */
int pcl_morton_keys(const float *xyz, int64_t n, const float *lo_host, const float *hi_host, int64_t *keys, void *stream);

size_t pcl_pano_bytes(int H, int W, int pano_format);
int pcl_pano_pack(const float *img_hwc, int H, int W, float *pano, void *stream);
int pcl_pano_pack_u8(const float *img_hwc, int H, int W, uint32_t *pano, int *not_exact, void *stream);
int pcl_pano_pack_u8p(const float *img_hwc, int H, int W, uint32_t *pano, int *not_exact, void *stream);
int pcl_pano_pack_u8v(const float *img_hwc, int H, int W, uint32_t *pano, int *not_exact, void *stream);
int pcl_pano_pack_f16(const float *img_hwc, int H, int W, void *pano, int *not_exact, void *stream);

/* ---- sampling loss (+ gradient) ----------------------------------------------------------------------------
 * Replaces SamplingLoss.forward (omniloc.py:171-202), BatchSamplingLoss.forward (omniloc.py:311-356), the forward
 * in trim_input_loss (utils.py:484-499) and sampling_loss (omniloc.py:105-157), plus — with with_grad != 0 — the
 * autograd backward the reference runs at omniloc.py:47,254, all fused in one pass over the cloud.
 *
 *   trans [B][3], rot [B][3] = (yaw, pitch, roll)
 *   result[B][PCL_RESULT_STRIDE] = loss, count, dL/dt(3), dL/d(yaw,pitch,roll)   (grad slots 0 if !with_grad)
 *   visible : nullable uint8 [B][n] in PACKED point order, multiplied into the mask (build-defined depth mask,
 *             off in the reference; see pcl_scatter_min_depth)
 * loss = sum_i mask_i ||c_i - rgb_i||_2 / sum_i mask_i, mask_i = sampled colour not exactly (0,0,0); 0/0 -> NaN.
 * Limits (32-bit buffer addressing): n <= 2^27 points, packed panorama < 2 GiB; PCL_EINVAL beyond.
 */
size_t pcl_loss_workspace_bytes(int64_t n, int B);
/* The same with the scatter-min depth mask OF THE SAME POSES multiplied into the mask (build-defined; see pcl_depth_mask below for
 * the definition): the z-buffers of the B poses are built on a depth_h x depth_w grid from every depth_stride-th point (0 x 0 / 0:
 * pcl_depth_default) and the loss kernel looks every point's cell up — what one depth-masked GD iteration evaluates.  Equal to
 * pcl_depth_mask on that grid / stride followed by pcl_sampling_loss(visible = that mask).  workspace: pcl_loss_depth_workspace_bytes
 * with the SAME depth_h / depth_w / depth_stride (the default grid depends on the stride; 0 for an invalid grid).  depth_stride 1, 2 and 4
 * have coalesced z passes; other strides (<= 64) are accepted and slower than reading every point.  depth_h, depth_w < 2^24. */
size_t pcl_loss_depth_workspace_bytes(int64_t n, int B, int H, int W, int depth_h, int depth_w, int depth_stride);
int pcl_sampling_loss_depth(const float *cloud, int64_t n, const void *pano, int pano_format, int H, int W, const float *trans,
                            const float *rot, int B, int with_grad, int depth_h, int depth_w, float tau, int depth_stride, float *result,
                            void *workspace, size_t workspace_bytes, void *stream);
int pcl_sampling_loss(const float *cloud, int64_t n, const void *pano, int pano_format, int H, int W, const float *trans,
                      const float *rot, int B, int with_grad, const uint8_t *visible, float *result, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- gradient-descent refinement ---------------------------------------------------------------------------
 * Replaces the optimisation loops of omniloc (omniloc.py:44-58) and omniloc_batch (omniloc.py:249-269): per
 * iteration one fused loss+gradient pass and one epilogue that does, per candidate and entirely on the device,
 * the final reduction, the chain rule to (t, yaw, pitch, roll), torch.optim.Adam (betas 0.9/0.999, eps 1e-8),
 * ReduceLROnPlateau(mode='min', threshold 1e-4 rel, cooldown 0, min_lr 0, eps 1e-8) and the clamp of t to `box`.
 *
 *   mode PCL_GD_SEQUENTIAL : clamp applies to the parameters the next forward sees          (omniloc.py:52-58)
 *   mode PCL_GD_BATCH      : the next forward sees the post-step PRE-clamp copy while Adam keeps updating the
 *                            clamped leaf (one-iteration lag of omniloc.py:260-269)
 *   box[6] = x_min, x_max, y_min, y_max, z_min, z_max  (device; quantile() of each xyz column, omniloc.py:53-55)
 *   state   : pcl_gd_state_bytes(B) bytes, opaque; pcl_gd_init fills it from the starting poses
 *   loss_history : nullable [num_iter][B]; loss of every forward
 * pcl_gd_run enqueues num_iter iterations back to back (no host synchronisation; capturable in a hipGraph) and may
 * be called repeatedly to continue.  pcl_gd_result writes, per candidate, PCL_GD_RESULT_STRIDE floats:
 *   fwd t(3), fwd ypr(3)  — the pose the reference returns (omniloc.py:102 / :272-275),
 *   leaf t(3), leaf ypr(3) — what the caller's input_trans/input_rot rows hold afterwards (omniloc.py:15-19,216-219),
 *   last loss (loss of the LAST forward, i.e. at the pose before the final update, omniloc.py:46,102,271,276), lr,
 *   ReduceLROnPlateau's num_bad_epochs and best (as floats; what the teacher-forced parity test compares with the reference's).
 */
#define PCL_GD_SEQUENTIAL 0
#define PCL_GD_BATCH 1
#define PCL_GD_RESULT_STRIDE 16

typedef struct pcl_gd_hyper {
    double lr;          /* cfg.lr        (omniloc.py:25)  */
    double factor;      /* cfg.factor    (omniloc.py:28)  */
    int32_t patience;   /* cfg.patience  (omniloc.py:27)  */
    int32_t mode;       /* PCL_GD_SEQUENTIAL | PCL_GD_BATCH */
    int32_t depth_mask; /* 0 = reference behaviour; 1 = the scatter-min depth mask of the poses each iteration evaluates multiplies
                           into the loss mask: before every loss pass the z-buffers of all B candidates are rebuilt (fill + z pass,
                           csrc/pcl_depth.hip) and the loss kernel looks each point's cell up (build-defined, cfg key `depth_mask`) */
    float depth_tau;    /* visibility tolerance: a point is visible iff its distance <= (1 + depth_tau) x the smallest distance in its cell */
    int32_t depth_h;    /* the z-buffer's grid (make_pano's pixel formula, utils.py:158-165, on depth_h x depth_w cells): chosen by point  */
    int32_t depth_w;    /* density, NOT the panorama's resolution.  0 x 0: pcl_depth_default(n, H, W, depth_stride).                         */
    int32_t depth_stride; /* the z-buffers are built from every depth_stride-th point of the packed cloud (every point is still TESTED       */
                        /* against them).  0: pcl_depth_default's choice with a default grid, 1 with a given grid.                           */
    int32_t fuse;          /* 0: pcl_gd_plan's rule (ONE launch per iteration when every block of the launch is resident at once).     */
                           /* < 0: never — always loss launch + epilogue launch.  Same bits either way (tests compare the two forms).  */
    int32_t images;        /* number of query images whose candidates share this launch chain (pcl_gd_set_panos / _set_pano_groups;   */
                           /* image i's candidates a contiguous range).  0 / 1: one image.  A hint for the block -> XCD mapping only    */
                           /* (with several panoramas every XCD takes a range of pose groups, i.e. of images, over the whole cloud      */
                           /* instead of a slice of the cloud for all of them): results do not depend on it.                            */
} pcl_gd_hyper;

size_t pcl_gd_state_bytes(int B);
/* workspace of pcl_gd_run: the loss partials, plus B z-buffers of depth_h x depth_w words when hyper->depth_mask is set (0 for an
 * invalid grid).  Pure scratch: every iteration refills what it reads, so a state may be continued with any workspace. */
size_t pcl_gd_workspace_bytes(int64_t n, int B, int H, int W, const pcl_gd_hyper *hyper_host);
int pcl_gd_init(void *state, const float *trans, const float *rot, int B, const pcl_gd_hyper *hyper_host, void *stream);
int pcl_gd_run(const float *cloud, int64_t n, const void *pano, int pano_format, int H, int W, void *state, int B, const float *box,
               const pcl_gd_hyper *hyper_host, int num_iter, float *loss_history, void *workspace,
               size_t workspace_bytes, void *timer, void *stream);
int pcl_gd_result(const void *state, int B, float *result, void *stream);
/* Teacher-forcing hook (parity tests): ONE optimiser step of every candidate from a GIVEN loss [B] and
 * gradient [B][6] = dL/d(t0, t1, t2, yaw, pitch, roll) — e.g. the reference's recorded loss_list and autograd gradients
 * (omniloc.py:253-254) — through the very update code of pcl_gd_run's epilogue: torch.optim.Adam, ReduceLROnPlateau, the clamp to
 * `box` in the given mode, the next forward pose.  Updates `state` in place (copy 0: follow with pcl_gd_result); scratch: B * 8
 * floats.  last loss / lr / the scheduler's counters advance exactly as in a run. */
int pcl_gd_step_from_grads(void *state, int B, const float *loss, const float *grad, const float *box, const pcl_gd_hyper *hyper_host,
                           float *scratch, void *stream);
/* How pcl_gd_run decomposes an n-point, B-candidate problem (host-only query, measurement aid): chunks of the cloud, poses per
 * block, and whether an iteration is ONE launch (the loss launch of iteration k + 1 finishes iteration k in the prologue of every
 * block: launches whose chunk x group blocks are all resident at once — the reference's shipped 167k-point / 6-candidate shape)
 * or two (loss + epilogue).  Any of the three outputs may be NULL. */
int pcl_gd_plan(int64_t n, int B, int *nchunks_host, int *poses_per_block_host, int *fused_host);
/* the same for the hyper-parameters a run will use: a depth-masked run (hyper->depth_mask) never fuses; hyper NULL = pcl_gd_plan */
int pcl_gd_plan_hyper(int64_t n, int B, const pcl_gd_hyper *hyper_host, int *nchunks_host, int *poses_per_block_host, int *fused_host);
/* Several query images against one shared cloud in ONE launch chain (BASELINE cfg 4: independent panoramas, shared
 * cloud): candidate b samples panos[b] (device array of B device addresses of packed panoramas; all the same H, W and
 * texel format as the `pano` passed to pcl_gd_run, which stays the default for entries that are 0).  Call after
 * pcl_gd_init (which resets every candidate to the default panorama); panos == NULL clears the table.  More poses per
 * launch share each cloud chunk in L2 and amortise the per-block costs: at cfg 2, 4 images x 32 candidates per launch
 * run at the efficiency of cfg 3. */
int pcl_gd_set_panos(void *state, const uint64_t *panos, int B, void *stream);
/* The same from a short HOST list: candidates [i * per_image, (i + 1) * per_image) sample panos_host[i] (device addresses of
 * packed panoramas; B = nimages * per_image).  The addresses travel as kernel arguments — nothing is copied to the device, so
 * nothing waits for the work the stream already holds (one launch per 64 images). */
int pcl_gd_set_pano_groups(void *state, const uint64_t *panos_host, int nimages, int per_image, void *stream);
/* The end of omniloc_batch, omniloc.py:271-277, for nimages x per_image candidates (B = nimages * per_image): per image the
 * candidate whose LAST forward had the smallest loss (torch.argmin semantics: first of equal minima, a NaN loss wins), as
 * winners [nimages][16] = post-step translation (3), R = RZ(yaw) RY(pitch) RX(roll) of the post-step angles (9, row-major),
 * that loss, yaw / pitch / roll.  leaf_trans / leaf_rot [B][3] (nullable) receive every candidate's leaf parameters — what the
 * reference leaves in the caller's input_trans / input_rot, whose rows it optimises in place (omniloc.py:216-219). */
int pcl_gd_winner(const void *state, int nimages, int per_image, float *winners, float *leaf_trans, float *leaf_rot, void *stream);

/* ---- kernel timer (measurement aid, HOST object) --------------------------------------------------------------
 * A pool of hipEvent pairs.  When a timer is passed to pcl_gd_run, every launch of the fused loss+gradient kernel is
 * bracketed by an event pair recorded on `stream` (no synchronisation inside the run).  pcl_timer_read synchronises
 * on the recorded events and returns the summed kernel time and the number of launches since the last reset.
 * Launches beyond `capacity` are simply not timed. */
void *pcl_timer_create(int capacity);
void pcl_timer_destroy(void *timer);
void pcl_timer_reset(void *timer);
void pcl_timer_set_stride(void *timer, int stride); /* time only every stride-th launch of a run (default 1) */
int pcl_timer_read(void *timer, double *total_ms_host, int *launches_host);
/* What an event pair reads with NOTHING between its two records: `reps` empty pairs are recorded on `stream`, the call
 * synchronises on them and returns their median elapsed time (ms).  A pair around a kernel over-reads the kernel's duration
 * by about this much (the two event packets' own processing); bench.py subtracts it per timed launch.  Uses its own events,
 * leaves the timer's recorded pairs untouched. */
int pcl_timer_calibrate(void *timer, int reps, double *pair_ms_host, void *stream);

/* ---- stand-alone ops of the path ---------------------------------------------------------------------------- */
/* utils.py:16-61 cloud2idx: xyz [n][3] -> coord [n][2] in [-1,1]^2 (batched form = same call on B*n points). */
int pcl_cloud2idx(const float *xyz, int64_t n, float *coord, void *stream);
/* utils.py:64-103 sample_from_img: clip to +-0.99, bilinear, zero padding, align_corners=False; rgb_out [n][3]. */
int pcl_sample_from_img(const void *pano, int pano_format, int H, int W, const float *coord, int64_t n, float *rgb_out,
                        void *stream);
/* Backward of the two ops above, for callers that differentiate through them outside SamplingLoss (the reference's are plain
 * autograd ops, utils.py:16-103):
 *   pcl_cloud2idx_backward        grad_xyz [n][3] = J^T grad_coord [n][2]  (atan2 / norm chain rule of utils.py:44-59)
 *   pcl_sample_from_img_backward  grad_coord [n][2] (through torch.clip and grid_sampler_2d, utils.py:96-98; nullable) and
 *                                 grad_img [H][W][3] fp32, ACCUMULATED with float atomics into a caller-zeroed buffer
 *                                 (nullable); grad_rgb [n][3] is the incoming gradient. */
int pcl_cloud2idx_backward(const float *xyz, const float *grad_coord, int64_t n, float *grad_xyz, void *stream);
int pcl_sample_from_img_backward(const void *pano, int pano_format, int H, int W, const float *coord, const float *grad_rgb,
                                 int64_t n, float *grad_coord, float *grad_img, void *stream);
/* utils.py:425-453 rot_from_ypr for B poses: rot [B][3] -> R [B][9] row-major. */
int pcl_rot_from_ypr(const float *rot, int B, float *R, void *stream);
/* utils.py:208-229 quantile on each of the 3 columns of xyz [n][3]: box[6] = x[int(n q)], x[int(n (1-q))], y.., z..
 * (exact order statistics by radix select; workspace pcl_quantile_workspace_bytes()). */
size_t pcl_quantile_workspace_bytes(void);
int pcl_quantile_box(const float *xyz, int64_t n, double q, float *box, void *workspace, void *stream);

/* Build-defined scatter-min visibility (the reference imports torch_scatter.scatter_min at utils.py:6 but never
 * calls it).  For camera-frame points xyz_cam [n][3]: pixel = make_pano's (utils.py:158-165), depth = ||p||
 * (utils.py:152); zbuf [H*W] uint64 = min over the pixel of (depth_bits << 32 | index), 0xFFFF... if empty.
 * pcl_scatter_min_unpack turns it into torch_scatter's (out, arg) convention: empty -> (0, n). */
int pcl_scatter_min_depth(const float *xyz_cam, int64_t n, int H, int W, uint64_t *zbuf, void *stream);
int pcl_scatter_min_unpack(const uint64_t *zbuf, int64_t n, int H, int W, float *zmin, int64_t *argmin, void *stream);
/* utils.py:134-205 make_pano: 3x3 splat, nearest point wins inside a pass, later passes overwrite earlier ones
 * (order idx8..idx1, centre; utils.py:190-198).  image [H][W][3] float = rgb*255 (0 where nothing projects).
 * workspace: H*W uint64. */
int pcl_make_pano(const float *xyz_cam, const float *rgb, int64_t n, int H, int W, float *image, uint64_t *workspace,
                  void *stream);
/* Second trimming stage of the initialisation for a batch of candidate poses (utils.py:510-588 with
 * color_utils.py:68-144): render the packed world-frame cloud (pcl_cloud_pack) from every candidate (make_pano semantics
 * at the image's resolution; among points at exactly equal distance the larger PACKED index wins), and per block j of the middle block rows (h = 1 + j / nsw in 1..nsh-2, w = j % nsw) intersect the
 * normalised 8x8x8 colour histogram of the rendered pixels (both render and query non-black) with that of the query
 * image's non-black pixels.  inter [ncand][(nsh-2)*nsw], nproj [ncand][..] = pixels histogrammed, nimg [..] likewise
 * for the query.  The caller forms score = sum_j inter / (nsh*nsw) with the reference's empty-block rule. */
size_t pcl_hist_trim_workspace_bytes(int ncand, int H, int W, int nsh, int nsw);
/* Same for a cloud of n points, large enough for the tile-binned render (ABI v4): every candidate's points are binned by the
 * 64 x 64-pixel image tile(s) their 3 x 3 splat touches and every tile is resolved and histogrammed in LDS — no z-buffer in
 * HBM, bit-identical scores.  pcl_hist_trim_scores takes that path when its workspace is at least this large (4 n list
 * entries of 12 bytes per candidate: the exact worst case; plus the tiles' run tables, 8 bytes x tiles x ceil(n / 2048) per candidate)
 * and the z-buffer splat otherwise. */
size_t pcl_hist_trim_workspace_bytes_n(int64_t n, int ncand, int H, int W, int nsh, int nsw);
int pcl_hist_trim_scores(const float *cloud, int64_t n, const float *img_hwc, int H, int W, const float *trans,
                         const float *rot, int ncand, int nsh, int nsw, float *inter, int32_t *nproj, int32_t *nimg,
                         void *workspace, size_t workspace_bytes, void *stream);
/* score[cand] of the trimming stage from pcl_hist_trim_scores' outputs: sum of the block intersections of every block row
 * up to its first empty block (utils.py:568-571), divided by nsh * nsw (utils.py:580). */
int pcl_hist_trim_reduce(const float *inter, const int32_t *nproj, const int32_t *nimg, int ncand, int nsh, int nsw, float *score,
                         void *stream);
/* The same for `nimages` query images of ONE room in one set of launches (<= 32 images): candidates [i * cand_per_image,
 * (i + 1) * cand_per_image) of trans / rot are rendered and scored against imgs_host[i] (HOST array of device addresses of (H, W, 3)
 * float images); inter / nproj [nimages * cand_per_image][nblk], nimg [nimages][nblk], score [nimages * cand_per_image].  Every image's
 * candidates form their own chain of the empty-block carry-over (one call of the reference's function per image); results are
 * those of the single-image entry points, bit for bit. */
size_t pcl_hist_trim_images_workspace_bytes(int64_t n, int nimages, int cand_per_image, int H, int W, int nsh, int nsw);
int pcl_hist_trim_scores_images(const float *cloud, int64_t n, const float *const *imgs_host, int nimages, int cand_per_image, int H, int W,
                                const float *trans, const float *rot, int nsh, int nsw, float *inter, int32_t *nproj, int32_t *nimg,
                                void *workspace, size_t workspace_bytes, void *stream);
int pcl_hist_trim_reduce_images(const float *inter, const int32_t *nproj, const int32_t *nimg, int nimages, int cand_per_image, int nsh,
                                int nsw, float *score, void *stream);
/* First trimming stage of the initialisation, utils.py:462-507 trim_input_loss: the forward-only sampling loss
 * (utils.py:484-499 = omniloc.py:171-202 without gradient) of ALL K x R pairs of trans [K][3] and rot [R][3] (yaw, pitch, roll),
 * loss_table [K][R] row-major like the reference's `loss_table[i, j]` (utils.py:497; its argsort / index decode stay with the
 * caller), count_table [K][R] (nullable) = points kept by the mask.  Rotations that differ only in YAW share the projection:
 * with q' = RY(pitch) RX(roll) (x - t), theta does not depend on yaw and phi = atan2(q'_y, q'_x) + yaw - 1e-6 p_y / rho^2 (first
 * order in the reference's `x + 1e-6`; points within 0.1 mm of the camera's vertical axis are evaluated exactly), so a block
 * rotates a point and takes both atan2s once for up to four yaws.
 *   pcl_trim_groups: classes of equal (pitch, roll) of the rotation table (bitwise), built on the device into
 *     `groups` (pcl_trim_groups_bytes(R) bytes; its first int32 is the number of 4-yaw groups, which a caller may read back ONCE
 *     per rotation grid and pass as `ngroups`; passing R is always valid: surplus blocks return at once).
 *     R <= 1024 (PCL_EINVAL beyond: callers fall back to pcl_sampling_loss over the K x R pairs, as piccolo_amd/utils.py does).
 *   pcl_trim_loss: workspace pcl_trim_loss_workspace_bytes(n, K, ngroups).  The table is filled with NaN first; if `groups` was
 *     built from a table of another size, or holds more groups than `ngroups`, NOTHING is written over them: an entry the
 *     launch did not compute ranks last in the caller's selection (NaN), never as stale memory. */
size_t pcl_trim_groups_bytes(int R);
/* The two selections of the initialisation stage in one launch each (one block per problem, nprob problems of M values):
 *   largest == 0, rot_per_trans == len(rot):  utils.py:500-505  min_inds = loss_table.flatten().argsort()[:n_keep];
 *                                             out_trans = trans[min_inds // len(rot)], out_rot = rot[min_inds % len(rot)]
 *   largest == 1, rot_per_trans == 0:         utils.py:583-586  min_inds = flip(hist_intersect.argsort()[-n_keep:]);
 *                                             out_trans = trans[min_inds], out_rot = rot[min_inds]
 * values [nprob][M]; problem p reads its pose rows from trans / rot + p * pose_stride * 3 floats (pose_stride 0: shared tables);
 * out_trans / out_rot [nprob][n_keep][3], out_idx [nprob][n_keep] (nullable).  Exact and deterministic: ascending by value with
 * ties by ascending index (a stable argsort); largest: descending, ties by descending index (the flipped tail of that argsort);
 * NaN ranks last in both.  n_keep <= min(M, 1024). */
int pcl_select_poses(const float *values, int nprob, int M, int n_keep, int largest, const float *trans, const float *rot,
                     int rot_per_trans, int64_t pose_stride, float *out_trans, float *out_rot, int *out_idx, void *stream);
int pcl_trim_groups(const float *rot, int R, void *groups, void *stream);
size_t pcl_trim_loss_workspace_bytes(int64_t n, int K, int ngroups);
int pcl_trim_loss(const float *cloud, int64_t n, const void *pano, int pano_format, int H, int W, const float *trans, int K,
                  const float *rot, int R, const void *groups, int ngroups, const void *order, float *loss_table, float *count_table,
                  void *workspace, size_t workspace_bytes, void *stream);
/* The launch's WORK LIST (ABI 9; `order`, nullable, of pcl_trim_loss / pcl_trim_loss_images).  Which block evaluates which (cloud chunk,
 * (translation, rotation class) slot) item is scheduling only — every item's partial sum has its own place, the tables are bit-identical
 * with any list or none — but it decides what the caches see: in plain (chunk, slot) order every pose streams its own region of the
 * panorama through the L2s (1M points x 1800 poses: 16.4 GB through the memory side for 41 MB of unique data).  pcl_trim_order ranks the
 * items by the panorama ROW their chunk's centroid projects to (the row does not depend on the yaw), cuts the ranking into bands of about
 * 2.5 MB of texture rows and deals the XCDs contiguous eighths: an XCD's L2 then holds one band of the texture for all the poses
 * (6.6 GB, L2 hit 0.70 -> 0.88).  The list depends on the cloud, the candidate grid and the panorama's size / texel layout, NOT on the query
 * image: build it once per room (two radix sorts of chunks x slots keys) and pass it to every image's launch.  A list built for another
 * cloud size or grid is ignored by the kernel (plain order).  order: pcl_trim_order_bytes(n, K, ngroups) bytes. */
size_t pcl_trim_order_bytes(int64_t n, int K, int ngroups);
size_t pcl_trim_order_workspace_bytes(int64_t n, int K, int ngroups);
int pcl_trim_order(const float *cloud, int64_t n, int pano_format, int H, int W, const float *trans, int K, const float *rot, int R,
                   const void *groups, int ngroups, void *order, void *workspace, size_t workspace_bytes, void *stream);
/* The same for `nimages` query images of ONE room in one launch (the image loop of localize.py:143-223: the candidate grid
 * depends on the cloud only, utils.py:613-616): panos_host = HOST array of nimages device addresses of packed panoramas (one
 * size and texel format; nimages <= 32), loss_tables / count_tables [nimages][K][R].  The cloud is cut into the chunks of the
 * single-image launch, so every image's table has the bits pcl_trim_loss gives it. */
size_t pcl_trim_loss_images_workspace_bytes(int64_t n, int K, int ngroups, int nimages);
int pcl_trim_loss_images(const float *cloud, int64_t n, const void *const *panos_host, int nimages, int pano_format, int H, int W,
                         const float *trans, int K, const float *rot, int R, const void *groups, int ngroups, const void *order,
                         float *loss_tables, float *count_tables, void *workspace, size_t workspace_bytes, void *stream);
/* Scatter-min depth mask on the PACKED cloud for B poses (build-defined: the reference imports torch_scatter.scatter_min at
 * utils.py:6 and never calls it; off by default in the loss).  Seen from pose b, every point falls into one cell of an H x W grid by
 * make_pano's pixel formula (utils.py:158-165) — the DEPTH grid, chosen by point density, not the panorama's resolution: a z-buffer
 * hides a point only when an occluder's point shares its cell, and at the panorama's resolution most cells hold one point —
 *   zmin[b][cell] = the smallest ||p|| among the OCCLUDER SAMPLES in the cell: every stride-th point of the packed cloud (1: all)
 *   visible[b][i] = 1 iff ||p_i|| <= (1 + tau) x zmin[b][cell of p_i]                  (every point i, packed point order)
 * Feeds the `visible` argument of pcl_sampling_loss; the GD loop and pcl_sampling_loss_depth look the z-buffer up in the loss kernel
 * instead and never build the byte mask.  workspace: pcl_depth_workspace_bytes(B, H, W).
 * pcl_depth_default (host-only): what is used when a caller does not name them.  Grid: at least 12 occluder samples per cell
 * (depth_w = 2 depth_h, depth_h a multiple of 8, never finer than the H x W panorama); tau = 3.5 pi / depth_h clipped to [0.02, 0.15]
 * (a coarser cell needs a larger tolerance: a surface seen at a grazing angle spans more depth inside it); stride (stride_in = 0): the
 * largest of 1, 2, 4 that keeps depth_h >= 128 — what the mask finds depends on the samples per cell, not on reading every point.
 * Measured against analytic occlusion on a furnished room (tools/depth_recall.py): recall 0.93-0.96, precision 0.93-0.99 for 167k-4M
 * points.  Any output may be NULL. */
int pcl_depth_default(int64_t n, int H, int W, int stride_in, int *depth_h_host, int *depth_w_host, float *tau_host, int *stride_host);
size_t pcl_depth_workspace_bytes(int B, int H, int W);
int pcl_depth_mask(const float *cloud, int64_t n, const float *trans, const float *rot, int B, int H, int W, float tau, int stride,
                   uint8_t *visible, void *workspace, size_t workspace_bytes, void *stream);
/* ---- colour preprocessing of the query panorama (color_utils.py; called at localize.py:173-179, :395-409) ----
 *
 * pcl_color_template_build: the point colours rgb [n][3] sorted per channel, tmpl [3][n] — once per cloud.
 * pcl_color_match (color_utils.py:146-234): histogram matching of the non-black pixels of img [H][W][3] (levels k/255,
 *   as decoded from an image file) to the point colours, per channel, with sin(latitude) pixel weights, the reference's
 *   wrapped interpolation (period 360) and its rank-indexed lookup.  out [H][W][3]; black pixels are copied.
 *   *not_exact (device int32, nullable) = 1 when some non-black pixel channel is not exactly k/255: the output is then
 *   not the reference's (its lookup is by distinct float value) and the Python layer raises.
 * pcl_color_mod (color_utils.py:7-65): joint luma equalisation.  Panorama (non-black pixels) and point colours go to
 *   8-bit YCrCb (OpenCV's fixed-point cvtColor, restated), the two Y histograms with num_bins levels are added, Y becomes
 *   the joint cumulative distribution at its level, back to RGB.  out_img [H][W][3], out_rgb [n][3].
 * workspace for both: pcl_color_workspace_bytes(). */
size_t pcl_color_template_bytes(int64_t n);
size_t pcl_color_template_workspace_bytes(int64_t n);
int pcl_color_template_build(const float *rgb, int64_t n, float *tmpl, void *workspace, size_t workspace_bytes, void *stream);
size_t pcl_color_workspace_bytes(void);
int pcl_color_match(const float *img_hwc, int H, int W, const float *tmpl, int64_t n, float *out, int32_t *not_exact,
                    void *workspace, size_t workspace_bytes, void *stream);
int pcl_color_mod(const float *img_hwc, int H, int W, const float *rgb, int64_t n, int num_bins, float *out_img,
                  float *out_rgb, void *workspace, size_t workspace_bytes, void *stream);
/* color_utils.py:68-118 histogram (unbatched form): colours img [npix][3] (scaled by 255 first if max(img) <= 1), pixels
 * with mask[p] != 0, c0 x c1 x c2 bins of size ceil(255 / c); hist [c0*c1*c2] float, index r + c0 g + c0 c1 b;
 * normalize: hist / (sum + eps) (eps 0 = unbatched form, 1e-6 = the batched form's).  color_utils.py:122-144
 * histogram_intersection: out[i] = sum_k min(a[i][k], b[i][k]). */
size_t pcl_histogram_workspace_bytes(int c0, int c1, int c2);
int pcl_histogram(const float *img, const uint8_t *mask, int64_t npix, int c0, int c1, int c2, int normalize, float eps,
                  float *hist, void *workspace, size_t workspace_bytes, void *stream);
int pcl_histogram_intersection(const float *a, const float *b, int batch, int nbins, float *out, void *stream);
/* p = R (x - t) for one pose: xyz [n][3] -> out [n][3] (feeds make_pano / scatter-min; localize.py:266-267). */
int pcl_transform_cloud(const float *xyz, int64_t n, const float *trans, const float *rot, float *out, void *stream);

/* ---- dataset text clouds (host side, no GPU involved) ----
 * data_utils.py:16-43 read_stanford / :138-163 read_omniscenes parse "x y z r g b" lines with
 * pandas.read_table(header=None, delim_whitespace=True).values.  pcl_cloud_txt_rows: number of non-blank lines
 * (< 0: -errno).  pcl_cloud_txt_read: rows x cols doubles, row-major, parsed by `nthreads` threads (0 = all cores)
 * from an mmap of the file; returns 0, -errno, PCL_EINVAL (bad arguments / row count mismatch) or -(1000 + line) for
 * the first malformed line (1-based). */
int64_t pcl_cloud_txt_rows(const char *path);
int64_t pcl_cloud_txt_read(const char *path, int64_t rows, int cols, double *out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif /* PICCOLO_HIP_H */
