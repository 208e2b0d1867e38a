/* include/avx.h -- C ABI of libavx.so: the MI355X (gfx950) per-frame animal-vision hot path.
 *
 * This is the drop-in boundary.  The reference (Kyaw-Thiha/animal-vision) is pure
 * Python and has no FFI of its own; the entry points below are what a ctypes binding added to the
 * reference's shared helpers would call instead of their NumPy/OpenCV bodies.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference repo).  The
 * reference-side stubs are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (AVX_OK) or a negative avx_status; nothing throws across the ABI;
 *     avx_last_error(ctx) returns a human-readable message for the last failure on that ctx;
 *   - an avx_ctx is bound to ONE device and is single-threaded (one caller at a time), matching the
 *     reference's one-frame-in-flight loop (main.py:60-72);
 *   - `stream` is a hipStream_t passed as void*, with HIP's own meaning: NULL is the null (default) stream,
 *     a torch.cuda.Stream.cuda_stream value is accepted as is; every launch is asynchronous on it, with ONE exception:
 *     the first avx_dichromat_u8 call for a (kernel configuration, batch, frame size) that is not in the seeded table
 *     times its candidate launch geometries on the caller's frames (~40 ms, blocking; the results are the same for every
 *     candidate) and remembers the winner in the ctx -- AVX_MARCH_CHUNKS / AVX_MARCH_NG pin a geometry and skip it;
 *   - frame/plane pointers are DEVICE pointers unless the parameter name ends in `_host`;
 *     the caller owns them; ctx owns only its scratch and its constant tables;
 *   - images are C-contiguous HWC (H x W x 3), batches are N such frames back to back.
 */
#ifndef AVX_H
#define AVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVX_ABI_VERSION 1

typedef struct avx_ctx avx_ctx;

typedef enum avx_status {
    AVX_OK = 0,
    AVX_ERR_INVALID = -1,     /* bad argument (shape, mode, NULL pointer) */
    AVX_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    AVX_ERR_HIP = -3,         /* a HIP runtime call failed; see avx_last_error */
    AVX_ERR_UNSUPPORTED = -4, /* valid request this build does not implement */
    AVX_ERR_NOMEM = -5
} avx_status;

/* ---- context, memory, streams ---------------------------------------------------------------- */
int avx_abi_version(void);
int avx_device_count(void);                       /* 0 when no GPU is visible; never fails        */
int avx_init(int device, avx_ctx** out_ctx);      /* AVX_ERR_NO_DEVICE when there is no such GPU  */
void avx_destroy(avx_ctx* ctx);
const char* avx_last_error(const avx_ctx* ctx);   /* ctx may be NULL: message of the last failed avx_init */

int avx_malloc(avx_ctx* ctx, size_t bytes, void** out_dptr);
int avx_free(avx_ctx* ctx, void* dptr);
int avx_host_alloc(avx_ctx* ctx, size_t bytes, void** out_hptr); /* pinned host memory */
int avx_host_free(avx_ctx* ctx, void* hptr);
int avx_memcpy_h2d(avx_ctx* ctx, void* dst, const void* src_host, size_t bytes, void* stream);
int avx_memcpy_d2h(avx_ctx* ctx, void* dst_host, const void* src, size_t bytes, void* stream);
int avx_memset(avx_ctx* ctx, void* dst, int value, size_t bytes, void* stream);
int avx_stream_create(avx_ctx* ctx, void** out_stream);
int avx_stream_destroy(avx_ctx* ctx, void* stream);
int avx_sync(avx_ctx* ctx, void* stream);         /* hipStreamSynchronize(stream) */
int avx_stream_wait(avx_ctx* ctx, void* waiter, void* signaler); /* device-side: waiter waits for signaler's work so far */
int avx_device_sync(avx_ctx* ctx);                /* hipDeviceSynchronize() */

/* HIP-event stopwatch ON `stream` (bench.py roofline leg: per-launch duration of the hot kernel). */
int avx_timer_start(avx_ctx* ctx, void* stream);
int avx_timer_stop(avx_ctx* ctx, void* stream, float* out_ms); /* records, synchronises, returns ms */

/* ---- dichromat path: animals/animal_utils.py + animals/<species>.py template -------------------
 *
 * One fused launch replaces, for N uint8 frames,
 *   get_normalized_image   animals/animal_utils.py:41-50   (uint8 -> f32, data-dependent /255)
 *   srgb_to_linear         animals/animal_utils.py:5-11    (256-entry table of the reference's values)
 *   pixels @ T.T           animals/dog.py:43-48            (FMA chain, f32)          [AVX_COLOR_MATRIX]
 *     or sRGB_to_LMS / L-M merge / LMS_to_RGB   animals/cat.py:95-101  (f64 tail)   [AVX_COLOR_CAT_MERGE]
 *   apply_acuity_blur      animals/animal_utils.py:121-145 (cv2.GaussianBlur semantics)   [AVX_POST_GAUSS]
 *     or apply_s_cone_vertical_gain  animal_utils.py:206-259 (per-row blue gain)          [AVX_POST_ROWGAIN]
 *     or apply_anisotropic_acuity_blur_with_streak  animal_utils.py:147-172 (as coded, Q3) [AVX_POST_STREAK]
 *   apply_chroma_compression  animal_utils.py:174-181 (optional)
 *   clip -> linear_to_srgb -> clip -> (x*255+0.5).astype(uint8)   animals/dog.py:54-57
 * Results are bit-identical to the reference executed in the build container (tests/golden). */
enum { AVX_COLOR_MATRIX = 0, AVX_COLOR_CAT_MERGE = 1 };
enum { AVX_POST_NONE = 0, AVX_POST_GAUSS = 1, AVX_POST_ROWGAIN = 2, AVX_POST_STREAK = 3 };

typedef struct avx_dichromat_desc {
    uint32_t struct_size;      /* sizeof(avx_dichromat_desc), for ABI growth                         */
    int32_t color_mode;        /* AVX_COLOR_*                                                        */
    float matrix[9];           /* AVX_COLOR_MATRIX: T row-major, out_i = sum_j T[i][j]*in_j (Q1)      */
    float cat_alpha;           /* AVX_COLOR_CAT_MERGE: LM = alpha*L + beta*M  (cat.py:98-99), both as   */
    float cat_beta;            /*   float32: alpha = f32(a), beta = f32(1.0 - a) computed in double     */
    int32_t post_mode;         /* AVX_POST_*                                                         */
    int32_t ksize;             /* AVX_POST_GAUSS: odd tap count (same in x and y), 1..AVX_MAX_KSIZE   */
    const double* taps_host;   /* AVX_POST_GAUSS: ksize normalised taps in double (getGaussianKernel);
                                  the library rounds them to the compute type (f32, or f64 for cat)  */
    const float* row_gain_host;/* AVX_POST_ROWGAIN: H per-row gains for channel 2 (host pointer)      */
    int32_t row_gain_clamp;    /* AVX_POST_ROWGAIN: clip channel 2 to [0,1] after the gain            */
    int32_t chroma_enable;     /* apply_chroma_compression after the post stage                      */
    float chroma_keep;         /*   float32(1 - strength), the factor animal_utils.py:181 multiplies by */
    int32_t variant;           /* 0 = auto; A/B only: 1 = reference kernel, 2 = 2-D tiled, 3 = marching strip */
    const float* streak_rows_host; /* AVX_POST_STREAK: H x streak_stride floats, per image row:
                                  [0] k1, [1] k2 (odd tap counts of the sigmaX / sigmaY kernels of that row),
                                  [2..15) k1 float32 taps, [15..48) k2 float32 taps                         */
    int32_t streak_stride;     /* >= 48                                                                  */
    int32_t in_f32;            /* 1: in_hwc is a float32 HWC frame already normalised to [0,1] (e.g. the output of
                                  avx_binocular_warp_u8); decoded with the sRGB EOTF in float32 (device powf)   */
} avx_dichromat_desc;

#define AVX_MAX_KSIZE 33

int avx_dichromat_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_hwc, int n_frames, int H, int W,
                     const avx_dichromat_desc* desc, void* stream);

/* ---- UV / spectral path: uv_helpers.py, uv_mappers.py, ml/classic_rgb_to_hsi, animals/honeybee.py ----
 *
 * Planes are float32, K x (H*W), struct-of-arrays.  Float contract: within 1e-4 relative of the reference
 * (device atan2f/powf); order statistics are exact; uint8 encode uses the reference's threshold table. */

/* np.percentile(data, q) with linear interpolation (uv_mappers.py:32,61-62,73,104,142): exact order
 * statistics by radix select on device, NumPy's _lerp formula; blocks until the value is on the host. */
int avx_percentile(avx_ctx* ctx, const float* data, size_t n, double q, double* out_host, void* stream);

/* integrate_band / cone catches (uv_helpers.py:142-146, honeybee.py:125-135): out_k = sum_b hsi_b * w[k][b].
 * layout 0 = HxWxB (NHWC), 1 = BxHxW (NCHW, what a torch model emits); dtype 0 = float32, 1 = float16.
 * weights_host: K x B with the illuminant already folded in.  stats_host (optional): K x {min,max,mean,1}. */
int avx_spectral_integrate(avx_ctx* ctx, const void* hsi, int layout, int dtype, int H, int W, int B,
                           const float* weights_host, int K, float* out_planes, float* stats_host, void* stream);

/* {min, max, mean, denominator} per plane (safe_norm uv_helpers.py:47-53, von_kries_* :195-206).
 * adapt: 0 none (den 1), 1 white_patch max(max,eps), 2 gray_world max(mean,eps), 3 safe_norm (max-min).
 * The result stays on the device for the next avx_planes_gaussian_blur; stats_host (optional) copies it. */
int avx_plane_stats(avx_ctx* ctx, const float* planes, int K, size_t n, int adapt, float eps, float* stats_host, void* stream);

/* uv_helpers.gaussian_blur (uv_helpers.py:67-73, cv2 semantics) on K planes, optionally rescaling each plane by
 * the statistics of the last avx_plane_stats / avx_spectral_integrate first: scale_mode 0 none, 1 x/den,
 * 3 safe_norm.  ksize == 1: rescale only.  taps_host: ksize normalised taps in double. */
int avx_planes_gaussian_blur(avx_ctx* ctx, const float* in, float* out, int K, int H, int W, int ksize,
                             const double* taps_host, int scale_mode, void* stream);

/* apply_anisotropic_acuity_blur_with_streak (animals/animal_utils.py:147-172, as coded: quirk Q3) on three float32
 * planes of linear light (3 x H x W in, 3 x H x W out): the float-frame form of the streak species; uint8 frames take
 * the fused avx_dichromat_u8.  rows_host: H x stride floats per image row [k1, k2, 13 taps(sigma_x), 33 taps(sigma_y)]. */
int avx_streak_planes_f32(avx_ctx* ctx, const float* src_planes, float* dst_planes, int H, int W, const float* rows_host,
                          int stride, void* stream);

/* classic_rgb_to_hsi analytic branch (ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py:47-82): HxWx3 frame (uint8, or
 * float32 that is linearised as coded) -> HxWxB float32 cube.  gains_host: B x 3 lobe gains in input-channel
 * order (quirk Q5), denom: mean lobe sum + 1e-8. */
int avx_rgb_to_hsi_lobes(avx_ctx* ctx, const void* in_hwc, int in_is_u8, int H, int W, int B, const float* gains_host,
                         float denom, float* out_hwb, void* stream);

/* HoneyBee.visualize steps 1-7 (animals/honeybee.py:99-175) for N uint8 frames, all passes on the device:
 * catches -> von Kries -> Gaussian blur -> percentiles -> map_* -> clip -> OETF -> uint8. */
enum { AVX_MAP_FALSECOLOR = 0, AVX_MAP_CUSTOM_MATRIX = 1, AVX_MAP_OPPONENT = 2, AVX_MAP_UV_PURPLE_YELLOW = 3, AVX_MAP_FALSECOLOR_UV_MIXED = 4 };
typedef struct avx_honeybee_desc {
    uint32_t struct_size;
    int32_t source;            /* 0: uint8 RGB frames + rgb_matrix (analytic lobes x illuminant x cone curves folded
                                  to 3x3, the route honeybee.py actually takes); 1: an HSI cube + weights_host;
                                  2: the catches themselves + their statistics (one frame)                        */
    float rgb_matrix[9];       /* source 0: [U,B,G]_k = sum_j rgb_matrix[k][j] * linear_channel_j                  */
    const void* hsi;           /* source 1: device cube(s), N of them back to back                                */
    int32_t hsi_layout, hsi_dtype, bands;
    const float* weights_host; /* source 1: 3 x bands (illuminant folded in)                                      */
    int32_t adaptation;        /* 0 none, 1 white_patch, 2 gray_world (honeybee.py:137-141)                       */
    float eps;
    int32_t blur_ksize;        /* 0: no blur; else odd taps of uv_helpers.gaussian_blur                            */
    const double* blur_taps_host;
    int32_t mapping;           /* AVX_MAP_*                                                                       */
    float custom_matrix[9];
    float mixed_alpha;         /* falsecolor_uv_mixed alpha (honeybee.py:162 passes 0.45)                          */
    int32_t out_float;         /* 1: out_hwc is a float32 HxWx3 buffer receiving linear_to_srgb(clip(rgb_lin)) (the
                                  reference's output for float frames, honeybee.py:172-173) instead of uint8 codes */
    const float* catches;      /* source 2: the three catch planes (device, 3 x H*W float32) already formed, e.g. by
                                  avx_mst_conv3x3_lds_spectral (the spectral integration as conv_out's epilogue) ...  */
    const void* catch_partials; /* ... with n_catch_partials x 3 records {float min, max; double sum} of them (device) */
    int32_t n_catch_partials;
} avx_honeybee_desc;

/* debug_planes (optional, device, N x 3 x H*W floats): receives U,B,G after adaptation + blur. */
int avx_honeybee_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_hwc, int n_frames, int H, int W,
                    const avx_honeybee_desc* desc, float* debug_planes, void* stream);

/* ---- geometric resampling: cv2.resize / cv2.remap / cv2.Sobel call sites ----------------
 * OpenCV semantics restated from its published algorithm (OpenCV is not available to pin against). */

/* cv2.resize(src, (Wd, Hd), interpolation) on an HWC image (uv_helpers.py:57-64,94,172,182;
 * cat_widevision_utils.py:26).  dtype 0 = float32 (interp 1 LINEAR, 2 CUBIC a=-0.75, 3 AREA), 2 = uint8 (LINEAR,
 * 11-bit fixed point). */
int avx_resize_hwc(avx_ctx* ctx, const void* src, int dtype, int H, int W, int C, void* dst, int Hd, int Wd, int interp, void* stream);

/* animal_fov_binocular_warp (cat_widevision_utils.py:46-99) on a uint8 frame: get_normalized_image, two
 * cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) with per-column x maps / per-row y map, cos^2 blend weights, clip:
 * -> float32 HWC in [0,1].  Host tables: xL, xR, wL, wR (Wo floats each), ymap (Ho floats). */
int avx_binocular_warp_u8(avx_ctx* ctx, const uint8_t* in_hwc, int H, int W, const float* xL_host, const float* xR_host, const float* ymap_host,
                          const float* wL_host, const float* wR_host, int Ho, int Wo, float* out_hwc_f32, void* stream);

/* VideoRenderer.make_split_frame (renderers/video.py:198-245) for two frames of the same size, without the
 * Hershey-font labels: left half `original`, right half `modified`, optional 1-px white seam at W//2.
 * out_hwc may alias modified_hwc (each byte is read before it is written by the same thread). */
int avx_split_compose_u8(avx_ctx* ctx, const uint8_t* original_hwc, const uint8_t* modified_hwc, uint8_t* out_hwc,
                         int H, int W, int draw_seam, void* stream);

/* VideoRenderer._draw_label (renderers/video.py:160-196) on a device-resident uint8 frame, in place: the 60 % black box
 * (inside box_xyxy, inclusive corners: out = saturate_cast<uchar>(0.4 * in)), then the text as stroke segments -- black at
 * outline_thickness, white at text_thickness (cv2.putText x 2, :194-195).  segments_host: n x 6 floats per segment
 * {ax, ay, bx - ax, by - ay, 1 / |b - a|^2 (0 for a point), 0} in pixel coordinates, built by the host from the Hershey
 * simplex stroke tables with the reference's geometry (renderers/labels.py).  Anti-aliasing is analytic distance coverage, not
 * OpenCV's LINE_AA scan converter (label pixels: parity unpinned; geometry pinned).  slot (0 | 1) names which of two cached
 * device copies of the segment table the call may reuse (a frame has two labels). */
int avx_draw_label_u8(avx_ctx* ctx, uint8_t* img_hwc, int H, int W, const int box_xyxy[4], const float* segments_host, int n_segments,
                      float outline_thickness, float text_thickness, int slot, void* stream);

/* cv2.remap(src, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT, borderValue) on K float32 planes that share two
 * per-pixel float32 maps (anableps.py:217-226): coordinates quantised to 1/32 px like OpenCV. */
int avx_remap_linear_planes(avx_ctx* ctx, const float* src_planes, int K, int H, int W, const float* mapx_dev,
                            const float* mapy_dev, float* dst_planes, float border_value, void* stream);

/* cv2.Sobel(plane, CV_32F, 1,0 / 0,1, ksize=3, BORDER_REFLECT101) -> gx, gy (mantis_shrimp.py:122-131). */
int avx_sobel3_plane(avx_ctx* ctx, const float* plane, int H, int W, float* gx, float* gy, void* stream);

/* MantisShrimp.visualize (animals/mantis_shrimp.py:143-279) for one uint8 frame, all passes on the device:
 * decode -> panorama warp -> baseline encode; RGB->HSI (optionally at hsi_scale) x band windows -> safe_norm stack
 * -> P95 -> barcode tint; red kill / haze / pre-soft blur; Sobel polarisation gain; unsharp; barcode blend;
 * scanlines; peripheral blur blend; encode.  All O(bands), O(H), O(W) tables are built by the host. */
typedef struct avx_mantis_desc {
    uint32_t struct_size;
    int32_t n_bands;                   /* N <= 16 (mantis_shrimp.py:49-60: 10)                                  */
    const float* band_matrix_host;     /* N x 3: band-pass windows folded with the analytic lobes (input-channel order) */
    const float* band_lut_host;        /* N x 3: the hue LUT of :175-197                                        */
    int32_t n_wavelengths;             /* B: the per-wavelength tables below serve pixels with a negative channel (cubic
                                          overshoot), where clamp_min(0) of classic_rgb_to_hsi.py:81 is not the identity */
    const float* lobe_gains_host;      /* B x 3 lobe gains (column j multiplies input channel j)                 */
    float lobe_denom;                  /* the scalar denominator of classic_rgb_to_hsi.py:73-79 (+1e-8)          */
    const float* band_weights_host;    /* N x B band-pass weights (uv_helpers.py:125-139)                        */
    int32_t pano_new_w;                /* widened width of panorama_warp (0 / W: no warp)                        */
    int32_t hsi_small_h, hsi_small_w;  /* classic_rgb_to_hsi_scaled size (0: full resolution)                    */
    float red_keep;                    /* float32(1 - red_kill)                                                 */
    float haze, haze_keep, haze_tint[3];
    int32_t pre_soft_ksize; const double* pre_soft_taps_host;
    float cos2_global, sin2_global;    /* float32((1 - mix) * cos/sin(2 * evec_angle))                          */
    float orientation_mix, pol_linear_strength, pol_linear_gamma, pol_circular_strength;
    int32_t unsharp_ksize; const double* unsharp_taps_host; float unsharp_amount;
    float barcode_saturation, barcode_opacity, winner_take_most;
    const float* rows_host;            /* H: 0.5 + 0.5 sin(2 pi f y) (:257-258)                                  */
    float scan_row_gain; int32_t scan_ksize; const double* scan_taps_host;
    int32_t periph_ksize; const double* periph_taps_host;
    const float* xx_host; const float* yy_host; /* W, H: linspace(-1, 1) (:270-271)                              */
    float periph_radius, periph_softness;
    const float* lin_hwc_in;           /* optional (float frames): device H x W x 3 float32, already srgb_to_linear(to_float01(frame));
                                          in_hwc is ignored then                                                        */
    int32_t out_float;                 /* 1: both outputs are float32 H x W x 3 buffers holding the sRGB value (float frames) */
} avx_mantis_desc;

int avx_mantis_u8(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_baseline_hwc, uint8_t* out_hwc, int H, int W,
                  const avx_mantis_desc* desc, void* stream);
/* The same for n_frames contiguous uint8 frames (baselines and outputs contiguous too): independent frames run on up to four
 * internal streams forked from / joined into `stream` (renderers/video.py's frame loop hands the library a batch). */
int avx_mantis_u8_batch(avx_ctx* ctx, const uint8_t* in_hwc, uint8_t* out_baseline_hwc, uint8_t* out_hwc, int n_frames, int H, int W,
                        const avx_mantis_desc* desc, void* stream);

/* ---- stages of the other UV species, each on device pointers so that a species is a sequence of asynchronous
 * calls on one stream (animal-vision_amd/planevm.py) --------------------------------------------------------- */
/* Steps 1-2 shared by every UV species but HoneyBee (animals/reindeer.py:83-98, goldfish.py:100-113, ...):
 * to_float01 + srgb_to_linear, panorama_warp to pano_new_w (<= W: no warp) and the uint8 baseline
 * (baseline_hwc_out may be NULL).  lin_hwc_out: H x W x 3 float32. */
int avx_uv_front_u8(avx_ctx* ctx, const uint8_t* in_hwc, int H, int W, int pano_new_w, float* lin_hwc_out,
                    uint8_t* baseline_hwc_out, void* stream);
/* uv_helpers.panorama_warp (:84-99) of an H x W x 3 float32 image: INTER_CUBIC widen to new_w (> W), centre crop. */
int avx_panorama_warp_f32(avx_ctx* ctx, const float* src_hwc, int H, int W, int new_w, float* dst_hwc, void* stream);
/* classic_rgb_to_hsi / classic_rgb_to_hsi_scaled (uv_helpers.py:155-183) followed by N x integrate_band
 * (:142-146): H x W x 3 linear float32 -> H x W x N float32 (raw band integrals; safe_norm is the caller's). */
typedef struct avx_band_stack_desc {
    uint32_t struct_size;
    int32_t n_bands;                   /* N <= 16                                                              */
    const float* band_matrix_host;     /* N x 3 (see avx_mantis_desc)                                          */
    int32_t n_wavelengths;
    const float* lobe_gains_host;      /* B x 3                                                                */
    float lobe_denom;
    const float* band_weights_host;    /* N x B                                                                */
    int32_t small_h, small_w;          /* reduced size of the hsi_scale route (0: full resolution)             */
} avx_band_stack_desc;
int avx_band_stack(avx_ctx* ctx, const float* lin_hwc, int H, int W, const avx_band_stack_desc* desc,
                   float* stack_hwk_out, void* stream);
/* avx_percentile with the result left on the device (one double, the float32 value NumPy would return). */
int avx_percentile_dev(avx_ctx* ctx, const float* data_dev, size_t n, double q, double* out_dev, void* stream);
/* `count` (<= 16) independent percentiles; up to four are resolved by the same three radix passes (the arrays of
 * pointers / sizes are host arrays of device pointers). */
int avx_percentiles_dev(avx_ctx* ctx, int count, const float* const* data_dev, const size_t* n, const double* q,
                        double* const* out_dev, void* stream);

/* ---- fused elementwise programs over float32 planes (csrc/ew.hip) --------------------------------------------
 * The remaining UV species (animals/reindeer.py:83-135, goldfish.py:86-180, ... ) are NumPy expression chains over
 * HxW planes.  One avx_ew_run = one launch evaluating a whole chain per pixel: a register program (32 float32
 * registers) whose LOAD/STORE touch only the planes that enter/leave the chain; frame-wide reductions (ndarray.min /
 * max / sum / mean) accumulate alongside and land in a device-side table of doubles that later programs read with
 * AVX_EW_SCALAR -- nothing visits the host.  Every instruction is one IEEE float32 operation (no fusing). */
enum { AVX_EW_MAX_INSN = 384, AVX_EW_MAX_PLANES = 24, AVX_EW_MAX_REGS = 32, AVX_EW_MAX_ACC = 16 };
enum {  /* opcodes: dst = op(a, b); imm = constant bits / plane index / scalar slot / third register of SELECT */
    AVX_EW_CONST = 1, AVX_EW_SCALAR, AVX_EW_LOAD, AVX_EW_STORE,
    AVX_EW_ADD, AVX_EW_SUB, AVX_EW_MUL, AVX_EW_DIV, AVX_EW_MIN, AVX_EW_MAX, AVX_EW_POW, AVX_EW_ATAN2,
    AVX_EW_NEG, AVX_EW_ABS, AVX_EW_SQRT, AVX_EW_EXP, AVX_EW_LOG, AVX_EW_SIN, AVX_EW_COS, AVX_EW_FLOOR, AVX_EW_CEIL,
    AVX_EW_CLIP01, AVX_EW_TANH,
    AVX_EW_LT, AVX_EW_LE, AVX_EW_GT, AVX_EW_GE, AVX_EW_EQ, AVX_EW_AND, AVX_EW_OR, AVX_EW_NOT,
    AVX_EW_SELECT,                      /* dst = a != 0 ? b : reg[imm & 0xff]  (np.where)                          */
    AVX_EW_ACCMIN, AVX_EW_ACCMAX, AVX_EW_ACCSUM   /* dst (an accumulator register) op= a                           */
};
/* Binary opcodes (ADD .. OR) may take ONE operand from the instruction itself: op | AVX_EW_IMM_A: a = float32(imm),
 * op | AVX_EW_IMM_B: b = float32(imm) (the weak Python scalars of the reference's expressions). */
enum { AVX_EW_OPCODE_MASK = 0x3f, AVX_EW_IMM_A = 0x40, AVX_EW_IMM_B = 0x80 };
enum {  /* plane kinds: element i of the frame (i = y*W + x) is at ptr[i*stride] unless stated                    */
    AVX_EW_PLANE_F32 = 0,               /* float32, load/store (stride 3 + offset pointer = one channel of HWC)      */
    AVX_EW_PLANE_U8,                    /* uint8 -> float value of the byte, load only                              */
    AVX_EW_PLANE_U8_LUT,                /* uint8 -> srgb_to_linear(to_float01(byte)) (uv_helpers.py:15-37), load only */
    AVX_EW_PLANE_COL,                   /* float32 vector of W entries broadcast down the rows: ptr[x]              */
    AVX_EW_PLANE_ROW,                   /* float32 vector of H entries broadcast along the rows: ptr[y]             */
    AVX_EW_PLANE_U8_ENC                 /* store only: from_float01(linear_to_srgb(clip(v,0,1)), uint8) (:25-44)    */
};
enum { AVX_EW_ACC_MIN = 0, AVX_EW_ACC_MAX, AVX_EW_ACC_SUM, AVX_EW_ACC_MEAN };
typedef struct avx_ew_insn { uint8_t op, dst, a, b; uint32_t imm; } avx_ew_insn;
typedef struct avx_ew_plane { void* ptr; int32_t stride; int32_t kind; } avx_ew_plane;
typedef struct avx_ew_program {
    uint32_t struct_size;
    int32_t H, W;
    int32_t n_insn; const avx_ew_insn* insn_host;
    int32_t n_planes; const avx_ew_plane* planes_host;   /* device pointers                                          */
    int32_t n_acc; const int32_t* acc_host;              /* n_acc x {register, AVX_EW_ACC_*, scalar slot}            */
    double* scalars_dev; int32_t n_scalars;              /* the device-side scalar table                             */
} avx_ew_program;
int avx_ew_run(avx_ctx* ctx, const avx_ew_program* program, void* stream);
/* Programs whose structure was recorded when csrc/ew_gen.hip was generated (tools/gen_ew_kernels.py) run as their own straight-line
 * kernels; others are interpreted.  Returns the number of generated kernels; *hits / *misses count avx_ew_run calls of either kind. */
int avx_ew_spec_stats(unsigned long long* hits, unsigned long long* misses);

/* ---- MST++ helpers (ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py) ----------------------------
 * Hand-written kernels for the memory-bound parts of the forward pass; PyTorch-ROCm keeps the dense GEMMs/convs.
 * Pointers are torch tensors' data_ptr(); dtype 0 = float32, 1 = float16; all on `stream` (torch's current stream). */

/* MS_MSA :127-129 in one pass over qkv (n_pix x 3C, row-major: q | k | v): per head h (d = C/heads <= 32)
 *   gram[h][i][j] = sum_n k[n][h*d+i] * q[n][h*d+j]   (32x32 padded, float32)
 *   nq[c] = ||q[:,c]||_2, nk[c] = ||k[:,c]||_2        (what F.normalize(dim=-1) divides by) */
int avx_mst_gram(avx_ctx* ctx, const void* qkv, int dtype, size_t n_pix, int C, int heads, float* gram, float* nq, float* nk, void* stream);

/* Fused matrix-core kernels for an MSAB block (csrc/mst_mfma.hip), float16 activations, channel groups 32 wide
 * (C = 32, 64 or 128).  wpack: the weights in MFMA fragment order, built by ml/mst_plus_plus.py::pack_fragments.
 *
 * avx_mst_qkv_gram: MS_MSA to_q/to_k/to_v (MST_Plus_Plus.py:118-120) + the Gram matrix k^T q over all n_pix pixels per
 * head and the L2 norm of every q / k column (:127-129) in one pass over x (n_pix x C); only v (n_pix x C) is
 * written.  gram: heads x 32 x 32 float32 ([i][j] = k_i . q_j), nq / nk: C float32.
 *
 * avx_mst_ln_gemm_gelu: PreNorm LayerNorm (:57-65; statistics over the 31 real channels of each group) -> FeedForward's
 * first 1x1 conv (:145, C -> 4C) -> GELU (:146); out: rows x 4C float16. */
int avx_mst_qkv_gram(avx_ctx* ctx, const void* x, const void* wpack, size_t n_pix, int C, void* v_out, float* gram, float* nq, float* nk,
                     void* stream);
/* avx_mst_qkv_gram on v_mfma_f32_32x32x16_f16: the same results (products are exact, float32 sums in another order) from wpack16 =
 * ml/mst_plus_plus.py::pack_qkv16 ([3C/32 tiles][C/16 steps][64 lanes] x 8 halves; q / k tiles in natural channel order, v tiles row-permuted). */
int avx_mst_qkv_gram16(avx_ctx* ctx, const void* x, const void* wpack16, size_t n_pix, int C, void* v_out, float* gram, float* nq, float* nk,
                       void* stream);
int avx_mst_ln_gemm_gelu(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* wpack, size_t rows, int C,
                         void* out, void* stream);

/* FeedForward's depthwise 3x3 -> GELU -> second 1x1 conv (4C -> C) (:147-149) plus the block's residual (:184):
 * out = residual + gelu(dwconv3x3(hidden)) @ W2, hidden B x H x W x 4C float16 (never rewritten), residual / out
 * B x H x W x C float16 (out may alias residual).  C = 32 or 64. */
int avx_mst_dw_gemm_add(avx_ctx* ctx, const void* hidden, const float* w_c9, const void* w2pack, const void* residual, void* out, int B, int H,
                        int W, int C, void* stream);

/* uint8 RGB frame -> the output of MST_Plus_Plus.conv_in (:275) in one kernel: x = frame / 255 rounded to float16, reflect-padded by
 * (pad_top, pad_bottom, pad_left, pad_right) like predict_torch.py:171-183, 3x3 conv with zero padding, 31 channels stored 32 wide:
 * out (H + pads, W + pads, 32) float16.  w_27x32_dev: device float32 [27][32], row k = (ky * 3 + kx) * 3 + c, column = output channel. */
int avx_mst_conv_in_u8(avx_ctx* ctx, const uint8_t* frame_hwc, int H, int W, int pad_top, int pad_bottom, int pad_left, int pad_right,
                       const float* w_27x32_dev, void* out, void* stream);

/* avx_mst_conv3x3_add through an LDS halo tile (16 x 16 pixels, every input pixel fetched once): wpack16 = the nine taps' C x C
 * weights as v_mfma_f32_32x32x16_f16 A fragments ([9][2][64][8] float16, ml/mst_plus_plus.py::pack_fragments16 per tap). */
int avx_mst_conv3x3_lds(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, void* out, int B, int H, int W, int C, void* stream);
/* avx_mst_conv3x3_lds with the Gram pass of the MSAB block that follows (avx_mst_qkv_gram16 with v_out = NULL on the conv's output, wqk16 = the q and k
 * tiles of pack_qkv16) as its epilogue: the output is not read again for it.  One frame (the Gram matrix is per frame). */
int avx_mst_conv3x3_lds_gram(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, void* out, int H, int W, int C, const void* wqk16, float* gram,
                             float* nq, float* nk, void* stream);
/* The same for the decoder's step back to full resolution: avx_mst_convt2x2_fuse at C = 64 with the following block's Gram pass as its epilogue. */
int avx_mst_convt2x2_fuse_gram(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int H, int W, int C,
                               const void* wqk16, float* gram, float* nq, float* nk, void* stream);
/* MST.encoder_layers[i][1] at full resolution (MST_Plus_Plus.py:206-208): Conv2d(C -> 2C, 4, stride 2, padding 1, bias=False) on (B, H, W, C)
 * float16 -> (B, H/2, W/2, 2C), C = 32; wpack16 = [16 taps][2C/32][C/16] fragments (pack_fragments16 of W[:, :, ky, kx]^T). */
int avx_mst_down4x4(avx_ctx* ctx, const void* x, const void* wpack16, void* out, int B, int H, int W, int C, void* stream);

/* The tail of the first half of an MSAB block in one pass (MS_MSA :104-106, :132-137; MSAB :183):
 * out = v @ M + bias + dw3x3(gelu(dw3x3(v))) + x on (B, H, W, C) float16 tensors, C = 32 or 64, with M the per-frame C x C matrix
 * of avx_mst_attn_pack16 (so B frames must share M: the host calls it per frame).  v on the tile's 20 x 20 halo region and the
 * GELU'd first conv live in LDS; v and x are read once, out written once.  taps*_9xc: depthwise weights, float16, tap-major
 * [9][C].  out may be x, not v. */
int avx_mst_attn_pack16(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                        void* stream);
int avx_mst_attn_tail(avx_ctx* ctx, const void* v, const void* x, const void* mpack16, const void* taps1_9xc, const void* taps2_9xc, const float* bias,
                      void* out, int B, int H, int W, int C, void* stream);
/* The same with v = float16(x W_v^T) formed inside the kernel (wvpack16: W_v in pack_fragments16(transposed) order): v never exists in
 * HBM and avx_mst_qkv_gram may be called with v_out = NULL.  out != x. */
int avx_mst_attn_tail_x(avx_ctx* ctx, const void* x, const void* wvpack16, const void* mpack16, const void* taps1_9xc, const void* taps2_9xc,
                        const float* bias, void* out, int B, int H, int W, int C, void* stream);
/* The tail with BOTH depthwise convs of pos_emb and the projection on the matrix pipe (csrc/mst_fused.hip::k_mst_attn_tail_mx; round 3): a
 * v_mfma_f32_16x16x32_f16 result tile is 8 channels x 2 vertically adjacent rows x 16 pixels, a depthwise 3x3 conv is three MFMAs per tile
 * (dw1pack / dw2pack: ml/mst_plus_plus.py::pack_dw_mfma fragments of MS_MSA.pos_emb's two weights, :104-106), and v @ M accumulates into the
 * same registers through block-diagonal fragments (mpack_mx: avx_mst_attn_pack_mx).  v = float16(x W_v^T) is formed on the tile's halo as in
 * avx_mst_attn_tail_x.  C = 32, 64 or 128; out != x. */
int avx_mst_attn_tail_mx(avx_ctx* ctx, const void* x, const void* wvpack16, const void* mpack_mx, const void* dw1pack, const void* dw2pack,
                         const float* bias, void* out, int B, int H, int W, int C, void* stream);
/* avx_mst_attn_pack16's matrix M in the fragment order avx_mst_attn_tail_mx takes: [C/8][C/16][64][8] float16 (4 C^2 entries, half of them
 * the zeros of the block-diagonal form). */
int avx_mst_attn_pack_mx(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C, void* mpack,
                         void* stream);


/* MST_Plus_Plus.conv_out + x (:289-292) with the spectral integration that follows it in the honeybee route (honeybee.py:126-135) as its epilogue: the cube is
 * never written.  x / add: (H, W, 32) float16; weights_host: 3 x 32 (illuminant folded in, band 31 zero); planes_out: 3 x Hc x Wc float32 of the frame cropped at
 * (crop_t, crop_l) (the predict harness pads frames to multiples of 16, predict_torch.py:171-183); partials_out: up to 3 x CUs x 3 records of 16 bytes, *n_partials
 * receives how many triples were written.  Feed both to avx_honeybee_u8 (source 2).  The planes equal avx_spectral_integrate's on the cube bit for bit. */
int avx_mst_conv3x3_lds_spectral(avx_ctx* ctx, const void* x, const void* wpack16, const void* add, int H, int W, int C, const float* weights_host, int crop_t,
                                 int crop_l, int Hc, int Wc, float* planes_out, void* partials_out, int* n_partials, void* stream);

/* The whole second half of an MSAB block in one kernel (MST_Plus_Plus.py:57-65 PreNorm, :141-158 FeedForward, :184 residual):
 * out = x + W2 gelu(dw3x3(gelu(W1 layernorm(x)))) on a (B, H, W, C) float16 tensor, C = 32 or 64 (31-channel groups stored
 * 32 wide; LayerNorm statistics over the real channels).  The 4C-channel hidden map lives in LDS only (16 x 16 pixel tiles,
 * 1-pixel halo recomputed).  w1pack / w2pack: the 1x1 convs' weights as v_mfma_f32_32x32x16_f16 A fragments
 * ([N/32][K/16][64][8] float16, ml/mst_plus_plus.py::pack_fragments16); taps_9xhid: the depthwise 3x3 weights, float16,
 * tap-major [9][4C].  out must not alias x (tiles read their neighbours' rows). */
int avx_mst_ffn_fused(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* taps_9xhid,
                      const void* w2pack, void* out, int B, int H, int W, int C, void* stream);
/* The same with the depthwise 3x3 conv (MST_Plus_Plus.py:151) on the matrix pipe: a result tile of v_mfma_f32_16x16x32_f16 is 8 channels x
 * 2 vertically adjacent output rows x 16 pixels, the K slots are 4 input rows x 3 column shifts of those 8 channels (three MFMAs per 256
 * outputs).  dwpack: the depthwise weights as A fragments, [4C/8 octets][3 shifts][64 lanes][8] float16
 * (ml/mst_plus_plus.py::pack_dw_mfma).  Same result up to the summation order of the nine taps (float32 accumulation in both). */
int avx_mst_ffn_fused_mx(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, float eps, const void* w1pack, const void* dwpack,
                         const void* w2pack, void* out, int B, int H, int W, int C, void* stream);
/* The matrix-pipe kernels evaluate GELU on PRESCALED operands (csrc/mst_common.h: x / s enters, gelu(x) / s leaves; two packed instructions fewer per pair).
 * s = avx_mst_gelu_prescale() (4: a power of two, exact in float16; 1 when the library was built without the prescaled form), and the caller scales the
 * weights around each GELU accordingly: avx_mst_ffn_fused_mx takes W1 / s in w1pack, the depthwise weights unscaled and W2 * s in w2pack -- and, since its
 * LayerNorm only centres and scales (round 3), LayerNorm's gamma and beta FOLDED into w1pack as well: row k of W1 (input channel k) times gamma[k], and row 31
 * (the first group's padding channel, whose operand the kernel sets to 1) = W1^T beta; its gamma / beta arguments are ignored;
 * avx_mst_attn_tail_mx takes pos_emb's first depthwise weights / s in dw1pack and the second * s in dw2pack. */
float avx_mst_gelu_prescale(void);

/* out = [add +] a @ W [+ a2 @ W2] for (rows x C) float16 tensors and C x C weights in fragment order (out may alias
 * add; a2 / W2 and add may be NULL): MS_MSA's `proj(attn @ v)` collapsed to one matrix per frame (:132-135)
 * accumulated onto pos_emb(v) + x, and the decoder's 1x1 fusion conv over [up | skip] (:257) without the concatenation. */
int avx_mst_rowgemm_add(avx_ctx* ctx, const void* a, const void* wpack, const void* a2, const void* wpack2, const void* add, void* out, size_t rows,
                        int C, void* stream);

/* MS_MSA :127-135 after avx_mst_qkv_gram: attn = softmax over the 31 real columns of gram / (nk nq^T) * rescale per
 * head, M = blockdiag(attn_h^T) @ W_proj^T (wproj_t: C x C float32, row = input channel), written as float16 in the
 * fragment order avx_mst_rowgemm_add takes. */
int avx_mst_attn_pack(avx_ctx* ctx, const float* gram, const float* nq, const float* nk, const float* rescale, const float* wproj_t, int C,
                      void* mpack, void* stream);

/* Dense 3x3 conv, zero padding 1, C -> C on B x H x W x C float16 (C = 32), optionally + add: MST.embedding / mapping
 * (:199, :228 with its `+ x`) and conv_out (:277, :291).  wpack: 9 taps x (C x C) in fragment order. */
int avx_mst_conv3x3_add(avx_ctx* ctx, const void* x, const void* wpack, const void* add, void* out, int B, int H, int W, int C, void* stream);

/* MS_MSA's pos_emb (:104-106: depthwise 3x3 -> GELU -> depthwise 3x3) in one pass, + residual + bias[c] (both optional):
 * out = dw2(gelu(dw1(v))) + residual + bias on B x H x W x C float16 (C = 32, 64, 128); taps C x 9 float32. */
int avx_mst_posemb(avx_ctx* ctx, const void* v, const float* w1_c9, const float* w2_c9, const void* residual, const float* bias, void* out, int B,
                   int H, int W, int C, void* stream);

/* ConvTranspose2d(C -> C/2, kernel 2, stride 2) + bias on B x H x W x C float16 -> B x 2H x 2W x C/2 (MST decoder, :214,
 * :256): four independent C x C/2 products, one per output-pixel parity.  wpack: 4 taps (dy*2 + dx) in fragment order. */
int avx_mst_convt2x2(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, void* out, int B, int H, int W, int C, void* stream);
/* The same with the decoder's 1x1 fusion conv over [up | skip] (:257) folded in: wpack / bias carry the taps composed with the
 * conv's `up` half (W_tap @ W_up^T, b @ W_up^T, built by the host), `skip` is the (B, 2H, 2W, C/2) skip tensor and wskip the conv's
 * `skip` half in fragment order ([C/64][C/16][64] x 4 float16, pack_fragments(..., True)): out = convT'(x) + skip @ W_skip^T. */
int avx_mst_convt2x2_fuse(avx_ctx* ctx, const void* x, const void* wpack, const float* bias, const void* skip, const void* wskip, void* out, int B, int H,
                          int W, int C, void* stream);

/* nn.Conv2d(C, C, 3, 1, 1, groups=C, bias=False) on a channels-last (B,H,W,C) tensor (pos_emb :104-106,
 * FeedForward :147), float32 accumulate; w_c9: C x 9 float32 (weight.reshape(C, 9)); gelu_out: exact-erf GELU. */
int avx_dwconv3x3_nhwc(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int dtype, int B, int H, int W, int C, int gelu_out, void* stream);
/* The same with a fused epilogue (float16 only, C % 8 == 0): y = dwconv(x) [GELU] + residual + bias[c]; residual (same
 * shape as y) and bias (C float32) may be NULL.  MS_MSA's `out_c + pos_emb(v)` and the block's `+ x` (:137, :183). */
int avx_dwconv3x3_nhwc_add(avx_ctx* ctx, const void* x, const float* w_c9, void* y, int B, int H, int W, int C, int gelu_out,
                           const void* residual, const float* bias, void* stream);

/* nn.LayerNorm(C) (PreNorm :57-65) over the last dim of (rows x C): biased variance, float32 statistics. */
int avx_layernorm_rows(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int dtype, size_t rows, int C, float eps, void* stream);
/* The same over channel groups stored with zero padding (MST++'s 31-channel groups kept 32 wide so that every row
 * is 64-byte aligned): channel c takes part iff (c % group) < real; padding channels are written as zero. */
int avx_layernorm_rows_grouped(avx_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, int dtype, size_t rows, int C, float eps,
                               int group, int real, void* stream);

/* The constant tables compiled into the library (reference outputs, see csrc/srgb_tables.h):
 * which = 0: 256 x f32 decode LUT; 1: 255 x f32 encode thresholds; 2: 255 x f64 encode thresholds.
 * Copies min(capacity, size) bytes to dst_host and returns the table's size in bytes. */
int avx_get_table(int which, void* dst_host, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* AVX_H */
