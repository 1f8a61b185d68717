/* gsraster.h -- C ABI of libgsraster.so, the MI355X (gfx950) differentiable 3D-Gaussian-splat rasteriser.
 *
 * This is the drop-in boundary for the one hot path of poloclub/3d-gaussian-splat-attack: what the
 * reference reaches through `from diff_gaussian_rasterization import GaussianRasterizationSettings,
 * GaussianRasterizer` (reference gaussian_renderer/__init__.py:14) and calls at
 * gaussian_renderer/__init__.py:36-49 (settings), :51 (constructor) and :86-95 (forward); the backward
 * is reached by `loss.backward()` at attack.py:494.  The third-party CUDA extension behind that import
 * is not vendored in the reference; the entry points below are what a Python binding for it needs.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers (float32 / int32, contiguous) + sizes + a hipStream_t
 *     passed as void*;
 *   - every call enqueues on the given stream of the CURRENT hip device; the only host synchronisation is
 *     inside gsr_forward: the host polls a pinned word for the number of (tile, Gaussian) pairs, which a kernel early
 *     in the forward writes there, to size the sort buffers (GSR_FLAG_ASYNC_COUNT removes even that);
 *   - return value 0 = success, otherwise a GSR_ERR_* code and gsr_last_error() describes it
 *     (thread-local string);
 *   - the caller owns all inputs / outputs / gradient buffers and must keep the INPUT tensors of
 *     gsr_forward alive and unmodified until gsr_backward / gsr_ctx_free for that context;
 *     the library owns an internal caching workspace pool per device and the opaque GsrCtx.
 *   - gradient outputs are fully written (zeros for culled Gaussians): no pre-zeroing needed.
 *
 * Non-finite and extreme inputs (a position or scale step of the attack can produce them: reference attack.py:500-511).
 * The published kernels turn them into undefined float -> int conversions and fully opaque garbage splats; here:
 *   - a Gaussian whose projected conic, pixel centre or view depth is not finite -- NaN or +-inf in its mean, scale,
 *     rotation or covariance, or finite values whose products overflow float32 (exp(log_scale) beyond ~1e19) -- and a
 *     Gaussian whose (activated) opacity is NaN is CULLED: radius 0, no pairs, zero gradients, exactly as if it were behind
 *     the camera.  Every other Gaussian renders and differentiates bit for bit as if the bad one were not in the scene;
 *   - a finite but enormous footprint keeps the published behaviour (its tile rect is the rect clamped to the image: it
 *     reaches every tile) with the radius saturated at 2^24 pixels; num_rendered never exceeds (Gaussians with radius > 0)
 *     x (tiles of the image), and more than 2^31 - 1 pairs in one forward is GSR_ERR_NOMEM, not a wrapped counter;
 *   - a NaN colour (NaN spherical-harmonics coefficients) composites as 0; an infinite one reaches the pixels of the tiles
 *     that Gaussian touches and nothing else; a zero quaternion is the identity rotation times zero (a point: the 0.3 px^2
 *     dilation renders it), as F.normalize's 1e-12 floor makes it in the reference.
 * tests/test_gpu_nonfinite.py; the scalar arithmetic also runs under -fsanitize=undefined,float-cast-overflow on the host
 * (tests/test_host_math.py).
 */
#ifndef GSRASTER_H_
#define GSRASTER_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 601 /* 0.6.1: the batch entry points, gsr_pgd_step_multi; + gsr_forward_raw2_batch, gsr_ctx_rerender on batch contexts */
#define GSR_NUM_OBJECTS 16 /* object-feature channels, reference scene/gaussian_model.py:52 */

enum {
  GSR_OK = 0,
  GSR_ERR_INVALID = 1,  /* bad argument combination / sizes */
  GSR_ERR_DEVICE = 2,   /* a HIP call failed (message names the stage) */
  GSR_ERR_NOMEM = 3,
  GSR_ERR_STATE = 4,    /* e.g. backward called twice on a released context */
  GSR_ERR_OVERFLOW = 5  /* GSR_FLAG_ASYNC_COUNT only: the forward emitted more pairs than the guessed capacity */
};

/* flags */
#define GSR_FLAG_NONE 0u
/* Keep every (tile, Gaussian) pair of the 3-sigma tile rect.  By default pairs whose Gaussian cannot reach
 * alpha >= 1/255 on any pixel of the tile are dropped before the sort: outputs are identical, only
 * num_rendered-internal work shrinks.  The flag exists for A/B tests of exactly that claim. */
#define GSR_FLAG_NO_CULL 1u
/* Per-call overrides of launch heuristics -- results do not depend on them (bitwise for the tile map; the tile split
 * only changes which wave owns which strip, and how many partial rows a pair has); they exist so that tests can drive
 * every kernel instantiation on small inputs.  0 in a field = the library's own choice from the tile count.
 *   GSR_FLAG_FWD_SPLIT(n)  n = 1|2|4 pixels per lane in the forward compositor (4|2|1 waves per 16x16 tile)
 *   GSR_FLAG_BWD_SPLIT(n)  n = 2|4   pixels per lane in the backward compositor
 *   GSR_FLAG_TILE_MAP(m)   m = 0..3  block -> tile map: 0 image order, 1 one band per XCD, 2 32-tile bands round
 *                                    robin, 3 longest list first (default)
 *   GSR_FLAG_NO_SEGMENTS             never split a long tile list over several waves (see gsr_backward)
 *   GSR_FLAG_FWD_SHARED              the forward waves of a tile form one workgroup that stages each batch of the list
 *                                    once for all of them (two or four waves per tile only): 38 % less gather traffic,
 *                                    14 % slower on the benchmark scene -- off by default */
#define GSR_FLAG_FWD_SPLIT(n) ((uint32_t)((n) == 1 ? 1u : (n) == 2 ? 2u : (n) == 4 ? 3u : 0u) << 4)
#define GSR_FLAG_BWD_SPLIT(n) ((uint32_t)((n) == 2 ? 1u : (n) == 4 ? 2u : 0u) << 8)
#define GSR_FLAG_TILE_MAP(m) ((uint32_t)(((m) & 3u) + 1u) << 12)
#define GSR_FLAG_NO_SEGMENTS (1u << 16)
#define GSR_FLAG_FWD_SHARED (1u << 17)
/* Asynchronous pair count.  By default gsr_forward waits (once, early, behind work it has already enqueued) for the
 * number of (tile, Gaussian) pairs before it sizes the pair buffers -- the only host synchronisation of the path.
 * With this flag a forward whose (P, H, W) has been rendered before on this device sizes them from the count that
 * earlier forward saw, plus 25 % + 64 K pairs of head-room, and never waits: *num_rendered is then -1 (ask
 * gsr_ctx_info(ctx, 0) later).  If the scene emits more pairs than that capacity, nothing is composited, out_color is
 * filled with NaN, gsr_backward on the context returns GSR_ERR_OVERFLOW, and the next forward counts synchronously
 * again.  Opt-in: a caller that never looks at the image or calls gsr_backward would not notice an overflow. */
#define GSR_FLAG_ASYNC_COUNT (1u << 18)
/* gsr_forward runs the colour half of its per-Gaussian stage (SH -> RGB) on a library-owned side stream, beside the
 * binning chain of the same view, and joins it before compositing (events; the caller's stream semantics are
 * unchanged).  This flag keeps everything on the caller's stream. */
#define GSR_FLAG_NO_SIDE_STREAM (1u << 19)
/* GSR_FLAG_NEEDLE_DOUBLE: splats whose dilated 2D covariance has eigenvalues more than 256 apart ("needles") get their
 * conic -- and, in the backward, their whole per-Gaussian chain rule -- from the published chain evaluated in double on
 * the same float32 inputs (the conic is rounded to its float32 record so that the form along the long axis is preserved).  The float32 chain leaves 1e-7 x that ratio in every conic entry, which a needle's exponent
 * (terms of radius^2 cancelling to O(1)) and gradients (cancelling once more) amplify: a 1500:1 needle's dL/dmean2D is 2.3 %
 * off in float32 -- in the reference's kernels as in any float32 statement of the formula.  Integer decisions (radius, tile
 * rect, culls) and every ordinary splat are unchanged.  Opt-in: it costs the geometry kernel 20 registers and a view
 * 1.3 % (one stream) to 2.8 % (four) on S-nyc-1M, whose splats are not needles (EXPERIMENTS.md, round 5). */
#define GSR_FLAG_NEEDLE_DOUBLE (1u << 20)

/* GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY: gsr_forward / gsr_forward_raw are given sh_objs but out_objects == NULL -- the object
 * channels are not composited (the caller knows they come out as zeros: all features are zero, which is what the reference's
 * combine_splats gives every Gaussian of an attack scene, scene/gaussian_model.py:528 -- or does not look at them), but the
 * context keeps the features, so that a gsr_backward WITH grad_objects still produces dL/dsh_objs (and the features'
 * share of dL/dalpha) exactly as after a forward that composited them.  Without the flag, sh_objs without out_objects is
 * ignored altogether (no object gradients).  The forward then runs the compositor without object channels: 0.16 ms
 * instead of 0.27 at 1 M Gaussians / 1080p, and 133 MB of output less. */
#define GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY (1u << 21)

/* Mirrors the 12 fields of GaussianRasterizationSettings in call-site order
 * (reference gaussian_renderer/__init__.py:36-49).  Tensor-valued fields are DEVICE pointers, read by the
 * kernels themselves (no host copy, no sync):
 *   bg          >= 3 floats (the reference passes 4 for black, attack.py:396: only the first 3 are read)
 *   viewmatrix  16 floats, row-major as stored by the reference, i.e. the TRANSPOSED world->view matrix:
 *               p_view = [x y z 1] * viewmatrix          (reference scene/cameras.py:54)
 *   projmatrix  16 floats, full projection in the same convention (reference scene/cameras.py:56)
 *   campos      3 floats                                  (reference scene/cameras.py:57)            */
typedef struct GsrSettings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  const float* bg;
  float scale_modifier;
  const float* viewmatrix;
  const float* projmatrix;
  int32_t sh_degree; /* active degree, 0..3 */
  const float* campos;
  int32_t prefiltered;
  int32_t debug;
  uint32_t flags; /* GSR_FLAG_* (extension; 0 reproduces the reference) */
} GsrSettings;

typedef struct GsrCtx GsrCtx; /* opaque: geometry / binning / per-pixel state of one forward */

/* Forward: replaces GaussianRasterizer.forward (call site reference gaussian_renderer/__init__.py:86-95).
 *   P              number of Gaussians
 *   K              SH coefficients stored per Gaussian and channel in `shs` (16 for max degree 3)
 *   means3D        [P,3]
 *   shs            [P,K,3] coefficient-major then channel (reference scene/gaussian_model.py:113-116), or NULL
 *   sh_objs        [P,16] object features (reference scene/gaussian_model.py:118-120), or NULL (objects = 0)
 *   colors_precomp [P,3] or NULL            (exactly one of shs / colors_precomp)
 *   opacities      [P]   (already sigmoid-activated)
 *   scales         [P,3] (already exp-activated), rotations [P,4] (w,x,y,z, used as given), or both NULL
 *   cov3D_precomp  [P,6] xx,xy,xz,yy,yz,zz or NULL     (exactly one of (scales,rotations) / cov3D_precomp)
 *   out_color      [3,H,W]  (background blended in, not clamped)
 *   out_objects    [16,H,W] or NULL
 *   radii          [P] int32, 0 for culled Gaussians
 *   ctx_out        receives the context for gsr_backward (pass NULL for a forward-only call: nothing is kept)
 *   num_rendered   receives the number of (tile, Gaussian) pairs, may be NULL                               */
int gsr_forward(const GsrSettings* settings, int32_t P, int32_t K, const float* means3D, const float* shs,
                const float* sh_objs, const float* colors_precomp, const float* opacities, const float* scales,
                const float* rotations, const float* cov3D_precomp, float* out_color, float* out_objects,
                int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream);

/* Backward: what loss.backward() (reference attack.py:494) reaches through the extension's autograd function.
 *   grad_color   [3,H,W]; grad_objects [16,H,W] or NULL (treated as zero)
 *   outputs (any may be NULL = not wanted):
 *   dmeans3D [P,3], dmeans2D [P,3] (gradient w.r.t. the screen-space means in NDC units, z = 0; this is what
 *   lands in viewspace_points.grad, reference gaussian_renderer/__init__.py:26-30), dshs [P,K,3],
 *   dsh_objs [P,16], dcolors_precomp [P,3], dopacities [P], dscales [P,3], drotations [P,4], dcov3D [P,6].
 * When every geometry-side output (dmeans3D, dmeans2D, dopacities, dscales, drotations, dcov3D) is NULL the
 * backward runs its colour-only kernels (no conic / mean / opacity sums, no projection chain rule); the colour-side
 * gradients are the same numbers either way.
 * May be called more than once on a context (retain_graph).                                                  */
int gsr_backward(GsrCtx* ctx, const float* grad_color, const float* grad_objects, float* dmeans3D, float* dmeans2D,
                 float* dshs, float* dsh_objs, float* dcolors_precomp, float* dopacities, float* dscales,
                 float* drotations, float* dcov3D, void* stream);

/* Fused-activation variants of the same path for callers that hold a reference-style GaussianModel: the seven RAW
 * parameter tensors go in (reference scene/gaussian_model.py:42-59) and the activation getters the reference's render()
 * applies first (exp / sigmoid / normalize / cat, scene/gaussian_model.py:97-124, gaussian_renderer/__init__.py:53-83)
 * and their chain rule run inside the per-Gaussian kernels, so no activated copy of the attributes is written to HBM.
 *   xyz [P,3]; features_dc [P,1,3]; features_rest [P,15,3]; objects_dc [P,16] or NULL; opacity_logit [P];
 *   log_scaling [P,3]; rotation_raw [P,4] (un-normalised; normalised as v / max(|v|, 1e-12)).
 * Results equal gsr_forward / gsr_backward composed with those PyTorch ops; gradients are w.r.t. the RAW tensors. */
int gsr_forward_raw(const GsrSettings* settings, int32_t P, const float* xyz, const float* features_dc,
                    const float* features_rest, const float* objects_dc, const float* opacity_logit,
                    const float* log_scaling, const float* rotation_raw, float* out_color, float* out_objects,
                    int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream);
int gsr_backward_raw(GsrCtx* ctx, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                     float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                     float* dlog_scaling, float* drotation_raw, void* stream);

/* gsr_backward_raw with a choice of what happens to the output buffers.  accumulate == 0: they are overwritten (zeros
 * for Gaussians without pairs), exactly gsr_backward_raw.  accumulate != 0: the 59 attribute gradients (dxyz, dfeatures_dc,
 * dfeatures_rest, dopacity_logit, dlog_scaling, drotation_raw) are ADDED to what the buffers hold and Gaussians without
 * pairs are not touched at all; dmeans2D and dobjects_dc, which belong to ONE view (the screen-space gradient the
 * reference reads from viewspace_points.grad, the object-feature gradient), are overwritten in either mode, zeros for
 * Gaussians without pairs included.  That is what a batch of views needs (reference
 * attack.py:476-494: the B renders' gradients add up in .grad): the first view of a PGD iteration overwrites a
 * caller-owned [P,59] bucket, the others add to it, and the per-view gradient buffer plus the framework's
 * read-modify-write accumulation of 236 bytes per Gaussian and view disappear.  Concurrent calls (views on different
 * streams) must use different buckets. */
int gsr_backward_raw_into(GsrCtx* ctx, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                          float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                          float* dlog_scaling, float* drotation_raw, int32_t accumulate, void* stream);

/* gsr_backward_raw_into with the per-Gaussian stage run as `nchunks` launches over consecutive ranges of Gaussians
 * (boundaries at multiples of 64).  After each range's launch has been enqueued, chunk_done(user, chunk, g_begin, g_end) is
 * called on the calling thread: every gradient of Gaussians [g_begin, g_end) is then complete in stream order, so the caller
 * can issue the multi-GPU sum of that range (six slices of the bucket) while the next range is still being computed
 * ("reduce while K9 finishes").  Results are bit for bit those of the unchunked call. */
typedef void (*gsr_chunk_fn)(void* user, int32_t chunk, int64_t g_begin, int64_t g_end);
int gsr_backward_raw_chunked(GsrCtx* ctx, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                             float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                             float* dlog_scaling, float* drotation_raw, int32_t accumulate, int32_t nchunks,
                             gsr_chunk_fn chunk_done, void* user, void* stream);

/* A BATCH of views of one set of raw parameters through ONE launch chain.  The reference's batch is a Python loop of B
 * render() calls on the same attributes (reference attack.py:476-485; :522-530 for the success renders) whose B backward
 * passes add up in .grad (:494); every call repeats the latency-bound binning chain (scan, two sorts, emission, schedule:
 * ~20 short launches), re-reads the 192-byte SH rows and re-writes 236 bytes of gradient per Gaussian.  Here the B views
 * form one virtual scene -- view v owns the virtual Gaussians [v * Ppad, v * Ppad + P) and the tiles [v * T, (v + 1) * T) --
 * and the scans, both sorts, the emission, the tile schedule and the two compositors run once over it.
 *   settings     [B] (1 <= B <= 16): per view the camera tensors, tan(fov / 2) and background; image size, scale_modifier,
 *                sh_degree and flags must be the same in all of them (GSR_FLAG_NEEDLE_DOUBLE is not available)
 *   out_color    [B,3,H,W]; radii [B,P]; no object channels
 *   num_rendered receives the (tile, Gaussian) pairs of all B views together
 * Every view's image and radii are bit for bit those of gsr_forward_raw on that view's settings.
 * gsr_backward_raw_batch_into: grad_color [B,3,H,W]; the 59 attribute gradients are the SUM over the B views, written once
 * (accumulate == 0; Gaussians without pairs in any view: zeros) or added to what the buffers hold (accumulate != 0) -- bit
 * for bit what B calls of gsr_backward_raw_into in view order leave, the first with the caller's `accumulate` and the
 * others adding; dmeans2D [B,P,3] or NULL is per view (overwritten).  gsr_backward_raw / _into / _chunked accept a batch
 * context with these shapes too, and so do gsr_ctx_request_sumsq (the batch's one fused per-Gaussian launch leaves the sums
 * of squares of the SUMMED gradient) and gsr_ctx_rerender (below: the batch's colour kernel + the compositor over the kept
 * lists of all B views). */
int gsr_forward_raw_batch(const GsrSettings* settings, int32_t B, int32_t P, const float* xyz, const float* features_dc,
                          const float* features_rest, const float* opacity_logit, const float* log_scaling,
                          const float* rotation_raw, float* out_color, int32_t* radii, GsrCtx** ctx_out,
                          int64_t* num_rendered, void* stream);
int gsr_backward_raw_batch_into(GsrCtx* ctx, const float* grad_color, float* dxyz, float* dmeans2D, float* dfeatures_dc,
                                float* dfeatures_rest, float* dopacity_logit, float* dlog_scaling, float* drotation_raw,
                                int32_t accumulate, void* stream);
/* The same backward with PER-VIEW attribute gradients, for callers that need every view's own gradient (independent views
 * that merely share a launch chain): the six attribute-gradient pointers are VIEW 0's buffers, view v's lie v * view_stride
 * floats further (e.g. B gradient buckets of 59 * P floats one after another: view_stride = 59 * P); dmeans2D is [B,P,3] as
 * above.  One forward-batch chain and one backward composite for all views, then one per-Gaussian launch per view: view v's
 * buffers receive bit for bit what gsr_backward_raw on that view alone writes (zeros for Gaussians the view does not see). */
int gsr_backward_raw_batch_views(GsrCtx* ctx, const float* grad_color, float* dxyz, float* dmeans2D, float* dfeatures_dc,
                                 float* dfeatures_rest, float* dopacity_logit, float* dlog_scaling, float* drotation_raw,
                                 int64_t view_stride, void* stream);

/* Forward-only render of TWO parameter sets as one scene: the attacked target (a) followed by the frozen background (b),
 * Gaussians numbered a then b (radii [Pa+Pb]).  Replaces what the reference does after every PGD step to check the
 * attack: deep-copy the attacked model, append the background to each of its seven tensors (seven concat_setup calls,
 * ~300 MB at 1 M Gaussians) and render the copy (reference attack.py:513-530).  Same image, bit for bit, as
 * gsr_forward_raw on the concatenated tensors; nothing is copied, no context is kept (the reference never
 * differentiates this render).  objects_dc_*: both or neither. */
int gsr_forward_raw2(const GsrSettings* settings, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                     const float* features_rest_a, const float* objects_dc_a, const float* opacity_logit_a,
                     const float* log_scaling_a, const float* rotation_raw_a, int32_t Pb, const float* xyz_b,
                     const float* features_dc_b, const float* features_rest_b, const float* objects_dc_b,
                     const float* opacity_logit_b, const float* log_scaling_b, const float* rotation_raw_b,
                     float* out_color, float* out_objects, int32_t* radii, int64_t* num_rendered, void* stream);

/* gsr_forward_raw2 that can keep its context -- for gsr_ctx_rerender ONLY: the context holds the binning and the splat
 * records but no backward state (gsr_backward* on it return GSR_ERR_STATE), as the reference never differentiates this
 * render.  ctx_out == NULL: exactly gsr_forward_raw2. */
int gsr_forward_raw2_keep(const GsrSettings* settings, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                          const float* features_rest_a, const float* objects_dc_a, const float* opacity_logit_a,
                          const float* log_scaling_a, const float* rotation_raw_a, int32_t Pb, const float* xyz_b,
                          const float* features_dc_b, const float* features_rest_b, const float* objects_dc_b,
                          const float* opacity_logit_b, const float* log_scaling_b, const float* rotation_raw_b,
                          float* out_color, float* out_objects, int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered,
                          void* stream);

/* gsr_forward_raw2_keep for a BATCH of views: the attacked target (a) followed by the frozen background (b) rendered from B
 * cameras through one launch chain -- the success renders of a batch of views (reference attack.py:513-530 renders the
 * combined scene once per camera of the batch, inside the loop of :476-485).  Forward only: settings [B], out_color
 * [B,3,H,W], radii [B,Pa+Pb], no object channels; Pa, Pb > 0.  Every image and radius is bit for bit gsr_forward_raw2's
 * for that view.  A kept context serves gsr_ctx_rerender ONLY (gsr_backward* return GSR_ERR_STATE). */
int gsr_forward_raw2_batch(const GsrSettings* settings, int32_t B, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                           const float* features_rest_a, const float* opacity_logit_a, const float* log_scaling_a,
                           const float* rotation_raw_a, int32_t Pb, const float* xyz_b, const float* features_dc_b,
                           const float* features_rest_b, const float* opacity_logit_b, const float* log_scaling_b,
                           const float* rotation_raw_b, float* out_color, int32_t* radii, GsrCtx** ctx_out,
                           int64_t* num_rendered, void* stream);

/* Re-render of a kept context after ONLY its colour inputs changed.  A colour attack (reference attack.py:25-49 steps
 * _features_dc / _features_rest and nothing else; configs/config.yaml attack groups = ["color"]; BASELINE configs 2, 3)
 * renders the same cameras iteration after iteration with the same means, scales, rotations and opacities: projection,
 * tile rects, depth order, the sorted (tile, Gaussian) lists, the tile schedule and the geometric words of the splat
 * records are then the same every time.  The reference's extension rebuilds them per call; a context kept in HBM
 * (about 250 MB per camera at 1 M Gaussians / 1080p, of 288 GB) makes a render of such a view two kernels: the colour half
 * of K1 (SH -> RGB over the Gaussians that emit pairs, into the colour words of the kept records) and the compositor K6
 * over the kept lists.  Image, radii-independent outputs and every gradient of a following gsr_backward* are bit for bit
 * those of a fresh gsr_forward* with the same inputs (tests/test_gpu_rerender.py).
 *   ctx            from gsr_forward / gsr_forward_raw (SH input: raw parameters, or shs with K = 16) with ctx_out, or from
 *                  gsr_forward_raw2_keep.  The CALLER guarantees that since that forward nothing but the SH coefficients
 *                  and bg changed: same means / opacities / scales / rotations contents, same camera, sizes, flags.
 *   features_dc, features_rest   the coefficients to render with ([P,1,3], [P,15,3]; a gsr_forward context: NULL and
 *                  shs [P,16,3]); NULL = the pointers of the previous render (their CONTENTS may have changed).  The
 *                  context re-reads them in gsr_backward*: keep them alive and unmodified until then, as after a forward.
 *   features_*_b   the same for the second segment of a gsr_forward_raw2_keep context (else NULL)
 *   bg             >= 3 floats, or NULL = the previous background pointer
 *   out_color      [3,H,W]; out_objects [16,H,W] or NULL (only if the context's forward composited them)
 *   flags          GSR_RERENDER_COLOR_GRADS_ONLY: the backward of this render will ask for colour-side gradients only
 *                  (dfeatures_*, dcolors): the colour kernel then skips the 36 bytes per Gaussian of d colour / d view
 *                  direction it leaves for dL/dmeans; a geometry backward on the context returns GSR_ERR_STATE until
 *                  the next re-render without the flag.
 *                  GSR_RERENDER_FIRST_SEGMENT_ONLY (two-segment contexts): the second segment's coefficients have not
 *                  changed since the context's last render (the frozen background of reference attack.py:513-530): the
 *                  colour kernel covers the first segment's Gaussians only; features_*_b must be NULL.
 * A BATCH context (gsr_forward_raw_batch): out_color [B,3,H,W]; bg [B,3] (view v's background at bg + 3 v) or NULL = the
 *                  views' previous background pointers, whose contents are read again; out_objects must be NULL
 *                  (features_*_b and GSR_RERENDER_FIRST_SEGMENT_ONLY as above for a gsr_forward_raw2_batch context).  One launch of the batch's colour kernel (every SH row read once for all views that see the
 *                  Gaussian) and one compositor launch over the B views' kept lists: images and the gradients of a following
 *                  gsr_backward_raw_batch_* are bit for bit those of a fresh gsr_forward_raw_batch with the same inputs.
 * The per-pixel state the backward reads (final T, last contributor, segment-boundary records) is overwritten: a
 * backward of the PREVIOUS render of this context must have been enqueued before, on the same stream or ordered
 * before it by the caller. */
#define GSR_RERENDER_COLOR_GRADS_ONLY 1u
#define GSR_RERENDER_FIRST_SEGMENT_ONLY 2u
int gsr_ctx_rerender(GsrCtx* ctx, const float* features_dc, const float* features_rest, const float* features_dc_b,
                     const float* features_rest_b, const float* bg, float* out_color, float* out_objects, uint32_t flags,
                     void* stream);

/* Releases the context's workspace back to the pool (stream-ordered: safe right after enqueueing backward). */
void gsr_ctx_free(GsrCtx* ctx);

/* Frustum test only (view-space z > 0.2): present[P] = 1/0.  Replaces GaussianRasterizer.markVisible. */
int gsr_mark_visible(const GsrSettings* settings, int32_t P, const float* means3D, uint8_t* present, void* stream);

/*
 * gsr_pgd_step: one projected-gradient update of a raw attribute tensor x[rows, cols] in place (reference
 * attack.py:25-173, the ten gaussian_*_{linf,l2}_attack functions; the colour rules call it once for _features_rest
 * [P,45] and once for _features_dc [P,3]).  l2 == 0: x += -alpha*sign(grad); x = clamp(x - x0, -eps, eps) + x0.
 * l2 != 0: x += -alpha*grad/||grad||_2 with the norm over the whole tensor (no step when it is 0), then every ROW of
 * x - x0 longer than eps is scaled by eps/(norm + 1e-7) (torch.renorm(p=2, dim=0, maxnorm=eps)).  cols <= 48.
 */
int gsr_pgd_step(float* x, const float* grad, const float* x0, int64_t rows, int32_t cols, float alpha, float epsilon,
                 int32_t l2, void* stream);

/*
 * The L2 rules' global norm without a second pass over the gradient (round 5).  gsr_ctx_request_sumsq arms the NEXT
 * overwrite-mode gsr_backward_raw / gsr_backward_raw_into (accumulate == 0, not chunked) of this context: besides the
 * gradients it leaves, in out6 (DEVICE memory, 6 doubles, written in stream order), the sums of squares of the gradients
 * it writes for xyz, features_dc, features_rest, opacity_logit, log_scaling, rotation_raw -- the quantities whose roots
 * reference attack.py:61-62,70-71,...,145-146,154-155 divide by (a tensor whose gradient pointer is NULL gets 0).  The
 * sums are those of THAT launch's output: a caller that adds other views' gradients to the buffers afterwards must not
 * use them.  One-shot: the request is consumed by the backward it serves.
 * gsr_pgd_step_normed is gsr_pgd_step's L2 rule with ||grad||^2 read from `sumsq` (device, one double) instead of being
 * summed by a launch of its own: one launch per tensor, one read of the gradient.
 */
int gsr_ctx_request_sumsq(GsrCtx* ctx, double* out6);
int gsr_pgd_step_normed(float* x, const float* grad, const float* x0, int64_t rows, int32_t cols, float alpha, float epsilon,
                        const double* sumsq, void* stream);

/*
 * gsr_pgd_step_multi (round 6): the update of gsr_pgd_step / gsr_pgd_step_normed on n <= 8 tensors of one model in ONE
 * launch (the attack steps _xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation in one iteration, reference
 * attack.py:496-520: six short bandwidth-bound launches whose tails nothing on the stream covers).  Arrays of n host
 * entries: x / grad / x0 device pointers, rows, cols (<= 48), alpha, epsilon; sumsq[t] = device pointer to ||grad_t||^2
 * (one double, as gsr_pgd_step_normed takes it) or NULL -- with l2 != 0 a tensor without one has its norm summed by a
 * launch of its own in front (sumsq == NULL: all of them).  Results are bit for bit those of the per-tensor calls.
 */
int gsr_pgd_step_multi(int32_t n, float* const* x, const float* const* grad, const float* const* x0, const int64_t* rows,
                       const int32_t* cols, const float* alpha, const float* epsilon, int32_t l2, const double* const* sumsq,
                       void* stream);

/* Mean squared distance of every point to its 3 nearest other points (exact): replaces the reference's second native
 * import, simple_knn._C.distCUDA2 (reference scene/gaussian_model.py:17, called at :144 to seed the initial scales).
 * points [P,3] float32 device, mean_dist2 [P] float32 device.  Synchronises the stream once (scene set-up routine, not
 * part of the per-view path).  With fewer than 4 points the missing neighbours count as FLT_MAX, like the original. */
int gsr_knn_dist2(const float* points, int32_t P, float* mean_dist2, void* stream);

/* Introspection. what: 0 version, 1 bytes held by the workspace pool on the current device,
 * 2 number of pairs of a context (ctx as int64 handle in *out on input is NOT used; see gsr_ctx_info). */
int gsr_query(int32_t what, int64_t* out);

/* Per-context numbers for roofline accounting: what 0 = num_rendered (N; waits for the forward's count if it was
 * asynchronous), 1 = visible Gaussians (V; -1: not counted on the host), 2 = workspace bytes of this context,
 * 3 = the pair capacity the forward's buffers were sized for (= N unless GSR_FLAG_ASYNC_COUNT).
 * A context kept for backward holds, besides 100 bytes per Gaussian and 16 per pixel, 4 bytes per pair and -- unless
 * object channels are composited or GSR_FLAG_NO_SEGMENTS is set -- (N/256 + min(tiles, N/256) + 1) boundary records of
 * 4 KB: 50-190 MB at N = 3-10 M pairs.  num_rendered counts the pairs of the TIGHTENED tile rects (the tiles the
 * alpha >= 1/255 footprint's bounding box touches); under GSR_FLAG_NO_CULL it is the reference's count. */
int gsr_ctx_info(const GsrCtx* ctx, int32_t what, int64_t* out);   /* also: 4 = views of the context's batch (1: an ordinary forward), 5 = Ppad */

/* Copies one internal array of a context into a caller DEVICE buffer (tests / diagnostics):
 * what 0 = tile ranges [T][2] u32, 1 = sorted pair list [N] u32 (Gaussian index | strip mask << 28, tile by tile, depth
 * order inside a tile), 2 = n_contrib [H*W] u32, 3 = final_T [H*W] f32, 4 = order (depth rank -> Gaussian; the first
 * V = scalars[1] entries are meaningful: only Gaussians that emit pairs are ranked) [P] u32, 5 = off [P+1] u32 (pairs
 * emitted in front of rank r; V+1 entries), 7 (and, for old callers, 6) = splat records [P][3] float4 in storage order
 * (layout: csrc/gsr_kernels.hip.h; written for Gaussians that emit pairs), 8 = the forward's device-side scalars [16]
 * u32 (0 pairs, 1 V, 2 smallest depth key, 3 depth digit width, 4 overflow flag, 5 boundary records, 6-7 64-bit pair
 * count), 9 = offg [P+1] u32 (storage-order scan of tiles touched). */
int gsr_ctx_export(const GsrCtx* ctx, int32_t what, void* dst, int64_t dst_bytes, void* stream);

/* Frees every cached workspace block of the current device (blocks in use by live contexts are kept). */
void gsr_trim_pool(void);

/* Per-stage timing of all calls of this process since the last reset, measured with hip events on the
 * stream the kernels were launched on.  Enable with gsr_profile(1): every stage is then bracketed by event
 * records (adds a few microseconds per stage); gsr_profile_read synchronises and fills
 * ms[GSR_STAGE_COUNT] with accumulated milliseconds and calls[GSR_STAGE_COUNT] with launch counts. */
enum {
  GSR_STAGE_PREPROCESS = 0,
  GSR_STAGE_DEPTH_SORT = 1,
  GSR_STAGE_BIN = 2,       /* pack + scan + emit */
  GSR_STAGE_TILE_SORT = 3, /* + tile ranges */
  GSR_STAGE_RENDER_FWD = 4,
  GSR_STAGE_RENDER_BWD = 5,
  GSR_STAGE_PREPROCESS_BWD = 6,
  GSR_STAGE_COUNT = 7
};
/* gsr_profile(mask): bit i of mask times stage i (0x7F = all stages, 0 = off; also resets the accumulators).  Each
 * timed stage costs two event records on the stream (a few microseconds of queue time each). */
void gsr_profile(int32_t stage_mask);
int gsr_profile_read(float* ms, int64_t* calls);

/* Diagnostic: writes (stage, start_ms, end_ms) triples of every span recorded since the last read, relative to the
 * first span's start, into out[3 * max_spans]; returns the number written and clears the spans. */
int gsr_profile_timeline(float* out, int max_spans);

const char* gsr_last_error(void);

/*
 * gsr_debug_wave_clock: diagnostic.  While `buf` (device, [ntiles][2] uint64) is set, every backward composite (K7)
 * stamps the 100 MHz wall clock at which each tile's wave started and ended; pass NULL to stop.  Process-wide.
 */
int gsr_debug_wave_clock(unsigned long long* buf);
/* Same for the forward composite (K6): [ntiles * waves per tile][2]. */
int gsr_debug_wave_clock_fwd(unsigned long long* buf);

/* Test hooks (used by tests/ only): the scan and sort primitives of the binning stage on caller buffers.
 * gsr_test_scan: out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total), uint32.
 * gsr_test_sort_pairs: stable ascending sort of (keys, vals) on key bits [begin_bit, end_bit), in place;
 *                      iota != 0: vals are ignored on input and the result is the sorting permutation. */
int gsr_test_scan(const uint32_t* in, uint32_t* out, uint32_t n, void* stream);
int gsr_test_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t n, int32_t begin_bit, int32_t end_bit, int32_t iota,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRASTER_H_ */
