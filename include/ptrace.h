/* ptrace.h — C-ABI of the MI355X ray-trace/shade hot path (libptrace.so).
 *
 * The reference (ziotom78/pytracer) has no FFI: its seam is the Python call
 *     ImageTracer.fire_all_rays(renderer)            src/pytracer/imagetracer.py:60-110
 * with `renderer` one of the solvers of                src/pytracer/render.py:42-193.
 * This header is what a ctypes binding for that seam binds (INTEGRATION.md shows the
 * reference-side stub).  Every entry point is `extern "C"`, takes plain pointers and sizes,
 * returns 0 or a negative error code, and never throws.
 *
 * Numerics contract: all scene/camera/param values are IEEE fp64 exactly as the reference
 * holds them (Python float); the PCG is the reference's PCG-XSH-RR 64/32 (src/pytracer/pcg.py:23-62).
 * The library is NOT thread-safe per handle: one call at a time per pt_scene.
 */
#ifndef PTRACE_H
#define PTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enumerations ------------------------------------------------------------------------ */
#define PT_SHAPE_SPHERE 0 /* shapes.py:86  unit sphere, transformed */
#define PT_SHAPE_PLANE 1  /* shapes.py:154 xy plane, transformed  */

#define PT_BRDF_DIFFUSE 0  /* materials.py:123 */
#define PT_BRDF_SPECULAR 1 /* materials.py:155 */

#define PT_PIGMENT_UNIFORM 0   /* materials.py:50 */
#define PT_PIGMENT_CHECKERED 1 /* materials.py:85 */
#define PT_PIGMENT_IMAGE 2     /* materials.py:62 */

#define PT_CAMERA_ORTHOGONAL 0  /* camera.py:42 */
#define PT_CAMERA_PERSPECTIVE 1 /* camera.py:81 */

#define PT_RENDERER_ONOFF 0      /* render.py:42  */
#define PT_RENDERER_FLAT 1       /* render.py:56  */
#define PT_RENDERER_PATHTRACER 2 /* render.py:77  */
#define PT_RENDERER_POINTLIGHT 3 /* render.py:142 */

/* PCG stream assignment ("seed alignment").
 * SEQ    : the reference's own two global sequential streams (ImageTracer.pcg for the jitter, imagetracer.py:84-101;
 *          PathTracer.pcg for the scattering, render.py:118,128), row-major pixel order.  For OnOff / Flat / PointLight
 *          only the jitter stream exists and every sample draws exactly two numbers from it, so sample k of pixel i
 *          starts 2 (i S^2 + k) draws into PCG(jitter_state, jitter_seq): the device enters the stream there by
 *          jump-ahead and renders the frame the reference's `ImageTracer(pcg=PCG(jitter_state, jitter_seq))` renders,
 *          bit for bit.  With the path tracer the scattering stream is serial by construction: PT_ERR_UNSUPPORTED on
 *          the device (the CPU oracle accepts it).
 * PIXEL  : pixel i = row*W+col owns PCG(path_state, path_seq + i) for jitter and scattering,
 *          consumed in program order.  Depends only on the global pixel index.
 * SAMPLE : sample k = sub_row*S+sub_col of pixel i owns PCG(path_state, path_seq + i*S*S + k).
 */
#define PT_PCG_SEQ 0
#define PT_PCG_PIXEL 1
#define PT_PCG_SAMPLE 2

#define PT_OUT_F64 0 /* W*H*3 doubles: the reference's in-memory HdrImage (hdrimages.py:59-94) */
#define PT_OUT_F32 1 /* W*H*3 floats : the reference's on-disk PFM precision (hdrimages.py:35-43) */

#define PT_OK 0
#define PT_ERR_INVALID (-1)     /* bad argument / inconsistent descriptor */
#define PT_ERR_HIP (-2)         /* a HIP runtime call failed (see pt_last_error) */
#define PT_ERR_UNSUPPORTED (-3) /* feature not available on the device path */
#define PT_ERR_NOMEM (-4)
#define PT_ERR_SIZE (-5)        /* output buffer too small */
#define PT_ERR_NODEVICE (-6)

/* ---- scene (flattened, structure-of-arrays, fp64) ------------------------------------------
 * Replaces: World.shapes (world.py:38-45) with each Shape's transformation.m / .invm
 * (transformations.py:48-56), Material.brdf / .emitted_radiance (materials.py:199-204).
 * 3x4 affine matrices are stored element-major: element (r,c) of shape i at a[(r*4+c)*n + i].
 * Colours are channel-major: channel k of shape i at a[k*n + i].
 */
typedef struct pt_scene_desc {
  int32_t n_shapes;
  const int32_t *kind;      /* [n] PT_SHAPE_*                                   */
  const double *invm;       /* [12][n] rows 0..2 of transformation.invm         */
  const double *m;          /* [12][n] rows 0..2 of transformation.m            */
  const int32_t *brdf_kind; /* [n] PT_BRDF_*                                    */
  const double *brdf_param; /* [n] SpecularBRDF.threshold_angle_rad (materials.py:158-162), else 0 */
  /* material.brdf.pigment */
  const int32_t *pig_kind;  /* [n] PT_PIGMENT_*                                 */
  const double *pig_c1;     /* [3][n] uniform colour / checkered color1         */
  const double *pig_c2;     /* [3][n] checkered color2                          */
  const double *pig_steps;  /* [n] checkered num_of_steps                       */
  const int32_t *pig_tex;   /* [n] texture index for PT_PIGMENT_IMAGE, else -1  */
  /* material.emitted_radiance */
  const int32_t *emi_kind;
  const double *emi_c1;
  const double *emi_c2;
  const double *emi_steps;
  const int32_t *emi_tex;
  /* World.point_lights (world.py:47-49, lights.py:25-39) */
  int32_t n_lights;
  const double *light_pos;    /* [3][n_lights] */
  const double *light_color;  /* [3][n_lights] */
  const double *light_radius; /* [n_lights] linear_radius */
  /* ImagePigment textures (materials.py:62-82): row-major RGB fp64, row 0 first */
  int32_t n_textures;
  const int32_t *tex_w;      /* [n_textures] */
  const int32_t *tex_h;      /* [n_textures] */
  const int64_t *tex_offset; /* [n_textures] offset (in doubles) into tex_data */
  const double *tex_data;
} pt_scene_desc;

/* Replaces: OrthogonalCamera / PerspectiveCamera fields (camera.py:48-57, 87-101). */
typedef struct pt_camera {
  int32_t kind; /* PT_CAMERA_* */
  int32_t _pad;
  double m[12]; /* rows 0..2 of camera.transformation.m, row-major */
  double screen_distance;
  double aspect_ratio;
} pt_camera;

/* Replaces: the arguments main.py:164-198 passes to ImageTracer(...) and the Renderer ctor. */
typedef struct pt_params {
  int32_t width, height;    /* HdrImage size (hdrimages.py:59-70)                        */
  int32_t samples_per_side; /* ImageTracer.samples_per_side (imagetracer.py:45); 0 = pixel centre */
  int32_t renderer;         /* PT_RENDERER_*                                             */
  double background[3];     /* Renderer.background_color (render.py:33)                  */
  double onoff_color[3];    /* OnOffRenderer.color (render.py:50)                        */
  double ambient[3];        /* PointLightRenderer.ambient_color (render.py:155)          */
  int32_t num_of_rays;      /* PathTracer.num_of_rays (render.py:95)                     */
  int32_t max_depth;        /* PathTracer.max_depth (render.py:96)                       */
  int32_t rr_limit;         /* PathTracer.russian_roulette_limit (render.py:97)          */
  int32_t pcg_mode;         /* PT_PCG_*                                                  */
  uint64_t jitter_state, jitter_seq; /* ImageTracer.pcg seeds (SEQ mode only; < 2^63)    */
  uint64_t path_state, path_seq;     /* PathTracer.pcg seeds (SEQ) / S0,Q0 (PIXEL, SAMPLE) */
  /* Pixel partition for multi-GPU: rows are cut in blocks of `row_block` rows, block b belongs
   * to rank b % n_ranks; a rank's output holds its rows compactly in ascending global order.
   * n_ranks = 1 renders the whole frame. */
  int32_t row_block, n_ranks, rank;
  int32_t out_format; /* PT_OUT_* */
} pt_params;

typedef struct pt_stats {
  uint64_t n_rays;      /* rays handed to a world query (primary + scattered + shadow), incl. n_rays_resolved */
  uint64_t n_pixels;    /* pixels written by this call                                   */
  double kernel_ms;     /* hipEvent time of the render kernel(s) only                    */
  double total_ms;      /* hipEvent time incl. D2H copy when the call copies             */
  int32_t vgprs, lds_bytes, grid, block; /* registers per lane / launch geometry of the render kernel
                                            (of its last launch when a frame takes several)   */
  uint64_t n_rays_resolved; /* of n_rays: primary rays whose hit was known without tracing them (tiles and
                               pixels that can only see a sphere around the camera) */
  int32_t kernel;           /* PT_KERNEL_*: which render kernel produced the frame (its last launch)          */
  int32_t _reserved;
} pt_stats;

/* pt_stats.kernel */
#define PT_KERNEL_NONE 0
#define PT_KERNEL_SIMPLE 1       /* one lane per pixel, every shape (tiny worlds, PTRACE_CULL=0)                  */
#define PT_KERNEL_TILE 2         /* 8x8 tiles with culled shape lists                                             */
#define PT_KERNEL_TILE4 3        /* 16x16 tiles, four pixels per lane (OnOff / Flat, pixel-centre rays)           */
#define PT_KERNEL_PATH 4         /* path tracer, one queue over all pixels                                        */
#define PT_KERNEL_PATH_REGIONS 5 /* path tracer in two passes: tile classification, then work units over regions */
#define PT_KERNEL_PATH_TREE 6    /* ... num_of_rays > 1: second pass with one pixel per wave, a node's children on lanes */

typedef struct pt_scene pt_scene; /* opaque: device-resident scene + workspace */

/* ---- entry points --------------------------------------------------------------------------*/
int pt_device_count(void);
/* Compute units and peak shader clock (kHz) of `device`: what a roofline is priced against. */
int pt_device_info(int device, int *compute_units, int *clock_khz);
/* Copy the flattened scene to the HBM of `device` (packed into the kernel's record layout). */
int pt_scene_upload(const pt_scene_desc *desc, int device, pt_scene **out);
/* A further handle on an uploaded scene, for frames IN FLIGHT (the reference renders an animation one frame per process;
 * here consecutive frames are independent launches, and a small frame leaves most of the chip idle): the new handle
 * shares the tables pt_scene_upload made -- nothing is uploaded again -- and has its own per-camera constants, queues,
 * counters and workspace.  One call at a time per HANDLE; frames rendered through different handles of a scene may run
 * concurrently, each on its own stream (pt_render_device).  Free every handle with pt_scene_free, in any order: the
 * last one frees the shared tables. */
int pt_scene_clone(pt_scene *scene, pt_scene **out);
void pt_scene_free(pt_scene *scene);
/* Number of image rows rank `p->rank` owns under p's partition. */
int pt_rows_for_rank(const pt_params *p);
/* Bytes the output of one call needs: rows_for_rank * W * 3 * sizeof(out_format). */
size_t pt_output_bytes(const pt_params *p);
/* Render into a caller-owned HOST buffer (kernel + D2H); row 0 = top of the image,
 * pixel (col,row) at out[(row*W+col)*3 + k]  (hdrimages.py:78-80). */
int pt_render(pt_scene *scene, const pt_camera *cam, const pt_params *p, void *out_host,
              size_t out_bytes);
/* Page-locked host memory for the output of pt_render (optional): with a destination obtained here the
 * device-to-host copy is a single DMA at link speed; any other host pointer works too, through the HIP
 * runtime's staging path.  (The reference's image lives in a Python list, hdrimages.py:70; this is the
 * buffer the binding hands to numpy.)  pt_host_free(NULL) is a no-op. */
int pt_host_alloc(size_t bytes, void **out);
int pt_host_free(void *p);
/* Device memory and streams for a caller WITHOUT a GPU framework of its own (ABI 1.5; the reference's frame lives in a Python
 * list, hdrimages.py:70, and its `render` command post-processes it in place, main.py:203-213: here the frame stays in HBM
 * between pt_render_device and pt_image_*, and these are the buffer and the stream that takes).  A caller that owns device
 * tensors (torch, a hipMalloc of its own) keeps handing their pointers to pt_render_device: nothing here is required.
 *   pt_device_alloc    bytes of HBM on `device` (0 bytes: *out = NULL, PT_OK); PT_ERR_NOMEM when the device is full.
 *   pt_device_free     waits for the device's work on the buffer; NULL is a no-op.
 *   pt_device_download blocking copy of `bytes` from HBM to host memory, ordered behind `stream` (NULL: the default stream).
 *   pt_stream_create   a non-blocking hipStream_t on `device`, for pt_render_device / pt_image_*: frames on different streams
 *                      (through different handles of a scene, pt_scene_clone) overlap.
 *   pt_stream_sync     block until everything enqueued on `stream` is done.
 *   pt_stream_destroy  drains, then destroys; NULL is a no-op. */
int pt_device_alloc(int device, size_t bytes, void **out);
int pt_device_free(int device, void *p);
int pt_device_download(int device, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int pt_stream_create(int device, void **out);
int pt_stream_sync(int device, void *stream);
int pt_stream_destroy(int device, void *stream);
/* Render into a caller-owned DEVICE buffer on `stream` (a hipStream_t, NULL = the library's own
 * stream); asynchronous when a stream is given. Nothing is copied to the host.
 * A scene's workspace is shared by all its launches, which are ordered by running on one stream: a
 * launch on a different stream than the scene's previous one first waits (on the host) until that
 * previous stream has drained. */
int pt_render_device(pt_scene *scene, const pt_camera *cam, const pt_params *p, void *out_dev,
                     size_t out_bytes, void *stream);
/* Statistics of the last completed pt_render / synchronised pt_render_device on this scene. */
int pt_get_stats(pt_scene *scene, pt_stats *out);
/* Enable (1) / disable (0) the in-kernel ray counter (default on). */
int pt_set_count_rays(pt_scene *scene, int enable);
/* Measurement switch: 0 = primary rays are always generated and traced, also where the image can only show
 * a sphere that encloses the camera (default 1: such tiles/pixels are resolved without rays; the image is
 * bit-identical either way). */
int pt_set_dome_shortcut(pt_scene *scene, int enable);
/* Block until the scene's last asynchronous render has finished and fold its statistics. */
int pt_sync(pt_scene *scene);
/* Enable (1, default) / disable (0) the hipEvent pair around each render kernel.  An event record is a
 * barrier packet on the stream: back-to-back frames run closer together without them. */
int pt_set_timing(pt_scene *scene, int enable);
/* Kernel-time accounting over many asynchronous launches (the reference only prints one
 * process_time() delta around fire_all_rays, main.py:196-200).  Between begin and end every render
 * on this scene brackets its render kernel with its own hipEvent pair, recorded on the stream the
 * kernel is launched on; end synchronises and returns the summed kernel time and the launch count. */
int pt_profile_begin(pt_scene *scene, int capacity);
int pt_profile_end(pt_scene *scene, double *total_kernel_ms, int *launches);
/* ---- HdrImage post-processing on the device (main.py:203-213) --------------
 * `img_dev` is a frame of H*W*3 values of `fmt` (PT_OUT_*), row 0 on top: either in HBM, as
 * pt_render_device leaves it, or in host memory (an HdrImage's array) — the library detects which and
 * stages host buffers through the device; the same holds for the output buffers.  `stream` as in
 * pt_render_device (NULL: the default stream; the call returns when done). */
/* HdrImage.write_pfm payload (hdrimages.py:113-118): W*H*3 float32, bottom row first, little (0) or
 * big (1) endian, into out_dev (W*H*12 bytes, device).  The ASCII header is the caller's. */
int pt_image_pack_pfm(int device, const void *img_dev, int fmt, int width, int height, int big_endian,
                      void *out_dev, void *stream);
/* HdrImage.average_luminosity (hdrimages.py:120-128) -> *out (host).  Synchronises. */
int pt_image_average_luminosity(int device, const void *img_dev, int fmt, int width, int height, double delta,
                                double *out, void *stream);
/* normalize_image (x * scale, scale = factor / luminosity; hdrimages.py:130-140), optionally clamp_image
 * (x / (1 + x); :142-146), optionally written back in place, optionally the LDR bytes of write_ldr_image
 * (int(255 * pow(x, 1/gamma)); :160-166) into rgb8_dev (W*H*3 bytes, row 0 on top; may be NULL). */
int pt_image_tonemap(int device, void *img_dev, int fmt, int width, int height, double scale, int clamp,
                     double gamma, unsigned char *rgb8_dev, int write_back, void *stream);
/* ---- a rank's shard in sparse form, for the gather of a sharded frame (the reference has no multi-GPU
 * path: this serves pytracer_amd/dist.py's gather, which replaces nothing in the reference) ----
 * A shard of n_pixels RGB pixels of `fmt` is cut into runs of 128 consecutive pixels; a run whose pixels all equal its
 * first one bit for bit is kept as that one pixel, the others whole and in order.  Lossless.
 *   fixed   (pt_image_sparse_fixed_bytes bytes): int64 count of runs that are not constant | int32 per run: its place
 *            among those, -1 = constant (padded to a multiple of 8 bytes) | the first pixel of every run
 *   payload ([count][128][3] values of fmt; capacity: every run): the runs that are not constant, the shard's last run
 *            filled up with its last pixel.
 * All pointers are device pointers; `stream` as in pt_render_device.  The count is read from fixed[0..8) by the caller. */
long long pt_image_sparse_fixed_bytes(long long n_pixels, int fmt);
int pt_image_sparse_encode(int device, const void *shard_dev, long long n_pixels, int fmt, void *fixed_dev,
                           void *payload_dev, void *stream);
/* ... and back, in one pass.  n_ranks <= 1: out_dev receives the shard's n_pixels pixels.  n_ranks > 1: out_dev is the
 * FRAME (rows of `width` pixels) the shard of rank `rank` belongs to under pt_params' row_block / n_ranks partition, and the
 * shard's rows are written where pt_rows_for_rank puts them.  payload_dev may be NULL when the count is 0. */
int pt_image_sparse_decode(int device, const void *fixed_dev, const void *payload_dev, long long n_pixels, int fmt,
                           void *out_dev, int width, int row_block, int n_ranks, int rank, void *stream);
/* ... the shards of up to 64 ranks into one frame, in ONE launch: fixed_dev[k] / payload_dev[k] / n_pixels[k] / ranks[k]
 * (host arrays of device pointers, pixel counts and ranks) as the single call takes them. */
int pt_image_sparse_decode_many(int device, int n_shards, const void *const *fixed_dev, const void *const *payload_dev,
                                const long long *n_pixels, const int *ranks, int fmt, void *frame_dev, int width,
                                int row_block, int n_ranks, void *stream);
/* Copy the last error message of the calling thread (NUL-terminated) into buf; returns its length. */
int pt_last_error(char *buf, size_t n);
/* 1 iff HIP_FORCE_DEV_KERNARG is set to a non-zero value in this process: the HIP runtime then keeps kernel ARGUMENTS in
 * device memory (it reads the variable once, when it initialises) and every launch is ~1 us shorter
 * (profiles/r04_dev_kernarg.txt).  The library never sets it: the caller does, before the process's first HIP call
 * (pytracer_amd.prefer_device_kernargs(), the `render` command and bench.py do).  A value set after the runtime came up
 * is reported here but has no effect. */
int pt_device_kernargs(void);
/* Library/ABI version: (major<<16)|minor; this header describes 1.5.  The minor grows whenever a struct here grows or an
 * entry point is added (1.2: pt_stats gained `kernel` and `_reserved` -- 56 bytes, which pt_get_stats writes in full --,
 * pt_scene_clone, pt_image_sparse_*; 1.3: PT_PCG_SEQ on the device for OnOff / Flat / PointLight; 1.4: pt_device_kernargs, no load-time setenv; 1.5: pt_device_alloc / _free / _download, pt_stream_create / _sync / _destroy): a caller built
 * against an older header must check pt_version() before it hands pt_get_stats its smaller struct. */
int pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_H */
