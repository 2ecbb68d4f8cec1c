// ptrace.hip — libptrace.so: the C-ABI of include/ptrace.h over the HIP kernels of pt_kernels.h.
//
// Boundary replaced: ImageTracer.fire_all_rays(renderer) — src/pytracer/imagetracer.py:60-110 with
// the solvers of src/pytracer/render.py:42-193.  No torch types, no exceptions across the ABI, every
// failure is a negative return code plus a thread-local message (pt_last_error).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/ptrace.h"
#include "../../include/ptrace_debug.h"
#include "pt_kernels.h"
#include "pt_layout.h"
#include "pt_plan.h"
#include "pt_post.h"

// major << 16 | minor.  The minor grows whenever a struct of include/ptrace.h grows or an entry point is added (minor 2:
// pt_stats gained `kernel` + `_reserved`, pt_scene_clone / pt_image_sparse_* arrived; minor 3: PT_PCG_SEQ accepted for
// OnOff / Flat / PointLight at any samples_per_side; minor 4: pt_device_kernargs, and the library no longer sets
// HIP_FORCE_DEV_KERNARG when it is loaded; minor 5: pt_device_alloc / pt_device_free / pt_device_download / pt_stream_*);
// a caller built against an older header checks pt_version() first.
#define PT_VERSION ((1 << 16) | 5)

// (Kernel arguments in device memory -- HIP_FORCE_DEV_KERNARG=1, ~1 us per launch, profiles/r04_dev_kernarg.txt -- are the
// CALLER's choice: the HIP runtime reads the variable when it initialises, and a library that edited the process environment
// at load time would change HIP for every other user of the runtime in the process.  pytracer_amd.prefer_device_kernargs(),
// the `render` command and bench.py set it before their first HIP call; pt_device_kernargs() reports what is in effect.)

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                 \
  } while (0)

// A ball of the scattered-ray filter as the kernels want it: r'^2 / (1 - 8e-6) rounded up; where there is no usable bound
// (r' not finite or >= 1e17, a centre that is not finite or beyond 1e17) the centre 0 and r'^2 = 1e38, which the filter's
// arithmetic never rejects and never overflows on (pt_query.h: world_query_lanes).
static void pt_ball_square(float *x, float *y, float *z, float *r, bool *ordinary) {
  const bool ok = std::isfinite(*r) && *r >= 0.0f && *r < 1e17f && std::isfinite(*x) && std::isfinite(*y) && std::isfinite(*z) &&
                  std::fabs(*x) < 1e17f && std::fabs(*y) < 1e17f && std::fabs(*z) < 1e17f;
  *ordinary = ok;
  if (ok) {
    *r = std::nextafter((float)((double)*r * (double)*r * (1.0 + 8.1e-6)), INFINITY);
  } else {
    *x = *y = *z = 0.0f;
    *r = 1e38f;
  }
}

struct pt_scene {
  int device = 0;
  int n_shapes = 0, n_spheres = 0, n_lights = 0, n_textures = 0;
  PtShapeRec *recs = nullptr;
  PtShapeAux *aux = nullptr;
  PtHoist *hoist = nullptr;
  PtDiagRec *diag = nullptr;
  PtHoistDiag *hoist_diag = nullptr;
  float4 *bounds = nullptr;
  unsigned *grid_cells = nullptr, *grid_occ = nullptr;  // uniform grid for scattered / shadow rays (see PtKArgs)
  float4 *grid_balls = nullptr;
  unsigned short *grid_slots = nullptr;
  int *grid_always = nullptr;
  int grid_n_always = 0, grid_n_cells = 0, grid_res[3] = {0, 0, 0};
  float grid_far_eo = INFINITY;  // rays with 1e-6 * max|origin component| above this do not walk the grid (see world_query_lanes)
  float grid_min[3] = {0, 0, 0}, grid_max[3] = {0, 0, 0}, grid_cell[3] = {0, 0, 0}, grid_inv[3] = {0, 0, 0};
  float *bsoa = nullptr;  // bounds as x[], y[], z[], r'[] (bs_stride floats each), then group and chunk balls
  int bs_stride = 0, gs_stride = 0, cs_stride = 0, bs_levels = 0;
  float bs_rmax[3] = {0.0f, 0.0f, 0.0f};
  int n_diag = 0;
  PtLight *lights = nullptr;
  PtTex *tex = nullptr;
  double *tex_data = nullptr;
  double *ws = nullptr;  // path-tracer frame stack, grown on demand
  size_t ws_bytes = 0;
  double *handover = nullptr;  // num_of_rays > 1: records of the pixels the one-queue kernel hands to the tree kernel
  size_t handover_doubles = 0;
  int4 *units_handed = nullptr;  // ... and their units
  size_t units_handed_n = 0;
  void *out_dev = nullptr;  // staging for pt_render (host output)
  size_t out_dev_bytes = 0;
  unsigned long long *ray_counter = nullptr;   // totals: all rays, rays resolved by the dome shortcut
  unsigned long long *ray_partials = nullptr;  // the same two per workgroup
  int ray_partials_n = 0;
  // path tracer: two queue blocks (pixel queue head / work units of the second pass, PT_QUEUE_WORDS each).  A frame
  // uses one and its last kernel zeroes the other for the next frame: no memset (and no launch gap after it) in front
  // of a frame.  queue_clean: the block the next frame will use is known to be zero.
  unsigned long long *queue = nullptr;
  unsigned long long *queue_last = nullptr;  // the block the most recent frame used (debug read-back)
  int queue_parity = 0;
  bool queue_clean = false;
  PtKArgs *args_dev = nullptr;          // device copy of the argument block (cold fields)
  PtKArgs *args_dev2 = nullptr;         // ... of pt_path_flagged_kernel's, when a frame enqueues both second-pass kernels (PT_Q_CHOICE)
  PtKArgs args2_last;
  bool args2_valid = false;
  hipStream_t args2_stream = nullptr;
  int last_handover_cap = 0;            // records the last num_of_rays > 1 frame's hand-over table held (pt_debug_handed_over)
  bool choice_pending = false;          // ray_counter_host[2] will hold the frame's PT_Q_CHOICE word once ev_count has passed
  unsigned char *region_keys = nullptr;  // path tracer region ordering
  struct DomeCand {
    int slot;
    double invm[12];
  };
  std::vector<DomeCand> dome_cands;  // spheres that may serve as "the dome" of a view: uniform pigments, sane scale
  int4 *units = nullptr;  // second pass: work units (pt_unit_scatter)
  int units_cap = 0;
  unsigned long long *region_mask = nullptr;
  int region_cap = 0;
  unsigned int *cell_list = nullptr;  // large scenes: per-cell survivor lists (pt_cell_kernel)
  int *cell_count = nullptr;
  size_t cell_list_cap = 0;
  int cell_count_cap = 0;
  PtKArgs args_last;                    // what args_dev holds
  bool args_valid = false;
  hipStream_t args_stream = nullptr;
  unsigned long long *ray_counter_host = nullptr;  // pinned
  hipStream_t stream = nullptr;
  hipStream_t last_stream = nullptr;  // stream of the most recent launch
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  hipEvent_t ev_count = nullptr;  // recorded behind the copy of the ray count into ray_counter_host
  bool count_pending = false;     // the last launch enqueued that copy: fold_stats waits for ev_count
  bool launched = false;          // last_stream names a stream this scene has work on
  int vgprs_last = 0;             // registers per lane of the render kernel launched last
  std::vector<hipEvent_t> prof;  // 2 * capacity events while profiling
  int prof_used = 0;             // pairs recorded
  bool profiling = false;
  bool count_rays = true;
  bool dome_shortcut = true;  // pt_set_dome_shortcut
  bool timing = true;  // bracket render kernels with hipEvents
  bool pending = false;  // an async render whose stats are not folded yet
  bool pending_copy = false;
  bool stats_valid = true;  // ev0/ev1 bracket the last launch (false while the profiling ring is used)
  bool hoist_valid = false;  // s->hoist holds the constants of hoist_cam
  pt_camera hoist_cam = {};
  hipStream_t hoist_stream = nullptr;  // the stream the constants were produced on
  int n_cu = 256;
  pt_stats stats = {};
  // handles made by pt_scene_clone share the scene's tables (everything pt_scene_upload wrote and no launch changes);
  // the last handle of the family to be freed frees them
  std::atomic<int> *family = nullptr;
};

extern "C" int pt_version(void) { return PT_VERSION; }

extern "C" int pt_device_kernargs(void) {
  const char *v = getenv("HIP_FORCE_DEV_KERNARG");
  return v && atoi(v) != 0 ? 1 : 0;
}

extern "C" int pt_last_error(char *buf, size_t n) {
  const size_t len = strlen(g_err);
  if (buf && n) {
    const size_t c = std::min(len, n - 1);
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)len;
}

extern "C" int pt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int pt_device_info(int device, int *compute_units, int *clock_khz) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  return PT_OK;
}

extern "C" int pt_rows_for_rank(const pt_params *p) {
  if (!p || p->height <= 0) return 0;
  const int rb = p->row_block > 0 ? p->row_block : 1;
  const int nr = p->n_ranks > 0 ? p->n_ranks : 1;
  int rows = 0;
  for (int b = 0; b * rb < p->height; ++b)
    if (b % nr == p->rank) rows += std::min(rb, p->height - b * rb);
  return rows;
}

extern "C" size_t pt_output_bytes(const pt_params *p) {
  if (!p) return 0;
  return (size_t)pt_rows_for_rank(p) * (size_t)p->width * 3 * (p->out_format == PT_OUT_F32 ? 4 : 8);
}

static int check_desc(const pt_scene_desc *d) {
  if (!d) return fail(PT_ERR_INVALID, "null scene descriptor");
  if (d->n_shapes < 0 || d->n_lights < 0 || d->n_textures < 0)
    return fail(PT_ERR_INVALID, "negative count in scene descriptor");
  if (d->n_shapes > 0 &&
      (!d->kind || !d->invm || !d->m || !d->brdf_kind || !d->brdf_param || !d->pig_kind ||
       !d->pig_c1 || !d->pig_c2 || !d->pig_steps || !d->pig_tex || !d->emi_kind || !d->emi_c1 ||
       !d->emi_c2 || !d->emi_steps || !d->emi_tex))
    return fail(PT_ERR_INVALID, "null array in scene descriptor");
  if (d->n_lights > 0 && (!d->light_pos || !d->light_color || !d->light_radius))
    return fail(PT_ERR_INVALID, "null light array in scene descriptor");
  if (d->n_textures > 0 && (!d->tex_w || !d->tex_h || !d->tex_offset || !d->tex_data))
    return fail(PT_ERR_INVALID, "null texture array in scene descriptor");
  for (int i = 0; i < d->n_shapes; ++i) {
    if (d->kind[i] != PT_SHAPE_SPHERE && d->kind[i] != PT_SHAPE_PLANE)
      return fail(PT_ERR_INVALID, "shape %d: unknown kind %d", i, d->kind[i]);
    if (d->brdf_kind[i] != PT_BRDF_DIFFUSE && d->brdf_kind[i] != PT_BRDF_SPECULAR)
      return fail(PT_ERR_INVALID, "shape %d: unknown BRDF kind %d", i, d->brdf_kind[i]);
    const int pk[2] = {d->pig_kind[i], d->emi_kind[i]};
    const int pt[2] = {d->pig_tex[i], d->emi_tex[i]};
    for (int k = 0; k < 2; ++k) {
      if (pk[k] < PT_PIGMENT_UNIFORM || pk[k] > PT_PIGMENT_IMAGE)
        return fail(PT_ERR_INVALID, "shape %d: unknown pigment kind %d", i, pk[k]);
      if (pk[k] == PT_PIGMENT_IMAGE && (pt[k] < 0 || pt[k] >= d->n_textures))
        return fail(PT_ERR_INVALID, "shape %d: texture index %d out of range", i, pt[k]);
    }
  }
  for (int t = 0; t < d->n_textures; ++t)
    if (d->tex_w[t] <= 0 || d->tex_h[t] <= 0 || d->tex_offset[t] < 0)
      return fail(PT_ERR_INVALID, "texture %d: bad size/offset", t);
  return PT_OK;
}

extern "C" void pt_scene_free(pt_scene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->launched && s->last_stream) (void)hipStreamSynchronize(s->last_stream);
  // the tables the family shares: freed by its last handle (a scene that was never cloned is a family of one)
  const bool last = !s->family || s->family->fetch_sub(1) == 1;
  if (last) {
    (void)hipFree(s->recs);
    (void)hipFree(s->aux);
    (void)hipFree(s->diag);
    (void)hipFree(s->bounds);
    (void)hipFree(s->bsoa);
    (void)hipFree(s->grid_cells);
    (void)hipFree(s->grid_occ);
    (void)hipFree(s->grid_balls);
    (void)hipFree(s->grid_slots);
    (void)hipFree(s->grid_always);
    (void)hipFree(s->lights);
    (void)hipFree(s->tex);
    (void)hipFree(s->tex_data);
    delete s->family;
  }
  (void)hipFree(s->hoist);
  (void)hipFree(s->hoist_diag);
  (void)hipFree(s->ws);
  (void)hipFree(s->handover);
  (void)hipFree(s->units_handed);
  (void)hipFree(s->out_dev);
  (void)hipFree(s->ray_counter);
  (void)hipFree(s->ray_partials);
  (void)hipFree(s->queue);
  (void)hipFree(s->args_dev);
  (void)hipFree(s->args_dev2);
  (void)hipFree(s->region_keys);
  (void)hipFree(s->units);
  (void)hipFree(s->region_mask);
  (void)hipFree(s->cell_list);
  (void)hipFree(s->cell_count);
  if (s->ray_counter_host) (void)hipHostFree(s->ray_counter_host);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->ev2) (void)hipEventDestroy(s->ev2);
  if (s->ev_count) (void)hipEventDestroy(s->ev_count);
  for (hipEvent_t e : s->prof) (void)hipEventDestroy(e);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src) {
  const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  HIP_TRY(hipMalloc((void **)dst, bytes));
  if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return PT_OK;
}

// what every handle has of its own besides the per-camera constants: counters, queue blocks, the argument block, a stream
// and events.  On failure the handle is freed.
static int handle_state(pt_scene *s) {
  auto hip_or_free = [&](hipError_t e, const char *what) -> int {
    if (e == hipSuccess) return PT_OK;
    int code = fail(PT_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    pt_scene_free(s);
    return code;
  };
  int rc;
  if ((rc = hip_or_free(hipMalloc((void **)&s->ray_counter, 2 * sizeof(unsigned long long)), "hipMalloc(counter)"))) return rc;
  if ((rc = hip_or_free(hipMalloc((void **)&s->queue, 2 * PT_QUEUE_WORDS * sizeof(unsigned long long)), "hipMalloc(queue)"))) return rc;
  if ((rc = hip_or_free(hipMalloc((void **)&s->args_dev, sizeof(PtKArgs)), "hipMalloc(args)"))) return rc;
  if ((rc = hip_or_free(hipMalloc((void **)&s->args_dev2, sizeof(PtKArgs)), "hipMalloc(args)"))) return rc;
  if ((rc = hip_or_free(hipHostMalloc((void **)&s->ray_counter_host, 3 * sizeof(unsigned long long)), "hipHostMalloc"))) return rc;
  s->ray_counter_host[0] = s->ray_counter_host[1] = s->ray_counter_host[2] = 0;
  if ((rc = hip_or_free(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate"))) return rc;
  if ((rc = hip_or_free(hipEventCreate(&s->ev0), "hipEventCreate"))) return rc;
  if ((rc = hip_or_free(hipEventCreate(&s->ev1), "hipEventCreate"))) return rc;
  if ((rc = hip_or_free(hipEventCreate(&s->ev2), "hipEventCreate"))) return rc;
  if ((rc = hip_or_free(hipEventCreateWithFlags(&s->ev_count, hipEventDisableTiming), "hipEventCreate"))) return rc;
  return PT_OK;
}

// What pt_scene_upload computes on the HOST before anything is uploaded: the records in slot order, the culling bounds, the
// ball hierarchy, the uniform grid.  No HIP call: pt_debug_plan runs it for a scene DESCRIPTION on any machine (the scalars
// land in `s`, used as a plain bag of facts there).
struct HostTables {
  std::vector<PtShapeRec> recs;
  std::vector<PtShapeAux> aux;
  std::vector<PtDiagRec> diag;
  std::vector<float4> bounds;
  std::vector<float> bsoa;
  std::vector<PtLight> lights;
  std::vector<PtTex> tex;
  std::vector<double> tex_data;
  bool has_grid = false;
  std::vector<unsigned> grid_cells, grid_occ;
  std::vector<unsigned short> grid_slots;
  std::vector<float4> grid_balls;
  std::vector<int> grid_always;
};

static void analyse_scene(const pt_scene_desc *d, const PtTuning &tn, pt_scene *s, HostTables &h) {
  s->n_shapes = d->n_shapes;
  s->n_lights = d->n_lights;
  s->n_textures = d->n_textures;
  const int n = d->n_shapes;

  // group the records: scale+translate spheres, other spheres, planes — each group in World.shapes order
  auto is_diag = [&](int i) {
    if (d->kind[i] != PT_SHAPE_SPHERE) return false;
    const int off[6] = {1, 2, 4, 6, 8, 9};
    for (int k : off)
      if (d->invm[(size_t)k * n + i] != 0.0) return false;
    const int dia[3] = {0, 5, 10};
    for (int k : dia) {
      const double v = std::fabs(d->invm[(size_t)k * n + i]);
      if (!(v >= 1e-100 && v <= 1e100)) return false;
    }
    for (int k : {3, 7, 11})
      if (!std::isfinite(d->invm[(size_t)k * n + i])) return false;
    return true;
  };
  std::vector<int> order;
  order.reserve(n);
  for (int i = 0; i < n; ++i)
    if (is_diag(i)) order.push_back(i);
  s->n_diag = (int)order.size();
  for (int i = 0; i < n; ++i)
    if (d->kind[i] == PT_SHAPE_SPHERE && !is_diag(i)) order.push_back(i);
  s->n_spheres = (int)order.size();
  for (int i = 0; i < n; ++i)
    if (d->kind[i] != PT_SHAPE_SPHERE) order.push_back(i);
  // Large scenes: within each sphere group the slots follow a Morton curve through the centres (the few
  // spheres much larger than the rest first), so that 8 and 64 consecutive slots are close in space and
  // a ball around them is tight (per-ray prefilter of scattered and shadow rays, world_query_lanes).
  // Any slot order gives the same image: ties in t go to the lower World.shapes index (r.index).
  s->bs_levels = s->n_spheres >= 128 ? 1 : 0;
  if (s->bs_levels) {
    auto centre = [&](int i, int k) { return d->m[(size_t)(3 + 4 * k) * n + i]; };
    auto radius2 = [&](int i) {  // squared Frobenius norm of M's 3x3 block: a size, not a bound
      double v = 0.0;
      for (int r_ = 0; r_ < 3; ++r_)
        for (int c_ = 0; c_ < 3; ++c_) v += d->m[(size_t)(r_ * 4 + c_) * n + i] * d->m[(size_t)(r_ * 4 + c_) * n + i];
      return v;
    };
    std::vector<double> sizes;
    for (int k = 0; k < s->n_spheres; ++k) sizes.push_back(radius2(order[k]));
    std::vector<double> sorted_sizes = sizes;
    std::nth_element(sorted_sizes.begin(), sorted_sizes.begin() + sorted_sizes.size() / 2, sorted_sizes.end());
    const double big = 64.0 * sorted_sizes[sorted_sizes.size() / 2];  // 8x the median radius
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < s->n_spheres; ++k) {
      if (!(sizes[k] <= big)) continue;
      for (int c_ = 0; c_ < 3; ++c_) {
        const double v = centre(order[k], c_);
        if (std::isfinite(v)) {
          lo[c_] = std::min(lo[c_], v);
          hi[c_] = std::max(hi[c_], v);
        }
      }
    }
    auto morton = [&](int i) {
      uint64_t code = 0;
      uint32_t q[3];
      for (int c_ = 0; c_ < 3; ++c_) {
        const double v = centre(i, c_), span = hi[c_] - lo[c_];
        double u = (span > 0.0 && std::isfinite(v)) ? (v - lo[c_]) / span : 0.0;
        u = std::min(1.0, std::max(0.0, u));
        q[c_] = (uint32_t)(u * 2097151.0);  // 21 bits
      }
      for (int b = 20; b >= 0; --b)
        for (int c_ = 0; c_ < 3; ++c_) code = (code << 1) | ((q[c_] >> b) & 1u);
      return code;
    };
    auto sort_range = [&](int a0, int a1) {
      std::vector<std::pair<std::pair<int, uint64_t>, int>> keyed;  // ((small?, code), shape)
      for (int k = a0; k < a1; ++k) {
        const bool small_ = sizes[k] <= big;
        // large spheres first, largest leading; then the Morton order of the rest
        const uint64_t code = small_ ? morton(order[k]) : (uint64_t)(k - a0);
        keyed.push_back({{small_ ? 1 : 0, code}, order[k]});
      }
      std::stable_sort(keyed.begin(), keyed.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
      for (int k = a0; k < a1; ++k) order[k] = keyed[k - a0].second;
    };
    sort_range(0, s->n_diag);
    sort_range(s->n_diag, s->n_spheres);
  }
  std::vector<PtShapeRec> recs(n);
  std::vector<PtShapeAux> aux(n);
  for (int slot = 0; slot < n; ++slot) {
    const int i = order[slot];
    PtShapeRec &r = recs[slot];
    PtShapeAux &x = aux[slot];
    memset(&r, 0, sizeof r);
    memset(&x, 0, sizeof x);
    for (int k = 0; k < 12; ++k) {
      r.invm[k] = d->invm[(size_t)k * n + i];
      x.m[k] = d->m[(size_t)k * n + i];
    }
    r.kind = d->kind[i];
    for (int k = 0; k < 3; ++k) {
      x.pig_c1[k] = d->pig_c1[(size_t)k * n + i];
      x.pig_c2[k] = d->pig_c2[(size_t)k * n + i];
      x.emi_c1[k] = d->emi_c1[(size_t)k * n + i];
      x.emi_c2[k] = d->emi_c2[(size_t)k * n + i];
    }
    x.pig_steps = d->pig_steps[i];
    x.emi_steps = d->emi_steps[i];
    x.brdf_param = d->brdf_param[i];
    x.brdf_kind = d->brdf_kind[i];
    x.pig_kind = d->pig_kind[i];
    x.emi_kind = d->emi_kind[i];
    x.pig_tex = d->pig_tex[i];
    x.emi_tex = d->emi_tex[i];
    x.needs_uv = (d->pig_kind[i] != PT_PIGMENT_UNIFORM || d->emi_kind[i] != PT_PIGMENT_UNIFORM) ? 1 : 0;
    // both pigments uniform: color2 of the (uniform) BRDF pigment is never read, and the slot carries what FlatRenderer
    // returns for the shape, pigment + emitted (render.py:65-74; the same fp64 addition the kernel would do per pixel)
    if (!x.needs_uv)
      for (int k = 0; k < 3; ++k) x.pig_c2[k] = x.pig_c1[k] + x.emi_c1[k];
    r.index = i;
    // |invm|_F^2 for the "camera inside this sphere" shortcut of the tile kernel; +inf disables it unless
    // every singular value of invm's 3x3 block is within 1e-6 .. 1e6 (Gershgorin bounds of invm^T invm)
    r.fro2 = INFINITY;
    if (r.kind == PT_SHAPE_SPHERE) {
      double A[3][3], fro2 = 0.0;
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          A[p][q] = 0.0;
          for (int k = 0; k < 3; ++k) A[p][q] += r.invm[k * 4 + p] * r.invm[k * 4 + q];
        }
      double lmin = INFINITY, lmax = 0.0;
      for (int p = 0; p < 3; ++p) {
        double off = 0.0;
        for (int q = 0; q < 3; ++q)
          if (q != p) off += std::fabs(A[p][q]);
        lmin = std::min(lmin, A[p][p] - off);
        lmax = std::max(lmax, A[p][p] + off);
        fro2 += A[p][p];
      }
      if (std::isfinite(fro2) && lmin >= 1e-12 && lmax <= 1e12) r.fro2 = fro2 * (1.0 + 1e-9);
    }
  }
  std::vector<PtLight> lights(d->n_lights);
  for (int l = 0; l < d->n_lights; ++l) {
    memset(&lights[l], 0, sizeof(PtLight));
    for (int k = 0; k < 3; ++k) {
      lights[l].pos[k] = d->light_pos[(size_t)k * d->n_lights + l];
      lights[l].color[k] = d->light_color[(size_t)k * d->n_lights + l];
    }
    lights[l].radius = d->light_radius[l];
  }
  std::vector<PtTex> tex(d->n_textures);
  size_t tex_doubles = 0;
  for (int t = 0; t < d->n_textures; ++t) {
    tex[t].w = d->tex_w[t];
    tex[t].h = d->tex_h[t];
    tex[t].offset = d->tex_offset[t];
    tex_doubles = std::max(tex_doubles, (size_t)d->tex_offset[t] + (size_t)d->tex_w[t] * d->tex_h[t] * 3);
  }
  std::vector<double> tex_data(d->tex_data, d->tex_data + tex_doubles);

  std::vector<PtDiagRec> diag(s->n_diag);
  for (int slot = 0; slot < s->n_diag; ++slot) {
    PtDiagRec &g = diag[slot];
    memset(&g, 0, sizeof g);
    const double *im = recs[slot].invm;
    g.s[0] = im[0];
    g.s[1] = im[5];
    g.s[2] = im[10];
    g.t[0] = im[3];
    g.t[1] = im[7];
    g.t[2] = im[11];
    g.tnz = (im[3] != 0.0 ? 1 : 0) | (im[7] != 0.0 ? 2 : 0) | (im[11] != 0.0 ? 4 : 0);
  }
  // bounding spheres for tile culling: radius = a rigorous upper bound of the spectral norm of M's 3x3 block
  struct Bound64 {
    double cx, cy, cz, r;
  };
  std::vector<float4> bounds(n);
  for (int slot = 0; slot < n; ++slot) {
    Bound64 b;
    const double *m = aux[slot].m;
    b.cx = m[3];
    b.cy = m[7];
    b.cz = m[11];
    b.r = -1.0;
    if (recs[slot].kind == PT_SHAPE_SPHERE) {
      double A[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A[i][j] = 0.0;
          for (int k = 0; k < 3; ++k) A[i][j] += m[k * 4 + i] * m[k * 4 + j];
        }
      // Gershgorin: lambda_max(M^T M) <= max_i sum_j |(M^T M)_ij|  (exact for scale/rotation blocks)
      double lam = 0.0;
      for (int i = 0; i < 3; ++i)
        lam = std::max(lam, std::fabs(A[i][0]) + std::fabs(A[i][1]) + std::fabs(A[i][2]));
      // The exact test uses invm, the bound uses m: check that m really inverts invm (the reference
      // stores both, transformations.py:48-56) and widen the radius by the residual; else never cull.
      const double *im = recs[slot].invm;
      double resid = 0.0;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
          double e = (i == j) ? -1.0 : 0.0;
          for (int k = 0; k < 3; ++k) e += im[i * 4 + k] * m[k * 4 + j];
          resid = std::max(resid, std::fabs(e));
        }
        double e = im[i * 4 + 3];
        for (int k = 0; k < 3; ++k) e += im[i * 4 + k] * m[k * 4 + 3];
        resid = std::max(resid, std::fabs(e));
      }
      const double r = std::sqrt(lam) * (1.0 + 1e-9 + 16.0 * resid);
      const bool finite = std::isfinite(r) && std::isfinite(b.cx) && std::isfinite(b.cy) &&
                          std::isfinite(b.cz) && resid < 1e-6;
      b.r = finite ? r : -1.0;
    }
    // to fp32: widen r by the rounding of the centre (<= 2^-24 relative per component) and of r itself
    float4 f;
    f.x = (float)b.cx;
    f.y = (float)b.cy;
    f.z = (float)b.cz;
    const double cabs = std::max(std::fabs(b.cx), std::max(std::fabs(b.cy), std::fabs(b.cz)));
    const double rw = b.r * (1.0 + 1e-6) + 2e-7 * cabs;
    f.w = (b.r >= 0.0 && std::isfinite(rw) && rw < 1e37 && cabs < 1e37) ? (float)rw * (1.0f + 1e-6f) : -1.0f;
    if (recs[slot].kind == PT_SHAPE_PLANE) {
      // planes have no bounding sphere; their slot carries what plane_keeps() needs instead: the z row of
      // invm (object-space d.z = row . d, o.z = row . o + invm[11]) rounded to fp32
      const double *im = recs[slot].invm;
      f.x = (float)im[8];
      f.y = (float)im[9];
      f.z = (float)im[10];
      f.w = (float)im[11];
    }
    bounds[slot] = f;
  }
  // ... and as structure-of-arrays for the per-ray prefilter of scattered rays (world_query_lanes): two
  // neighbouring spheres per packed fp32 instruction.  r' = r*(1 + 1e-5) + 1e-6*max|c| rounded up;
  // +inf where there is no bound (the test then always keeps the shape).
  s->bs_stride = (n + 8 + 7) / 8 * 8;  // 8 floats of slack: the prefilter reads eight at a time, 32-byte aligned
  const int n_groups = (s->n_spheres + 7) / 8, n_chunks = (s->n_spheres + 63) / 64;
  s->gs_stride = (n_chunks * 8 + 8 + 7) / 8 * 8;  // eight groups per chunk, read eight at a time
  s->cs_stride = (n_chunks + 8 + 7) / 8 * 8;
  std::vector<float> bsoa((size_t)4 * (s->bs_stride + s->gs_stride + s->cs_stride), 0.0f);
  for (int slot = 0; slot < s->bs_stride; ++slot) {
    float rk = INFINITY;
    if (slot < n) {
      const float4 f = bounds[slot];
      bsoa[slot] = f.x;
      bsoa[(size_t)s->bs_stride + slot] = f.y;
      bsoa[(size_t)2 * s->bs_stride + slot] = f.z;
      if (slot < s->n_spheres && f.w >= 0.0f) {
        const double cabs = std::max(std::fabs((double)f.x), std::max(std::fabs((double)f.y), std::fabs((double)f.z)));
        const double v = (double)f.w * (1.0 + 1e-5) + 1e-6 * cabs;
        rk = std::nextafter((float)v, INFINITY);
      }
    }
    bsoa[(size_t)3 * s->bs_stride + slot] = rk;
  }
  // a ball around the (already inflated) balls of slots [a0, a1): centre = middle of the centres' box,
  // radius = max_i(|c_i - centre| + r'_i), rounded up; +inf as soon as one member has no bound
  auto ball_around = [&](int a0, int a1, float *out4) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bounded = a1 > a0;
    for (int k = a0; k < a1; ++k) {
      bounded = bounded && std::isfinite(bsoa[(size_t)3 * s->bs_stride + k]);
      for (int c_ = 0; c_ < 3; ++c_) {
        const double v = bsoa[(size_t)c_ * s->bs_stride + k];
        lo[c_] = std::min(lo[c_], v);
        hi[c_] = std::max(hi[c_], v);
      }
    }
    out4[0] = out4[1] = out4[2] = 0.0f;
    out4[3] = INFINITY;
    if (!bounded) return;
    float c[3];
    for (int c_ = 0; c_ < 3; ++c_) c[c_] = (float)(0.5 * (lo[c_] + hi[c_]));
    double rad = 0.0;
    for (int k = a0; k < a1; ++k) {
      double d2 = 0.0;
      for (int c_ = 0; c_ < 3; ++c_) {
        const double dv = (double)bsoa[(size_t)c_ * s->bs_stride + k] - (double)c[c_];
        d2 += dv * dv;
      }
      rad = std::max(rad, std::sqrt(d2) * (1.0 + 1e-12) + (double)bsoa[(size_t)3 * s->bs_stride + k]);
    }
    const double cabs = std::max(std::fabs((double)c[0]), std::max(std::fabs((double)c[1]), std::fabs((double)c[2])));
    const double v = rad * (1.0 + 1e-5) + 1e-6 * cabs;
    if (!std::isfinite(v) || v > 1e37) return;
    out4[0] = c[0];
    out4[1] = c[1];
    out4[2] = c[2];
    out4[3] = std::nextafter((float)v, INFINITY);
  };
  {
    float *gs = bsoa.data() + (size_t)4 * s->bs_stride, *cs = gs + (size_t)4 * s->gs_stride;
    for (int k = 0; k < s->gs_stride; ++k) {
      float b4[4] = {0.0f, 0.0f, 0.0f, INFINITY};
      if (k < n_groups) ball_around(k * 8, std::min(k * 8 + 8, s->n_spheres), b4);
      for (int c_ = 0; c_ < 4; ++c_) gs[(size_t)c_ * s->gs_stride + k] = b4[c_];
    }
    for (int k = 0; k < s->cs_stride; ++k) {
      float b4[4] = {0.0f, 0.0f, 0.0f, INFINITY};
      if (k < n_chunks) ball_around(k * 64, std::min(k * 64 + 64, s->n_spheres), b4);
      for (int c_ = 0; c_ < 4; ++c_) cs[(size_t)c_ * s->cs_stride + k] = b4[c_];
    }
  }
  for (int slot = 0; slot < s->n_spheres; ++slot)
    if (aux[slot].needs_uv == 0 && std::isfinite(recs[slot].fro2)) {
      pt_scene::DomeCand dc;
      dc.slot = slot;
      memcpy(dc.invm, recs[slot].invm, sizeof dc.invm);
      s->dome_cands.push_back(dc);
    }
  h.recs = recs;
  h.bounds = bounds;
  // ---- uniform grid over the ordinary spheres (scenes of >= 128 spheres) ----
  // A sphere is entered into every cell that the box around its ball (the r' of the per-ray prefilter, already
  // inflated) overlaps after widening it by eps = 2e-3 cell + 1e-4 max|coordinate|.  The walk (world_query_lanes)
  // runs a 3D-DDA in fp32 on the fp32 copy of the ray: that copy stays within ~1e-7 |coordinate| x a few of the
  // true ray, and the accumulated rounding of the DDA's crossing parameters (<= 200 steps x 2^-24) can make it
  // enter a face or skip a corner cell up to ~1.2e-5 x the grid's extent early or late; eps (>= 3e-5 extent, since
  // a cell is >= 1/64 of it) covers both, so a point where the true ray meets a sphere always lies within eps of a
  // visited cell, i.e. in a cell the sphere is entered in.  Spheres much larger than the rest (8x the median
  // radius: a dome would be in every cell) or without a bound go to the "always" list.
  if (tn.grid && s->bs_levels && s->n_spheres <= 65535) {
    auto ball = [&](int k, int q) { return bsoa[(size_t)q * s->bs_stride + k]; };  // q: 0..2 centre, 3 radius r'
    std::vector<float> radii;
    for (int k = 0; k < s->n_spheres; ++k)
      if (std::isfinite(ball(k, 3))) radii.push_back(ball(k, 3));
    std::vector<int> always, inside;
    float big = INFINITY;
    if (!radii.empty()) {
      std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
      big = 8.0f * radii[radii.size() / 2];
    }
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, cmax = 0.0;
    for (int k = 0; k < s->n_spheres; ++k) {
      const float r = ball(k, 3);
      if (!std::isfinite(r) || r > big) {
        always.push_back(k);
        continue;
      }
      inside.push_back(k);
      for (int q = 0; q < 3; ++q) {
        lo[q] = std::min(lo[q], (double)ball(k, q) - r);
        hi[q] = std::max(hi[q], (double)ball(k, q) + r);
      }
    }
    // (below ~1000 spheres the exhaustive packed prefilter and the cell walk cost the same -- measured on C4's 256 --
    //  and the prefilter has the sparse path for the deep stragglers; the grid wins 2.5-3.3x at 10 000)
    if ((int)inside.size() >= std::max(64, (int)tn.grid_min)) {
      double ext[3], vol = 1.0;
      for (int q = 0; q < 3; ++q) {
        const double pad = 1e-3 * (hi[q] - lo[q]) + 1e-4 * (1.0 + std::max(std::fabs(lo[q]), std::fabs(hi[q])));
        lo[q] -= pad;
        hi[q] += pad;
        ext[q] = hi[q] - lo[q];
        vol *= ext[q];
        cmax = std::max(cmax, std::max(std::fabs(lo[q]), std::fabs(hi[q])));
      }
      const double target = std::min<double>(32768.0, std::max<double>(64.0, tn.grid_density * (double)inside.size()));  // (cells per sphere)
      const double side = std::cbrt(vol / target);
      long long ncell = 1;
      for (int q = 0; q < 3; ++q) {
        s->grid_res[q] = (int)std::min(64.0, std::max(1.0, std::ceil(ext[q] / side)));
        s->grid_min[q] = (float)lo[q];
        s->grid_max[q] = (float)hi[q];
        s->grid_cell[q] = (float)(ext[q] / s->grid_res[q]);
        s->grid_inv[q] = (float)(s->grid_res[q] / ext[q]);
        ncell *= s->grid_res[q];
      }
      std::vector<std::vector<unsigned short>> cells((size_t)ncell);
      size_t items = 0;
      bool ok = std::isfinite(vol) && vol > 0.0 && ncell <= 65535;  // (cell ids travel in 16 bits)
      for (int k : inside) {
        if (!ok) break;
        int c0[3], c1[3];
        for (int q = 0; q < 3; ++q) {
          const double eps = 2e-3 * s->grid_cell[q] + 1e-4 * cmax, c = ball(k, q), r = ball(k, 3);
          c0[q] = std::max(0, std::min(s->grid_res[q] - 1, (int)std::floor((c - r - eps - lo[q]) * s->grid_res[q] / ext[q])));
          c1[q] = std::max(0, std::min(s->grid_res[q] - 1, (int)std::floor((c + r + eps - lo[q]) * s->grid_res[q] / ext[q])));
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
          for (int y = c0[1]; y <= c1[1]; ++y)
            for (int x = c0[0]; x <= c1[0]; ++x) {
              auto &cell = cells[((size_t)z * s->grid_res[1] + y) * s->grid_res[0] + x];
              cell.push_back((unsigned short)k);
              ++items;
              ok = ok && cell.size() <= 255 && items <= ((size_t)1 << 23);
            }
      }
      if (ok) {
        std::vector<unsigned> words((size_t)ncell), occ((size_t)(ncell + 31) / 32 + 1, 0u);
        std::vector<unsigned short> slots;
        std::vector<float4> balls;
        for (size_t cidx = 0; cidx < (size_t)ncell; ++cidx) {
          words[cidx] = ((unsigned)slots.size() << 8) | (unsigned)cells[cidx].size();
          if (!cells[cidx].empty()) occ[cidx >> 5] |= 1u << (cidx & 31);
          for (unsigned short k : cells[cidx]) {
            slots.push_back(k);
            float4 b;
            b.x = ball(k, 0);
            b.y = ball(k, 1);
            b.z = ball(k, 2);
            b.w = ball(k, 3);
            bool ordinary;
            pt_ball_square(&b.x, &b.y, &b.z, &b.w, &ordinary);
            balls.push_back(b);
          }
        }
        h.grid_cells = words;
        h.grid_occ = occ;
        h.grid_slots = slots;
        h.grid_balls = balls;
        h.grid_always = always;
        h.has_grid = true;
        s->grid_n_always = (int)always.size();
        // the margin a sphere is entered with, >= 1e-4 * cmax, covers the fp32 copy of a ray whose origin lies within
        // ~100 x the grid's coordinates (1.2e-7 |o| <= a quarter of the margin); a ray from farther away takes the
        // exhaustive filter instead of the walk
        s->grid_far_eo = (float)(1e-4 * cmax);
        s->grid_n_cells = (int)ncell;
      }
    }
  }
  // the filter compares squares (world_query_lanes): r' -> r'^2 rounded up, in all three tables
  {
    float *tab[3] = {bsoa.data(), bsoa.data() + (size_t)4 * s->bs_stride, bsoa.data() + (size_t)4 * (s->bs_stride + s->gs_stride)};
    const int stride[3] = {s->bs_stride, s->gs_stride, s->cs_stride};
    for (int lv = 0; lv < 3; ++lv) {
      float rmax = 0.0f;
      for (int k = 0; k < stride[lv]; ++k) {
        float *x = tab[lv] + k, *y = x + stride[lv], *z = y + stride[lv], *r = z + stride[lv];
        const float rp = *r;
        bool ordinary;
        pt_ball_square(x, y, z, r, &ordinary);
        if (ordinary) rmax = std::max(rmax, rp);
      }
      s->bs_rmax[lv] = rmax;
    }
  }
  h.bsoa = bsoa;
  h.diag = diag;
  h.aux = aux;
  h.lights = lights;
  h.tex = tex;
  h.tex_data = tex_data;
}

static PtSceneFacts scene_facts(const pt_scene *s) {
  PtSceneFacts f;
  f.n_shapes = s->n_shapes;
  f.n_spheres = s->n_spheres;
  f.n_diag = s->n_diag;
  f.n_lights = s->n_lights;
  f.bs_levels = s->bs_levels;
  f.has_grid = s->grid_n_cells > 0 ? 1 : 0;
  f.grid_n_cells = s->grid_n_cells;
  f.n_cu = s->n_cu;
  f.dome_shortcut = s->dome_shortcut ? 1 : 0;
  return f;
}

extern "C" int pt_scene_upload(const pt_scene_desc *d, int device, pt_scene **out) {
  if (!out) return fail(PT_ERR_INVALID, "null output handle");
  *out = nullptr;
  int rc = check_desc(d);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PT_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  pt_scene *s = new (std::nothrow) pt_scene();
  if (!s) return fail(PT_ERR_NOMEM, "out of host memory");
  s->device = device;
  HostTables h;
  analyse_scene(d, pt_tuning(), s, h);
#define UP(call)            \
  do {                      \
    rc = (call);            \
    if (rc) {               \
      pt_scene_free(s);     \
      return rc;            \
    }                       \
  } while (0)
  UP(upload(&s->recs, h.recs));
  UP(upload(&s->bounds, h.bounds));
  if (h.has_grid) {
    UP(upload(&s->grid_cells, h.grid_cells));
    UP(upload(&s->grid_occ, h.grid_occ));
    UP(upload(&s->grid_slots, h.grid_slots));
    UP(upload(&s->grid_balls, h.grid_balls));
    UP(upload(&s->grid_always, h.grid_always));
  }
  UP(upload(&s->bsoa, h.bsoa));
  UP(upload(&s->diag, h.diag));
  {
    std::vector<PtHoistDiag> hd(std::max(s->n_diag, 1));
    UP(upload(&s->hoist_diag, hd));
  }
  UP(upload(&s->aux, h.aux));
  UP(upload(&s->lights, h.lights));
  UP(upload(&s->tex, h.tex));
  UP(upload(&s->tex_data, h.tex_data));
  {
    std::vector<PtHoist> hh(std::max(s->n_shapes, 1));
    UP(upload(&s->hoist, hh));
  }
#undef UP
  if ((rc = handle_state(s))) return rc;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->n_cu = prop.multiProcessorCount;
  s->family = new std::atomic<int>(1);
  *out = s;
  return PT_OK;
}

// A second handle on the same scene: it shares the tables pt_scene_upload made (nothing a launch writes) and has its
// own per-camera constants, queues, counters, workspace, stream and events -- so frames rendered through different
// handles of a family may be in flight at the same time, each on its own stream (include/ptrace.h).
extern "C" int pt_scene_clone(pt_scene *src, pt_scene **out) {
  if (!src || !out) return fail(PT_ERR_INVALID, "null argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(src->device));
  pt_scene *s = new pt_scene();
  // what describes the scene
  s->device = src->device;
  s->n_shapes = src->n_shapes; s->n_spheres = src->n_spheres; s->n_lights = src->n_lights; s->n_textures = src->n_textures;
  s->recs = src->recs; s->aux = src->aux; s->diag = src->diag; s->bounds = src->bounds;
  s->grid_cells = src->grid_cells; s->grid_occ = src->grid_occ; s->grid_balls = src->grid_balls; s->grid_slots = src->grid_slots;
  s->grid_always = src->grid_always; s->grid_n_always = src->grid_n_always; s->grid_n_cells = src->grid_n_cells;
  s->grid_far_eo = src->grid_far_eo;
  for (int q = 0; q < 3; ++q) {
    s->grid_res[q] = src->grid_res[q]; s->grid_min[q] = src->grid_min[q]; s->grid_max[q] = src->grid_max[q];
    s->grid_cell[q] = src->grid_cell[q]; s->grid_inv[q] = src->grid_inv[q]; s->bs_rmax[q] = src->bs_rmax[q];
  }
  s->bsoa = src->bsoa; s->bs_stride = src->bs_stride; s->gs_stride = src->gs_stride; s->cs_stride = src->cs_stride;
  s->bs_levels = src->bs_levels; s->n_diag = src->n_diag;
  s->lights = src->lights; s->tex = src->tex; s->tex_data = src->tex_data;
  s->dome_cands = src->dome_cands;
  s->n_cu = src->n_cu;
  s->count_rays = src->count_rays; s->dome_shortcut = src->dome_shortcut; s->timing = src->timing;
  s->family = src->family;
  s->family->fetch_add(1);
  // what is this handle's own
  int rc = PT_OK;
  const size_t nh = (size_t)std::max(s->n_shapes, 1), nd = (size_t)std::max(s->n_diag, 1);
  if (hipMalloc((void **)&s->hoist, nh * sizeof(PtHoist)) != hipSuccess || hipMalloc((void **)&s->hoist_diag, nd * sizeof(PtHoistDiag)) != hipSuccess) {
    pt_scene_free(s);
    return fail(PT_ERR_NOMEM, "hipMalloc(per-camera constants) failed");
  }
  if ((rc = handle_state(s))) return rc;
  *out = s;
  return PT_OK;
}

extern "C" int pt_set_dome_shortcut(pt_scene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  s->dome_shortcut = enable != 0;
  return PT_OK;
}

extern "C" int pt_set_count_rays(pt_scene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  s->count_rays = enable != 0;
  return PT_OK;
}

static int check_params(const pt_scene *s, const pt_camera *cam, const pt_params *p) {
  if (!s || !cam || !p) return fail(PT_ERR_INVALID, "null argument");
  if (p->width <= 0 || p->height <= 0) return fail(PT_ERR_INVALID, "bad image size %dx%d", p->width, p->height);
  if (p->samples_per_side < 0 || p->samples_per_side > 1024)
    return fail(PT_ERR_INVALID, "bad samples_per_side %d", p->samples_per_side);
  if (p->renderer < PT_RENDERER_ONOFF || p->renderer > PT_RENDERER_POINTLIGHT)
    return fail(PT_ERR_INVALID, "unknown renderer %d", p->renderer);
  if (cam->kind != PT_CAMERA_ORTHOGONAL && cam->kind != PT_CAMERA_PERSPECTIVE)
    return fail(PT_ERR_INVALID, "unknown camera kind %d", cam->kind);
  if (p->out_format != PT_OUT_F64 && p->out_format != PT_OUT_F32)
    return fail(PT_ERR_INVALID, "unknown output format %d", p->out_format);
  const int nr = p->n_ranks > 0 ? p->n_ranks : 1;
  if (p->rank < 0 || p->rank >= nr) return fail(PT_ERR_INVALID, "rank %d outside [0,%d)", p->rank, nr);
  // PT_PCG_SEQ, the reference's own streams: the JITTER stream is one sequential generator from which every sample draws
  // exactly two numbers (imagetracer.py:84-101), so sample k of pixel i starts 2 (i S^2 + k) draws in -- parallel by
  // jump-ahead, and exact.  The path tracer's SCATTERING stream (render.py:118,128) is consumed in the order the paths of
  // ALL pixels end in, each taking a number of draws only known once it has been traced: serial by construction.
  if (p->pcg_mode == PT_PCG_SEQ && p->renderer == PT_RENDERER_PATHTRACER)
    return fail(PT_ERR_UNSUPPORTED,
                "PT_PCG_SEQ with the path tracer: its scattering stream is ONE generator consumed in the order the paths "
                "of all pixels end in (render.py:118,128), which only a serial program reproduces; the device path "
                "implements PT_PCG_PIXEL and PT_PCG_SAMPLE for it (PT_PCG_SEQ is exact for OnOff, Flat and PointLight)");
  if (p->pcg_mode == PT_PCG_SEQ && p->samples_per_side > 0 &&
      (unsigned long long)p->width * p->height > (~0ULL >> 2) / ((unsigned long long)p->samples_per_side * p->samples_per_side))
    return fail(PT_ERR_INVALID, "PT_PCG_SEQ: the frame draws more than 2^63 jitter numbers");
  if (p->pcg_mode < PT_PCG_SEQ || p->pcg_mode > PT_PCG_SAMPLE)
    return fail(PT_ERR_INVALID, "unknown pcg_mode %d", p->pcg_mode);
  if (p->renderer == PT_RENDERER_PATHTRACER) {
    if (p->num_of_rays < 1)
      return fail(PT_ERR_INVALID, "num_of_rays must be >= 1 (the reference divides by it, render.py:139)");
    if (p->max_depth > 4096) return fail(PT_ERR_INVALID, "max_depth %d too large", p->max_depth);
  }
  if ((long long)p->width * p->height > (1LL << 40)) return fail(PT_ERR_INVALID, "image too large");
  return PT_OK;
}

// dynamic LDS above the default 64 KiB has to be asked for, once per kernel
static hipError_t path_lds_limit(const void *kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PT_LDS_BUDGET);
}

// The fp32 model of the primary rays that the culling cones are built from (tile_cone, pixel_cone):
// camera.py:116-124 + imagetracer.py:56-58 folded into d(x, y) = d0 + x * dx + y * dy (perspective: directions,
// common origin in `apex`; orthogonal, camera.py:59-78: the same affine model describes the ORIGINS and the
// "apex" slot carries the common direction).
static void fill_cone_model(PtKArgs &a, const pt_camera *cam, int width, int height) {
  const double *m = cam->m, dist = cam->screen_distance, asp = cam->aspect_ratio;
  for (int r = 0; r < 3; ++r) {
    a.cone_d0[r] = (float)(m[r * 4 + 0] * dist + m[r * 4 + 1] * asp + m[r * 4 + 2]);
    a.cone_dx[r] = (float)(m[r * 4 + 1] * (-2.0 * asp / width));
    a.cone_dy[r] = (float)(m[r * 4 + 2] * (-2.0 / height));
    a.cone_apex[r] = (float)(m[r * 4 + 0] * -dist + m[r * 4 + 3]);
    if (cam->kind != PT_CAMERA_PERSPECTIVE) {
      a.cone_d0[r] = (float)(-m[r * 4 + 0] + m[r * 4 + 1] * asp + m[r * 4 + 2] + m[r * 4 + 3]);
      a.cone_apex[r] = (float)m[r * 4 + 0];
    }
  }
  a.cam_kind = cam->kind;
}

// Launch a kernel of the frame on `st`; the LAST one is the render kernel proper (pt_stats.vgprs).  Timing: the
// frame's first launch carries the start event and its last one the stop event IN the dispatch itself
// (hipExtLaunchKernelGGL: the events take the kernel's own begin / end timestamps), so a timed frame costs no
// barrier packet on the stream and the interval holds the frame's kernels, not the launch gaps around them.
#ifndef PT_WAVES_POINTLIGHT
#define PT_WAVES_POINTLIGHT 3  // waves per SIMD pt_tile_kernel<POINTLIGHT> is compiled for (register budget 512 / that)
#endif
#define PT_LAUNCH(KERNEL, GRID, LDS, LAST, ...)                                                              \
  do {                                                                                                       \
    main_fn = (const void *)(KERNEL);                                                                        \
    hipExtLaunchKernelGGL((KERNEL), dim3(GRID), dim3(PT_BLOCK), LDS, st, ev_started ? nullptr : ev_a,        \
                          (LAST) ? ev_b : nullptr, 0, __VA_ARGS__);                                          \
    ev_started = true;                                                                                       \
  } while (0)

// the scene's tables as the kernels see them (launch(); the probes of ptrace_debug.h)
static void fill_scene_args(const pt_scene *s, PtKArgs &a) {
  a.recs = s->recs;
  a.aux = s->aux;
  a.hoist = s->hoist;
  a.diag = s->diag;
  a.hoist_diag = s->hoist_diag;
  a.queue = s->queue;
  a.bounds = s->bounds;
  a.bsoa = s->bsoa;
  a.bs_stride = s->bs_stride;
  for (int q = 0; q < 3; ++q) a.bs_rmax[q] = s->bs_rmax[q];
  a.gs_stride = s->gs_stride;
  a.cs_stride = s->cs_stride;
  a.bs_levels = s->bs_levels && s->n_spheres >= pt_tuning().levels_min;
  a.grid_cells = s->grid_cells;
  a.grid_occ = s->grid_occ;
  a.grid_balls = s->grid_balls;
  a.grid_slots = s->grid_slots;
  a.grid_always = s->grid_always;
  a.grid_n_always = s->grid_n_always;
  a.grid_far_eo = s->grid_far_eo;
  a.grid_occ_lds = -1;
  a.scene_lds = -1;
  for (int q = 0; q < 3; ++q) {
    a.grid_res[q] = s->grid_res[q];
    a.grid_min[q] = s->grid_min[q];
    a.grid_max[q] = s->grid_max[q];
    a.grid_cell[q] = s->grid_cell[q];
    a.grid_inv[q] = s->grid_inv[q];
  }
  a.n_diag = s->n_diag;
  a.lights = s->lights;
  a.tex = s->tex;
  a.tex_data = s->tex_data;
  a.n_shapes = s->n_shapes;
  a.n_spheres = s->n_spheres;
  a.n_lights = s->n_lights;
}

static_assert(PT_PLAN_BLOCK == PT_BLOCK && PT_PLAN_REGION == PT_REGION && PT_PLAN_CELL == PT_CELL && PT_PLAN_CELL_CHUNK == PT_CELL_CHUNK &&
                  PT_PLAN_TREE_FRAME == PT_TREE_FRAME && PT_PLAN_SCATTER_BLOCK == PT_SCATTER_BLOCK && PT_PLAN_DIAG_BYTES == sizeof(PtDiagRec) &&
                  PT_PLAN_REC_BYTES == sizeof(PtShapeRec) && PT_PLAN_AUX_BYTES == sizeof(PtShapeAux) && PT_PLAN_HANDOVER_HEADER == PT_HANDOVER_HEADER &&
                  PT_TREE_FRAME == 20,
              "pt_plan.h sizes the kernels' tiles and records by number: keep them equal to the kernels' own");

// grow-only device buffers of a handle: a launch that needs more lets the stream drain first
template <typename T>
static int ensure(T **ptr, size_t *have, size_t need, hipStream_t st) {
  if (need <= *have) return PT_OK;
  HIP_TRY(hipStreamSynchronize(st));
  if (*ptr) HIP_TRY(hipFree(*ptr));
  *ptr = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc((void **)ptr, need * sizeof(T)));
  *have = need;
  return PT_OK;
}

// launch() = plan (pt_plan.h: a pure function of the scene's facts, the camera, the parameters and the tuning table) +
// enqueue (this function: buffers, the argument block, the launches the plan names -- no decision is taken here).
static int launch(pt_scene *s, const pt_camera *cam, const pt_params *p, void *out_dev, hipStream_t st) {
  const PtTuning &tn = pt_tuning();
  PtPlan pl;
  pt_make_plan(scene_facts(s), cam, p, tn, pl);

  PtKArgs a;
  memset(&a, 0, sizeof a);
  fill_scene_args(s, a);
  a.bs_levels = pl.bs_levels;
  a.dome_shortcut = s->dome_shortcut ? 1 : 0;
  a.out = out_dev;
  a.cam_kind = cam->kind;
  memcpy(a.cam_m, cam->m, sizeof a.cam_m);
  a.cam_dist = cam->screen_distance;
  a.cam_aspect = cam->aspect_ratio;
  fill_cone_model(a, cam, p->width, p->height);
  a.W = p->width;
  a.H = p->height;
  a.S = p->samples_per_side;
  a.N = p->num_of_rays;
  a.D = p->max_depth;
  a.rr = p->rr_limit;
  a.pcg_mode = p->pcg_mode;
  a.s0 = p->path_state;
  a.q0 = p->path_seq;
  if (p->pcg_mode == PT_PCG_SEQ) {  // (OnOff / Flat / PointLight only: check_params)
    if (p->samples_per_side > 0) {  // the jitter stream of the reference's ImageTracer, entered by jump-ahead per pixel
      a.s0 = p->jitter_state;
      a.q0 = p->jitter_seq;
    } else {
      a.pcg_mode = PT_PCG_PIXEL;  // pixel-centre rays: no random number is drawn, the alignments coincide
    }
  }
  a.row_block = p->row_block > 0 ? p->row_block : 1;
  a.n_ranks = p->n_ranks > 0 ? p->n_ranks : 1;
  a.rank = p->rank;
  a.out_f32 = p->out_format == PT_OUT_F32;
  for (int k = 0; k < 3; ++k) {
    a.bg[k] = p->background[k];
    a.onoff[k] = p->onoff_color[k];
    a.ambient[k] = p->ambient[k];
  }
  // The scene's workspace (argument block copy, hoisted constants, region/cell tables, frame stack, ray
  // partials) is shared by all its launches, which are ordered by being on ONE stream.  A launch on another
  // stream than the previous one first lets that stream drain (host-side wait; a rare event).
  if (s->launched && s->last_stream != st) {
    if (hipStreamSynchronize(s->last_stream) != hipSuccess) (void)hipGetLastError();  // (a destroyed stream has nothing in flight)
  }
  s->last_stream = st;
  s->launched = true;
  s->count_pending = false;
  s->choice_pending = false;
  a.rows_local = pl.rows;
  a.npass = pl.npass;
  a.npix = pl.npix;
  s->stats.n_pixels = (uint64_t)a.npix;
  if (a.npix == 0) return PT_OK;

  a.frame_doubles = pl.frame_doubles;
  a.diag_lds = pl.diag_lds;
  a.grid_occ_lds = pl.grid_occ_lds;
  a.scene_lds = pl.scene_lds;
  a.nthreads = pl.nthreads;
  a.block_h = pl.block_h;
  s->stats.grid = pl.grid;
  s->stats.block = PT_BLOCK;
  s->stats.lds_bytes = 0;
  s->stats.kernel = pl.kernel;
  const int n_count_slots = pl.grid + pl.grid_first + pl.grid_q;

  if (s->count_rays) {
    size_t have = (size_t)s->ray_partials_n * 2;
    int rc = ensure(&s->ray_partials, &have, (size_t)n_count_slots * 2, st);
    s->ray_partials_n = (int)(have / 2);
    if (rc) return rc;
    a.ray_counter = s->ray_partials;
  }

  if (pl.hoist && !(s->hoist_valid && s->hoist_stream == st && memcmp(&s->hoist_cam, cam, sizeof(pt_camera)) == 0)) {
    V3 o = {-cam->screen_distance, 0.0, 0.0};
    // camera.py:116-124: origin (-d, 0, 0) through the camera transformation, reference order
    V3 w;
    w.x = o.x * cam->m[0] + o.y * cam->m[1] + o.z * cam->m[2] + cam->m[3];
    w.y = o.x * cam->m[4] + o.y * cam->m[5] + o.z * cam->m[6] + cam->m[7];
    w.z = o.x * cam->m[8] + o.y * cam->m[9] + o.z * cam->m[10] + cam->m[11];
    hipLaunchKernelGGL(pt_prep_hoist, dim3((s->n_shapes + 255) / 256), dim3(256), 0, st, s->recs, s->hoist,
                       s->hoist_diag, s->n_shapes, s->n_diag, w);
    s->hoist_cam = *cam;
    s->hoist_valid = true;
    s->hoist_stream = st;
  }

  if (pl.zero_frame) {
    // render.py:100-101: every primary call returns Color(0, 0, 0) without a world query
    HIP_TRY(hipEventRecord(s->ev0, st));
    HIP_TRY(hipMemsetAsync(out_dev, 0, pt_output_bytes(p), st));
    HIP_TRY(hipEventRecord(s->ev1, st));
    if (s->count_rays) {
      HIP_TRY(hipMemsetAsync(s->ray_counter, 0, 2 * sizeof(unsigned long long), st));
      HIP_TRY(hipMemcpyAsync(s->ray_counter_host, s->ray_counter, 2 * sizeof(unsigned long long),
                             hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(s->ev_count, st));
      s->count_pending = true;
    }
    return PT_OK;
  }
  if (p->renderer == PT_RENDERER_PATHTRACER) {
    // the queue block (head, unit counts, F; pt_path.h: pt_unit_scatter) starts every frame zeroed: by the path
    // kernel of the frame before (it clears the OTHER block), by a memset after anything went wrong in between
    a.qpar = s->queue_parity;
    s->queue_last = s->queue + (size_t)a.qpar * PT_QUEUE_WORDS;
    if (!s->queue_clean) HIP_TRY(hipMemsetAsync(s->queue_last, 0, PT_QUEUE_WORDS * sizeof(unsigned long long), st));
    s->queue_clean = false;  // (true again once this frame's path kernel is enqueued)
    a.p_max_path = pl.p_max_path;
    a.s_min_path = pl.s_min_path;
    size_t have = s->ws_bytes / sizeof(double);
    int rc = ensure(&s->ws, &have, (pl.ws_bytes + sizeof(double) - 1) / sizeof(double), st);
    s->ws_bytes = have * sizeof(double);
    if (rc) return rc;
    a.ws = s->ws;
  }

  // path tracer: per-region masks/keys from the first pass, work units for the second from pt_unit_scatter
  if (pl.path_tiled) {
    if (pl.nregions > s->region_cap || pl.units_need > s->units_cap) {
      HIP_TRY(hipStreamSynchronize(st));
      if (s->region_keys) HIP_TRY(hipFree(s->region_keys));
      if (s->units) HIP_TRY(hipFree(s->units));
      if (s->region_mask) HIP_TRY(hipFree(s->region_mask));
      s->region_keys = nullptr;
      s->units = nullptr;
      s->region_mask = nullptr;
      s->region_cap = 0;
      s->units_cap = 0;
      HIP_TRY(hipMalloc((void **)&s->region_keys, (size_t)pl.nregions));
      HIP_TRY(hipMalloc((void **)&s->units, (size_t)pl.units_need * sizeof(int4)));
      HIP_TRY(hipMalloc((void **)&s->region_mask, (size_t)pl.nregions * sizeof(unsigned long long)));
      s->region_cap = pl.nregions;
      s->units_cap = pl.units_need;
    }
    a.units = s->units;
    a.region_keys = s->region_keys;
    a.region_mask = s->region_mask;
    a.spec_draws = pl.spec_draws;
    a.tree_jump_lds = pl.tree_jump_lds;
    if (pl.q_alt) {
      int rc = ensure(&s->handover, &s->handover_doubles, pl.handover_doubles, st);
      if (rc) return rc;
      rc = ensure(&s->units_handed, &s->units_handed_n, (size_t)pl.handover_cap, st);
      if (rc) return rc;
    }
    a.handover = s->handover;
    a.units_handed = s->units_handed;
    a.handover_cap = pl.q_alt ? pl.handover_cap : 0;
    s->last_handover_cap = a.handover_cap;
    a.q_budget = 0;  // (the one-queue kernel's block carries these)
    a.q_tail_budget = 0;
    a.q_few_lanes = 0;
    a.dbg_trace_unit = (int)tn.trace_unit;
    // The sphere the camera is deepest inside (object-space |o'|^2 - 1 most negative, and below -0.5): the
    // first pass settles, per pixel, what can only hit that sphere (pt_tile_kernel re-checks every condition
    // from the exact hoisted constants; this only names the candidate).
    a.dome_slot = -1;
    if (!pl.ortho && tn.pixel_dome) {
      const double ox = -cam->screen_distance * cam->m[0] + cam->m[3], oy = -cam->screen_distance * cam->m[4] + cam->m[7],
                   oz = -cam->screen_distance * cam->m[8] + cam->m[11];
      double best = -0.5;
      for (const auto &dc : s->dome_cands) {
        const double *m = dc.invm;
        const double px = ox * m[0] + oy * m[1] + oz * m[2] + m[3], py = ox * m[4] + oy * m[5] + oz * m[6] + m[7],
                     pz = ox * m[8] + oy * m[9] + oz * m[10] + m[11];
        const double c = px * px + py * py + pz * pz - 1.0;
        if (c < best) {
          best = c;
          a.dome_slot = dc.slot;
        }
      }
    }
  }
  // large scenes: two-level culling (cells of PT_CELL x PT_CELL global pixels, then 8x8 tiles)
  if (pl.hier) {
    const size_t need = (size_t)pl.ncells * pl.cell_stride;
    if (need > s->cell_list_cap || pl.ncells > s->cell_count_cap) {
      HIP_TRY(hipStreamSynchronize(st));
      if (s->cell_list) HIP_TRY(hipFree(s->cell_list));
      if (s->cell_count) HIP_TRY(hipFree(s->cell_count));
      s->cell_list = nullptr;
      s->cell_count = nullptr;
      s->cell_list_cap = 0;
      s->cell_count_cap = 0;
      HIP_TRY(hipMalloc((void **)&s->cell_list, need * sizeof(unsigned int)));
      HIP_TRY(hipMalloc((void **)&s->cell_count, (size_t)pl.ncells * sizeof(int)));
      s->cell_list_cap = need;
      s->cell_count_cap = pl.ncells;
    }
    a.cell_list = s->cell_list;
    a.cell_count = s->cell_count;
    a.cells_x = pl.cells_x;
    a.cell_stride = pl.cell_stride;
  }
  // the cold half of the argument block is read from device memory: refresh the copy when it changed
  // (the output pointer stays a by-value argument: double-buffered frames alternate it every launch)
  a.cold = s->args_dev;
  PtKArgs cold = a;
  cold.out = nullptr;
  cold.qpar = 0;
  if (!(s->args_valid && s->args_stream == st && memcmp(&s->args_last, &cold, sizeof cold) == 0)) {
    // pageable source: the runtime stages the bytes before returning, so `cold` may go out of scope
    HIP_TRY(hipMemcpyAsync(s->args_dev, &cold, sizeof cold, hipMemcpyHostToDevice, st));
    s->args_last = cold;
    s->args_valid = true;
    s->args_stream = st;
  }
  const void *main_fn = nullptr;  // the render kernel proper (the last one launched), for pt_stats.vgprs
  const bool prof = s->timing && s->profiling && (size_t)(2 * s->prof_used + 1) < s->prof.size();
  const hipEvent_t ev_a = s->timing ? (prof ? s->prof[2 * s->prof_used] : s->ev0) : nullptr;
  const hipEvent_t ev_b = s->timing ? (prof ? s->prof[2 * s->prof_used + 1] : s->ev1) : nullptr;
  bool ev_started = false;
  const int R = p->renderer;
#ifdef PT_DEBUG_TIME
  if (pl.tile4 || ((pl.tile || pl.path_tiled) && R != PT_RENDERER_PATHTRACER)) {
    s->queue_last = s->queue;  // (a.qpar = 0: the section sums land in block 0)
    HIP_TRY(hipMemsetAsync(s->queue, 0, 16 * sizeof(unsigned long long), st));
    if (s->queue_parity == 0) s->queue_clean = false;
  }
#endif
  if (pl.tile4) {
    // small worlds: the shapes' records ride in LDS for shading (Flat; OnOff reads none of them)
    s->stats.lds_bytes = (int)pl.lds_main;
    const dim3 grid4(pl.grid4_x, pl.grid4_y, 1);
#define PT_LAUNCH4(R_, L_)                                                                                                  \
  do {                                                                                                                      \
    main_fn = (const void *)pt_tile4_kernel<R_, L_>;                                                                        \
    hipExtLaunchKernelGGL((pt_tile4_kernel<R_, L_>), grid4, dim3(PT_BLOCK), pl.lds_main, st, ev_a, ev_b, 0, a);             \
  } while (0)
    if (R == PT_RENDERER_ONOFF)
      PT_LAUNCH4(PT_RENDERER_ONOFF, false);
    else if (pl.t4lds)
      PT_LAUNCH4(PT_RENDERER_FLAT, true);
    else
      PT_LAUNCH4(PT_RENDERER_FLAT, false);
#undef PT_LAUNCH4
  } else if (pl.tile || pl.path_tiled) {
    const size_t lds = pl.lds_tile;
    s->stats.lds_bytes = (int)lds;
    const int tgrid = pl.path_tiled ? pl.grid_first : pl.grid;
    const bool last = !pl.path_tiled;
    const int pgrid = pl.path_tiled ? pl.grid : 0;  // (the first pass is told the second pass's grid)
    if (pl.hier) {
      if (pl.cell_chunk_len > PT_CELL_CHUNK) return fail(PT_ERR_INVALID, "internal: cell chunk exceeds its LDS staging");
      HIP_TRY(hipMemsetAsync(s->cell_count, 0, (size_t)pl.ncells * sizeof(int), st));
      PT_LAUNCH(pt_cell_kernel, pl.cell_groups * pl.cell_chunks, 0, false, a, pl.cell_chunks, pl.cell_chunk_len);
    }
    // pt_tile_kernel<renderer, waves, HIER, ORTHO, BLOCKS>
#define PT_TILE_MODES(R_, W_)                                                                              \
  do {                                                                                                     \
    if (pl.tile_mode == PT_TILE_HIER)                                                                      \
      PT_LAUNCH((pt_tile_kernel<R_, W_, true>), tgrid, lds, last, a, pgrid);                               \
    else if (pl.tile_mode == PT_TILE_ORTHO)                                                                \
      PT_LAUNCH((pt_tile_kernel<R_, W_, false, true>), tgrid, lds, last, a, pgrid);                        \
    else                                                                                                   \
      PT_LAUNCH((pt_tile_kernel<R_, W_, false>), tgrid, lds, last, a, pgrid);                              \
  } while (0)
    if (R == PT_RENDERER_ONOFF)
      PT_TILE_MODES(PT_RENDERER_ONOFF, 4);
    else if (R == PT_RENDERER_FLAT)
      PT_TILE_MODES(PT_RENDERER_FLAT, 4);
    else if (R == PT_RENDERER_POINTLIGHT)
      PT_TILE_MODES(PT_RENDERER_POINTLIGHT, PT_WAVES_POINTLIGHT);
    else if (pl.tile_mode == PT_TILE_BLOCKS)
      PT_LAUNCH((pt_tile_kernel<PT_RENDERER_PATHTRACER, 4, false, false, true>), tgrid, lds, last, a, pgrid);
    else
      PT_TILE_MODES(PT_RENDERER_PATHTRACER, 4);
#undef PT_TILE_MODES
    if (pl.path_tiled) {
      // second pass: the pixels the first one flagged, fullest regions first
      if (pl.tree)  // one pixel per unit: "64 lanes per pixel, whatever the number of flagged pixels"
        hipLaunchKernelGGL(pt_unit_scatter, dim3(pl.grid_scatter), dim3(PT_SCATTER_BLOCK), 0, st, s->region_keys, s->region_mask, pl.nregions,
                           s->units, s->units_cap, s->queue_last, (long long)1 << 60, 64, 1, pl.q_min, pl.q_budget_per_flagged, pl.q_budget_min);
      else
        hipLaunchKernelGGL(pt_unit_scatter, dim3(pl.grid_scatter), dim3(PT_SCATTER_BLOCK), 0, st, s->region_keys, s->region_mask, pl.nregions,
                           s->units, s->units_cap, s->queue_last, pl.lanes_cap, pl.nsamp, (long long)pl.min_rounds);
      if (pl.q_alt) {
        // num_of_rays > 1: the one-queue kernel IN FRONT of the tree kernel, with an argument block of its own (a lane per
        // pixel: 20 doubles per depth and lane, its own grid, its own slots for the ray counts); it returns at once unless
        // PT_Q_CHOICE says 1, and then leaves the pixels over its budget to the tree kernel behind it (PT_Q_HEAVY)
        PtKArgs aq = a;
        aq.cold = s->args_dev2;
        aq.nthreads = pl.grid_q * PT_BLOCK;
        aq.frame_doubles = 20;
        aq.p_max_path = pl.q_p_max_path;
        aq.s_min_path = pl.q_s_min_path;
        aq.count_base = pl.grid + pl.grid_first;
        aq.q_budget = pl.q_budget;
        aq.q_tail_budget = pl.q_tail_budget;
        aq.q_few_lanes = pl.q_few_lanes;
        aq.scene_lds = -1;
        aq.grid_occ_lds = -1;
        aq.diag_lds = pl.q_diag_lds;
        aq.ws = s->ws;
        PtKArgs cold2 = aq;
        cold2.out = nullptr;
        cold2.qpar = 0;
        if (!(s->args2_valid && s->args2_stream == st && memcmp(&s->args2_last, &cold2, sizeof cold2) == 0)) {
          HIP_TRY(hipMemcpyAsync(s->args_dev2, &cold2, sizeof cold2, hipMemcpyHostToDevice, st));
          s->args2_last = cold2;
          s->args2_valid = true;
          s->args2_stream = st;
        }
#define PT_ALT(K_)                                              \
  do {                                                          \
    HIP_TRY(path_lds_limit((const void *)(K_), pl.lds_q));      \
    PT_LAUNCH((K_), pl.grid_q, pl.lds_q, false, aq);            \
  } while (0)
        switch (pl.alt) {
          case PT_ALT_FLAGGED_LEAN_LDS: PT_ALT((pt_path_flagged_kernel<1, 1>)); break;
          case PT_ALT_FLAGGED_LEAN_SPLIT: PT_ALT((pt_path_flagged_kernel<1, 2>)); break;
          case PT_ALT_FLAGGED_SPLIT: PT_ALT((pt_path_flagged_kernel<0, 2>)); break;
          default: PT_ALT((pt_path_flagged_kernel<0, 1>)); break;
        }
#undef PT_ALT
      }
      const bool last2 = true;
#define PT_SECOND(K_)                                               \
  do {                                                              \
    HIP_TRY(path_lds_limit((const void *)(K_), pl.lds_main));       \
    PT_LAUNCH((K_), pl.grid, pl.lds_main, last2, a);                \
  } while (0)
      switch (pl.second) {
        case PT_SECOND_TREE_LEAN: PT_SECOND((pt_path_tree_kernel<true>)); break;
        case PT_SECOND_TREE_LEAN_SCENE: PT_SECOND((pt_path_tree_kernel<true, true>)); break;
        case PT_SECOND_TREE: PT_SECOND((pt_path_tree_kernel<false>)); break;
        case PT_SECOND_REGIONS_LDS_SCENE_LEAN: PT_SECOND((pt_path_regions_kernel<true, true, 1>)); break;
        case PT_SECOND_REGIONS_LDS_SCENE: PT_SECOND((pt_path_regions_kernel<true, true>)); break;
        case PT_SECOND_REGIONS_LDS_NOGRID: PT_SECOND((pt_path_regions_kernel<true, false, 2>)); break;
        case PT_SECOND_REGIONS_LDS: PT_SECOND((pt_path_regions_kernel<true>)); break;
        default: PT_SECOND((pt_path_regions_kernel<false>)); break;
      }
#undef PT_SECOND
      // (pt_stats.vgprs: the tree kernel's; pt_stats.kernel follows the device's choice, see fold_stats -- only when the frame
      // is measured at all: a frame loop with timing and counting off enqueues neither the copy nor its event)
      if (pl.q_alt && (s->timing || s->count_rays || prof)) {
        HIP_TRY(hipMemcpyAsync(s->ray_counter_host + 2, s->queue_last + PT_Q_CHOICE, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        s->choice_pending = true;
      }
    }
  } else if (R == PT_RENDERER_PATHTRACER) {
    PT_LAUNCH(pt_path_kernel, pl.grid, 0, true, a);
  } else {
#define PT_SIMPLE(R_)                                                              \
  do {                                                                             \
    if (pl.simple_hoist)                                                           \
      PT_LAUNCH((pt_simple_kernel<R_, true>), pl.grid, 0, true, a);                \
    else                                                                           \
      PT_LAUNCH((pt_simple_kernel<R_, false>), pl.grid, 0, true, a);               \
  } while (0)
    if (R == PT_RENDERER_ONOFF)
      PT_SIMPLE(PT_RENDERER_ONOFF);
    else if (R == PT_RENDERER_FLAT)
      PT_SIMPLE(PT_RENDERER_FLAT);
    else
      PT_SIMPLE(PT_RENDERER_POINTLIGHT);
#undef PT_SIMPLE
  }
  HIP_TRY(hipGetLastError());
  if (p->renderer == PT_RENDERER_PATHTRACER) {  // its path kernel is enqueued: the other queue block will be zero
    s->queue_parity ^= 1;
    s->queue_clean = true;
  }
  if (main_fn) {
    // registers per lane of that kernel (hipFuncGetAttributes; looked up once per kernel)
    static std::vector<std::pair<const void *, int>> known;
    int regs = -1;
    for (const auto &kv : known)
      if (kv.first == main_fn) regs = kv.second;
    if (regs < 0) {
      hipFuncAttributes fa;
      regs = hipFuncGetAttributes(&fa, main_fn) == hipSuccess ? fa.numRegs : 0;
      (void)hipGetLastError();
      known.push_back({main_fn, regs});
    }
    s->stats.vgprs = regs;
  }
  if (prof) {
    s->prof_used++;
    s->stats_valid = false;
  } else if (s->timing) {
    s->stats_valid = true;
  } else {
    s->stats_valid = false;
  }
  if (s->count_rays) {
    hipLaunchKernelGGL(pt_sum_counts, dim3(1), dim3(256), 0, st, s->ray_partials, n_count_slots, s->ray_counter);
    HIP_TRY(hipMemcpyAsync(s->ray_counter_host, s->ray_counter, 2 * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, st));
    // ev1 was recorded BEFORE the count left the device: whoever reads ray_counter_host waits for this one
    HIP_TRY(hipEventRecord(s->ev_count, st));
    s->count_pending = true;
  } else if (s->choice_pending) {
    HIP_TRY(hipEventRecord(s->ev_count, st));
  }
  return PT_OK;
}

static int fold_stats(pt_scene *s) {
  if (!s->pending) return PT_OK;
  float ms = 0.f;
  if (s->stats_valid) HIP_TRY(hipEventSynchronize(s->pending_copy ? s->ev2 : s->ev1));
  if (!s->stats_valid) {
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    s->stats.kernel_ms = s->stats.total_ms = 0.0;
  } else if (s->stats.n_pixels > 0) {
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->stats.kernel_ms = ms;
    if (s->pending_copy) {
      HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev2));
    }
    s->stats.total_ms = ms;
  } else {
    s->stats.kernel_ms = s->stats.total_ms = 0.0;
  }
  if (s->count_pending || s->choice_pending) HIP_TRY(hipEventSynchronize(s->ev_count));
  if (s->count_pending) {
    s->count_pending = false;
    s->stats.n_rays = s->ray_counter_host[0];
    s->stats.n_rays_resolved = s->ray_counter_host[1];
  } else {
    s->stats.n_rays = s->stats.n_rays_resolved = 0;  // (no pixels, or counting off)
  }
  if (s->choice_pending) {  // which of the two enqueued second-pass kernels the device let work (PT_Q_CHOICE)
    s->choice_pending = false;
    s->stats.kernel = s->ray_counter_host[2] ? PT_KERNEL_PATH : PT_KERNEL_PATH_TREE;
  }
  s->pending = false;
  s->pending_copy = false;
  return PT_OK;
}

extern "C" int pt_render_device(pt_scene *s, const pt_camera *cam, const pt_params *p, void *out_dev,
                                size_t out_bytes, void *stream) {
  int rc = check_params(s, cam, p);
  if (rc) return rc;
  const size_t need = pt_output_bytes(p);
  if (out_bytes < need) return fail(PT_ERR_SIZE, "output buffer too small: %zu < %zu bytes", out_bytes, need);
  if (need > 0 && !out_dev) return fail(PT_ERR_INVALID, "null output buffer");
  HIP_TRY(hipSetDevice(s->device));
  // an earlier asynchronous render that was never synchronised simply loses its statistics:
  // folding them here would block the host on the device every frame
  s->pending = false;
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  rc = launch(s, cam, p, out_dev, st);
  if (rc) return rc;
  if (s->stats.n_pixels > 0) {
    s->pending = true;
    s->pending_copy = false;
  }
  if (!stream) return pt_sync(s);
  return PT_OK;
}

extern "C" int pt_sync(pt_scene *s) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  return fold_stats(s);
}

extern "C" int pt_render(pt_scene *s, const pt_camera *cam, const pt_params *p, void *out_host,
                         size_t out_bytes) {
  int rc = check_params(s, cam, p);
  if (rc) return rc;
  const size_t need = pt_output_bytes(p);
  if (out_bytes < need) return fail(PT_ERR_SIZE, "output buffer too small: %zu < %zu bytes", out_bytes, need);
  if (need > 0 && !out_host) return fail(PT_ERR_INVALID, "null output buffer");
  HIP_TRY(hipSetDevice(s->device));
  if (s->pending) {
    rc = fold_stats(s);
    if (rc) return rc;
  }
  if (need > s->out_dev_bytes) {
    if (s->out_dev) HIP_TRY(hipFree(s->out_dev));
    s->out_dev = nullptr;
    s->out_dev_bytes = 0;
    HIP_TRY(hipMalloc(&s->out_dev, need));
    s->out_dev_bytes = need;
  }
  rc = launch(s, cam, p, s->out_dev, s->stream);
  if (rc) return rc;
  if (need > 0) {
    HIP_TRY(hipMemcpyAsync(out_host, s->out_dev, need, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(s->ev2, s->stream));
    s->pending = true;
    s->pending_copy = true;
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return fold_stats(s);
}

// Page-locked host memory for pt_render's output: the D2H copy then is one DMA at link speed into the
// caller's buffer (a pageable destination goes through the runtime's staging path at roughly 2/3 of it).
extern "C" int pt_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(PT_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (bytes == 0) return PT_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device visible");
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes,
                hipGetErrorString(e));
  }
  return PT_OK;
}

extern "C" int pt_host_free(void *p) {
  if (!p) return PT_OK;
  HIP_TRY(hipHostFree(p));
  return PT_OK;
}

// ---- device memory and streams for callers WITHOUT a GPU framework of their own (ABI 1.5) ----
// The `render` command leaves its frame in HBM (pt_render_device) and post-processes it there (pt_image_*): with these it needs
// no torch for the buffer.  Thin and error-coded; a caller that owns device tensors keeps passing their pointers.
static int device_ok(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
  return PT_OK;
}

extern "C" int pt_device_alloc(int device, size_t bytes, void **out) {
  if (!out) return fail(PT_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (bytes == 0) return PT_OK;
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return PT_OK;
}

extern "C" int pt_device_free(int device, void *p) {
  if (!p) return PT_OK;
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(p));  // (waits for the device's work on it)
  return PT_OK;
}

extern "C" int pt_device_download(int device, void *dst_host, const void *src_dev, size_t bytes, void *stream) {
  if (bytes == 0) return PT_OK;
  if (!dst_host || !src_dev) return fail(PT_ERR_INVALID, "null argument");
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

extern "C" int pt_stream_create(int device, void **out) {
  if (!out) return fail(PT_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));  // (the legacy default stream serialises the host with the device)
  *out = (void *)st;
  return PT_OK;
}

extern "C" int pt_stream_sync(int device, void *stream) {
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return PT_OK;
}

extern "C" int pt_stream_destroy(int device, void *stream) {
  if (!stream) return PT_OK;
  int rc = device_ok(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return PT_OK;
}

extern "C" int pt_get_stats(pt_scene *s, pt_stats *out) {
  if (!s || !out) return fail(PT_ERR_INVALID, "null argument");
  if (s->pending) {
    int rc = pt_sync(s);
    if (rc) return rc;
  }
  *out = s->stats;
  return PT_OK;
}

extern "C" int pt_set_timing(pt_scene *s, int enable) {
  if (!s) return fail(PT_ERR_INVALID, "null scene");
  s->timing = enable != 0;
  return PT_OK;
}

extern "C" int pt_profile_begin(pt_scene *s, int capacity) {
  if (!s || capacity <= 0 || capacity > (1 << 20)) return fail(PT_ERR_INVALID, "bad profiling capacity");
  HIP_TRY(hipSetDevice(s->device));
  while ((int)s->prof.size() < 2 * capacity) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    s->prof.push_back(e);
  }
  s->prof_used = 0;
  s->profiling = true;
  return PT_OK;
}

extern "C" int pt_profile_end(pt_scene *s, double *total_kernel_ms, int *launches) {
  if (!s || !total_kernel_ms || !launches) return fail(PT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  double sum = 0.0;
  for (int k = 0; k < s->prof_used; ++k) {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(s->prof[2 * k + 1]));
    HIP_TRY(hipEventElapsedTime(&ms, s->prof[2 * k], s->prof[2 * k + 1]));
    sum += ms;
  }
  *total_kernel_ms = sum;
  *launches = s->prof_used;
  s->profiling = false;
  s->prof_used = 0;
  return PT_OK;
}

// ---- HdrImage post-processing (pt_post.h) -------------------------------------------------------------
static int post_check(int device, const void *img, int fmt, int w, int h) {
  if (!img) return fail(PT_ERR_INVALID, "null image");
  if (w <= 0 || h <= 0) return fail(PT_ERR_INVALID, "bad image size %dx%d", w, h);
  if (fmt != PT_OUT_F64 && fmt != PT_OUT_F32) return fail(PT_ERR_INVALID, "unknown pixel format %d", fmt);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  return PT_OK;
}

static int post_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096)); }

static bool is_device_ptr(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Image and output buffers may be device memory (a frame left in HBM by pt_render_device) or plain host
// memory (an HdrImage's array); host buffers are staged through a temporary device copy.
struct Staged {
  void *dev = nullptr;
  void *host = nullptr;
  size_t bytes = 0;
  bool owned = false;
  ~Staged() {
    if (owned && dev) (void)hipFree(dev);
  }
  int in(const void *p, size_t n, bool copy_in, hipStream_t st) {
    bytes = n;
    if (is_device_ptr(p)) {
      dev = const_cast<void *>(p);
      return PT_OK;
    }
    host = const_cast<void *>(p);
    owned = true;
    HIP_TRY(hipMalloc(&dev, n));
    if (copy_in) HIP_TRY(hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, st));
    return PT_OK;
  }
  int out(hipStream_t st) {
    if (owned) {
      HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    return PT_OK;
  }
};

extern "C" int pt_image_pack_pfm(int device, const void *img, int fmt, int width, int height, int big_endian,
                                 void *out, void *stream) {
  int rc = post_check(device, img, fmt, width, height);
  if (rc) return rc;
  if (!out) return fail(PT_ERR_INVALID, "null output");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)width * height * 3;
  Staged si, so;
  if ((rc = si.in(img, (size_t)n * (fmt == PT_OUT_F32 ? 4 : 8), true, st))) return rc;
  if ((rc = so.in(out, (size_t)n * 4, false, st))) return rc;
  hipLaunchKernelGGL(pt_post_pfm_kernel, dim3(post_grid(n)), dim3(256), 0, st, si.dev, fmt == PT_OUT_F32 ? 1 : 0,
                     width, height, big_endian ? 1 : 0, (uint32_t *)so.dev);
  HIP_TRY(hipGetLastError());
  if ((rc = so.out(st))) return rc;
  if (!stream || si.owned) HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

// ---- sparse shards for the multi-GPU gather (pt_post.h) ----
extern "C" long long pt_image_sparse_fixed_bytes(long long n_pixels, int fmt) {
  if (n_pixels <= 0 || (fmt != PT_OUT_F64 && fmt != PT_OUT_F32)) return -1;
  const long long nt = (n_pixels + PT_SPARSE_RUN - 1) / PT_SPARSE_RUN;
  return 8 + (nt + 1) / 2 * 8 + nt * 3 * (fmt == PT_OUT_F32 ? 4 : 8);
}

static int sparse_check(int device, const void *a, const void *b, long long n_pixels, int fmt) {
  if (!a || !b) return fail(PT_ERR_INVALID, "null buffer");
  if (n_pixels <= 0 || n_pixels > (1LL << 37)) return fail(PT_ERR_INVALID, "bad pixel count %lld", n_pixels);
  if (fmt != PT_OUT_F64 && fmt != PT_OUT_F32) return fail(PT_ERR_INVALID, "unknown pixel format %d", fmt);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  return PT_OK;
}

extern "C" int pt_image_sparse_encode(int device, const void *shard_dev, long long n_pixels, int fmt, void *fixed_dev,
                                      void *payload_dev, void *stream) {
  int rc = sparse_check(device, shard_dev, fixed_dev, n_pixels, fmt);
  if (rc) return rc;
  if (!payload_dev) return fail(PT_ERR_INVALID, "null buffer");
  hipStream_t st = (hipStream_t)stream;
  const long long nt = (n_pixels + PT_SPARSE_RUN - 1) / PT_SPARSE_RUN, ntp = (nt + 1) / 2 * 2;
  int *place = (int *)((unsigned char *)fixed_dev + 8);
  void *firsts = (unsigned char *)fixed_dev + 8 + ntp * 4;
  const unsigned grid = (unsigned)((nt + 3) / 4);
  if (fmt == PT_OUT_F32)
    hipLaunchKernelGGL(pt_sparse_classify_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint32_t *)shard_dev, n_pixels, nt, place, (uint32_t *)firsts);
  else
    hipLaunchKernelGGL(pt_sparse_classify_kernel<uint64_t>, dim3(grid), dim3(256), 0, st, (const uint64_t *)shard_dev, n_pixels, nt, place, (uint64_t *)firsts);
  hipLaunchKernelGGL(pt_sparse_scan_kernel, dim3(1), dim3(1024), 0, st, place, nt, ntp, (long long *)fixed_dev);
  if (fmt == PT_OUT_F32)
    hipLaunchKernelGGL(pt_sparse_pack_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint32_t *)shard_dev, n_pixels, nt, (const int *)place, (uint32_t *)payload_dev);
  else
    hipLaunchKernelGGL(pt_sparse_pack_kernel<uint64_t>, dim3(grid), dim3(256), 0, st, (const uint64_t *)shard_dev, n_pixels, nt, (const int *)place, (uint64_t *)payload_dev);
  HIP_TRY(hipGetLastError());
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

extern "C" int pt_image_sparse_decode(int device, const void *fixed_dev, const void *payload_dev, long long n_pixels, int fmt,
                                      void *out_dev, int width, int row_block, int n_ranks, int rank, void *stream) {
  int rc = sparse_check(device, fixed_dev, out_dev, n_pixels, fmt);
  if (rc) return rc;
  if (n_ranks > 1 && (width <= 0 || row_block <= 0 || rank < 0 || rank >= n_ranks || n_pixels % width != 0 || n_pixels >= (1LL << 31)))
    return fail(PT_ERR_INVALID, "bad placement: width %d, row_block %d, rank %d of %d", width, row_block, rank, n_ranks);
  hipStream_t st = (hipStream_t)stream;
  const long long nt = (n_pixels + PT_SPARSE_RUN - 1) / PT_SPARSE_RUN, ntp = (nt + 1) / 2 * 2;
  const int *place = (const int *)((const unsigned char *)fixed_dev + 8);
  const void *firsts = (const unsigned char *)fixed_dev + 8 + ntp * 4;
  const unsigned grid = (unsigned)((nt + 3) / 4);
  // (payload_dev may be null when no run needs it: then no run reads it)
  if (fmt == PT_OUT_F32)
    hipLaunchKernelGGL(pt_sparse_unpack_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, place, (const uint32_t *)firsts, (const uint32_t *)payload_dev,
                       n_pixels, nt, (uint32_t *)out_dev, width, row_block, n_ranks, rank);
  else
    hipLaunchKernelGGL(pt_sparse_unpack_kernel<uint64_t>, dim3(grid), dim3(256), 0, st, place, (const uint64_t *)firsts, (const uint64_t *)payload_dev,
                       n_pixels, nt, (uint64_t *)out_dev, width, row_block, n_ranks, rank);
  HIP_TRY(hipGetLastError());
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

extern "C" int pt_image_sparse_decode_many(int device, int n_shards, const void *const *fixed_dev, const void *const *payload_dev,
                                           const long long *n_pixels, const int *ranks, int fmt, void *frame_dev, int width,
                                           int row_block, int n_ranks, void *stream) {
  if (n_shards <= 0 || n_shards > PT_SPARSE_MANY || !fixed_dev || !payload_dev || !n_pixels || !ranks)
    return fail(PT_ERR_INVALID, "bad shard list (%d shards, at most %d)", n_shards, PT_SPARSE_MANY);
  if (n_ranks <= 1 || width <= 0 || row_block <= 0) return fail(PT_ERR_INVALID, "bad placement: width %d, row_block %d, %d ranks", width, row_block, n_ranks);
  PtSparseMany m;
  memset(&m, 0, sizeof m);
  long long most = 0;
  for (int k = 0; k < n_shards; ++k) {
    int rc = sparse_check(device, fixed_dev[k], frame_dev, n_pixels[k], fmt);
    if (rc) return rc;
    if (ranks[k] < 0 || ranks[k] >= n_ranks || n_pixels[k] % width != 0 || n_pixels[k] >= (1LL << 31))
      return fail(PT_ERR_INVALID, "bad shard %d: rank %d of %d, %lld pixels", k, ranks[k], n_ranks, n_pixels[k]);
    m.fixed[k] = fixed_dev[k];
    m.payload[k] = payload_dev[k];
    m.npix[k] = n_pixels[k];
    m.rank[k] = ranks[k];
    most = std::max(most, n_pixels[k]);
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)(((most + PT_SPARSE_RUN - 1) / PT_SPARSE_RUN + 3) / 4);
  if (fmt == PT_OUT_F32)
    hipLaunchKernelGGL(pt_sparse_unpack_many_kernel<uint32_t>, dim3(gx, n_shards), dim3(256), 0, st, m, (uint32_t *)frame_dev, width, row_block, n_ranks);
  else
    hipLaunchKernelGGL(pt_sparse_unpack_many_kernel<uint64_t>, dim3(gx, n_shards), dim3(256), 0, st, m, (uint64_t *)frame_dev, width, row_block, n_ranks);
  HIP_TRY(hipGetLastError());
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

extern "C" int pt_image_average_luminosity(int device, const void *img, int fmt, int width, int height,
                                           double delta, double *out, void *stream) {
  int rc = post_check(device, img, fmt, width, height);
  if (rc) return rc;
  if (!out) return fail(PT_ERR_INVALID, "null output");
  hipStream_t st = (hipStream_t)stream;
  const long long npix = (long long)width * height;
  Staged si;
  if ((rc = si.in(img, (size_t)npix * 3 * (fmt == PT_OUT_F32 ? 4 : 8), true, st))) return rc;
  const int nblocks = (int)((npix + PT_POST_CHUNK - 1) / PT_POST_CHUNK);
  double *partials = nullptr;
  HIP_TRY(hipMalloc((void **)&partials, (size_t)(nblocks + 1) * sizeof(double)));
  hipLaunchKernelGGL(pt_post_loglum_kernel, dim3(nblocks), dim3(256), 0, st, si.dev, fmt == PT_OUT_F32 ? 1 : 0, npix,
                     delta, partials);
  hipLaunchKernelGGL(pt_post_sum_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, partials + nblocks);
  double sum = 0.0;
  hipError_t e = hipMemcpyAsync(&sum, partials + nblocks, sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(partials);
  if (e != hipSuccess) return fail(PT_ERR_HIP, "average_luminosity failed: %s", hipGetErrorString(e));
  *out = std::pow(10.0, sum / (double)npix);  // hdrimages.py:128
  return PT_OK;
}

extern "C" int pt_image_tonemap(int device, void *img, int fmt, int width, int height, double scale, int clamp,
                                double gamma, unsigned char *rgb8, int write_back, void *stream) {
  int rc = post_check(device, img, fmt, width, height);
  if (rc) return rc;
  if (rgb8 && !(gamma > 0.0)) return fail(PT_ERR_INVALID, "gamma must be positive");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)width * height * 3;
  Staged si, so;
  if ((rc = si.in(img, (size_t)n * (fmt == PT_OUT_F32 ? 4 : 8), true, st))) return rc;
  if (rgb8 && (rc = so.in(rgb8, (size_t)n, false, st))) return rc;
  hipLaunchKernelGGL(pt_post_tonemap_kernel, dim3(post_grid(n)), dim3(256), 0, st, si.dev, fmt == PT_OUT_F32 ? 1 : 0, n,
                     scale, clamp ? 1 : 0, rgb8 ? 1.0 / gamma : 1.0, rgb8 ? (unsigned char *)so.dev : nullptr,
                     write_back ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (write_back && (rc = si.out(st))) return rc;
  if (rgb8 && (rc = so.out(st))) return rc;
  if (!stream || si.owned || so.owned) HIP_TRY(hipStreamSynchronize(st));
  return PT_OK;
}

static void plan_info(const PtPlan &pl, const PtSceneFacts &f, const pt_params *p, pt_plan_info *out) {
  memset(out, 0, sizeof *out);
  out->kernel = pl.kernel;
  out->rows = pl.rows;
  out->npix = pl.npix;
  pt_plan_kernel_name(pl, p->renderer, 0, out->pre_kernel, sizeof out->pre_kernel);
  pt_plan_kernel_name(pl, p->renderer, 1, out->first_kernel, sizeof out->first_kernel);
  pt_plan_kernel_name(pl, p->renderer, 2, out->main_kernel, sizeof out->main_kernel);
  pt_plan_kernel_name(pl, p->renderer, 3, out->alt_kernel, sizeof out->alt_kernel);
  out->grid = pl.grid;
  out->grid_first = pl.grid_first;
  out->grid_alt = pl.grid_q;
  out->grid4_x = (int)pl.grid4_x;
  out->grid4_y = (int)pl.grid4_y;
  out->npx = 4;  // (pt_tile4_kernel: four pixels per lane; the two-pixel variant is gone)
  out->lds_first = pl.path_tiled ? (long long)pl.lds_tile : 0;
  out->lds_main = (long long)pl.lds_main;
  out->lds_alt = (long long)pl.lds_q;
  const bool path = p->renderer == PT_RENDERER_PATHTRACER && !pl.zero_frame && pl.npix > 0;
  out->frame_stack_home = !path ? 0 : (pl.lds_frames ? 1 : 2);
  out->alt_frame_stack_home = !pl.q_alt ? 0 : (pl.q_home == 1 ? 1 : (pl.q_home == 2 ? 3 : 2));
  out->frame_doubles = path ? pl.frame_doubles : 0;
  out->workspace_bytes = (long long)pl.ws_bytes;
  out->q_min_flagged = pl.q_min;
  out->wg_per_cu = pl.wg_per_cu;
  out->block_h = pl.block_h;
  out->hier = pl.hier;
  out->ortho = pl.ortho;
  out->hoist = pl.hoist;
  out->tile4_lds = pl.t4lds;
  out->n_spheres = f.n_spheres;
  out->n_diag = f.n_diag;
  out->has_grid = f.has_grid;
  out->ball_levels = pl.bs_levels;
  out->units_need = pl.units_need;
  out->alt_budget = pl.q_alt ? pl.q_budget : 0;
  out->nregions = pl.nregions;
  out->min_rounds = pl.min_rounds;
  out->spec_draws = pl.spec_draws;
}

extern "C" int pt_debug_plan(const pt_scene_desc *desc, const pt_camera *cam, const pt_params *p, int n_cu, int dome_shortcut,
                             pt_plan_info *out) {
  if (!out) return fail(PT_ERR_INVALID, "null argument");
  int rc = check_desc(desc);
  if (rc) return rc;
  pt_scene tmp;  // (a bag of facts here: no HIP call touches it)
  HostTables h;
  analyse_scene(desc, pt_tuning(), &tmp, h);
  tmp.n_cu = n_cu > 0 ? n_cu : 256;
  tmp.dome_shortcut = dome_shortcut != 0;
  rc = check_params(&tmp, cam, p);
  if (rc) return rc;
  const PtSceneFacts f = scene_facts(&tmp);
  PtPlan pl;
  pt_make_plan(f, cam, p, pt_tuning(), pl);
  plan_info(pl, f, p, out);
  return PT_OK;
}

extern "C" int pt_debug_plan_scene(pt_scene *s, const pt_camera *cam, const pt_params *p, pt_plan_info *out) {
  if (!out) return fail(PT_ERR_INVALID, "null argument");
  int rc = check_params(s, cam, p);
  if (rc) return rc;
  const PtSceneFacts f = scene_facts(s);
  PtPlan pl;
  pt_make_plan(f, cam, p, pt_tuning(), pl);
  plan_info(pl, f, p, out);
  return PT_OK;
}

extern "C" int pt_debug_set_tuning(const char *name, long long value) {
  if (!name || !pt_tuning_set(pt_tuning(), name, value)) return fail(PT_ERR_INVALID, "unknown tuning switch %s", name ? name : "(null)");
  return PT_OK;
}

extern "C" int pt_debug_get_tuning(const char *name, long long *value) {
  if (!name || !value) return fail(PT_ERR_INVALID, "null argument");
  PtTuning &t = pt_tuning();
#define X(field, env, dflt)                                    \
  if (strcmp(name, #field) == 0 || strcmp(name, env) == 0) {   \
    *value = t.field;                                          \
    return PT_OK;                                              \
  }
  PT_TUNING_TABLE(X)
#undef X
  return fail(PT_ERR_INVALID, "unknown tuning switch %s", name);
}

// debug: the 16 words of the path-tracer queue block (word 0 = queue head, 1..8 = section cycle sums of
// a -DPT_DEBUG_TIME build)
extern "C" int pt_debug_read_queue(pt_scene *s, unsigned long long *out16) {
  if (!s || !out16) return fail(PT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out16, s->queue_last ? s->queue_last : s->queue, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return PT_OK;
}

extern "C" int pt_debug_handed_over(pt_scene *s, unsigned long long *pixels, unsigned long long *budget) {
  if (!s || !pixels || !budget) return fail(PT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long w[3] = {0, 0, 0};  // PT_Q_CHOICE, PT_Q_HEAVY, PT_Q_BUDGET
  static_assert(PT_Q_HEAVY == PT_Q_CHOICE + 1 && PT_Q_BUDGET == PT_Q_CHOICE + 2, "read as one block");
  HIP_TRY(hipMemcpy(w, (s->queue_last ? s->queue_last : s->queue) + PT_Q_CHOICE, sizeof w, hipMemcpyDeviceToHost));
  // (PT_Q_HEAVY keeps counting the lanes that ASKED once the record table is full, pt_path.h: handed over are at most its capacity)
  *pixels = w[0] ? std::min<unsigned long long>(w[1], (unsigned long long)s->last_handover_cap) : 0ULL;
  *budget = w[2];
  return PT_OK;
}

#ifdef PT_DEBUG_TIME
extern "C" int pt_debug_read_dbg(unsigned long long *out8, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pt_dbg), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pt_dbg), z, sizeof z));
  }
  return PT_OK;
}
extern "C" int pt_debug_read_lat_hist(unsigned long long *out160, int clear) {
#ifdef PT_DEBUG_TIME
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out160, HIP_SYMBOL(pt_lat_hist), 160 * sizeof(unsigned long long)));
  if (clear) {
    static const unsigned long long zeros[160] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pt_lat_hist), zeros, sizeof(zeros)));
  }
  return PT_OK;
#else
  (void)out160;
  (void)clear;
  return PT_ERR_INVALID;
#endif
}
extern "C" int pt_debug_read_lat_events(unsigned long long *out, int clear) {  // out: PT_LAT_EVENTS * 3 + 1 words
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_lat_events), (PT_LAT_EVENTS * 3 + 1) * sizeof(unsigned long long)));
  if (clear) {
    const unsigned long long zero = 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pt_lat_events), &zero, sizeof(zero), (PT_LAT_EVENTS * 3) * sizeof(unsigned long long)));
  }
  return PT_OK;
}
extern "C" int pt_debug_read_unitlog(unsigned long long *out, int n_units) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_unitlog), (size_t)std::min(n_units, PT_UNITLOG_LEN) * 8 * sizeof(unsigned long long)));
  return PT_OK;
}
extern "C" int pt_debug_read_trace(unsigned long long *out, int n) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_trace), (size_t)std::min(n, PT_TRACE_LEN + 64 * 80) * sizeof(unsigned long long)));
  return PT_OK;
}
#endif

// ---- diagnostics (not part of the reference seam): the culling predicate on its own ------------------------
// keep[i] (by World.shapes index) = would the cull of the primary rays through the image rectangle
// [x0, x1] x [row0, row1 + 1] keep shape i?  pixel_x >= 0: the same for the cone of the single pixel
// (pixel_x, pixel_row) inside that tile (pixel_cone, perspective only).  Planes answer 1.
extern "C" int pt_debug_cull_probe(pt_scene *s, const pt_camera *cam, int width, int height, int x0, int x1, int row0,
                                   int row1, int pixel_x, int pixel_row, int *keep) {
  if (!s || !cam || !keep || width <= 0 || height <= 0) return fail(PT_ERR_INVALID, "bad probe arguments");
  HIP_TRY(hipSetDevice(s->device));
  PtKArgs a;
  memset(&a, 0, sizeof a);
  fill_cone_model(a, cam, width, height);
  PtKArgs *a_dev = nullptr;
  int *keep_dev = nullptr;
  HIP_TRY(hipMalloc((void **)&a_dev, sizeof a));
  HIP_TRY(hipMalloc((void **)&keep_dev, sizeof(int) * std::max(1, s->n_shapes)));
  a.cold = a_dev;
  a.recs = s->recs;
  a.bounds = s->bounds;
  a.n_shapes = s->n_shapes;
  a.n_spheres = s->n_spheres;
  HIP_TRY(hipMemcpy(a_dev, &a, sizeof a, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pt_cull_probe_kernel, dim3(1), dim3(64), 0, 0, a, x0, x1, row0, row1, pixel_x, pixel_row, keep_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(keep, keep_dev, sizeof(int) * s->n_shapes, hipMemcpyDeviceToHost));
  (void)hipFree(a_dev);
  (void)hipFree(keep_dev);
  return PT_OK;
}

// ---- diagnostics (not part of the reference seam): device primitive probe used by the tests -------------
extern "C" int pt_debug_probe(int op, const double *x, const double *y, double *out, int n) {
  if (n <= 0 || !x || !out) return fail(PT_ERR_INVALID, "bad probe arguments");
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  HIP_TRY(hipMalloc((void **)&dx, bytes));
  HIP_TRY(hipMalloc((void **)&dy, bytes));
  HIP_TRY(hipMalloc((void **)&dout, bytes));
  HIP_TRY(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dy, y ? y : x, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pt_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, dx, dy, dout, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(dy);
  (void)hipFree(dout);
  return PT_OK;
}

// ---- diagnostics: the hit record, camera rays and BRDF scattering of the device path on their own ----------------
extern "C" int pt_debug_hit_probe(pt_scene *s, int shape_index, const double *rays, int n, double *out) {
  if (!s || !rays || !out || n <= 0 || shape_index >= s->n_shapes) return fail(PT_ERR_INVALID, "bad probe arguments");
  HIP_TRY(hipSetDevice(s->device));
  PtKArgs a;
  memset(&a, 0, sizeof a);
  a.recs = s->recs;
  a.aux = s->aux;
  a.diag = s->diag;
  a.n_shapes = s->n_shapes;
  a.n_spheres = s->n_spheres;
  a.n_diag = s->n_diag;
  double *rd = nullptr, *od = nullptr;
  HIP_TRY(hipMalloc((void **)&rd, (size_t)n * 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&od, (size_t)n * 12 * sizeof(double)));
  HIP_TRY(hipMemcpy(rd, rays, (size_t)n * 8 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pt_hit_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, a, shape_index, rd, n, od);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, od, (size_t)n * 12 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(rd);
  (void)hipFree(od);
  return PT_OK;
}

extern "C" int pt_debug_lanes_probe(pt_scene *s, int anyhit, const double *rays, int n, double *out) {
  if (!s || !rays || !out || n <= 0) return fail(PT_ERR_INVALID, "bad probe arguments");
  HIP_TRY(hipSetDevice(s->device));
  PtKArgs a;
  memset(&a, 0, sizeof a);
  fill_scene_args(s, a);
  PtKArgs *a_dev = nullptr;
  double *rd = nullptr, *od = nullptr;
  HIP_TRY(hipMalloc((void **)&a_dev, sizeof a));
  HIP_TRY(hipMalloc((void **)&rd, (size_t)n * 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&od, (size_t)n * 4 * sizeof(double)));
  a.cold = a_dev;
  HIP_TRY(hipMemcpy(a_dev, &a, sizeof a, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(rd, rays, (size_t)n * 8 * sizeof(double), hipMemcpyHostToDevice));
  if (anyhit)
    hipLaunchKernelGGL(pt_lanes_probe_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, 0, a, rd, n, od);
  else
    hipLaunchKernelGGL(pt_lanes_probe_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, 0, a, rd, n, od);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, od, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(a_dev);
  (void)hipFree(rd);
  (void)hipFree(od);
  return PT_OK;
}

extern "C" int pt_debug_camera_probe(const pt_camera *cam, int width, int height, const double *pix, int n, double *out) {
  if (!cam || !pix || !out || n <= 0 || width <= 0 || height <= 0) return fail(PT_ERR_INVALID, "bad probe arguments");
  PtKArgs a;
  memset(&a, 0, sizeof a);
  a.cam_kind = cam->kind;
  memcpy(a.cam_m, cam->m, sizeof a.cam_m);
  a.cam_dist = cam->screen_distance;
  a.cam_aspect = cam->aspect_ratio;
  a.W = width;
  a.H = height;
  PtKArgs *a_dev = nullptr;
  double *pd = nullptr, *od = nullptr;
  HIP_TRY(hipMalloc((void **)&a_dev, sizeof a));
  HIP_TRY(hipMalloc((void **)&pd, (size_t)n * 4 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&od, (size_t)n * 7 * sizeof(double)));
  a.cold = a_dev;
  HIP_TRY(hipMemcpy(a_dev, &a, sizeof a, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(pd, pix, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pt_camera_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, a, pd, n, od);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, od, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(a_dev);
  (void)hipFree(pd);
  (void)hipFree(od);
  return PT_OK;
}

extern "C" int pt_debug_scatter_probe(const double *in, int n, double *out, unsigned long long *state_after) {
  if (!in || !out || !state_after || n <= 0) return fail(PT_ERR_INVALID, "bad probe arguments");
  double *id = nullptr, *od = nullptr;
  unsigned long long *sd = nullptr;
  HIP_TRY(hipMalloc((void **)&id, (size_t)n * 12 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&od, (size_t)n * 7 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&sd, (size_t)n * sizeof(unsigned long long)));
  HIP_TRY(hipMemcpy(id, in, (size_t)n * 12 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pt_scatter_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, id, n, od, sd);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, od, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(state_after, sd, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(id);
  (void)hipFree(od);
  (void)hipFree(sd);
  return PT_OK;
}
