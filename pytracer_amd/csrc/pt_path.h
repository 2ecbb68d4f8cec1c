// pt_path.h -- PathTracer: work units and the per-lane state machine (pt_path_kernel, pt_path_regions_kernel).
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
// ---- work units for the path tracer's second pass ---------------------------------------------------------
// The first pass (pt_tile_kernel<PATHTRACER>) leaves, per 8x8 region, the mask of the pixels that need real
// path tracing and their number as a key.  The second pass works in UNITS: a unit is up to `ppu` flagged
// pixels of one region, rendered by one wave whose 64 lanes are shared out L = min(S*S, 64 / pixels) to a
// pixel -- the lanes of a pixel trace different samples of it at the same time (path_trace).  ppu is chosen
// from the frame's total F of flagged pixels: with few of them (a rank's share of a frame, a sparse frame) a
// region is cut into several units so that the whole chip works on samples in parallel instead of a few
// waves walking their pixels' S*S samples one after the other; with many, ppu = 64 (a unit = a region) and
// nothing is spent on idle lanes.  Regions without flagged pixels yield nothing.  The order of the units only
// changes WHEN a pixel is rendered, never its value.
// Units with the most pixels come first (they have the fewest lanes per pixel, hence the longest chains): a
// counting sort by size over any number of workgroups, one region per thread -- the first pass counts the regions
// by their number of flagged pixels, pt_unit_scatter turns the counts into descending offsets of the unit sizes
// (every workgroup for itself: 64 numbers) and places the units.
// queue[0] = queue head, [9] = number of units, [10] = ppu (for the statistics), [11] = F (summed up by the first
// pass), [16 + k] = regions with k flagged pixels (first pass), [96 + s] = units of s pixels placed so far; [PT_QUEUE_HEADS + 32 s] = head of shard s
// of the unit list (the second pass pulls units through PT_UNIT_SHARDS heads, 256 B apart: one word takes ~88
// dequeues/us, and thousands of waves pull); all zeroed before the first pass.
PT_DEV int unit_ppu(const unsigned long long *queue, long long lanes_cap, int nsamp, int min_rounds) {
  // lanes per pixel every unit gets at least: the largest power of two (<= S*S, <= 64) at which all flagged
  // pixels together still fit the lanes the launch keeps resident ...
  const unsigned long long total = queue[11];
  int lg = 1;
  while (lg * 2 <= 64 && lg * 2 <= nsamp && (long long)total * (lg * 2) <= lanes_cap) lg *= 2;
  // ... or more, up to four units per resident wave, as long as a unit keeps `min_rounds` rounds of work: many
  // short units spread over the chip more evenly than few long ones (a unit's time varies a lot with what its
  // pixels see), but every unit costs a fetch and a cull, and lanes beyond what speculation can use are wasted
  // (PT_PCG_PIXEL asks for more rounds per unit than PT_PCG_SAMPLE for that reason).  min_rounds < 0: -min_rounds
  // rounds, and single-round units where even those come to three or more per resident wave (a full frame of
  // PT_PCG_SAMPLE: the fetch is cheap next to what finer balancing saves; with fewer units it is not).
  const int mr = min_rounds < 0 ? -min_rounds : min_rounds;
  while (lg * 2 <= 64 && lg * 2 * mr <= nsamp && (long long)total * (lg * 2) <= 4 * lanes_cap) lg *= 2;
  if (min_rounds < 0 && lg * 2 <= 64 && lg * 2 <= nsamp && (long long)total * (lg * 2) <= 4 * lanes_cap &&
      (long long)total * (lg * 2) >= 3 * lanes_cap)
    lg *= 2;
  return 64 / lg;
}
#ifndef PT_SCATTER_BLOCK
#define PT_SCATTER_BLOCK 256
#endif
__global__ void pt_unit_scatter(const unsigned char *keys, const unsigned long long *masks, int n, int4 *units, int units_cap,
                                unsigned long long *queue, long long lanes_cap, int nsamp, int min_rounds, long long q_min_flagged = -1,
                                double q_budget_per_flagged = 0.0, int q_budget_min = 0) {
  __shared__ int cnt[65], offs[65];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = i < n ? keys[i] : 0;
  const unsigned long long m = i < n ? masks[i] : 0ULL;  // carried in the unit: one dependent load less when a wave fetches it
  const int h = threadIdx.x < 64 ? (int)queue[16 + threadIdx.x + 1] : 0;  // (requested together with F: one round trip)
  const int ppu = unit_ppu(queue, lanes_cap, nsamp, min_rounds);
  // regions with k flagged pixels (counted by the first pass) -> units of s pixels: a region yields k / ppu units
  // of ppu pixels and one of k % ppu.  The first wave does it, lane k - 1 for the regions of k pixels.
  if (threadIdx.x < 65) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int kk = threadIdx.x + 1;
    if (h) {
      if (kk >= ppu) atomicAdd(&cnt[ppu], h * (kk / ppu));
      if (kk % ppu) atomicAdd(&cnt[kk % ppu], h);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // offs[s] = units of more than s pixels (descending order of size)
    const int sz = 64 - threadIdx.x;  // lane 0 holds the largest size
    const int c = cnt[sz];
    int upto = c;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(upto, off, 64);
      if ((int)threadIdx.x >= off) upto += v;
    }
    offs[sz] = upto - c;
    if (threadIdx.x == 63 && blockIdx.x == 0) {
      queue[9] = (unsigned long long)(upto < units_cap ? upto : units_cap);
      queue[10] = (unsigned long long)ppu;
      // num_of_rays > 1: which second-pass kernel works on this frame (PT_Q_CHOICE); < 0: the tree kernel, always
      queue[PT_Q_CHOICE] = (q_min_flagged >= 0 && (long long)queue[11] >= q_min_flagged) ? 1ULL : 0ULL;
      // ... and after how many rays a lane of the one-queue kernel hands its pixel to the tree kernel: about what a lane
      // traces in the whole frame if the work were spread evenly (F x mean rays of a pixel / lanes) -- a pixel's own chain
      // should not outlast that by much
      const double qb = (double)queue[11] * q_budget_per_flagged;
      queue[PT_Q_BUDGET] = (unsigned long long)(qb > (double)q_budget_min ? (qb < 1e9 ? qb : 1e9) : (double)q_budget_min);
    }
  }
  __syncthreads();
  const int full = k / ppu, rem = k - full * ppu;
  // the units of ppu pixels (most of them): one returning atomic per wave, the lanes share out what it reserved
  const int lane = threadIdx.x & 63;
  int upto = full;  // inclusive prefix sum over the wave
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(upto, off, 64);
    if (lane >= off) upto += v;
  }
  const int wave_total = __shfl(upto, 63, 64);
  int base = 0;
  if (wave_total) {
    if (lane == 63) base = (int)atomicAdd(queue + 96 + ppu, (unsigned long long)wave_total);
    base = __shfl(base, 63, 64);
  }
  if (!k) return;
  const int mlo = (int)(unsigned)m, mhi = (int)(unsigned)(m >> 32);
  if (full) {
    const int at = offs[ppu] + base + upto - full;
    for (int g = 0; g < full; ++g)
      if (at + g < units_cap) units[at + g] = make_int4(i, (g * ppu) | (ppu << 8), mlo, mhi);  // (region, first | count << 8, mask)
  }
  if (rem) {
    const int at = offs[rem] + (int)atomicAdd(queue + 96 + rem, 1ULL);
    if (at < units_cap) units[at] = make_int4(i, (full * ppu) | (rem << 8), mlo, mhi);
  }
}

// position of the n-th (0-based) set bit of m (which has more than n bits set)
PT_DEV int nth_set_bit(unsigned long long m, int n) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll((m >> pos) & ((1ULL << w) - 1ULL));
    if (n >= c) {
      pos += w;
      n -= c;
    }
  }
  return pos & 63;
}

#ifdef PT_DEBUG_TIME
#define PT_TRACE_LEN 8192
__device__ unsigned long long pt_trace[PT_TRACE_LEN + 64 * 80];  // (+ the traced unit's validated draw counts: [pixel][sample], tools/dbgdraws.py)
#endif

// ---- PathTracer (render.py:99-139) as a per-lane state machine ----------------------------------------
// The reference recursion is depth-first; frame `k` of the explicit stack is the call at depth k.
// Frame fields (in ws, [slot][field][thread] so a wave's accesses are contiguous):
//   0..2 hit_color (after Russian roulette)   3..5 emitted
//   N > 1 only: 6..8 cum_radiance, 9 children done, 10..12 hit point, 13..15 normal,
//               16..18 incoming direction, 19 brdf kind
// Pixels are handed out dynamically (one wave-aggregated atomic per refill): a lane that finishes
// a cheap pixel (sky) immediately takes the next one, so a few expensive pixels (deep recursion,
// num_of_rays > 1) do not hold 63 idle lanes hostage.  Per-pixel seeds make the image independent
// of which lane renders which pixel.
struct PathCtx {
  double *ws;
  size_t stride;  // frame_doubles * nthreads
  size_t nthreads;
  int gtid;
  int lds_base, lds_frame;  // LDS frames: first double of the frame area, doubles per frame
  int deep_slot;            // HOME 2: the one stack slot that lives in LDS (the deepest: max_depth - 1)
};
// The frame stack lives in LDS whenever (max_depth x frame) x 256 lanes fits beside the survivor masks
// (LDSF): the second pass is a chain of dependent steps per pixel, and a frame access that goes to
// HBM costs more than the step's arithmetic.  Same [slot][field][lane] layout in both homes.
// HOME 2 (round 5, the one-queue kernel of num_of_rays > 1): SPLIT -- only the DEEPEST slot lives in LDS, the shallower ones
// in HBM.  A tree of N children per node touches the frame at depth d once per node below it, N^(d+1) times per pixel: at the
// CLI's N = 10, D = 3 the deepest frame takes 1000 of a pixel's 1110 frame visits, the one above it 100, the root's 10.  One
// frame of 20 doubles per lane is 40 KB per workgroup instead of 120: two workgroups per CU -- two waves per SIMD, where
// all three frames in LDS allow one -- and nine visits in ten still never leave the CU.
extern __shared__ double pt_lds_f64[];  // the same dynamic LDS block as pt_lds_masks
// A frame of the split stack is reached through ONE generic pointer picked per lane when the frame is entered (a flat
// access resolves to LDS or to memory by its address): the two homes then share every load and store instead of doubling
// them behind a per-lane branch (which cost 180 - 350 bytes of scratch at the 256 registers two waves per SIMD leave).
struct FrameRef {
  double *p;        // field 0 of this lane's frame
  size_t fstride;   // doubles between consecutive fields
};
PT_DEV FrameRef frame_ref_split(const PathCtx &w, int slot) {
  FrameRef r;
  double *lds = (double *)(pt_lds_f64 + w.lds_base + (int)threadIdx.x);
  double *mem = w.ws + (size_t)slot * w.stride + w.gtid;
  const bool deep = slot == w.deep_slot;
  r.p = deep ? lds : mem;
  r.fstride = deep ? (size_t)PT_BLOCK : w.nthreads;
  return r;
}
template <int HOME>
PT_DEV double ws_get(const PathCtx &w, int slot, int field) {
  static_assert(HOME != 2, "the split stack goes through frame_ref_split");
  if (HOME == 1) return pt_lds_f64[w.lds_base + (slot * w.lds_frame + field) * PT_BLOCK + (int)threadIdx.x];
  return w.ws[(size_t)slot * w.stride + (size_t)field * w.nthreads + w.gtid];
}
template <int HOME>
PT_DEV void ws_put(const PathCtx &w, int slot, int field, double v) {
  static_assert(HOME != 2, "the split stack goes through frame_ref_split");
  if (HOME == 1)
    pt_lds_f64[w.lds_base + (slot * w.lds_frame + field) * PT_BLOCK + (int)threadIdx.x] = v;
  else
    w.ws[(size_t)slot * w.stride + (size_t)field * w.nthreads + w.gtid] = v;
}

// next pixel for every lane with `need` set; returns -1 when the frame is exhausted
PT_DEV long long next_pixel(const PtKArgs &a, bool need, long long npix) {
  const unsigned long long mask = __ballot(need);
  long long pix = -1;
  if (need) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    const int rank = __popcll(mask & ((1ULL << lane) - 1ULL));
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(pt_queue(a), (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
    const long long p = (long long)(base + rank);
    pix = p < npix ? p : -1;
  }
  return pix;
}

// Two kinds of work alternate inside a wave, each executed only by the lanes that need it and only
// when enough of them do (the bodies are skipped wave-wide otherwise):
//   P  lanes starting a sample (mode 0): jitter draws, primary ray, query against the survivors.
//   S  lanes holding a scattered ray (mode 1): query against ALL shapes.
// Both kinds then share one shade + unwind block (deliver radiance up the frame stack, scatter the
// next child) which leaves each lane with a ray to query (mode 1), a finished sample (mode 0 / 3) or a
// finished pixel (mode 2).  S queries are batched until >= 16 lanes wait, so the 32..10k-shape loop
// does not run for one or two lanes at a time; regions of pure background never run it.
//
// !TILED (orthogonal camera): 1 lane = 1 pixel, pixels come from one global queue, a lane walks its pixel's
// samples one after the other and P-steps run the full shape loop.
//
// TILED (perspective camera, second pass): a wave works through UNITS (pt_unit_scatter): up to 64 flagged pixels
// of one 8x8 region.  The unit's P-steps use the hoisted, culled tile query against the region's survivor
// masks.  The wave's lanes are shared out L = min(S*S, 64 / pixels) to a pixel, and the L lanes of a pixel
// trace L consecutive samples of it AT THE SAME TIME (a "round"):
//   PT_PCG_SAMPLE  every sample owns its generator: the L samples are independent, all of them count.
//   PT_PCG_PIXEL   the samples of a pixel share ONE generator, consumed in program order: where sample k+1
//     starts in the stream depends on how many numbers sample k drew, which is only known once its path has
//     ended.  Lane j therefore SPECULATES: it guesses what each of the j samples before it draws -- what the last
//     validated sample of the pixel drew, or, where that has been the better guess for this pixel so far, what each
//     sample's upper neighbour in the S x S grid of strata drew (a pixel across an edge repeats its row of short and
//     long paths; `hist`, `pscore`) -- and starts from the state that many draws ahead (pcg_advance).  After the round the samples are validated in order: sample j counts if and only
//     if the state it started from IS the state sample j-1 ended with -- then everything it computed is what
//     the sequential program computes -- and the first one that started elsewhere is thrown away together
//     with everything behind it and repeated in the next round, now from the right state.  The first lane
//     always starts from the validated state, so every round completes at least one sample.
// A round's radiances are added to the pixel's sum in sample order (imagetracer.py:97: cum_color += ...), one
// lane after the other through wave shuffles, so the sum is the sequential one bit for bit; rays are counted
// for validated samples only.  Per-pixel / per-sample seeds depend on the global pixel index alone: the image
// does not depend on how regions are cut into units or how many lanes a pixel gets.
// LAT: the second pass is built for few waves per SIMD; its time is set by chains of dependent steps:
// everything inline, registers no object.
#ifndef PT_REGIONS_INLINE
#define PT_REGIONS_INLINE 1  // second pass: HitRecord / scatter / transcendental code inline (1) or behind calls (0)
#endif
// FLAGGED (!TILED only): the kernel runs BEHIND the first pass, as the alternative to pt_path_tree_kernel (PT_Q_CHOICE): it
// returns at once unless the device chose it, and a lane keeps only pixels the first pass flagged (the others are settled).
template <bool TILED, int LDSF, bool LAT, bool SLDS = false, int LEAN = 0, bool FLAGGED = false>
PT_DEV void path_trace(const PtKArgs &a) {
  constexpr bool INL = LAT && PT_REGIONS_INLINE;
  static_assert(LDSF != 2 || !TILED, "the split frame stack belongs to the one-queue kernel");
  static_assert(!SLDS || INL, "the scene is staged in LDS for the second pass only");
  static_assert(!FLAGGED || !TILED, "the flagged-pixel filter belongs to the one-queue kernel");
  PathCtx w;
  int S, nsamp, N, W = 0, rows_local = 0, npass = 0, D = 0, rr = 0, diag_lds = -1, pcg_mode = PT_PCG_PIXEL;
  bool ortho = false;
  {
    pt_kargs c = cold_args(a);
    w.ws = c->ws;
    w.nthreads = (size_t)c->nthreads;
    w.stride = (size_t)c->frame_doubles * w.nthreads;
    w.gtid = blockIdx.x * PT_BLOCK + threadIdx.x;
    w.lds_frame = c->frame_doubles;
    w.lds_base = TILED ? 4 * c->npass : 0;  // behind the four waves' survivor masks (8-byte units)
    w.deep_slot = (c->D > 1 ? c->D : 1) - 1;
    ortho = c->cam_kind != PT_CAMERA_PERSPECTIVE;
    diag_lds = c->diag_lds;
    pcg_mode = c->pcg_mode;
    S = c->S;
    N = c->N;
    W = c->W;
    rows_local = c->rows_local;
    npass = c->npass;
    D = c->D;
    rr = c->rr;
  }
  if (blockIdx.x == gridDim.x - 1) {  // the next frame's queue block (nothing of this frame reads it)
    unsigned long long *qn = pt_queue_next(a);
    for (int k = threadIdx.x; k < PT_QUEUE_WORDS; k += PT_BLOCK) qn[k] = 0ULL;
  }
  if (FLAGGED && pt_queue(a)[PT_Q_CHOICE] != 1ULL) {  // (uniform over the grid: the tree kernel renders this frame)
    add_ray_count(a, 0ULL, cold_args(a)->count_base);
    return;
  }
  if (LAT && diag_lds >= 0) {
    // scale+translate records into LDS: world_query_lanes fetches them by lane-private index
    const unsigned long long *src = (const unsigned long long *)a.diag;
    for (int k = threadIdx.x; k < a.n_diag * 8; k += PT_BLOCK) pt_lds_masks[diag_lds + k] = src[k];
    __syncthreads();
  }
  int scene_lds = 0;
  if (SLDS) {
    // the shapes' records (128 B + 256 B each) into LDS: shading gathers ~20 values of the hit shape per lane, and a
    // gather from LDS costs a fraction of one through the vector memory path (8 waves of a CU share one of those)
    scene_lds = cold_args(a)->scene_lds;
    const unsigned long long *src = (const unsigned long long *)a.recs;
    for (int k = threadIdx.x; k < a.n_shapes * 16; k += PT_BLOCK) pt_lds_masks[scene_lds + k] = src[k];
    src = (const unsigned long long *)a.aux;
    for (int k = threadIdx.x; k < a.n_shapes * 32; k += PT_BLOCK) pt_lds_masks[scene_lds + a.n_shapes * 16 + k] = src[k];
    __syncthreads();
  }
  if (LAT) {  // the grid's occupancy bits into LDS: the cell walk of world_query_lanes reads one per step
    pt_kargs c = cold_args(a);
    const int occ_lds = c->grid_occ_lds;
    if (occ_lds >= 0) {
      const int nwords = (c->grid_res[0] * c->grid_res[1] * c->grid_res[2] + 31) / 32;
      unsigned *dst = (unsigned *)pt_lds_masks;
      for (int k = threadIdx.x; k < nwords; k += PT_BLOCK) dst[occ_lds + k] = c->grid_occ[k];
      __syncthreads();
    }
  }
  nsamp = S > 0 ? S * S : 1;
  const double invN = 1.0 / (double)N;
  const PcgJump j2n = pcg_jump_coeffs(2u * (unsigned)N);  // (wave-uniform: the scatter draws of N children beyond max_depth)
  const int lane = threadIdx.x & 63;
  const int mbase = (threadIdx.x >> 6) * npass;
  const int regions_x = (W + PT_REGION - 1) / PT_REGION;
  bool exhausted = false;           // !TILED: the global queue is empty
  // (q_budget < 0 in the argument block: the budget pt_unit_scatter derived from the frame's flagged pixels)
  const int q_budget = !FLAGGED ? 0 : (cold_args(a)->q_budget >= 0 ? cold_args(a)->q_budget : (int)pt_queue(a)[PT_Q_BUDGET]), q_tail = FLAGGED ? cold_args(a)->q_tail_budget : 0,
            q_few = FLAGGED ? cold_args(a)->q_few_lanes : 0;
  unsigned qtail = 0;               // FLAGGED: rays of the lane's pixel since the queue ran dry
  bool q_full = false;              // FLAGGED: the record table was full when this lane last asked
  unsigned qrays = 0;               // FLAGGED: rays of the lane's pixel so far (counted when the pixel is done: a pixel over budget is the tree kernel's)
  bool first_unit = true;           // TILED (wave-uniform)
  // TILED: units of the frame (written by pt_unit_scatter before this kernel started; read once -- not from the heads' lines)
  const int n_units = TILED ? (int)pt_queue(a)[9] : 0;
  // FLAGGED: flagged pixels of the frame (one unit each: pt_unit_scatter as for the tree kernel) and how the queue deals
  const int n_flagged = FLAGGED ? (int)pt_queue(a)[9] : 0;
  const bool deal_units = FLAGGED && 4LL * (long long)n_flagged < a.npix;
  unsigned long long nrays = 0;

  // lane state.  mode 0: starts a sample at the next P-step; 1: inside a path (S-steps); 2: nothing to do;
  // 3 (TILED): sample finished, waits for the end of the round
  int mode = 2;
  long long pix = -1;
  Pcg pcg;
  pcg.state = 0;
  pcg.inc = 1;
  pcg.n = 0;
  int samp = 0, sp = 0, col = 0, grow = 0;
  V3 cum = {0.0, 0.0, 0.0};
  Ray ray;
  ray.o = {0.0, 0.0, 0.0};
  ray.d = {1.0, 0.0, 0.0};
  ray.tmin = 1e-5;
  // what shade() hands to the unwind loop of the same step: a value to deliver, or a child to spawn
  V3 ret = {0.0, 0.0, 0.0};
  bool spawn = false;
  V3 f_wp = {0.0, 0.0, 0.0}, f_n = {0.0, 0.0, 1.0}, f_in = {1.0, 0.0, 0.0};
  int f_brdf = 0;
  // TILED: the unit (wave-uniform) and this lane's place in it
  int L = 1;                        // lanes per pixel
  int leader = lane, jlane = 0;     // first lane of this lane's pixel; this lane's sample slot in a round
  bool in_unit = false;             // the lane belongs to a pixel of the unit
  int vbase = 0;                    // samples of the pixel validated so far (same in all lanes of the pixel)
  uint64_t vstate = 0;              // PT_PCG_PIXEL: generator state behind the last validated sample
  // PT_PCG_PIXEL: what the pixel's last eight validated samples drew, a byte each, the latest in the low byte.  The guess
  // for sample k is what sample k - S drew -- its neighbour one row up in the S x S grid of strata (imagetracer.py:86-93):
  // a pixel across an edge repeats its pattern of short and long paths row after row, where "what the last sample drew"
  // is wrong twice per row.  (S > 8: the sample before it.)
  uint64_t hist = 0;
  int pscore = 0;                   // ... how much more often the upper neighbour was the better guess than the predecessor (per pixel)
  const int hperiod = (S >= 1 && S <= 8) ? S : 1;
  // (Round 4 also let pixels with lanes to spare trace the next samples from a WINDOW of start states; it moved a full frame
  //  by nothing -- four lanes per pixel -- and was deleted in round 5: profiles/DROPPED_VARIANTS.md.)
  uint64_t st_start = 0;            // state this lane's sample started from
  unsigned srays = 0, prays = 0;    // rays of the current sample; of the pixel's validated samples
  unsigned long long gpix = 0;      // global pixel index (seeds)

  // pixel coordinates + seeds + the sample's primary ray (imagetracer.py:86-97)
  auto start_sample = [&]() {
    pt_kargs c = cold_args(a);
    if (!TILED) {
      if (samp == 0) {
        pixel_coords(a, pix, col, grow);
        if (c->pcg_mode == PT_PCG_PIXEL) pcg_seed(pcg, c->s0, c->q0 + ((unsigned long long)grow * c->W + col));
      }
      if (c->pcg_mode == PT_PCG_SAMPLE)
        pcg_seed(pcg, c->s0, c->q0 + ((unsigned long long)grow * c->W + col) * (unsigned)nsamp + (unsigned)samp);
    }
    double up = 0.5, vp = 0.5;
    if (S > 0) {
      const int sr = samp / S, sc = samp - sr * S;
      up = ((double)sc + pcg_float(pcg)) / (double)S;
      vp = ((double)sr + pcg_float(pcg)) / (double)S;
    }
    ray = primary_ray(a, col, grow, up, vp);
  };

  // TILED: the sample of the round this lane traces (vbase + jlane) and the generator state it starts from.
  // -> whether that sample exists (vbase + jlane < nsamp)
  auto seed_round = [&]() -> bool {
    pt_kargs c = cold_args(a);
    if (pcg_mode == PT_PCG_SAMPLE) {
      samp = vbase + jlane;
      if (samp < nsamp) pcg_seed(pcg, c->s0, c->q0 + gpix * (unsigned)nsamp + (unsigned)samp);
    } else {
      // (sample vbase + i: what its upper neighbour vbase + i - period drew, if the pixel has got that far and that guess
      //  has been the better one so far; else what the last validated sample drew)
      const int period = hperiod;
      unsigned ahead = (unsigned)jlane * ((unsigned)hist & 0xffu);
      if (pscore > 0) {
        ahead = 0;
        for (int i = 0; i < jlane; ++i)
          ahead += (unsigned)(hist >> (vbase + (i % period) >= period ? 8 * (period - 1 - (i % period)) : 0)) & 0xffu;
      }
      samp = vbase + jlane;
      pcg.state = pcg_advance(vstate, pcg.inc, ahead);
    }
    pcg.n = 0;
    st_start = pcg.state;
    srays = 0;
    return samp < nsamp;
  };

  // render.py:103-139 up to (not including) the recursion: sets `ret`, or pushes frame `sp` and asks
  // for child 0 (`spawn`).  `ray` is the ray that was queried, at depth `sp`.
  auto shade_hit = [&](auto rec, auto ax, double best_t) {
    V3 hc, em;
    double lum;
    Hit h;
    h.u = 0.0;
    h.v = 0.0;
    const bool uv = ax->needs_uv != 0;
    bool details = false;
    if (uv) {
      if constexpr (INL)
        hit_details<true>(rec, ax, ray, best_t, h, true);
      else
        hit_details_call(rec, ax, &ray, best_t, &h, true);
      details = true;
    }
    hc = brdf_pigment(a, ax, h.u, h.v);
    em = emitted_pigment(a, ax, h.u, h.v);
    lum = max2(max2(hc.x, hc.y), hc.z);
    if (sp >= rr) {  // render.py:116-123
      const double q = max2(0.05, 1.0 - lum);
      if (pcg_float(pcg) > q) {
        const double k = 1.0 / (1.0 - q);
        hc.x = hc.x * k;
        hc.y = hc.y * k;
        hc.z = hc.z * k;
      } else {
        ret = em;
        return;
      }
    }
    if (!(lum > 0.0)) {  // render.py:139 with cum_radiance = 0
      ret.x = em.x + 0.0 * invN;
      ret.y = em.y + 0.0 * invN;
      ret.z = em.z + 0.0 * invN;
      return;
    }
    if (sp + 1 > D) {
      // Every child of this hit would be beyond max_depth: the reference still calls scatter_ray for each
      // (consuming its draws: 2 for a diffuse BRDF, none for a mirror) and each child returns black at
      // render.py:100-101 without a world query.  No ray, no frame, no geometry is needed: advance the
      // generator and accumulate hit_color * 0 exactly as render.py:135-139 does.
      // (Round 5: the N x 2 draws are ONE jump of the generator -- nobody reads their outputs --, and the N additions of
      //  hit_color * 0 are one: 0 + z + z + ... = 0 + z for z = +-0 and for a NaN, whatever N >= 1.  Exact, and a fifth of
      //  what a leaf hit used to cost: 2 N dependent 64-bit multiply-adds and 3 N dependent additions.)
      const bool diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
      if (diffuse) pcg_jump(pcg, j2n, 2u * (unsigned)N);
      const V3 fc = {0.0 + hc.x * 0.0, 0.0 + hc.y * 0.0, 0.0 + hc.z * 0.0};
      ret.x = em.x + fc.x * invN;
      ret.y = em.y + fc.y * invN;
      ret.z = em.z + fc.z * invN;
      return;
    }
    // render.py:126-137: push the frame, child 0 is scattered at the next S-step
    if (!details) {
      if constexpr (INL)
        hit_details<true>(rec, ax, ray, best_t, h, false);
      else
        hit_details_call(rec, ax, &ray, best_t, &h, false);
    }
    FrameRef fr = {nullptr, 0};
    if constexpr (LDSF == 2) fr = frame_ref_split(w, sp);
    auto fput = [&](int field, double v) {
      if constexpr (LDSF == 2)
        fr.p[(size_t)field * fr.fstride] = v;
      else
        ws_put<LDSF>(w, sp, field, v);
    };
    fput(0, hc.x);
    fput(1, hc.y);
    fput(2, hc.z);
    fput(3, em.x);
    fput(4, em.y);
    fput(5, em.z);
    if (N > 1) {
      fput(6, 0.0);
      fput(7, 0.0);
      fput(8, 0.0);
      fput(9, 0.0);
      fput(10, h.wp.x);
      fput(11, h.wp.y);
      fput(12, h.wp.z);
      fput(13, h.n.x);
      fput(14, h.n.y);
      fput(15, h.n.z);
      fput(16, ray.d.x);
      fput(17, ray.d.y);
      fput(18, ray.d.z);
      fput(19, (double)ax->brdf_kind);
    }
    f_wp = h.wp;
    f_n = h.n;
    f_in = ray.d;
    f_brdf = ax->brdf_kind;
    sp++;
    spawn = true;
  };
  auto shade = [&](int hit, double best_t) {
    spawn = false;
    if (hit < 0) {  // render.py:103-105
      pt_kargs c = cold_args(a);
      ret.x = c->bg[0];
      ret.y = c->bg[1];
      ret.z = c->bg[2];
      return;
    }
    if constexpr (SLDS)
      shade_hit((pt_lds_rec)(const void *)(pt_lds_f64 + scene_lds) + hit,
                (pt_lds_aux)(const void *)(pt_lds_f64 + scene_lds + a.n_shapes * 16) + hit, best_t);
    else
      shade_hit(a.recs + hit, cold_args(a)->aux + hit, best_t);
  };

  // the primary call returned `ret`: one sample done (imagetracer.py:94-104)
  auto finish_sample = [&]() {
    if (TILED) {  // the radiance stays in `ret` until the round is validated
      mode = 3;
      return;
    }
    if (S > 0) {
      cum.x = cum.x + ret.x;
      cum.y = cum.y + ret.y;
      cum.z = cum.z + ret.z;
    } else {
      cum = ret;
    }
    mode = 0;
    if (++samp == nsamp) {
      if (S > 0) {
        const double k = 1.0 / (double)(S * S);
        cum.x = cum.x * k;
        cum.y = cum.y * k;
        cum.z = cum.z * k;
      }
      store_pixel(a, pix, cum);
      if (FLAGGED) {
        nrays += qrays;
        qrays = 0;
        qtail = 0;
      }
      cum.x = 0.0;
      cum.y = 0.0;
      cum.z = 0.0;
      samp = 0;
      mode = 2;
    }
  };

#ifdef PT_DEBUG_TIME
  // section sums for every wave, plus a step-by-step trace of the wave that drew the first (fullest) unit
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
  bool tracing = false;
  int trace_n = 0;
  int ulog_seq = -1, ulog_rounds = 0, ulog_iters = 0;
  unsigned long long ulog_t[4] = {0, 0, 0, 0};
  unsigned long long dbg_q[3] = {0, 0, 0};
#define PT_STAMP(k)                                                                        \
  do {                                                                                     \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                            \
    tsum[k] += tn - tprev;                                                                 \
    const unsigned long long np_ = (unsigned long long)__popcll(__ballot(mode == 1));     \
    if (tracing && lane == 0 && trace_n < PT_TRACE_LEN)                                    \
      pt_trace[trace_n] = ((tn - tprev) << 16) | (np_ << 8) | (k);                         \
    if (tracing) trace_n++;                                                                \
    tprev = tn;                                                                            \
  } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif
  for (;;) {
    PT_STAMP(7);
    // (values that never flow from one iteration into the next: said explicitly, so that they hold no
    //  registers across the queries)
    spawn = false;
    f_wp = {0.0, 0.0, 0.0};
    f_n = {0.0, 0.0, 1.0};
    f_in = {1.0, 0.0, 0.0};
    f_brdf = 0;
    // ---- work for idle lanes ----
    if (TILED) {
      if (!__any(mode == 0 || mode == 1)) {
        if (__any(mode == 3)) {
          // ---- end of a round: validate the pixel's samples in order, add them up in order ----
          // (every lane of a pixel runs the same loop over the pixel's L lanes and ends with the same
          //  vbase / vstate / hist; only the values in the leader are used for the pixel's result)
          const bool fin = mode == 3;
          bool chain = true;
#ifdef PT_DEBUG_TIME
          ulog_rounds++;
          const int dbg_vbase0 = vbase;
          int dbg_fin = 0;
#endif
          // What the walk reads of a lane: radiance, whether its sample finished, its rays, and (PT_PCG_PIXEL) the sample's
          // index, draws, start and end state.  With the frame stack in LDS the lanes PARK these in slot 0 of it -- no lane is
          // inside a path at the end of a round, the stack is empty -- and the walk reads them from there: one LDS read per
          // value instead of two cross-lane permutes per double (the walk was 17 - 22 % of the second pass's cycles under
          // PT_PCG_SAMPLE: 16 turns for a pixel with 16 lanes).
          const int park = w.lds_base + (int)(threadIdx.x & ~63u);  // field f of lane l of this wave: park + f * PT_BLOCK + l
          if (LDSF == 1) {
            const int me = park + lane;
            pt_lds_f64[me] = ret.x;
            pt_lds_f64[me + PT_BLOCK] = ret.y;
            pt_lds_f64[me + 2 * PT_BLOCK] = ret.z;
            // (fin | sample index, 23 bits: S <= 1024 | the draws' low byte, all that `hist` keeps | rays of the sample)
            pt_lds_masks[me + 3 * PT_BLOCK] = (unsigned long long)(fin ? 1u : 0u) | ((unsigned long long)((unsigned)samp & 0x7fffffu) << 1) |
                                              ((unsigned long long)(pcg.n & 0xffu) << 24) | ((unsigned long long)srays << 32);
            if (pcg_mode != PT_PCG_SAMPLE) {
              pt_lds_masks[me + 4 * PT_BLOCK] = st_start;
              pt_lds_masks[me + 5 * PT_BLOCK] = pcg.state;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          for (int jj = 0; jj < L; ++jj) {
            const int src = (leader + jj) & 63;
            uint64_t s_from = 0, s_to = 0;  // (PT_PCG_SAMPLE validates nothing: no states, no draws)
            unsigned s_draws = 0, s_rays;
            int s_fin, s_samp = 0;
            double rx_, ry_, rz_;
            if (LDSF == 1) {
              const int at = park + src;
              const unsigned long long meta = pt_lds_masks[at + 3 * PT_BLOCK];
              rx_ = pt_lds_f64[at];
              ry_ = pt_lds_f64[at + PT_BLOCK];
              rz_ = pt_lds_f64[at + 2 * PT_BLOCK];
              s_fin = (int)(meta & 1ULL);
              s_samp = (int)((meta >> 1) & 0x7fffffULL);
              s_draws = (unsigned)(meta >> 24) & 0xffu;
              s_rays = (unsigned)(meta >> 32);
              if (pcg_mode != PT_PCG_SAMPLE) {
                s_from = pt_lds_masks[at + 4 * PT_BLOCK];
                s_to = pt_lds_masks[at + 5 * PT_BLOCK];
              }
            } else {
              if (pcg_mode != PT_PCG_SAMPLE) {
                s_from = __shfl((unsigned long long)st_start, src, 64);
                s_to = __shfl((unsigned long long)pcg.state, src, 64);
                s_draws = (unsigned)__shfl((int)pcg.n, src, 64);
                s_samp = __shfl(samp, src, 64);
              }
              s_fin = __shfl((int)fin, src, 64);
              s_rays = (unsigned)__shfl((int)srays, src, 64);
              rx_ = __shfl(ret.x, src, 64);
              ry_ = __shfl(ret.y, src, 64);
              rz_ = __shfl(ret.z, src, 64);
            }
            if (pcg_mode == PT_PCG_SAMPLE) {
              chain = chain && s_fin != 0;
            } else {
              // the lane's sample counts iff it is the NEXT one and it started from the state the sequential program is in
              chain = s_fin != 0 && s_samp == vbase && s_from == vstate;
            }
#ifdef PT_DEBUG_TIME
            dbg_fin += s_fin;
#endif
            if (chain) {
              if (S > 0) {  // imagetracer.py:97
                cum.x = cum.x + rx_;
                cum.y = cum.y + ry_;
                cum.z = cum.z + rz_;
              } else {
                cum.x = rx_;
                cum.y = ry_;
                cum.z = rz_;
              }
              if (pcg_mode != PT_PCG_SAMPLE) {
                vstate = s_to;
                if (vbase >= hperiod)  // which guess would have been right for this sample: its upper neighbour's draws, or its predecessor's?
                  pscore += (int)(((unsigned)(hist >> (8 * (hperiod - 1))) & 0xffu) == (s_draws & 0xffu)) - (int)(((unsigned)hist & 0xffu) == (s_draws & 0xffu));
                hist = (hist << 8) | (uint64_t)(s_draws & 0xffu);
              }
              prays += s_rays;
              vbase++;
#ifdef PT_DEBUG_TIME
              if (tracing && in_unit && lane == leader && (leader / L) < 64 && vbase <= 80)
                pt_trace[PT_TRACE_LEN + (leader / L) * 80 + (vbase - 1)] = ((unsigned long long)ulog_rounds << 32) | ((unsigned long long)s_draws << 16) | ((unsigned long long)(leader / L) << 8) | 0xEEULL;
#endif
            }
          }
          mode = 2;
#ifdef PT_DEBUG_TIME
          {  // speculation statistics: pixel-rounds, samples traced, samples kept
            const bool lead = in_unit && lane == leader && pix >= 0;
            unsigned long long r4 = lead ? 1ULL : 0ULL, r5 = lead ? (unsigned long long)dbg_fin : 0ULL,
                               r6 = lead ? (unsigned long long)(vbase - dbg_vbase0) : 0ULL;
            for (int off = 32; off > 0; off >>= 1) {
              r4 += __shfl_down(r4, off, 64);
              r5 += __shfl_down(r5, off, 64);
              r6 += __shfl_down(r6, off, 64);
            }
            if (lane == 0) {
              unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
              wv[4] += r4;
              wv[5] += r5;
              wv[6] += r6;
            }
          }
#endif
          if (in_unit) {
            if (vbase >= nsamp) {
              if (lane == leader && pix >= 0) {  // imagetracer.py:99-104
                if (S > 0) {
                  const double k = 1.0 / (double)(S * S);
                  cum.x = cum.x * k;
                  cum.y = cum.y * k;
                  cum.z = cum.z * k;
                }
                store_pixel(a, pix, cum);
                nrays += prays;
              }
              pix = -1;  // this pixel is done (in every lane of it)
            } else if (seed_round()) {
              mode = 0;
            }
          }
        }
        PT_STAMP(3);
        if (!__any(mode == 0)) {
          // next unit for this wave, then its region's cone and survivor masks.  The sorted unit list is dealt out
          // to PT_UNIT_SHARDS shards (unit u belongs to shard u % shards: every shard the same mix of sizes) and a
          // workgroup pulls from shard blockIdx % shards only: a returning atomic on ONE head word saturates near 88
          // dequeues/us -- with thousands of waves pulling, queueing at the head costs more than a unit's work.  A
          // wave's first unit is its own rank in the shard (no atomic at all), later ones come from the shard's head,
          // one atomic by lane 0.
          unsigned uid = 0;
          const unsigned nsh = gridDim.x < PT_UNIT_SHARDS ? gridDim.x : PT_UNIT_SHARDS;  // (every shard needs a puller)
          const unsigned shard = blockIdx.x % nsh;
          if (first_unit) {
            uid = (blockIdx.x / nsh) * (PT_BLOCK / 64) + (threadIdx.x >> 6);
            first_unit = false;
          } else {
            const unsigned pullers = (gridDim.x - shard + nsh - 1) / nsh * (PT_BLOCK / 64);
#ifdef PT_DEBUG_TIME
            PT_VM_DRAIN();
            const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
#endif
            if (lane == 0) uid = pullers + (unsigned)atomicAdd(pt_queue(a) + PT_QUEUE_HEADS + 32 * shard, 1ULL);
#ifdef PT_DEBUG_TIME
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(uid) : "memory");
            lat_note(2, __builtin_amdgcn_s_memtime() - lt0);
#endif
          }
          uid = uid * nsh + shard;
          const int seq = (int)__builtin_amdgcn_readfirstlane((int)uid);
          PT_STAMP(6);
#ifdef PT_DEBUG_TIME
          tracing = seq == cold_args(a)->dbg_trace_unit;  // (or, below, the unit that starts at a given flagged pixel of a given region)
          if (LAT && tracing && lane == 0) {
            const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
            for (int q = 0; q < 3; ++q) dbg_q[q] = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane == 0 && ulog_seq >= 0 && ulog_seq < PT_UNITLOG_LEN) {  // close the log entry of the unit just finished
            pt_unitlog[ulog_seq * 8 + 1] = __builtin_amdgcn_s_memtime();
            pt_unitlog[ulog_seq * 8 + 2] = (unsigned long long)ulog_rounds | ((unsigned long long)ulog_iters << 32);
            pt_unitlog[ulog_seq * 8 + 4] = tsum[4] - ulog_t[0];
            pt_unitlog[ulog_seq * 8 + 5] = tsum[5] - ulog_t[1];
            pt_unitlog[ulog_seq * 8 + 6] = tsum[1] + tsum[2] - ulog_t[2];
            pt_unitlog[ulog_seq * 8 + 7] = tsum[0] - ulog_t[3];
          }
          ulog_t[0] = tsum[4];
          ulog_t[1] = tsum[5];
          ulog_t[2] = tsum[1] + tsum[2];
          ulog_t[3] = tsum[0];
          ulog_seq = seq;
          ulog_rounds = 0;
          ulog_iters = 0;
#endif
          pt_kargs ca = cold_args(a);
          if (seq >= n_units) break;
#ifdef PT_DEBUG_TIME
          if (lane == 0) pt_dbg_wave[(size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8 + 7] += 1ULL;
#endif
#ifdef PT_DEBUG_TIME
          const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
          PT_VM_DRAIN();
          const unsigned long long lt1 = __builtin_amdgcn_s_memtime();
          const int4 unit = ca->units[seq];
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : "v"(unit.x) : "memory");
          lat_note(0, __builtin_amdgcn_s_memtime() - lt1);
          lat_note(1, lt1 - lt0);
#else
          const int4 unit = ca->units[seq];
#endif
          const int region = unit.x, first = unit.y & 0xff, count = (unit.y >> 8) & 0xff;
#ifdef PT_DEBUG_TIME
          if (cold_args(a)->dbg_trace_unit <= -2) tracing = (-2 - cold_args(a)->dbg_trace_unit) == region * 64 + first;  // (unit numbers vary from frame to frame)
#endif
          const unsigned long long todo = (unsigned long long)(unsigned)unit.z | ((unsigned long long)(unsigned)unit.w << 32);  // the region's flagged pixels
          const int ry = region / regions_x, rx = region - ry * regions_x;
          const int gr0 = global_row(a, ry * PT_REGION);
          const int gr1 = global_row(a, (ry * PT_REGION + PT_REGION - 1 < rows_local) ? ry * PT_REGION + PT_REGION - 1 : rows_local - 1);
          const TileCone tc = tile_cone(a, rx * PT_REGION, (rx * PT_REGION + PT_REGION < W) ? rx * PT_REGION + PT_REGION : W, gr0, gr1);
          __builtin_amdgcn_wave_barrier();
          for (int p = 0; p < npass; ++p) {
            const int slot = p * 64 + lane;
            bool keep = false;
            if (slot < a.n_shapes) keep = slot >= a.n_spheres || cone_keeps(tc, a.bounds[slot]);  // planes: always
            const unsigned long long m = __ballot(keep);
            if (lane == 0) pt_lds_masks[mbase + p] = m;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // lanes [p * L, (p + 1) * L) take the unit's p-th pixel = flagged pixel `first + p` of the region
          L = 64 / count;
          if (L > nsamp) L = nsamp;
#ifdef PT_DEBUG_TIME
          if (lane == 0 && seq < PT_UNITLOG_LEN) {
            pt_unitlog[seq * 8 + 0] = __builtin_amdgcn_s_memtime();
            pt_unitlog[seq * 8 + 3] = (unsigned long long)count | ((unsigned long long)L << 8) | ((unsigned long long)(blockIdx.x & 0x3ff) << 16) |
                                      ((unsigned long long)first << 26) | ((unsigned long long)region << 32);
          }
#endif
          const int pidx = lane / L;
          in_unit = pidx < count;
          leader = in_unit ? pidx * L : lane;
          jlane = lane - leader;
          mode = 2;
          pix = -1;
          if (in_unit) {
            const int bit = nth_set_bit(todo, first + pidx);
            const int lrow = ry * PT_REGION + (bit >> 3);
            col = rx * PT_REGION + (bit & 7);
            pix = (long long)lrow * W + col;
            grow = global_row(a, lrow);  // (pixel_coords would divide the 64-bit index by W to find what is known here)
            gpix = (unsigned long long)grow * ca->W + col;
            if (pcg_mode != PT_PCG_SAMPLE) {
              pcg_seed(pcg, ca->s0, ca->q0 + gpix);
              vstate = pcg.state;
            }
            hist = 0x0101010101010101ULL * (uint64_t)(ca->spec_draws & 0xff);
            pscore = 0;
            vbase = 0;
            prays = 0;
            cum.x = 0.0;
            cum.y = 0.0;
            cum.z = 0.0;
            if (seed_round()) mode = 0;
          }
        }
      }
    } else {
      for (;;) {
        const bool need = mode == 2 && !exhausted;
        if (!__any(need)) break;
        // FLAGGED: the queue deals out the frame's pixel indices and a lane that draws a settled pixel draws again -- or, where
        // flagged pixels are few (under a quarter of the frame), the FLAGGED pixels themselves, from pt_unit_scatter's list of
        // one-pixel units: skipping cost a returning atomic per wave and settled pixel (~1 M atomics on one word for a 4K
        // frame with 3 % flagged pixels, at ~90 per us: C3 at 4K 10.5 -> 3.0 ms).  Frames FULL of flagged pixels keep the
        // row-major order: the list's order (fullest regions first) costs them 15 - 30 % (profiles/r05_queue_dealing.txt).
        const long long np = next_pixel(a, need, deal_units ? (long long)n_flagged : a.npix);
        if (need && np >= 0) {
          bool take = true;
          long long p = np;
          if (FLAGGED) {
            if (deal_units) {
              const int4 unit = cold_args(a)->units[np];
              const unsigned long long todo = (unsigned long long)(unsigned)unit.z | ((unsigned long long)(unsigned)unit.w << 32);
              const int bit = nth_set_bit(todo, unit.y & 0xff);
              const int ry = unit.x / regions_x, rx = unit.x - ry * regions_x;
              p = (long long)(ry * PT_REGION + (bit >> 3)) * W + (rx * PT_REGION + (bit & 7));
            } else {  // pixels the first pass settled are not this kernel's (pt_tile_kernel: rmask)
              const int lr = (int)(np / W), c0 = (int)(np - (long long)lr * W);
              const unsigned long long m = cold_args(a)->region_mask[(lr / PT_REGION) * regions_x + c0 / PT_REGION];
              take = ((m >> ((lr % PT_REGION) * PT_REGION + (c0 % PT_REGION))) & 1ULL) != 0ULL;
            }
          }
          if (take) {
            pix = p;
            mode = 0;
          }
        }
        exhausted = __any(need && np < 0);
        if (!FLAGGED || deal_units) break;  // (dealing indices: lanes that drew a settled pixel draw again)
      }
      if (!__any(mode != 2)) break;
    }

    PT_STAMP(0);
#ifdef PT_DEBUG_TIME
    ulog_iters++;
#endif
    const int n_start = __popcll(__ballot(mode == 0));
    const int n_path = __popcll(__ballot(mode == 1));
    if (n_start == 0 && n_path == 0) continue;  // TILED: nothing in flight, the round / unit logic above decides
    const bool do_p = n_start > 0 && n_path < cold_args(a)->p_max_path;
    const bool do_s = n_path >= cold_args(a)->s_min_path || (n_path > 0 && !do_p);

    // ---- queries: primary rays against the region's survivors, scattered rays against everything ----
    const bool prim = do_p && mode == 0;
    const bool scat = do_s && mode == 1;
    double best_t = INFINITY;
    int hit = -1;
    if (do_p) {
      if (prim) start_sample();
      PT_STAMP(1);
      double tp = INFINITY;
      int hp;
      if (TILED)
        hp = ortho ? world_query_tile<false, false, false>(a, ray, mbase, npass, tp, prim)
                   : world_query_tile<false, false, true>(a, ray, mbase, npass, tp, prim);
      else
        hp = world_query<false, false>(a, ray, INFINITY, tp, prim);
      if (prim) {
        hit = hp;
        best_t = tp;
      }
      PT_STAMP(2);
    }
    if (do_s) {
      double ts;
      const int hs = LAT ? world_query_lanes<false, LEAN>(a, ray, INFINITY, ts, scat, diag_lds) : world_query<false, false>(a, ray, INFINITY, ts, scat);
      if (scat) {
        hit = hs;
        best_t = ts;
      }
      PT_STAMP(4);
#ifdef PT_DEBUG_TIME
      if (LAT && tracing && lane == 0) {  // the traced unit: this query's prefilter cycles (8), walk cycles (9), walk turns (10)
        const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
        for (int q = 0; q < 3; ++q) {
          const unsigned long long now = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (trace_n < PT_TRACE_LEN) pt_trace[trace_n] = ((now - dbg_q[q]) << 16) | (unsigned long long)(8 + q);
          dbg_q[q] = now;
          trace_n++;
        }
      } else if (LAT && tracing) {
        trace_n += 3;
      }
#endif
    }

    // ---- shade the hit, then unwind: deliver radiance up the stack / scatter the next child, until
    //      this lane has a ray that needs a query (mode 1) or its sample is complete (mode 0 / 2 / 3) ----
    const bool work = prim || scat;
    if (work) {
      if (TILED)
        srays++;
      else if (FLAGGED) {
        qrays++;
        if (exhausted) qtail++;
      }
      else
        nrays++;
      shade(hit, best_t);
      mode = 1;
    }
    bool unwinding = work;
    while (unwinding) {
      // (a lane with a child to scatter leaves the loop; the scatter itself -- two draws, sin / cos, two square roots -- runs
      //  ONCE behind the loop for every lane of the wave that spawns in this step, whether its node was pushed by shade() or
      //  reached by a child's return: inside the loop the wave ran it once per turn that any lane spawned in)
      if (spawn) break;
      if (sp == 0) {
        finish_sample();  // mode 0 (next sample), 2 (pixel done) or 3 (TILED: wait for the round's end)
        break;
      }
      // a child of frame sp-1 returned `ret` (render.py:135-137)
      const int fs = sp - 1;
      FrameRef fr = {nullptr, 0};
      if constexpr (LDSF == 2) fr = frame_ref_split(w, fs);
      auto fget = [&](int field) -> double {
        if constexpr (LDSF == 2)
          return fr.p[(size_t)field * fr.fstride];
        else
          return ws_get<LDSF>(w, fs, field);
      };
      auto fput = [&](int field, double v) {
        if constexpr (LDSF == 2)
          fr.p[(size_t)field * fr.fstride] = v;
        else
          ws_put<LDSF>(w, fs, field, v);
      };
      const V3 hc = {fget(0), fget(1), fget(2)};
      V3 fc = {0.0, 0.0, 0.0};
      int done = 0;
      if (N > 1) {
        fc.x = fget(6);
        fc.y = fget(7);
        fc.z = fget(8);
        done = (int)fget(9);
      }
      fc.x = fc.x + hc.x * ret.x;
      fc.y = fc.y + hc.y * ret.y;
      fc.z = fc.z + hc.z * ret.z;
      done++;
      if (done < N) {
        fput(6, fc.x);
        fput(7, fc.y);
        fput(8, fc.z);
        fput(9, (double)done);
        f_wp = {fget(10), fget(11), fget(12)};
        f_n = {fget(13), fget(14), fget(15)};
        f_in = {fget(16), fget(17), fget(18)};
        f_brdf = (int)fget(19);
        spawn = true;
        continue;
      }
      // render.py:139
      ret.x = fget(3) + fc.x * invN;
      ret.y = fget(4) + fc.y * invN;
      ret.z = fget(5) + fc.z * invN;
      sp = fs;
    }
    if constexpr (FLAGGED) {
      // A lane walks its pixel's rays one after the other: a tree of 1 111 rays (the CLI's N = 10, D = 3) is 1 111 turns of
      // this loop, ~6 us each, and once the pixel queue has run dry nothing fills the lanes that finish: the frame waits for
      // its heaviest pixels while most of the chip idles.  The tree kernel behind this one traces a node's children at the
      // same time, so a pixel's remaining rays take a sixth of the time there.  A lane therefore HANDS ITS PIXEL OVER, at the
      // point where its next ray would be scattered: the node stack, the generator, the sums and the ray count go into a
      // record (PT_HANDOVER_HEADER + 20 doubles per node, in the tree kernel's node layout) and the pixel becomes a unit of the
      // tree kernel (PT_Q_HEAVY), which goes on exactly where the lane stopped -- nothing is traced twice, and every draw
      // happens at the state the sequential program has there.  When: the queue dry and `q_few` or fewer lanes of the wave
      // still hold a pixel (the plan's default); or the pixel has traced q_budget rays, or q_tail rays since the queue ran
      // dry (measurement switches).  A full record table leaves the pixel with its lane.
      const bool few = q_few > 0 && exhausted && __popcll(__ballot(mode != 2)) <= q_few;  // (wave-uniform)
      if (spawn && !q_full && (few || (q_budget > 0 && qrays >= (unsigned)q_budget) || (q_tail > 0 && qtail >= (unsigned)q_tail))) {
        pt_kargs ca = cold_args(a);
        const unsigned long long k = atomicAdd(pt_queue(a) + PT_Q_HEAVY, 1ULL);
        q_full = k >= (unsigned long long)ca->handover_cap;  // (the lane keeps this pixel and asks no more)
        if (!q_full) {
          const int lr = (int)(pix / W), c0 = (int)(pix - (long long)lr * W);
          const int region = (lr / PT_REGION) * regions_x + c0 / PT_REGION;
          const unsigned long long m = ca->region_mask[region];
          const int bit = (lr % PT_REGION) * PT_REGION + (c0 % PT_REGION);
          const int first = __popcll(m & ((1ULL << bit) - 1ULL));
          ca->units_handed[k] = make_int4(region, first | (1 << 8) | (1 << 16), (int)(unsigned)m, (int)(unsigned)(m >> 32));
          double *rec = ca->handover + (size_t)k * (size_t)(PT_HANDOVER_HEADER + 20 * (D > 1 ? D : 1));
          rec[0] = __longlong_as_double((long long)pcg.state);
          rec[1] = __longlong_as_double((long long)pcg.inc);
          rec[2] = (double)samp;
          rec[3] = (double)sp;
          rec[4] = (double)qrays;
          rec[5] = cum.x;
          rec[6] = cum.y;
          rec[7] = cum.z;
          for (int d = 0; d < sp; ++d) {
            FrameRef fr = {nullptr, 0};
            if constexpr (LDSF == 2) fr = frame_ref_split(w, d);
            auto fget = [&](int field) -> double {
              if constexpr (LDSF == 2)
                return fr.p[(size_t)field * fr.fstride];
              else
                return ws_get<LDSF>(w, d, field);
            };
            double *t = rec + PT_HANDOVER_HEADER + 20 * d;
            for (int f = 0; f < 9; ++f) t[f] = fget(f);           // hit_color, emitted radiance, the children's sum so far
            for (int f = 0; f < 9; ++f) t[9 + f] = fget(10 + f);  // hit point, normal, incoming direction
            t[18] = fget(19);                                     // BRDF
            // children traced: the tree kernel counts the one whose subtree is being walked (every node but the innermost)
            t[19] = fget(9) + (d < sp - 1 ? 1.0 : 0.0);
          }
          spawn = false;
          qrays = 0;
          qtail = 0;
          cum.x = 0.0;
          cum.y = 0.0;
          cum.z = 0.0;
          samp = 0;
          sp = 0;
          mode = 2;
        }
      }
    }
    if (spawn) {
      // scatter_ray (materials.py:132-152, 175-196); the child is at depth sp <= max_depth (a hit whose children would lie
      // beyond it never pushes a frame: shade_hit), so it is queried at the next S-step (mode 1)
      if (INL)
        ray = scatter_ray<true>(f_brdf, pcg, f_in, f_wp, f_n);
      else
        scatter_ray_call(f_brdf, &pcg, &f_in, &f_wp, &f_n, &ray);
      spawn = false;
    }
    PT_STAMP(5);
  }
#ifdef PT_DEBUG_TIME
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
  pt_dbg_flush();
#endif
  add_ray_count(a, nrays, TILED ? 0 : cold_args(a)->count_base);
}

// every pixel of the frame, one lane per pixel, pixels from one queue, every shape tested by the wave-uniform loop, the frame
// stack in HBM: worlds the tiled kernels do not take (no shape at all) and the measurement switch PTRACE_CULL=0 -- the
// brute-force device path the culled kernels are checked against
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_WAVES_PATH, 8))) void pt_path_kernel(const PtKArgs a) {
  path_trace<false, 0, false, false, 0, false>(a);
}
// The flagged pixels of a perspective frame of num_of_rays > 1 when the device chose this kernel (PT_Q_CHOICE): a lane per
// pixel from one queue, the scattered rays on per-lane candidate lists (world_query_lanes) and everything inline, like the
// second pass by regions: it runs with 20 doubles per depth and lane of frame stack in LDS -- one workgroup per CU at the CLI's
// D = 3, one wave per SIMD --, so registers are no object and a step's latency is what counts
// (HOME 1: the whole stack in LDS; 2: only the deepest slot in LDS -- see frame_ref_split; two waves per SIMD)
template <int LEAN, int HOME = 1>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(HOME == 1 ? 1 : 2, 2))) void pt_path_flagged_kernel(const PtKArgs a) {
  static_assert(HOME == 1 || HOME == 2, "the one-queue kernel's stack: all in LDS, or split");
  path_trace<false, HOME, true, false, LEAN, true>(a);
}
// second pass behind pt_tile_kernel<PATHTRACER> (perspective camera): the flagged pixels, by region
#ifndef PT_WAVES_REGIONS
#define PT_WAVES_REGIONS 2
#endif
template <bool LDSF, bool SLDS = false, int LEAN = 0>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_WAVES_REGIONS, 8))) void pt_path_regions_kernel(const PtKArgs a) {
  path_trace<true, LDSF, true, SLDS, LEAN>(a);
}
