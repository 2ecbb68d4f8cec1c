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
// LAT: the second pass is built for few waves per SIMD; its time is set by chains of dependent steps: scattered rays walk
// per-lane candidate lists (world_query_lanes), registers are no object.  HitRecord, scatter and their sin / cos are inline in
// every variant (the one-lane-per-pixel kernel kept them behind calls until round 6: 256 - 544 B of stack for the callees'
// frames); a sphere's atan2 / acos stay behind calls everywhere (hit_details).
// ---- path_trace in parts ---------------------------------------------------------------------------------------------------
// The state machine is ONE loop (path_trace, at the end) over named phases; each phase is a function over the same three
// records: PathCtx (where the frame stack lives), PathK (what is constant for the launch or the wave) and PathLane (this
// lane's state).  Everything is force-inlined and the records are scalars only, so the compiler sees what it saw when the
// phases were lambdas of one 850-line function (round 5) -- same registers, same code --; a reader sees which phase touches
// what.  The phases, in loop order:
//   path_commit_round   TILED: the samples a round traced are validated and added up in sample order
//   path_next_unit      TILED: the wave's next unit -- region, cone, survivor masks, lanes shared out to its pixels
//   path_next_pixels    one queue: idle lanes draw pixels (FLAGGED: only flagged ones, or the flagged-pixel list itself)
//   path_start_sample   P step: jitter draws and the sample's primary ray (+ path_seed_round: where a round's samples start)
//   (the two queries: primary rays against the region's survivors, scattered rays against everything -- in path_trace itself)
//   path_shade          render.py:103-139 up to the recursion: a value to deliver, or a frame pushed and a child to scatter
//   path_unwind         deliver radiance up the frame stack until the lane has a child to scatter or its sample is complete
//   path_hand_over      FLAGGED: a lane gives its pixel, where it stands, to the tree kernel behind this one
struct PathK {
  int S, nsamp, N, W, rows_local, npass, D, rr, diag_lds, pcg_mode, scene_lds;
  int lane, mbase, regions_x, hperiod;
  bool ortho;
  double invN;
  PcgJump j2n;  // the scatter draws of N children beyond max_depth, as one jump
  // FLAGGED: when a lane hands its pixel over (path_hand_over), and how the queue deals (path_next_pixels)
  int q_budget, q_tail, q_few, n_flagged;
  bool deal_units;
  int n_units;  // TILED: units of the frame (written by pt_unit_scatter before this kernel started)
};
struct PathLane {
  // mode 0: starts a sample at the next P step; 1: inside a path (S steps); 2: nothing to do; 3 (TILED): sample finished,
  // waits for the end of the round
  int mode;
  long long pix;
  Pcg pcg;
  int samp, sp, col, grow;
  V3 cum;
  Ray ray;
  // what path_shade hands to path_unwind of the same step: a value to deliver, or a child to spawn (and the node it leaves from)
  V3 ret;
  bool spawn;
  V3 f_wp, f_n, f_in;
  int f_brdf;
  // TILED: the unit (wave-uniform) and this lane's place in it
  int L;               // lanes per pixel
  int leader, jlane;   // first lane of this lane's pixel; this lane's sample slot in a round
  bool in_unit;        // the lane belongs to a pixel of the unit
  int vbase;           // samples of the pixel validated so far (same in all lanes of the pixel)
  uint64_t vstate;     // PT_PCG_PIXEL: generator state behind the last validated sample
  // PT_PCG_PIXEL: what the pixel's last eight validated samples drew, a byte each, the latest in the low byte.  The guess
  // for sample k is what sample k - S drew -- its neighbour one row up in the S x S grid of strata (imagetracer.py:86-93):
  // a pixel across an edge repeats its pattern of short and long paths row after row, where "what the last sample drew"
  // is wrong twice per row.  (S > 8: the sample before it.)
  uint64_t hist;
  int pscore;          // ... how much more often the upper neighbour was the better guess than the predecessor (per pixel)
  uint64_t st_start;   // state this lane's sample started from
  unsigned srays, prays;   // rays of the current sample; of the pixel's validated samples
  unsigned long long gpix;  // global pixel index (seeds)
  bool first_unit;     // (wave-uniform)
  // one queue
  bool exhausted;      // the global queue is empty
  unsigned qtail;      // FLAGGED: rays of the lane's pixel since the queue ran dry
  bool q_full;         // FLAGGED: the record table was full when this lane last asked
  unsigned qrays;      // FLAGGED: rays of the lane's pixel so far (counted when the pixel is done: a pixel over budget is the tree kernel's)
  unsigned long long nrays;
};

// section timers and the step trace of -DPT_DEBUG_TIME builds (tools/dbgtime.py, dbgunits.py, dbgdraws.py); nothing otherwise
#ifdef PT_DEBUG_TIME
struct PathDbg {
  // section sums for every wave, plus a step-by-step trace of the wave that drew the first (fullest) unit
  unsigned long long tsum[8], tprev;
  bool tracing;
  int trace_n;
  int ulog_seq, ulog_rounds, ulog_iters;
  unsigned long long ulog_t[4];
  unsigned long long dbg_q[3];
};
#define PT_STAMP(k)                                                                          \
  do {                                                                                       \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                              \
    dbg.tsum[k] += tn - dbg.tprev;                                                           \
    const unsigned long long np_ = (unsigned long long)__popcll(__ballot(s.mode == 1));     \
    if (dbg.tracing && K.lane == 0 && dbg.trace_n < PT_TRACE_LEN)                            \
      pt_trace[dbg.trace_n] = ((tn - dbg.tprev) << 16) | (np_ << 8) | (k);                   \
    if (dbg.tracing) dbg.trace_n++;                                                          \
    dbg.tprev = tn;                                                                          \
  } while (0)
#else
struct PathDbg {};
#define PT_STAMP(k) do { } while (0)
#endif

// one frame of the stack, wherever it lives (LDSF 0: HBM, 1: LDS, 2: split -- one generic pointer picked per lane)
template <int LDSF>
struct FrameAt {
  const PathCtx &w;
  int slot;
  FrameRef fr;
  __device__ __forceinline__ FrameAt(const PathCtx &w_, int slot_) : w(w_), slot(slot_), fr{nullptr, 0} {
    if constexpr (LDSF == 2) fr = frame_ref_split(w, slot);
  }
  __device__ __forceinline__ double get(int field) const {
    if constexpr (LDSF == 2)
      return fr.p[(size_t)field * fr.fstride];
    else
      return ws_get<LDSF>(w, slot, field);
  }
  __device__ __forceinline__ void put(int field, double v) const {
    if constexpr (LDSF == 2)
      fr.p[(size_t)field * fr.fstride] = v;
    else
      ws_put<LDSF>(w, slot, field, v);
  }
};

// the tables a latency-bound kernel keeps in LDS: scale+translate records (world_query_lanes fetches them by lane-private
// index), the shapes' records (shading gathers ~20 values of the hit shape per lane, and a gather from LDS costs a fraction
// of one through the vector memory path), the grid's occupancy bits (the cell walk reads one per step).  -> scene_lds
template <bool LAT, bool SLDS>
PT_DEV int path_stage_tables(const PtKArgs &a, int diag_lds) {
  if (LAT && diag_lds >= 0) {
    const unsigned long long *src = (const unsigned long long *)a.diag;
    for (int k = threadIdx.x; k < a.n_diag * 8; k += PT_BLOCK) pt_lds_masks[diag_lds + k] = src[k];
    __syncthreads();
  }
  int scene_lds = 0;
  if (SLDS) {  // recs[] (128 B each) then aux[] (256 B each), as 8-byte words
    scene_lds = cold_args(a)->scene_lds;
    const unsigned long long *src = (const unsigned long long *)a.recs;
    for (int k = threadIdx.x; k < a.n_shapes * 16; k += PT_BLOCK) pt_lds_masks[scene_lds + k] = src[k];
    src = (const unsigned long long *)a.aux;
    for (int k = threadIdx.x; k < a.n_shapes * 32; k += PT_BLOCK) pt_lds_masks[scene_lds + a.n_shapes * 16 + k] = src[k];
    __syncthreads();
  }
  if (LAT) {
    pt_kargs c = cold_args(a);
    const int occ_lds = c->grid_occ_lds;
    if (occ_lds >= 0) {
      const int nwords = (c->grid_res[0] * c->grid_res[1] * c->grid_res[2] + 31) / 32;
      unsigned *dst = (unsigned *)pt_lds_masks;
      for (int k = threadIdx.x; k < nwords; k += PT_BLOCK) dst[occ_lds + k] = c->grid_occ[k];
      __syncthreads();
    }
  }
  return scene_lds;
}

// pixel coordinates + seeds + the sample's primary ray (imagetracer.py:86-97)
template <bool TILED>
PT_DEV void path_start_sample(const PtKArgs &a, const PathK &K, PathLane &s) {
  pt_kargs c = cold_args(a);
  if (!TILED) {
    if (s.samp == 0) {
      pixel_coords(a, s.pix, s.col, s.grow);
      if (c->pcg_mode == PT_PCG_PIXEL) pcg_seed(s.pcg, c->s0, c->q0 + ((unsigned long long)s.grow * c->W + s.col));
    }
    if (c->pcg_mode == PT_PCG_SAMPLE)
      pcg_seed(s.pcg, c->s0, c->q0 + ((unsigned long long)s.grow * c->W + s.col) * (unsigned)K.nsamp + (unsigned)s.samp);
  }
  double up = 0.5, vp = 0.5;
  if (K.S > 0) {
    const int sr = s.samp / K.S, sc = s.samp - sr * K.S;
    up = ((double)sc + pcg_float(s.pcg)) / (double)K.S;
    vp = ((double)sr + pcg_float(s.pcg)) / (double)K.S;
  }
  s.ray = primary_ray(a, s.col, s.grow, up, vp);
}

// TILED: the sample of the round this lane traces (vbase + jlane) and the generator state it starts from.
// -> whether that sample exists (vbase + jlane < nsamp)
PT_DEV bool path_seed_round(const PtKArgs &a, const PathK &K, PathLane &s) {
  pt_kargs c = cold_args(a);
  if (K.pcg_mode == PT_PCG_SAMPLE) {
    s.samp = s.vbase + s.jlane;
    if (s.samp < K.nsamp) pcg_seed(s.pcg, c->s0, c->q0 + s.gpix * (unsigned)K.nsamp + (unsigned)s.samp);
  } else {
    // (sample vbase + i: what its upper neighbour vbase + i - period drew, if the pixel has got that far and that guess
    //  has been the better one so far; else what the last validated sample drew)
    const int period = K.hperiod;
    unsigned ahead = (unsigned)s.jlane * ((unsigned)s.hist & 0xffu);
    if (s.pscore > 0) {
      ahead = 0;
      for (int i = 0; i < s.jlane; ++i)
        ahead += (unsigned)(s.hist >> (s.vbase + (i % period) >= period ? 8 * (period - 1 - (i % period)) : 0)) & 0xffu;
    }
    s.samp = s.vbase + s.jlane;
    s.pcg.state = pcg_advance(s.vstate, s.pcg.inc, ahead);
  }
  s.pcg.n = 0;
  s.st_start = s.pcg.state;
  s.srays = 0;
  return s.samp < K.nsamp;
}

// render.py:103-139 up to (not including) the recursion: sets `ret`, or pushes frame `sp` and asks for child 0 (`spawn`).
// `ray` is the ray that was queried, at depth `sp`.
template <int LDSF, typename RP, typename AP>
PT_DEV void path_shade_hit(const PtKArgs &a, const PathCtx &w, const PathK &K, PathLane &s, RP rec, AP ax, double best_t) {
  V3 hc, em;
  double lum;
  Hit h;
  h.u = 0.0;
  h.v = 0.0;
  const bool uv = ax->needs_uv != 0;
  bool details = false;
  if (uv) {
    hit_details<true>(rec, ax, s.ray, best_t, h, true);
    details = true;
  }
  hc = brdf_pigment(a, ax, h.u, h.v);
  em = emitted_pigment(a, ax, h.u, h.v);
  lum = max2(max2(hc.x, hc.y), hc.z);
  if (s.sp >= K.rr) {  // render.py:116-123
    const double q = max2(0.05, 1.0 - lum);
    if (pcg_float(s.pcg) > q) {
      const double k = 1.0 / (1.0 - q);
      hc.x = hc.x * k;
      hc.y = hc.y * k;
      hc.z = hc.z * k;
    } else {
      s.ret = em;
      return;
    }
  }
  if (!(lum > 0.0)) {  // render.py:139 with cum_radiance = 0
    s.ret.x = em.x + 0.0 * K.invN;
    s.ret.y = em.y + 0.0 * K.invN;
    s.ret.z = em.z + 0.0 * K.invN;
    return;
  }
  if (s.sp + 1 > K.D) {
    // Every child of this hit would be beyond max_depth: the reference still calls scatter_ray for each
    // (consuming its draws: 2 for a diffuse BRDF, none for a mirror) and each child returns black at
    // render.py:100-101 without a world query.  No ray, no frame, no geometry is needed: advance the
    // generator and accumulate hit_color * 0 exactly as render.py:135-139 does.
    // (Round 5: the N x 2 draws are ONE jump of the generator -- nobody reads their outputs --, and the N additions of
    //  hit_color * 0 are one: 0 + z + z + ... = 0 + z for z = +-0 and for a NaN, whatever N >= 1.  Exact, and a fifth of
    //  what a leaf hit used to cost: 2 N dependent 64-bit multiply-adds and 3 N dependent additions.)
    const bool diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
    if (diffuse) pcg_jump(s.pcg, K.j2n, 2u * (unsigned)K.N);
    const V3 fc = {0.0 + hc.x * 0.0, 0.0 + hc.y * 0.0, 0.0 + hc.z * 0.0};
    s.ret.x = em.x + fc.x * K.invN;
    s.ret.y = em.y + fc.y * K.invN;
    s.ret.z = em.z + fc.z * K.invN;
    return;
  }
  // render.py:126-137: push the frame, child 0 is scattered at the next S-step
  if (!details) hit_details<true>(rec, ax, s.ray, best_t, h, false);
  const FrameAt<LDSF> f(w, s.sp);
  f.put(0, hc.x);
  f.put(1, hc.y);
  f.put(2, hc.z);
  f.put(3, em.x);
  f.put(4, em.y);
  f.put(5, em.z);
  if (K.N > 1) {
    f.put(6, 0.0);
    f.put(7, 0.0);
    f.put(8, 0.0);
    f.put(9, 0.0);
    f.put(10, h.wp.x);
    f.put(11, h.wp.y);
    f.put(12, h.wp.z);
    f.put(13, h.n.x);
    f.put(14, h.n.y);
    f.put(15, h.n.z);
    f.put(16, s.ray.d.x);
    f.put(17, s.ray.d.y);
    f.put(18, s.ray.d.z);
    f.put(19, (double)ax->brdf_kind);
  }
  s.f_wp = h.wp;
  s.f_n = h.n;
  s.f_in = s.ray.d;
  s.f_brdf = ax->brdf_kind;
  s.sp++;
  s.spawn = true;
}
template <int LDSF, bool SLDS>
PT_DEV void path_shade(const PtKArgs &a, const PathCtx &w, const PathK &K, PathLane &s, int hit, double best_t) {
  s.spawn = false;
  if (hit < 0) {  // render.py:103-105
    pt_kargs c = cold_args(a);
    s.ret.x = c->bg[0];
    s.ret.y = c->bg[1];
    s.ret.z = c->bg[2];
    return;
  }
  if constexpr (SLDS)
    path_shade_hit<LDSF>(a, w, K, s, (pt_lds_rec)(const void *)(pt_lds_f64 + K.scene_lds) + hit,
                              (pt_lds_aux)(const void *)(pt_lds_f64 + K.scene_lds + a.n_shapes * 16) + hit, best_t);
  else
    path_shade_hit<LDSF>(a, w, K, s, a.recs + hit, cold_args(a)->aux + hit, best_t);
}

// the primary call returned `ret`: one sample done (imagetracer.py:94-104)
template <bool TILED, bool FLAGGED>
PT_DEV void path_finish_sample(const PtKArgs &a, const PathK &K, PathLane &s) {
  if (TILED) {  // the radiance stays in `ret` until the round is validated
    s.mode = 3;
    return;
  }
  if (K.S > 0) {
    s.cum.x = s.cum.x + s.ret.x;
    s.cum.y = s.cum.y + s.ret.y;
    s.cum.z = s.cum.z + s.ret.z;
  } else {
    s.cum = s.ret;
  }
  s.mode = 0;
  if (++s.samp == K.nsamp) {
    if (K.S > 0) {
      const double k = 1.0 / (double)(K.S * K.S);
      s.cum.x = s.cum.x * k;
      s.cum.y = s.cum.y * k;
      s.cum.z = s.cum.z * k;
    }
    store_pixel(a, s.pix, s.cum);
    if (FLAGGED) {
      s.nrays += s.qrays;
      s.qrays = 0;
      s.qtail = 0;
    }
    s.cum.x = 0.0;
    s.cum.y = 0.0;
    s.cum.z = 0.0;
    s.samp = 0;
    s.mode = 2;
  }
}

// ---- TILED, end of a round: validate the pixel's samples in order, add them up in order ----
// (every lane of a pixel runs the same loop over the pixel's L lanes and ends with the same vbase / vstate / hist; only the
//  values in the leader are used for the pixel's result)
template <int LDSF>
PT_DEV void path_commit_round(const PtKArgs &a, const PathCtx &w, const PathK &K, PathLane &s, PathDbg &dbg) {
  const int lane = K.lane;
  const bool fin = s.mode == 3;
  bool chain = true;
#ifdef PT_DEBUG_TIME
  dbg.ulog_rounds++;
  const int dbg_vbase0 = s.vbase;
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
    pt_lds_f64[me] = s.ret.x;
    pt_lds_f64[me + PT_BLOCK] = s.ret.y;
    pt_lds_f64[me + 2 * PT_BLOCK] = s.ret.z;
    // (fin | sample index, 23 bits: S <= 1024 | the draws' low byte, all that `hist` keeps | rays of the sample)
    pt_lds_masks[me + 3 * PT_BLOCK] = (unsigned long long)(fin ? 1u : 0u) | ((unsigned long long)((unsigned)s.samp & 0x7fffffu) << 1) |
                                      ((unsigned long long)(s.pcg.n & 0xffu) << 24) | ((unsigned long long)s.srays << 32);
    if (K.pcg_mode != PT_PCG_SAMPLE) {
      pt_lds_masks[me + 4 * PT_BLOCK] = s.st_start;
      pt_lds_masks[me + 5 * PT_BLOCK] = s.pcg.state;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  for (int jj = 0; jj < s.L; ++jj) {
    const int src = (s.leader + jj) & 63;
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
      if (K.pcg_mode != PT_PCG_SAMPLE) {
        s_from = pt_lds_masks[at + 4 * PT_BLOCK];
        s_to = pt_lds_masks[at + 5 * PT_BLOCK];
      }
    } else {
      if (K.pcg_mode != PT_PCG_SAMPLE) {
        s_from = __shfl((unsigned long long)s.st_start, src, 64);
        s_to = __shfl((unsigned long long)s.pcg.state, src, 64);
        s_draws = (unsigned)__shfl((int)s.pcg.n, src, 64);
        s_samp = __shfl(s.samp, src, 64);
      }
      s_fin = __shfl((int)fin, src, 64);
      s_rays = (unsigned)__shfl((int)s.srays, src, 64);
      rx_ = __shfl(s.ret.x, src, 64);
      ry_ = __shfl(s.ret.y, src, 64);
      rz_ = __shfl(s.ret.z, src, 64);
    }
    if (K.pcg_mode == PT_PCG_SAMPLE) {
      chain = chain && s_fin != 0;
    } else {
      // the lane's sample counts iff it is the NEXT one and it started from the state the sequential program is in
      chain = s_fin != 0 && s_samp == s.vbase && s_from == s.vstate;
    }
#ifdef PT_DEBUG_TIME
    dbg_fin += s_fin;
#endif
    if (chain) {
      if (K.S > 0) {  // imagetracer.py:97
        s.cum.x = s.cum.x + rx_;
        s.cum.y = s.cum.y + ry_;
        s.cum.z = s.cum.z + rz_;
      } else {
        s.cum.x = rx_;
        s.cum.y = ry_;
        s.cum.z = rz_;
      }
      if (K.pcg_mode != PT_PCG_SAMPLE) {
        s.vstate = s_to;
        if (s.vbase >= K.hperiod)  // which guess would have been right for this sample: its upper neighbour's draws, or its predecessor's?
          s.pscore += (int)(((unsigned)(s.hist >> (8 * (K.hperiod - 1))) & 0xffu) == (s_draws & 0xffu)) - (int)(((unsigned)s.hist & 0xffu) == (s_draws & 0xffu));
        s.hist = (s.hist << 8) | (uint64_t)(s_draws & 0xffu);
      }
      s.prays += s_rays;
      s.vbase++;
#ifdef PT_DEBUG_TIME
      if (dbg.tracing && s.in_unit && lane == s.leader && (s.leader / s.L) < 64 && s.vbase <= 80)
        pt_trace[PT_TRACE_LEN + (s.leader / s.L) * 80 + (s.vbase - 1)] = ((unsigned long long)dbg.ulog_rounds << 32) | ((unsigned long long)s_draws << 16) | ((unsigned long long)(s.leader / s.L) << 8) | 0xEEULL;
#endif
    }
  }
  s.mode = 2;
#ifdef PT_DEBUG_TIME
  {  // speculation statistics: pixel-rounds, samples traced, samples kept
    const bool lead = s.in_unit && lane == s.leader && s.pix >= 0;
    unsigned long long r4 = lead ? 1ULL : 0ULL, r5 = lead ? (unsigned long long)dbg_fin : 0ULL,
                       r6 = lead ? (unsigned long long)(s.vbase - dbg_vbase0) : 0ULL;
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
  if (s.in_unit) {
    if (s.vbase >= K.nsamp) {
      if (lane == s.leader && s.pix >= 0) {  // imagetracer.py:99-104
        if (K.S > 0) {
          const double k = 1.0 / (double)(K.S * K.S);
          s.cum.x = s.cum.x * k;
          s.cum.y = s.cum.y * k;
          s.cum.z = s.cum.z * k;
        }
        store_pixel(a, s.pix, s.cum);
        s.nrays += s.prays;
      }
      s.pix = -1;  // this pixel is done (in every lane of it)
    } else if (path_seed_round(a, K, s)) {
      s.mode = 0;
    }
  }
}

// ---- TILED: next unit for this wave, then its region's cone and survivor masks ----
// The sorted unit list is dealt out to PT_UNIT_SHARDS shards (unit u belongs to shard u % shards: every shard the same mix of
// sizes) and a workgroup pulls from shard blockIdx % shards only: a returning atomic on ONE head word saturates near 88
// dequeues/us -- with thousands of waves pulling, queueing at the head costs more than a unit's work.  A wave's first unit is
// its own rank in the shard (no atomic at all), later ones come from the shard's head, one atomic by lane 0.
// -> false when the list is exhausted (the wave is done)
template <bool LAT>
PT_DEV bool path_next_unit(const PtKArgs &a, const PathK &K, PathLane &s, PathDbg &dbg) {
  const int lane = K.lane;
  unsigned uid = 0;
  const unsigned nsh = gridDim.x < PT_UNIT_SHARDS ? gridDim.x : PT_UNIT_SHARDS;  // (every shard needs a puller)
  const unsigned shard = blockIdx.x % nsh;
  if (s.first_unit) {
    uid = (blockIdx.x / nsh) * (PT_BLOCK / 64) + (threadIdx.x >> 6);
    s.first_unit = false;
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
  dbg.tracing = seq == cold_args(a)->dbg_trace_unit;  // (or, below, the unit that starts at a given flagged pixel of a given region)
  if (LAT && dbg.tracing && lane == 0) {
    const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
    for (int q = 0; q < 3; ++q) dbg.dbg_q[q] = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0 && dbg.ulog_seq >= 0 && dbg.ulog_seq < PT_UNITLOG_LEN) {  // close the log entry of the unit just finished
    pt_unitlog[dbg.ulog_seq * 8 + 1] = __builtin_amdgcn_s_memtime();
    pt_unitlog[dbg.ulog_seq * 8 + 2] = (unsigned long long)dbg.ulog_rounds | ((unsigned long long)dbg.ulog_iters << 32);
    pt_unitlog[dbg.ulog_seq * 8 + 4] = dbg.tsum[4] - dbg.ulog_t[0];
    pt_unitlog[dbg.ulog_seq * 8 + 5] = dbg.tsum[5] - dbg.ulog_t[1];
    pt_unitlog[dbg.ulog_seq * 8 + 6] = dbg.tsum[1] + dbg.tsum[2] - dbg.ulog_t[2];
    pt_unitlog[dbg.ulog_seq * 8 + 7] = dbg.tsum[0] - dbg.ulog_t[3];
  }
  dbg.ulog_t[0] = dbg.tsum[4];
  dbg.ulog_t[1] = dbg.tsum[5];
  dbg.ulog_t[2] = dbg.tsum[1] + dbg.tsum[2];
  dbg.ulog_t[3] = dbg.tsum[0];
  dbg.ulog_seq = seq;
  dbg.ulog_rounds = 0;
  dbg.ulog_iters = 0;
#endif
  pt_kargs ca = cold_args(a);
  if (seq >= K.n_units) return false;
#ifdef PT_DEBUG_TIME
  if (lane == 0) pt_dbg_wave[(size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8 + 7] += 1ULL;
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
  if (cold_args(a)->dbg_trace_unit <= -2) dbg.tracing = (-2 - cold_args(a)->dbg_trace_unit) == region * 64 + first;  // (unit numbers vary from frame to frame)
#endif
  const unsigned long long todo = (unsigned long long)(unsigned)unit.z | ((unsigned long long)(unsigned)unit.w << 32);  // the region's flagged pixels
  const int ry = region / K.regions_x, rx = region - ry * K.regions_x;
  const int gr0 = global_row(a, ry * PT_REGION);
  const int gr1 = global_row(a, (ry * PT_REGION + PT_REGION - 1 < K.rows_local) ? ry * PT_REGION + PT_REGION - 1 : K.rows_local - 1);
  const TileCone tc = tile_cone(a, rx * PT_REGION, (rx * PT_REGION + PT_REGION < K.W) ? rx * PT_REGION + PT_REGION : K.W, gr0, gr1);
  __builtin_amdgcn_wave_barrier();
  for (int p = 0; p < K.npass; ++p) {
    const int slot = p * 64 + lane;
    bool keep = false;
    if (slot < a.n_shapes) keep = slot >= a.n_spheres || cone_keeps(tc, a.bounds[slot]);  // planes: always
    const unsigned long long m = __ballot(keep);
    if (lane == 0) pt_lds_masks[K.mbase + p] = m;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // lanes [p * L, (p + 1) * L) take the unit's p-th pixel = flagged pixel `first + p` of the region
  s.L = 64 / count;
  if (s.L > K.nsamp) s.L = K.nsamp;
#ifdef PT_DEBUG_TIME
  if (lane == 0 && seq < PT_UNITLOG_LEN) {
    pt_unitlog[seq * 8 + 0] = __builtin_amdgcn_s_memtime();
    pt_unitlog[seq * 8 + 3] = (unsigned long long)count | ((unsigned long long)s.L << 8) | ((unsigned long long)(blockIdx.x & 0x3ff) << 16) |
                              ((unsigned long long)first << 26) | ((unsigned long long)region << 32);
  }
#endif
  const int pidx = lane / s.L;
  s.in_unit = pidx < count;
  s.leader = s.in_unit ? pidx * s.L : lane;
  s.jlane = lane - s.leader;
  s.mode = 2;
  s.pix = -1;
  if (s.in_unit) {
    const int bit = nth_set_bit(todo, first + pidx);
    const int lrow = ry * PT_REGION + (bit >> 3);
    s.col = rx * PT_REGION + (bit & 7);
    s.pix = (long long)lrow * K.W + s.col;
    s.grow = global_row(a, lrow);  // (pixel_coords would divide the 64-bit index by W to find what is known here)
    s.gpix = (unsigned long long)s.grow * ca->W + s.col;
    if (K.pcg_mode != PT_PCG_SAMPLE) {
      pcg_seed(s.pcg, ca->s0, ca->q0 + s.gpix);
      s.vstate = s.pcg.state;
    }
    s.hist = 0x0101010101010101ULL * (uint64_t)(ca->spec_draws & 0xff);
    s.pscore = 0;
    s.vbase = 0;
    s.prays = 0;
    s.cum.x = 0.0;
    s.cum.y = 0.0;
    s.cum.z = 0.0;
    if (path_seed_round(a, K, s)) s.mode = 0;
  }
  return true;
}

// ---- one queue: idle lanes (mode 2) draw their next pixel ----
// FLAGGED: the queue deals out the frame's pixel indices and a lane that draws a settled pixel draws again -- or, where
// flagged pixels are few (under a quarter of the frame), the FLAGGED pixels themselves, from pt_unit_scatter's list of
// one-pixel units: skipping cost a returning atomic per wave and settled pixel (~1 M atomics on one word for a 4K
// frame with 3 % flagged pixels, at ~90 per us: C3 at 4K 10.5 -> 3.0 ms).  Frames FULL of flagged pixels keep the
// row-major order: the list's order (fullest regions first) costs them 15 - 30 % (profiles/r05_queue_dealing.txt).
template <bool FLAGGED>
PT_DEV void path_next_pixels(const PtKArgs &a, const PathK &K, PathLane &s) {
  for (;;) {
    const bool need = s.mode == 2 && !s.exhausted;
    if (!__any(need)) break;
    const long long np = next_pixel(a, need, K.deal_units ? (long long)K.n_flagged : a.npix);
    if (need && np >= 0) {
      bool take = true;
      long long p = np;
      if (FLAGGED) {
        if (K.deal_units) {
          const int4 unit = cold_args(a)->units[np];
          const unsigned long long todo = (unsigned long long)(unsigned)unit.z | ((unsigned long long)(unsigned)unit.w << 32);
          const int bit = nth_set_bit(todo, unit.y & 0xff);
          const int ry = unit.x / K.regions_x, rx = unit.x - ry * K.regions_x;
          p = (long long)(ry * PT_REGION + (bit >> 3)) * K.W + (rx * PT_REGION + (bit & 7));
        } else {  // pixels the first pass settled are not this kernel's (pt_tile_kernel: rmask)
          const int lr = (int)(np / K.W), c0 = (int)(np - (long long)lr * K.W);
          const unsigned long long m = cold_args(a)->region_mask[(lr / PT_REGION) * K.regions_x + c0 / PT_REGION];
          take = ((m >> ((lr % PT_REGION) * PT_REGION + (c0 % PT_REGION))) & 1ULL) != 0ULL;
        }
      }
      if (take) {
        s.pix = p;
        s.mode = 0;
      }
    }
    s.exhausted = __any(need && np < 0);
    if (!FLAGGED || K.deal_units) break;  // (dealing indices: lanes that drew a settled pixel draw again)
  }
}

// ---- unwind: deliver radiance up the stack until this lane has a child to scatter (`spawn`) or its sample is complete ----
// (a lane with a child to scatter leaves the loop; the scatter itself -- two draws, sin / cos, two square roots -- runs ONCE
//  behind the loop for every lane of the wave that spawns in this step, whether its node was pushed by path_shade or reached
//  by a child's return: inside the loop the wave ran it once per turn that any lane spawned in)
template <bool TILED, int LDSF, bool FLAGGED>
PT_DEV void path_unwind(const PtKArgs &a, const PathCtx &w, const PathK &K, PathLane &s, bool unwinding) {
  while (unwinding) {
    if (s.spawn) break;
    if (s.sp == 0) {
      path_finish_sample<TILED, FLAGGED>(a, K, s);  // mode 0 (next sample), 2 (pixel done) or 3 (TILED: wait for the round's end)
      break;
    }
    // a child of frame sp-1 returned `ret` (render.py:135-137)
    const int fs = s.sp - 1;
    const FrameAt<LDSF> f(w, fs);
    const V3 hc = {f.get(0), f.get(1), f.get(2)};
    V3 fc = {0.0, 0.0, 0.0};
    int done = 0;
    if (K.N > 1) {
      fc.x = f.get(6);
      fc.y = f.get(7);
      fc.z = f.get(8);
      done = (int)f.get(9);
    }
    fc.x = fc.x + hc.x * s.ret.x;
    fc.y = fc.y + hc.y * s.ret.y;
    fc.z = fc.z + hc.z * s.ret.z;
    done++;
    if (done < K.N) {
      f.put(6, fc.x);
      f.put(7, fc.y);
      f.put(8, fc.z);
      f.put(9, (double)done);
      s.f_wp = {f.get(10), f.get(11), f.get(12)};
      s.f_n = {f.get(13), f.get(14), f.get(15)};
      s.f_in = {f.get(16), f.get(17), f.get(18)};
      s.f_brdf = (int)f.get(19);
      s.spawn = true;
      continue;
    }
    // render.py:139
    s.ret.x = f.get(3) + fc.x * K.invN;
    s.ret.y = f.get(4) + fc.y * K.invN;
    s.ret.z = f.get(5) + fc.z * K.invN;
    s.sp = fs;
  }
}

// ---- FLAGGED: hand the pixel over to the tree kernel ----
// A lane walks its pixel's rays one after the other: a tree of 1 111 rays (the CLI's N = 10, D = 3) is 1 111 turns of the
// loop, ~6 us each, and once the pixel queue has run dry nothing fills the lanes that finish: the frame waits for its
// heaviest pixels while most of the chip idles.  The tree kernel behind this one traces a node's children at the same time,
// so a pixel's remaining rays take a sixth of the time there.  A lane therefore HANDS ITS PIXEL OVER, at the point where its
// next ray would be scattered: the node stack, the generator, the sums and the ray count go into a record
// (PT_HANDOVER_HEADER + 20 doubles per node, in the tree kernel's node layout) and the pixel becomes a unit of the tree
// kernel (PT_Q_HEAVY), which goes on exactly where the lane stopped -- nothing is traced twice, and every draw happens at the
// state the sequential program has there.  When: the queue dry and `q_few` or fewer lanes of the wave still hold a pixel
// (the plan's default); or the pixel has traced q_budget rays, or q_tail rays since the queue ran dry (measurement
// switches).  A full record table leaves the pixel with its lane.
template <int LDSF>
PT_DEV void path_hand_over(const PtKArgs &a, const PathCtx &w, const PathK &K, PathLane &s) {
  const bool few = K.q_few > 0 && s.exhausted && __popcll(__ballot(s.mode != 2)) <= K.q_few;  // (wave-uniform)
  if (s.spawn && !s.q_full && (few || (K.q_budget > 0 && s.qrays >= (unsigned)K.q_budget) || (K.q_tail > 0 && s.qtail >= (unsigned)K.q_tail))) {
    pt_kargs ca = cold_args(a);
    const unsigned long long k = atomicAdd(pt_queue(a) + PT_Q_HEAVY, 1ULL);
    s.q_full = k >= (unsigned long long)ca->handover_cap;  // (the lane keeps this pixel and asks no more)
    if (!s.q_full) {
      const int lr = (int)(s.pix / K.W), c0 = (int)(s.pix - (long long)lr * K.W);
      const int region = (lr / PT_REGION) * K.regions_x + c0 / PT_REGION;
      const unsigned long long m = ca->region_mask[region];
      const int bit = (lr % PT_REGION) * PT_REGION + (c0 % PT_REGION);
      const int first = __popcll(m & ((1ULL << bit) - 1ULL));
      ca->units_handed[k] = make_int4(region, first | (1 << 8) | (1 << 16), (int)(unsigned)m, (int)(unsigned)(m >> 32));
      double *rec = ca->handover + (size_t)k * (size_t)(PT_HANDOVER_HEADER + 20 * (K.D > 1 ? K.D : 1));
      rec[0] = __longlong_as_double((long long)s.pcg.state);
      rec[1] = __longlong_as_double((long long)s.pcg.inc);
      rec[2] = (double)s.samp;
      rec[3] = (double)s.sp;
      rec[4] = (double)s.qrays;
      rec[5] = s.cum.x;
      rec[6] = s.cum.y;
      rec[7] = s.cum.z;
      for (int d = 0; d < s.sp; ++d) {
        const FrameAt<LDSF> f(w, d);
        double *t = rec + PT_HANDOVER_HEADER + 20 * d;
        for (int q = 0; q < 9; ++q) t[q] = f.get(q);           // hit_color, emitted radiance, the children's sum so far
        for (int q = 0; q < 9; ++q) t[9 + q] = f.get(10 + q);  // hit point, normal, incoming direction
        t[18] = f.get(19);                                     // BRDF
        // children traced: the tree kernel counts the one whose subtree is being walked (every node but the innermost)
        t[19] = f.get(9) + (d < s.sp - 1 ? 1.0 : 0.0);
      }
      s.spawn = false;
      s.qrays = 0;
      s.qtail = 0;
      s.cum.x = 0.0;
      s.cum.y = 0.0;
      s.cum.z = 0.0;
      s.samp = 0;
      s.sp = 0;
      s.mode = 2;
    }
  }
}

// FLAGGED (!TILED only): the kernel runs BEHIND the first pass, as the alternative to pt_path_tree_kernel (PT_Q_CHOICE): it
// returns at once unless the device chose it, and a lane keeps only pixels the first pass flagged (the others are settled).
template <bool TILED, int LDSF, bool LAT, bool SLDS = false, int LEAN = 0, bool FLAGGED = false>
PT_DEV void path_trace(const PtKArgs &a) {
  static_assert(LDSF != 2 || !TILED, "the split frame stack belongs to the one-queue kernel");
  static_assert(!SLDS || LAT, "the scene is staged in LDS for the second pass only");
  static_assert(!FLAGGED || !TILED, "the flagged-pixel filter belongs to the one-queue kernel");
  PathCtx w;
  PathK K;
  {
    pt_kargs c = cold_args(a);
    w.ws = c->ws;
    w.nthreads = (size_t)c->nthreads;
    w.stride = (size_t)c->frame_doubles * w.nthreads;
    w.gtid = blockIdx.x * PT_BLOCK + threadIdx.x;
    w.lds_frame = c->frame_doubles;
    w.lds_base = TILED ? 4 * c->npass : 0;  // behind the four waves' survivor masks (8-byte units)
    w.deep_slot = (c->D > 1 ? c->D : 1) - 1;
    K.ortho = c->cam_kind != PT_CAMERA_PERSPECTIVE;
    K.diag_lds = c->diag_lds;
    K.pcg_mode = c->pcg_mode;
    K.S = c->S;
    K.N = c->N;
    K.W = c->W;
    K.rows_local = c->rows_local;
    K.npass = c->npass;
    K.D = c->D;
    K.rr = c->rr;
  }
  if (blockIdx.x == gridDim.x - 1) {  // the next frame's queue block (nothing of this frame reads it)
    unsigned long long *qn = pt_queue_next(a);
    for (int k = threadIdx.x; k < PT_QUEUE_WORDS; k += PT_BLOCK) qn[k] = 0ULL;
  }
  if (FLAGGED && pt_queue(a)[PT_Q_CHOICE] != 1ULL) {  // (uniform over the grid: the tree kernel renders this frame)
    add_ray_count(a, 0ULL, cold_args(a)->count_base);
    return;
  }
  K.scene_lds = path_stage_tables<LAT, SLDS>(a, K.diag_lds);
  K.nsamp = K.S > 0 ? K.S * K.S : 1;
  K.invN = 1.0 / (double)K.N;
  K.j2n = pcg_jump_coeffs(2u * (unsigned)K.N);  // (wave-uniform)
  K.lane = threadIdx.x & 63;
  K.mbase = (threadIdx.x >> 6) * K.npass;
  K.regions_x = (K.W + PT_REGION - 1) / PT_REGION;
  K.hperiod = (K.S >= 1 && K.S <= 8) ? K.S : 1;
  // (q_budget < 0 in the argument block: the budget pt_unit_scatter derived from the frame's flagged pixels)
  K.q_budget = !FLAGGED ? 0 : (cold_args(a)->q_budget >= 0 ? cold_args(a)->q_budget : (int)pt_queue(a)[PT_Q_BUDGET]);
  K.q_tail = FLAGGED ? cold_args(a)->q_tail_budget : 0;
  K.q_few = FLAGGED ? cold_args(a)->q_few_lanes : 0;
  // TILED: units of the frame (read once -- not from the heads' lines); FLAGGED: flagged pixels of the frame (one unit each:
  // pt_unit_scatter as for the tree kernel) and how the queue deals
  K.n_units = TILED ? (int)pt_queue(a)[9] : 0;
  K.n_flagged = FLAGGED ? (int)pt_queue(a)[9] : 0;
  K.deal_units = FLAGGED && 4LL * (long long)K.n_flagged < a.npix;

  PathLane s;
  s.mode = 2;
  s.pix = -1;
  s.pcg.state = 0;
  s.pcg.inc = 1;
  s.pcg.n = 0;
  s.samp = 0;
  s.sp = 0;
  s.col = 0;
  s.grow = 0;
  s.cum = {0.0, 0.0, 0.0};
  s.ray.o = {0.0, 0.0, 0.0};
  s.ray.d = {1.0, 0.0, 0.0};
  s.ray.tmin = 1e-5;
  s.ret = {0.0, 0.0, 0.0};
  s.spawn = false;
  s.f_wp = {0.0, 0.0, 0.0};
  s.f_n = {0.0, 0.0, 1.0};
  s.f_in = {1.0, 0.0, 0.0};
  s.f_brdf = 0;
  s.L = 1;
  s.leader = K.lane;
  s.jlane = 0;
  s.in_unit = false;
  s.vbase = 0;
  s.vstate = 0;
  s.hist = 0;
  s.pscore = 0;
  // (Round 4 also let pixels with lanes to spare trace the next samples from a WINDOW of start states; it moved a full frame
  //  by nothing -- four lanes per pixel -- and was deleted in round 5: profiles/DROPPED_VARIANTS.md.)
  s.st_start = 0;
  s.srays = 0;
  s.prays = 0;
  s.gpix = 0;
  s.first_unit = true;
  s.exhausted = false;
  s.qtail = 0;
  s.q_full = false;
  s.qrays = 0;
  s.nrays = 0;

  PathDbg dbg;
#ifdef PT_DEBUG_TIME
  for (int q = 0; q < 8; ++q) dbg.tsum[q] = 0;
  dbg.tprev = __builtin_amdgcn_s_memtime();
  dbg.tracing = false;
  dbg.trace_n = 0;
  dbg.ulog_seq = -1;
  dbg.ulog_rounds = 0;
  dbg.ulog_iters = 0;
  for (int q = 0; q < 4; ++q) dbg.ulog_t[q] = 0;
  for (int q = 0; q < 3; ++q) dbg.dbg_q[q] = 0;
#endif
  for (;;) {
    PT_STAMP(7);
    // (values that never flow from one iteration into the next: said explicitly, so that they hold no
    //  registers across the queries)
    s.spawn = false;
    s.f_wp = {0.0, 0.0, 0.0};
    s.f_n = {0.0, 0.0, 1.0};
    s.f_in = {1.0, 0.0, 0.0};
    s.f_brdf = 0;
    // ---- work for idle lanes ----
    if (TILED) {
      if (!__any(s.mode == 0 || s.mode == 1)) {
        if (__any(s.mode == 3)) path_commit_round<LDSF>(a, w, K, s, dbg);
        PT_STAMP(3);
        if (!__any(s.mode == 0))
          if (!path_next_unit<LAT>(a, K, s, dbg)) break;
      }
    } else {
      path_next_pixels<FLAGGED>(a, K, s);
      if (!__any(s.mode != 2)) break;
    }

    PT_STAMP(0);
#ifdef PT_DEBUG_TIME
    dbg.ulog_iters++;
#endif
    const int n_start = __popcll(__ballot(s.mode == 0));
    const int n_path = __popcll(__ballot(s.mode == 1));
    if (n_start == 0 && n_path == 0) continue;  // TILED: nothing in flight, the round / unit logic above decides
    const bool do_p = n_start > 0 && n_path < cold_args(a)->p_max_path;
    const bool do_s = n_path >= cold_args(a)->s_min_path || (n_path > 0 && !do_p);

    // ---- queries: primary rays against the region's survivors, scattered rays against everything ----
    const bool prim = do_p && s.mode == 0;
    const bool scat = do_s && s.mode == 1;
    double best_t = INFINITY;
    int hit = -1;
    if (do_p) {
      if (prim) path_start_sample<TILED>(a, K, s);
      PT_STAMP(1);
      double tp = INFINITY;
      int hp;
      if (TILED)
        hp = K.ortho ? world_query_tile<false, false, false>(a, s.ray, K.mbase, K.npass, tp, prim)
                     : world_query_tile<false, false, true>(a, s.ray, K.mbase, K.npass, tp, prim);
      else
        hp = world_query<false, false>(a, s.ray, INFINITY, tp, prim);
      if (prim) {
        hit = hp;
        best_t = tp;
      }
      PT_STAMP(2);
    }
    if (do_s) {
      double ts;
      const int hs = LAT ? world_query_lanes<false, LEAN>(a, s.ray, INFINITY, ts, scat, K.diag_lds) : world_query<false, false>(a, s.ray, INFINITY, ts, scat);
      if (scat) {
        hit = hs;
        best_t = ts;
      }
      PT_STAMP(4);
#ifdef PT_DEBUG_TIME
      if (LAT && dbg.tracing && K.lane == 0) {  // the traced unit: this query's prefilter cycles (8), walk cycles (9), walk turns (10)
        const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
        for (int q = 0; q < 3; ++q) {
          const unsigned long long now = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (dbg.trace_n < PT_TRACE_LEN) pt_trace[dbg.trace_n] = ((now - dbg.dbg_q[q]) << 16) | (unsigned long long)(8 + q);
          dbg.dbg_q[q] = now;
          dbg.trace_n++;
        }
      } else if (LAT && dbg.tracing) {
        dbg.trace_n += 3;
      }
#endif
    }

    // ---- shade the hit, then unwind: deliver radiance up the stack / scatter the next child, until
    //      this lane has a ray that needs a query (mode 1) or its sample is complete (mode 0 / 2 / 3) ----
    const bool work = prim || scat;
    if (work) {
      if (TILED)
        s.srays++;
      else if (FLAGGED) {
        s.qrays++;
        if (s.exhausted) s.qtail++;
      }
      else
        s.nrays++;
      path_shade<LDSF, SLDS>(a, w, K, s, hit, best_t);
      s.mode = 1;
    }
    path_unwind<TILED, LDSF, FLAGGED>(a, w, K, s, work);
    if constexpr (FLAGGED) path_hand_over<LDSF>(a, w, K, s);
    if (s.spawn) {
      // scatter_ray (materials.py:132-152, 175-196); the child is at depth sp <= max_depth (a hit whose children would lie
      // beyond it never pushes a frame: path_shade_hit), so it is queried at the next S-step (mode 1)
      s.ray = scatter_ray<true>(s.f_brdf, s.pcg, s.f_in, s.f_wp, s.f_n);
      s.spawn = false;
    }
    PT_STAMP(5);
  }
#ifdef PT_DEBUG_TIME
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, dbg.tsum[q]);
  pt_dbg_flush();
#endif
  add_ray_count(a, s.nrays, TILED ? 0 : cold_args(a)->count_base);
}

// every pixel of the frame, one lane per pixel, pixels from one queue, every shape tested by the wave-uniform loop, the frame
// stack in HBM: worlds the tiled kernels do not take (no shape at all) and the measurement switch PTRACE_CULL=0 -- the
// brute-force device path the culled kernels are checked against
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 8))) void pt_path_kernel(const PtKArgs a) {
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
