// pt_tree.h -- PathTracer with num_of_rays > 1: one pixel per wave (pt_path_tree_kernel).
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
// ---- PathTracer with num_of_rays > 1 (second pass behind pt_tile_kernel<PATHTRACER>): ONE pixel per wave, a node's children on lanes --
// render.py:126-139 runs the N children of a hit one after the other, each with its whole subtree, all drawing from one
// generator: where child k starts in the stream is known only when child k-1 has returned.  path_trace gives such a
// pixel one lane, which walks the tree ray by ray: up to sum N^d dependent steps (1 111 for the CLI's N = 10, D = 3)
// while a frame's worst pixel sets the launch time (profiles/r03_units_n10_before.log: 8.3 ms, 1 111 iterations).
// Here a wave owns a pixel and works on one NODE at a time (explicit stack of nodes, depth first, so the order of
// draws is the reference's): the node's next children are scattered and traced AT THE SAME TIME on different lanes, each
// from a SPECULATED generator state, and then committed in child order by comparing states -- child k counts iff the
// state it started from is the state child k-1 ended with, in which case everything it computed is what the sequential
// program computes; the first child that started elsewhere (and everything behind it) is simply done again in the next
// round from the right state.  Child 0 always starts right, so a round commits at least one child.
//   * A child that needs children of its own (hit, lum > 0, survived roulette, depth < D) is committed by pushing its
//     node; its later siblings wait for the state its subtree leaves behind.
//   * Ordinary families speculate a chain: child r starts r * cpred draws ahead, cpred = what the last committed child
//     without a subtree drew (initially: its scatter draws, plus the roulette draw where depth >= rr).
//   * LEAF families (children at depth D: traced, but THEIR children are beyond max_depth and only consume draws,
//     render.py:100-101) have few outcomes: c0 draws (hit and killed, black or specular surface) or c0 + 2N (a diffuse hit
//     that survives).  So child r is traced for EVERY start state it can have, r * c0 + b * 2N for b = 0..r: 55 lanes
//     settle ten leaves in one round whatever mix of outcomes they have.  (A miss draws c0 - 1: the chain then breaks
//     there and resumes next round -- slower, never wrong.)
// The sum a node keeps (cum_radiance += hit_color * child, render.py:137) is formed in child order, so the frame is
// the sequential one bit for bit; rays are counted for committed children only.
// LDS: per wave max(D, 1) node records of PT_TREE_FRAME doubles; of the innermost node hit_color, the running sum, the child
// counter and the BRDF kind are also kept in registers (wave-uniform).
#define PT_TREE_FRAME 20  // hc 0..2, em 3..5, cum 6..8, wp 9..11, n 12..14, in 15..17, brdf 18, next child 19

// SLDS (round 5): the shapes' records (128 B + 256 B each) staged in LDS by the workgroup, as the second pass by regions does:
// shading gathers ~25 values of the hit shape per lane through dependent loads (needs_uv -> pigments -> matrices), and a
// round of this kernel is one dependent chain: every trip to L2 is in it.
template <bool SMALL, bool SLDS = false>
PT_DEV void path_tree(const PtKArgs &a) {
  int S, nsamp, N, W, rows_local, npass, D, rr, diag_lds, pcg_mode, frames_lds;
  bool ortho;
  {
    pt_kargs c = cold_args(a);
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
    frames_lds = 4 * c->npass + (int)(threadIdx.x >> 6) * (c->D > 1 ? c->D : 1) * PT_TREE_FRAME;  // (doubles)
  }
  if (blockIdx.x == gridDim.x - 1) {  // the next frame's queue block (nothing of this frame reads it)
    unsigned long long *qn = pt_queue_next(a);
    for (int k = threadIdx.x; k < PT_QUEUE_WORDS; k += PT_BLOCK) qn[k] = 0ULL;
  }
  // (uniform over the grid) a frame full of flagged pixels: pt_path_flagged_kernel rendered it, except the pixels it
  // handed over (PT_Q_HEAVY units from unit 0 of the list)
  const bool handed = pt_queue(a)[PT_Q_CHOICE] != 0ULL;
  int n_units = (int)pt_queue(a)[handed ? PT_Q_HEAVY : 9];
  if (handed && n_units > cold_args(a)->handover_cap) n_units = cold_args(a)->handover_cap;  // (a full table: the lane kept its pixel)
  if (handed && n_units == 0) {
    add_ray_count(a, 0ULL);
    return;
  }
  if (diag_lds >= 0) {  // scale+translate records into LDS: world_query_lanes fetches them by lane-private index
    const unsigned long long *src = (const unsigned long long *)a.diag;
    for (int k = threadIdx.x; k < a.n_diag * 8; k += PT_BLOCK) pt_lds_masks[diag_lds + k] = src[k];
    __syncthreads();
  }
  // Leaf rounds start every lane a FIXED number of draws ahead of the sequential state -- row * c0 + column * 2N for the
  // hypothesis lanes, N * c0 + 2N * b for the lanes that trace the parent's next child -- so the jump's coefficients
  // (state' = A * state + inc * G, both powers of the multiplier) are tabulated once per workgroup for the four values c0
  // can take, instead of ~8 rounds of 64-bit multiplications by repeated squaring in every lane and round.
  const int jump_lds = cold_args(a)->tree_jump_lds;
  if (jump_lds >= 0) {
    const int Nn = cold_args(a)->N;
    for (int e = threadIdx.x; e < 2 * 4 * 64; e += PT_BLOCK) {
      const int which = e >> 8, c0e = (e >> 6) & 3, idx = e & 63;
      unsigned delta;
      if (which == 0) {
        int row = 0;
        while ((row + 1) * (row + 2) / 2 <= idx) ++row;
        delta = (unsigned)row * (unsigned)c0e + (unsigned)(idx - row * (row + 1) / 2) * 2u * (unsigned)Nn;
      } else {
        delta = (unsigned)Nn * (unsigned)c0e + 2u * (unsigned)Nn * (unsigned)idx;
      }
      uint64_t acc_mul = 1ULL, acc_g = 0ULL, cur_mul = 6364136223846793005ULL, cur_g = 1ULL;
      while (delta) {  // (pcg_advance with the increment factored out)
        if (delta & 1u) {
          acc_mul *= cur_mul;
          acc_g = acc_g * cur_mul + cur_g;
        }
        cur_g = (cur_mul + 1ULL) * cur_g;
        cur_mul *= cur_mul;
        delta >>= 1;
      }
      pt_lds_masks[jump_lds + 2 * e] = acc_mul;
      pt_lds_masks[jump_lds + 2 * e + 1] = acc_g;
    }
    __syncthreads();
  }
  int scene_lds = 0;
  if (SLDS) {  // recs[] then aux[] (8-byte words)
    scene_lds = cold_args(a)->scene_lds;
    const unsigned long long *src = (const unsigned long long *)a.recs;
    for (int k = threadIdx.x; k < a.n_shapes * 16; k += PT_BLOCK) pt_lds_masks[scene_lds + k] = src[k];
    src = (const unsigned long long *)a.aux;
    for (int k = threadIdx.x; k < a.n_shapes * 32; k += PT_BLOCK) pt_lds_masks[scene_lds + a.n_shapes * 16 + k] = src[k];
    __syncthreads();
  }
  {  // the grid's occupancy bits into LDS: the cell walk of world_query_lanes reads one per step
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
  bool first_unit = true;
  unsigned long long nrays = 0;  // (wave-uniform: committed rays of this wave's pixels)
  // small worlds: the wave-uniform loop over every shape (records through the scalar cache) has a shorter critical
  // path than per-lane candidate lists -- and a round's latency, not its throughput, is what a pixel's tree waits for
  // lane b: complete leaf families so far (of this wave, whatever the pixel) that had b survivors -- where the next family's
  // guesses go (an exponentially fading count)
  int b_count = 0;
  // lane r of a leaf round: row = child offset in the round, col = hypothesis b (0..row); rows with row(row+1)/2 + row < 64
  int tri_row = 0;
  while ((tri_row + 1) * (tri_row + 2) / 2 <= lane) ++tri_row;
  const int tri_col = lane - tri_row * (tri_row + 1) / 2;
  int tri_rows = 0;  // rows that fit the wave: 10
  while ((tri_rows + 1) * (tri_rows + 2) / 2 <= 64) ++tri_rows;

  // node records live in LDS (frame d = the node at depth d of the current path through the tree); a wave's DS
  // operations execute in order, so a record written by one lane is what every lane reads afterwards
  auto frame = [&](int d) -> double * { return pt_lds_f64 + frames_lds + d * PT_TREE_FRAME; };
  auto rfl_f64 = [&](double v) -> double { return rl_f64(v, 0); };  // (a broadcast LDS read, made scalar)

  // what a lane found out about the ray it traced (render.py:103-139 up to the recursion)
  bool o_term = true;            // the call returns without children of its own
  V3 o_ret = {0.0, 0.0, 0.0};    // ... this value
  V3 o_hc = {0.0, 0.0, 0.0}, o_em = {0.0, 0.0, 0.0}, o_wp = {0.0, 0.0, 0.0}, o_n = {0.0, 0.0, 1.0};  // else: its node
  int o_brdf = 0;
  Pcg pcg;
  pcg.state = 0;
  pcg.inc = 1;
  pcg.n = 0;
  Ray ray;
  ray.o = {0.0, 0.0, 0.0};
  ray.d = {1.0, 0.0, 0.0};
  ray.tmin = 1e-5;
  auto shade_hit = [&](auto rec, auto ax, double best_t, int depth) {
    Hit h;
    h.u = 0.0;
    h.v = 0.0;
    bool details = false;
    if (ax->needs_uv != 0) {
      hit_details<true>(rec, ax, ray, best_t, h, true);
      details = true;
    }
    V3 hc = brdf_pigment(a, ax, h.u, h.v);
    const V3 em = emitted_pigment(a, ax, h.u, h.v);
    const double lum = max2(max2(hc.x, hc.y), hc.z);
    if (depth >= rr) {  // render.py:116-123
      const double q = max2(0.05, 1.0 - lum);
      if (pcg_float(pcg) > q) {
        const double k = 1.0 / (1.0 - q);
        hc.x = hc.x * k;
        hc.y = hc.y * k;
        hc.z = hc.z * k;
      } else {
        o_ret = em;
        return;
      }
    }
    if (!(lum > 0.0)) {  // render.py:139 with cum_radiance = 0
      o_ret = {em.x + 0.0 * invN, em.y + 0.0 * invN, em.z + 0.0 * invN};
      return;
    }
    if (depth + 1 > D) {  // every child is beyond max_depth: its scatter draws are consumed, it returns black (render.py:100-101)
      // (one jump of 2 N draws, one addition of hit_color * 0: see path_trace)
      const bool diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
      if (diffuse) pcg_jump(pcg, j2n, 2u * (unsigned)N);
      const V3 fc = {0.0 + hc.x * 0.0, 0.0 + hc.y * 0.0, 0.0 + hc.z * 0.0};
      o_ret = {em.x + fc.x * invN, em.y + fc.y * invN, em.z + fc.z * invN};
      return;
    }
    if (!details) hit_details<true>(rec, ax, ray, best_t, h, false);
    o_term = false;
    o_hc = hc;
    o_em = em;
    o_wp = h.wp;
    o_n = h.n;
    o_brdf = ax->brdf_kind;
  };
  auto shade_ray = [&](int hit, double best_t, int depth) {
    o_term = true;
    if (hit < 0) {  // render.py:103-105
      pt_kargs c = cold_args(a);
      o_ret = {c->bg[0], c->bg[1], c->bg[2]};
      return;
    }
    if constexpr (SLDS)
      shade_hit((pt_lds_rec)(const void *)(pt_lds_f64 + scene_lds) + hit,
                (pt_lds_aux)(const void *)(pt_lds_f64 + scene_lds + a.n_shapes * 16) + hit, best_t, depth);
    else
      shade_hit(a.recs + hit, cold_args(a)->aux + hit, best_t, depth);
  };

#ifdef PT_DEBUG_TIME
  // cycles of this wave in: 0 fetch + cull, 1 primary ray, 2 state jump + scatter, 3 scattered-ray query, 4 shade,
  // 5 commit, 6 node returns; 7: rounds
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
  unsigned long long dbg_leaf_rounds = 0, dbg_committed = 0, dbg_traced = 0, dbg_max_rounds = 0, dbg_fused = 0, dbg_fused_hit = 0, dbg_fused_incomplete = 0;
#define PT_TT(k) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[k] += tn - tprev; tprev = tn; } while (0)
#else
#define PT_TT(k) do { } while (0)
#endif
  for (;;) {
    // ---- next pixel: the unit list, one pixel per unit (see path_trace for the sharded heads) ----
    unsigned uid = 0;
    const unsigned nsh = gridDim.x < PT_UNIT_SHARDS ? gridDim.x : PT_UNIT_SHARDS;
    const unsigned shard = blockIdx.x % nsh;
    if (first_unit) {
      uid = (blockIdx.x / nsh) * (PT_BLOCK / 64) + (threadIdx.x >> 6);
      first_unit = false;
    } else {
      const unsigned pullers = (gridDim.x - shard + nsh - 1) / nsh * (PT_BLOCK / 64);
      if (lane == 0) uid = pullers + (unsigned)atomicAdd(pt_queue(a) + PT_QUEUE_HEADS + 32 * shard, 1ULL);
    }
    uid = uid * nsh + shard;
    const int seq = (int)__builtin_amdgcn_readfirstlane((int)uid);
    if (seq >= n_units) break;
    pt_kargs ca = cold_args(a);
    const int4 unit = handed ? ca->units_handed[seq] : ca->units[seq];
    const int region = __builtin_amdgcn_readfirstlane(unit.x), first = __builtin_amdgcn_readfirstlane(unit.y) & 0xff;
    // a pixel the one-queue kernel handed over in the middle of its tree (path_trace): record `seq` says where it stands
    const bool resumed = ((__builtin_amdgcn_readfirstlane(unit.y) >> 16) & 1) != 0;
    const unsigned long long todo = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(unit.z) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(unit.w) << 32);
    const int ry = region / regions_x, rx = region - ry * regions_x;
    {  // the region's cone and survivor masks, for the primary rays
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
    }
    const int bit = nth_set_bit(todo, first);
    const long long pix = (long long)(ry * PT_REGION + (bit >> 3)) * W + (rx * PT_REGION + (bit & 7));
    int col, grow;
    pixel_coords(a, pix, col, grow);
    const unsigned long long gpix = (unsigned long long)grow * ca->W + col;
    // the pixel's generator (PT_PCG_PIXEL) -- wave-uniform: `gstate` is the state the sequential program is in
    unsigned long long gstate = 0, ginc = 1;
    if (pcg_mode != PT_PCG_SAMPLE) {
      Pcg g;
      pcg_seed(g, ca->s0, ca->q0 + gpix);
      gstate = g.state;
      ginc = g.inc;
    }
    V3 cum_pix = {0.0, 0.0, 0.0};
    unsigned long long prays = 0;
    int samp0 = 0, resume_sp = 0;  // (resume_sp > 0: the first sample of this unit goes on in the middle of its tree)
    if (resumed) {
      // what the lane had: the samples it finished and their sum, the rays it traced, the generator where the sequential
      // program's is, and the node stack as the lane left it (its records are in this kernel's node layout) -- the next
      // thing the sequential program does is scatter child `next` of the innermost node
      const double *hrec = ca->handover + (size_t)seq * (size_t)(PT_HANDOVER_HEADER + PT_TREE_FRAME * (D > 1 ? D : 1));
      gstate = (unsigned long long)__double_as_longlong(rl_f64(hrec[0], 0));
      ginc = (unsigned long long)__double_as_longlong(rl_f64(hrec[1], 0));
      samp0 = (int)rl_f64(hrec[2], 0);
      resume_sp = (int)rl_f64(hrec[3], 0);
      prays = (unsigned long long)rl_f64(hrec[4], 0);
      cum_pix = {rl_f64(hrec[5], 0), rl_f64(hrec[6], 0), rl_f64(hrec[7], 0)};
      for (int e = lane; e < resume_sp * PT_TREE_FRAME; e += 64) frame(0)[e] = hrec[PT_HANDOVER_HEADER + e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef PT_DEBUG_TIME
    unsigned long long dbg_rounds_pix = 0;
#endif
    PT_TT(0);
    for (int samp = samp0; samp < nsamp; ++samp) {
      const bool resume_now = resume_sp > 0;  // (wave-uniform)
      if (pcg_mode == PT_PCG_SAMPLE && !resume_now) {
        Pcg g;
        pcg_seed(g, ca->s0, ca->q0 + gpix * (unsigned)nsamp + (unsigned)samp);
        gstate = g.state;
        ginc = g.inc;
      }
      // ---- the sample's primary ray (imagetracer.py:86-97): lane 0 ----
      V3 sample_ret = {0.0, 0.0, 0.0};
      if (!resume_now) {
        pcg.state = gstate;
        pcg.inc = ginc;
        pcg.n = 0;
        double up = 0.5, vp = 0.5;
        if (S > 0) {
          const int sr = samp / S, sc = samp - sr * S;
          up = ((double)sc + pcg_float(pcg)) / (double)S;
          vp = ((double)sr + pcg_float(pcg)) / (double)S;
        }
        ray = primary_ray(a, col, grow, up, vp);
        {
          double tp = INFINITY;
          // (an orthogonal camera's rays have no common origin: nothing is hoisted)
          const int hp = ortho ? world_query_tile<false, false, false>(a, ray, mbase, npass, tp, lane == 0)
                               : world_query_tile<false, false, true>(a, ray, mbase, npass, tp, lane == 0);
          shade_ray(hp, tp, 0);
        }
        prays += 1ULL;
        PT_TT(1);
        sample_ret = rl_v3(o_ret, 0);
        gstate = rl_u64(pcg.state, 0);
      }
      int sp = 0;  // nodes on the stack; the innermost one (frame sp - 1) is the node whose children are being traced
      // of that node, in registers (wave-uniform): hit_color, the sum of its children so far, how many are done, its BRDF
      V3 t_hc = {0.0, 0.0, 0.0}, t_cum = {0.0, 0.0, 0.0};
      int t_next = 0, t_brdf = 0;
      unsigned cpred = 0;
      // lane `src` traced a ray that needs children of its own: its node becomes frame sp (written by that lane itself)
      auto push_node = [&](int src, V3 in_dir) {
        if (sp > 0 && lane == 0) {  // the parent's running sum and child counter wait in its record
          double *f = frame(sp - 1);
          f[6] = t_cum.x; f[7] = t_cum.y; f[8] = t_cum.z;
          f[19] = (double)t_next;
        }
        if (lane == src) {
          double *f = frame(sp);
          f[0] = o_hc.x; f[1] = o_hc.y; f[2] = o_hc.z; f[3] = o_em.x; f[4] = o_em.y; f[5] = o_em.z;
          f[6] = 0.0; f[7] = 0.0; f[8] = 0.0; f[9] = o_wp.x; f[10] = o_wp.y; f[11] = o_wp.z;
          f[12] = o_n.x; f[13] = o_n.y; f[14] = o_n.z; f[15] = in_dir.x; f[16] = in_dir.y; f[17] = in_dir.z;
          f[18] = (double)o_brdf; f[19] = 0.0;
        }
        // (the record is read by every lane later on: the compiler may neither move those loads above this store nor
        //  feed them from this lane's registers)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        t_hc = rl_v3(o_hc, src);
        t_cum = {0.0, 0.0, 0.0};
        t_next = 0;
        t_brdf = __builtin_amdgcn_readlane(o_brdf, src);
        sp++;
      };
      // render.py:139 for the innermost node, then render.py:137 in its parent, which becomes the innermost one
      auto pop_node = [&]() -> V3 {
        const double *f = frame(sp - 1);
        const V3 val = {rfl_f64(f[3]) + t_cum.x * invN, rfl_f64(f[4]) + t_cum.y * invN, rfl_f64(f[5]) + t_cum.z * invN};
        sp--;
        if (sp > 0) {
          const double *g = frame(sp - 1);
          t_hc = {rfl_f64(g[0]), rfl_f64(g[1]), rfl_f64(g[2])};
          t_cum = {rfl_f64(g[6]), rfl_f64(g[7]), rfl_f64(g[8])};
          t_brdf = (int)rfl_f64(g[18]);
          t_next = (int)rfl_f64(g[19]);
          t_cum.x = t_cum.x + t_hc.x * val.x;
          t_cum.y = t_cum.y + t_hc.y * val.y;
          t_cum.z = t_cum.z + t_hc.z * val.z;
        }
        return val;
      };
      auto base_draws = [&]() -> unsigned {  // what a child of the innermost node draws when it needs no children: scatter + roulette
        return (t_brdf == PT_BRDF_DIFFUSE ? 2u : 0u) + (sp >= rr ? 1u : 0u);
      };
      if (resume_now) {
        sp = resume_sp;
        resume_sp = 0;
        const double *g = frame(sp - 1);
        t_hc = {rfl_f64(g[0]), rfl_f64(g[1]), rfl_f64(g[2])};
        t_cum = {rfl_f64(g[6]), rfl_f64(g[7]), rfl_f64(g[8])};
        t_brdf = (int)rfl_f64(g[18]);
        t_next = (int)rfl_f64(g[19]);
        cpred = base_draws();
      } else if (!__builtin_amdgcn_readfirstlane((int)o_term)) {
        push_node(0, ray.d);
        cpred = base_draws();
      }
      // ---- the tree under the primary hit ----
      while (sp > 0) {
        const int remaining = N - t_next;
        if (remaining <= 0) {
          const V3 val = pop_node();
          if (sp == 0) {
            sample_ret = val;
            break;
          }
          cpred = base_draws();
          PT_TT(6);
          continue;
        }
        // ---- a round: children t_next .. of this node, depth sp, each from a speculated state ----
        const unsigned c0 = base_draws();
        const bool leaf = sp == D;  // (children of the children are beyond max_depth)
        int row, nrows;
        unsigned ahead;
        if (leaf) {
          nrows = remaining < tri_rows ? remaining : tri_rows;
          row = tri_row;
          ahead = (unsigned)tri_row * c0 + (unsigned)tri_col * 2u * (unsigned)N;
        } else {
          nrows = remaining < 64 ? remaining : 64;
          row = lane;
          ahead = (unsigned)lane * cpred;
        }
        bool act = row < nrows;
        // A WHOLE leaf family leaves lanes over (ten leaves: 55 of 64).  Where the family ends is known up to the number
        // b of its members that survive roulette on a diffuse surface -- N * c0 + 2N * b draws -- so the spare lanes
        // trace the NEXT sibling of this node (a child of its parent, one level up) from those states in the same round:
        // when the family commits in full and b is among the guesses, the sibling's ray is already traced when the node
        // returns, and a parent whose children all branch costs one round per child instead of two.
        const int leaf_lanes = nrows * (nrows + 1) / 2;
        // (round 5: when this node is its parent's LAST child the parent returns with it -- at no draws --, and the ray that
        //  follows is the next child of the nearest ancestor that has children left: `up` levels above this node.  A full
        //  tree of the CLI's N = 10, D = 3 saved one round per leaf family this way in round 3 and saves the ten re-traced
        //  children of the root now.)
        int up = 0;
        if (leaf && sp >= 2 && t_next == 0 && nrows == N && leaf_lanes < 64)
          for (int u = 1; u <= sp - 1; ++u)
            if ((int)rfl_f64(frame(sp - 1 - u)[19]) < N) {
              up = u;
              break;
            }
        const bool fused = up > 0;
        // (the spare lanes cover the nh values of b -- a family's survivors -- that were the most FREQUENT so far: lane b < N + 1
        //  counts the families that had b survivors, and the nh lanes of highest count, the lower b first on a tie, give the
        //  guesses.  Ten leaves leave nine lanes for eleven values, and the counts are not of one kind: a node under open sky
        //  has 0 - 2 survivors, one in a crevice 5 - 8.  Round 4's nine values around the LAST family's count missed one time
        //  in ten, nine around the running mean one in thirteen -- the window then slides off b = 0 --; every miss is a round.)
        const int nh = (64 - leaf_lanes) < (N + 1) ? (64 - leaf_lanes) : (N + 1);
        unsigned long long b_set = (N + 1 <= 63) ? ((1ULL << (N + 1)) - 1ULL) : ~0ULL;  // (wave-uniform) the values guessed
        if (fused && nh < N + 1) {
          int rank = 0;  // lanes 0 .. N: how many values come before this one
          for (int j = 0; j <= N; ++j) {
            const int cj = __builtin_amdgcn_readlane(b_count, j);
            rank += (cj > b_count || (cj == b_count && j < lane)) ? 1 : 0;
          }
          b_set = __ballot(lane <= N && rank < nh);
        }
        const int sib_h = (fused && lane >= leaf_lanes && lane - leaf_lanes < nh) ? lane - leaf_lanes : -1;
        const bool sib = sib_h >= 0;
        const int sib_b = sib ? nth_set_bit(b_set, sib_h) : 0;  // this lane's guess of the family's survivors
        if (sib) {
          act = true;
          row = -1;
          ahead = (unsigned)N * c0 + 2u * (unsigned)N * (unsigned)sib_b;
        }
        // (what the last round found out is dead: said explicitly, so that it holds no registers across the query)
        o_term = true;
        o_ret = o_hc = o_em = o_wp = {0.0, 0.0, 0.0};
        o_n = {0.0, 0.0, 1.0};
        o_brdf = 0;
        if (leaf && jump_lds >= 0 && c0 < 4u) {
          const int e = sib ? 256 + (int)c0 * 64 + sib_b : (int)c0 * 64 + lane;
          const uint64_t A = pt_lds_masks[jump_lds + 2 * (act ? e : 0)], G = pt_lds_masks[jump_lds + 2 * (act ? e : 0) + 1];
          pcg.state = act ? A * gstate + ginc * G : gstate;
        } else {
          pcg.state = act ? pcg_advance(gstate, ginc, ahead) : gstate;
        }
        pcg.inc = ginc;
        pcg.n = 0;
        const unsigned long long st_start = pcg.state;
        {
          const double *f = frame(sib ? sp - 1 - up : sp - 1);  // the node the ray leaves from
          const V3 n_wp = {f[9], f[10], f[11]}, n_n = {f[12], f[13], f[14]}, n_in = {f[15], f[16], f[17]};
          ray = scatter_ray<true>((int)f[18], pcg, n_in, n_wp, n_n);  // materials.py:132-152, 175-196
        }
        PT_TT(2);
        double ts = INFINITY;
        // (the wave-uniform loop over all shapes was measured against the per-lane candidate lists on C3, N = 10 -- 2.48 against
        //  1.92 ms -- and deleted in round 6)
        const int hs = world_query_lanes<false, SMALL ? 1 : 0>(a, ray, INFINITY, ts, act, diag_lds);
        PT_TT(3);
        if (act) shade_ray(hs, ts, sib ? sp - up : sp);
        PT_TT(4);
#ifdef PT_DEBUG_TIME
        tsum[7] += 1;
        dbg_rounds_pix += 1;
        if (leaf) dbg_leaf_rounds += 1;
        dbg_traced += (unsigned long long)__popcll(__ballot(act));
#endif
        // ---- commit in child order ----
        unsigned long long expect = gstate;
        bool pushed = false;
        unsigned fam_draws = 0;
        // child `src` of the innermost node counts: add its value up, or put its node on the stack
        auto commit_child = [&](int src) {
          prays += 1ULL;
          t_next++;
          expect = rl_u64(pcg.state, src);
          if (__builtin_amdgcn_readlane((int)o_term, src)) {
            const V3 val = rl_v3(o_ret, src);
            t_cum.x = t_cum.x + t_hc.x * val.x;  // render.py:137
            t_cum.y = t_cum.y + t_hc.y * val.y;
            t_cum.z = t_cum.z + t_hc.z * val.z;
            cpred = (unsigned)__builtin_amdgcn_readlane((int)pcg.n, src);
            fam_draws += cpred;
          } else {  // the child has children of its own: its node goes on the stack, the siblings wait
            push_node(src, ray.d);
            pushed = true;
          }
        };
        if (leaf) {
          // A leaf family commits along ONE path through the triangle of hypotheses: child r was traced by lane (r, b) from
          // the state r * c0 + b * 2N draws on, and the state it ends in is the start state of (r + 1, b) if it drew c0
          // numbers, of (r + 1, b + 1) if it drew c0 + 2N (a diffuse hit that survives roulette), of nobody otherwise (a
          // miss draws c0 - 1: the chain resumes next round).  So every lane knows its successor from its own draw count
          // (no states compared, no ballots), the wave follows the path with one lane read per child, and the products
          // hit_color * value (render.py:137) are formed by all lanes at once and only ADDED in child order.  (Leaf
          // children never need children of their own: render.py:100-101.)  Round 5: the commit was 5 200 of a round's
          // 32 000 cycles as a loop of ballot, find-first, ten lane reads and three dependent multiply-adds per child.
          int nxt = -1;  // successor hypothesis lane; -2: the family ends with this child; -1: no lane traced the next child from here
          {
            const unsigned extra = pcg.n - c0;
            if (act && !sib && (extra == 0u || extra == 2u * (unsigned)N)) {
              const int r2 = tri_row + 1, b2 = tri_col + (extra ? 1 : 0);
              nxt = r2 < nrows ? r2 * (r2 + 1) / 2 + b2 : -2;
            }
          }
          const V3 prod = {t_hc.x * o_ret.x, t_hc.y * o_ret.y, t_hc.z * o_ret.z};
          int cur = 0, k = 0;  // (wave-uniform) the lane of the child being committed; children committed
          for (;;) {
            const V3 pv = rl_v3(prod, cur);
            t_cum.x = t_cum.x + pv.x;
            t_cum.y = t_cum.y + pv.y;
            t_cum.z = t_cum.z + pv.z;
            ++k;
            const int nx = __builtin_amdgcn_readlane(nxt, cur);
            if (nx < 0) break;
            cur = nx;
          }
          prays += (unsigned long long)k;
          t_next += k;
          expect = rl_u64(pcg.state, cur);
          cpred = (unsigned)__builtin_amdgcn_readlane((int)pcg.n, cur);
          fam_draws = (unsigned)__builtin_amdgcn_readlane((int)ahead, cur) + cpred;
        } else {
          for (int r = 0; r < nrows && !pushed; ++r) {
            const unsigned long long m = __ballot(act && row == r && st_start == expect);
            if (!m) break;  // nobody traced child r from the right state: next round
            commit_child(__ffsll((long long)m) - 1);
          }
        }
        if (leaf && t_next == N && nrows == N && fam_draws >= (unsigned)N * c0)
        {
          const int bf = (int)((fam_draws - (unsigned)N * c0) / (2u * (unsigned)N));
          b_count -= b_count >> 4;  // (the recent past counts: sixteen families' memory)
          if (lane == bf) b_count += 256;
        }
#ifdef PT_DEBUG_TIME
        if (fused) dbg_fused += 1;
        if (fused && t_next != N) dbg_fused_incomplete += 1;
#endif
        if (fused && t_next == N) {
          // the leaf family is complete: its node returns now -- and every ancestor it completes --, and the next child of the
          // node that is then innermost may be there already
          for (int u = 0; u < up; ++u) (void)pop_node();
          cpred = base_draws();
          const unsigned long long m = __ballot(sib && st_start == expect);
          if (m) commit_child(__ffsll((long long)m) - 1);
#ifdef PT_DEBUG_TIME
          if (m) dbg_fused_hit += 1;
#endif
        }
        gstate = expect;
        if (pushed) cpred = base_draws();
        PT_TT(5);
      }
      // imagetracer.py:94-97
      if (S > 0) {
        cum_pix.x = cum_pix.x + sample_ret.x;
        cum_pix.y = cum_pix.y + sample_ret.y;
        cum_pix.z = cum_pix.z + sample_ret.z;
      } else {
        cum_pix = sample_ret;
      }
    }
    if (S > 0) {  // imagetracer.py:99-101
      const double k = 1.0 / (double)(S * S);
      cum_pix.x = cum_pix.x * k;
      cum_pix.y = cum_pix.y * k;
      cum_pix.z = cum_pix.z * k;
    }
    if (lane == 0) store_pixel(a, pix, cum_pix);
    nrays += prays;
#ifdef PT_DEBUG_TIME
    dbg_committed += prays;
    if (dbg_rounds_pix > dbg_max_rounds) dbg_max_rounds = dbg_rounds_pix;
#endif
  }
#ifdef PT_DEBUG_TIME
  if (lane == 0) {
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
    atomicAdd(pt_queue(a) + 12, dbg_leaf_rounds | (dbg_fused << 24) | (dbg_fused_hit << 44));
    atomicAdd(pt_queue(a) + 13, dbg_committed);
    atomicAdd(pt_queue(a) + 14, dbg_traced | (dbg_fused_incomplete << 40));
    atomicMax(pt_queue(a) + 15, dbg_max_rounds);
  }
#endif
  add_ray_count(a, lane == 0 ? nrays : 0ULL);
}

#ifndef PT_TREE_WAVES
#define PT_TREE_WAVES 2
#endif
template <bool SMALL = false, bool SLDS = false>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_TREE_WAVES, 8))) void pt_path_tree_kernel(const PtKArgs a) {
  path_tree<SMALL, SLDS>(a);
}
