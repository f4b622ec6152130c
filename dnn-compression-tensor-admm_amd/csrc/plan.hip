// Host side of libtadmm_hip.so: the C ABI of include/tadmm.h and the phase scheduler that turns a set
// of layers into grouped launches.
//
// One ADMM projection (reference ADMM.update, admm.py:42-78) over L layers runs as
//     unfold (1 launch)                                   T0 = unfold(W+U)
//     for TT step s = 0 .. max(d)-2, over every layer that has that step:
//         gram_partial, gram_reduce                       G = A A^T | A^T A              (fp64 MFMA)
//         jacobi_init, jacobi_tick x (sweeps*(nb-1))      eigen-decomposition of G       (fp64 MFMA)
//         eig_norms, eig_sort, eig_extract                top-r vectors, sigma
//         gemm                                            T_{s+1} = U_r^T A | core = A V S^-1 (fp32 MFMA)
//     gemm x (chain depth)                                Zmat = core_0 (core_1 (... T_{d-1}))
//     fold_update, resid_reduce                           Z, U += W-Z, ||W-Z||^2
// Layers are independent, so every launch is *grouped*: its blocks are mapped to
// (layer, local block) through a BlockRef table built once at plan creation.  All descriptors live in
// the caller-provided workspace; tadmm_plan_run allocates nothing.
#include "host.h"
#include "filter_host.h"

#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace tadmm;

// ------------------------------------------------------------------------------------------------
// layer geometry
// ------------------------------------------------------------------------------------------------
struct StepGeom {
  int m = 0;            // rows of the unfolding  r_s * n_s
  int64_t cols = 0;     // columns of the unfolding
  int r = 0;            // kept rank r_{s+1}
  bool skip = false;    // identity step (Z-only mode)
  bool trans = false;   // m > cols : eigen-solve on A^T A
  int N = 0, Npad = 0, ld = 0, nb = 0;
  int nt = 0, ksplit = 0, kchunk = 0;
};

struct LayerGeom {
  tadmm_layer_desc desc;
  int64_t numel = 0;
  int O = 0, I = 0, K2 = 1;   // K2 > 1 -> conv permutation
  int d = 0;
  std::vector<StepGeom> steps;  // d-1 entries
};

static int clamp_ranks(tadmm_layer_desc* d) {
  int changed = 0;
  int64_t tail = 1;
  for (int i = 0; i < d->d; ++i) tail *= d->tt_shapes[i];
  // reference ttd.py:15-19: unfolding i is (r_i*n_i) x (tail / n_i ...) with the *already clamped* r_i
  int64_t rest = tail;
  for (int i = 0; i + 1 < d->d; ++i) {
    rest /= d->tt_shapes[i];
    const int64_t m = (int64_t)d->ranks[i] * d->tt_shapes[i];
    const int64_t ns = std::min<int64_t>(m, rest);
    if (ns < d->ranks[i + 1]) { d->ranks[i + 1] = (int32_t)ns; ++changed; }
  }
  return changed;
}

static int build_geom(tadmm_handle h, const tadmm_layer_desc& din, LayerGeom& g) {
  g.desc = din;
  tadmm_layer_desc& d = g.desc;
  if (d.ndim != 2 && d.ndim != 4) CTX_FAIL(h, TADMM_ERR_INVALID, "ndim must be 2 or 4 (got %d)", d.ndim);
  g.numel = 1;
  for (int i = 0; i < d.ndim; ++i) {
    if (d.dims[i] <= 0) CTX_FAIL(h, TADMM_ERR_INVALID, "non-positive dim");
    g.numel *= d.dims[i];
  }
  if (d.kind == TADMM_KIND_TUCKER2) CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "Tucker layers use tadmm_tucker_* (not in a TT plan)");
  g.O = (int)d.dims[0];
  g.I = (int)d.dims[1];
  g.K2 = 1;
  if (d.kind == TADMM_KIND_TT_CONV) {
    if (d.ndim != 4) CTX_FAIL(h, TADMM_ERR_INVALID, "TT_CONV needs a 4-D weight");
    g.K2 = (int)(d.dims[2] * d.dims[3]);
  } else if (d.kind == TADMM_KIND_SVD) {
    // admm.py:129-149: squeeze to (O,I); a 4-D weight must be 1x1
    if (d.ndim == 4 && d.dims[2] * d.dims[3] != 1) CTX_FAIL(h, TADMM_ERR_INVALID, "SVD format needs a 1x1 kernel");
    d.d = 2;
    d.tt_shapes[0] = g.O; d.tt_shapes[1] = g.I;
    const int r = d.ranks[0];
    d.ranks[0] = 1; d.ranks[1] = r; d.ranks[2] = 1;
  }
  if (d.d < 2 || d.d > TADMM_MAX_MODES) CTX_FAIL(h, TADMM_ERR_INVALID, "number of TT modes must be in [2,%d]", TADMM_MAX_MODES);
  int64_t prod = 1;
  for (int i = 0; i < d.d; ++i) {
    if (d.tt_shapes[i] <= 0) CTX_FAIL(h, TADMM_ERR_INVALID, "non-positive tt_shape");
    prod *= d.tt_shapes[i];
  }
  if (prod != g.numel) CTX_FAIL(h, TADMM_ERR_INVALID, "prod(tt_shapes)=%lld != numel=%lld", (long long)prod, (long long)g.numel);
  if (d.ranks[0] != 1 || d.ranks[d.d] != 1) CTX_FAIL(h, TADMM_ERR_INVALID, "boundary TT ranks must be 1");
  for (int i = 0; i <= d.d; ++i) if (d.ranks[i] <= 0) CTX_FAIL(h, TADMM_ERR_INVALID, "non-positive rank");
  clamp_ranks(&d);
  g.d = d.d;
  g.steps.resize(d.d - 1);
  int64_t rest = g.numel;
  for (int s = 0; s + 1 < d.d; ++s) {
    StepGeom& st = g.steps[s];
    rest /= d.tt_shapes[s];
    st.m = d.ranks[s] * d.tt_shapes[s];
    st.cols = rest;
    st.r = d.ranks[s + 1];
    st.trans = (int64_t)st.m > st.cols;
    st.N = (int)std::min<int64_t>(st.m, st.cols);
    st.skip = (d.flags & TADMM_FLAG_SKIP_ROTATIONS) && !st.trans && st.r == st.m;
    st.Npad = (int)align_up(st.N, 4 * kJB);     // whole super-pairs of 2 x 16 columns
    st.nb = st.Npad / kJB;
    // row length of the eigen-solver's X image: whole 1 KiB chunks (tick3 wants ld % 64 == 0)
    st.ld = eig_ld(st.N);
    st.nt = (st.N + 31) / 32;
    if (!st.skip && !jacobi_size_supported(st.N))
      CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "TT step %d: eigen-problem of size %d exceeds the Jacobi kernels (max %d)", s,
               st.N, kJacobiMaxN);
    const int64_t K = st.trans ? st.m : st.cols;
    const int ntp = st.nt * (st.nt + 1) / 2;
    // split-K only where a problem has too few tiles to matter beside the others of its level (levels batch
    // 15-30 problems): >= 64 workgroups per problem; big problems (ks = 1) write G directly, no reduce pass
    int ks = (64 + ntp - 1) / ntp;
    const int maxks = (int)std::max<int64_t>(1, (K + 255) / 256);
    ks = std::max(1, std::min(ks, maxks));
    // a workgroup's duration grows with its K chunk and it shares the CU's matrix cores with its neighbours:
    // cap the chunk so that the long reductions (K = 4608 beside K = 512) do not form the tail of the launch
    ks = std::max(ks, (int)((K + 2047) / 2048));
    st.kchunk = (int)align_up((K + ks - 1) / ks, 64);
    st.ksplit = (int)((K + st.kchunk - 1) / st.kchunk);
  }
  return TADMM_OK;
}

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
struct StepPlan {
  Phase gram_p, gram_r, eig_tick, eig_norm, eig_ext, proj;
  size_t eig_desc_off = 0;
  int neig = 0;
  int gsteps = 0;                 // ticks per global sweep = max(nb-1)
  size_t tick_lds = 0;            // dynamic LDS of the tick launches of this step
  int last_sweeps = 0;            // global sweeps the previous run needed (polls start 2 sweeps before that)
  bool super = false;             // LDS-resident super-pair kernel (all problems of the level fit)
  int mode = 0;                   // 0: pairs (tick1), 1: LDS super-pairs (tick2), 3: tick3 + self pass
  Phase eig_self;                 // mode 2: block map of the once-per-sweep self kernel (nb/2 workgroups per problem)
  int ld_max = 0;
  int npad_max = 0;
  size_t off_off = 0, done_off = 0;   // contiguous [neig][3] doubles / [neig] ints
  size_t prev_off_dev = 0;            // [neig] doubles of the convergence kernel
  std::vector<int> nb;            // per problem
  std::vector<int> row_len;       // per problem: ld of the X image (instrumented runs)
  std::vector<int> mid;           // per problem: 128 / 192 when it may take the direct route of tridiag_mid.hip, else 0
  std::vector<int> layer_of;      // problem -> layer
  // filtered eigen-solver (filter_host.h): the problems of this level it serves keep their slot in the eig group,
  // where their r' x r' Rayleigh-Ritz problem replaces the full N x N one; the full variants form the fallback group
  FilterGroup fg;
  std::vector<int> filt_of;       // filtered problem -> problem of the level
  size_t skip_off = 0;            // [neig] ints: eig group (set for filtered problems that went bad)
  size_t fb_skip_off = 0;         // [nf] ints: fallback group (1 = filtered result accepted)
  struct Fallback {
    Phase tick, self, norm, ext;
    int neig = 0, gsteps = 0, mode = 0, ld_max = 0, npad_max = 0, last_sweeps = 0;
    size_t tick_lds = 0, prev_off_dev = 0;
    std::vector<int> nb, row_len;
  } fb;
};

struct tadmm_plan_s {
  tadmm_handle h = nullptr;
  int n = 0;
  std::vector<LayerGeom> layers;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  Phase unfold, fold;
  size_t sweep_desc_off = 0;
  size_t resid_partial_off = 0;
  int total_sweep_blocks = 0;
  std::vector<StepPlan> steps;
  std::vector<Phase> recon;       // chain levels
  // per layer / step bookkeeping for queries
  std::vector<std::vector<size_t>> sigma_off;   // [layer][step] -> offset of sigma doubles (or SIZE_MAX)
  // timing
  bool timing = false;
  hipEvent_t ev[16];
  bool ev_made = false;
  PollCtx poll;                   // pipelined convergence poll (host.h)
  double last_ms[8] = {0};
  int last_sweeps = 0;
  double tol = 1e-9;
  int inner_sweeps = 1;
  bool debug = false;
  int max_global_sweeps = 40;
  // filtered eigen-solver statistics of the last run
  int filt_problems = 0, filt_fallbacks = 0, filt_stages = 0;
  FilterTiming ftm;
  JacobiTiming jtm;
  const int32_t* resid_index = nullptr;   // device map local layer -> slot of the caller's residual array (lanes)
  struct Lanes* lanes = nullptr;          // set on a parent plan that runs its layers as two concurrent lanes
};

// ------------------------------------------------------------------------------------------------
// Lanes.  One eigen-solve is a chain of dependent launches, and the grouped launches of a plan put the short chains
// of most layers behind the few long ones (ResNet-50: the 3x3 kernels of layer3/layer4 need ~3/4 of the launches).
// A plan whose table has both kinds is therefore run as TWO sub-plans on two streams of the device: lane 0 holds the
// long chains and runs on a high-priority stream at the pace it would have alone, lane 1 (everything else) fills the
// CUs those launches leave idle.  Each sub-plan polls its own convergence words, so lane 1 is driven by a worker thread
// that lives as long as the plan.  Measured on MI355X, ResNet-50 table: 10.2 ms as one plan, 8.6 ms as two lanes.
// ------------------------------------------------------------------------------------------------
struct Lanes {
  tadmm_plan_s* sub[2] = {nullptr, nullptr};
  std::vector<int> lane_of, local_of;
  hipStream_t st[2] = {nullptr, nullptr};
  hipEvent_t ev_begin = nullptr, ev_end[2] = {nullptr, nullptr};
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  int job = 0;                 // 0 idle, 1 run posted, 2 quit
  bool done = true;
  int a_update_u = 0, a_use_u = 0;
  double* a_resid = nullptr;
  int rc = 0;
};

namespace {

struct Built {
  // per layer buffers (offsets in the arena)
  std::vector<size_t> tbuf0, tbuf1, xt, gpart, vs, cores_ws;
};

}  // namespace

// Lays out the whole plan.  If `img` is null only sizes are computed.
static int layout_plan(tadmm_plan_s* P, const float* const* W, float* const* U, float* const* Z, float* const* cores,
                       HostImage* img, size_t desc_region, size_t* desc_bytes, size_t* total_bytes) {
  tadmm_handle h = P->h;
  const int n = P->n;
  Arena da(P->ws);             // descriptors + block maps: [0, desc_region)
  Arena ar(P->ws);             // data buffers: [desc_region, ...)
  ar.off = desc_region;
  auto dev = [&](size_t off) -> char* { return P->ws ? P->ws + off : nullptr; };

  // ---- data buffers per layer ----
  std::vector<size_t> tb0(n), tb1(n), xt(n), gp(n), vs(n), cw(n);
  std::vector<std::vector<size_t>> core_off(n);     // float offsets (bytes) of core_s inside arena or user buf
  std::vector<std::vector<float*>> core_ptr(n);     // device pointers of cores (incl. last = T_{d-1} when user buf)
  P->sigma_off.assign(n, {});
  for (int l = 0; l < n; ++l) {
    const LayerGeom& g = P->layers[l];
    tb0[l] = ar.take((size_t)g.numel * 4);
    tb1[l] = ar.take((size_t)g.numel * 4);
    size_t xtb = 0, gpb = 0, vsb = 0, cb = 0;
    for (const StepGeom& st : g.steps) {
      if (st.skip) continue;
      xtb = std::max(xtb, (size_t)st.Npad * st.ld * 8);
      gpb = std::max(gpb, (size_t)st.ksplit * (st.nt * (st.nt + 1) / 2) * 1024 * 8);
      if (st.trans) vsb = std::max(vsb, (size_t)st.N * st.r * 4);
      cb += align_up((size_t)st.m * st.r * 4, 256);
    }
    xt[l] = ar.take(xtb ? xtb : 256);
    gp[l] = ar.take(gpb ? gpb : 256);
    vs[l] = ar.take(vsb ? vsb : 256);
    cw[l] = ar.take(cb ? cb : 256);
    P->sigma_off[l].assign(g.steps.size(), (size_t)-1);
  }

  // ---- sweep descriptors (unfold / fold_update) ----
  std::vector<SweepDesc> sd(n);
  std::vector<BlockRef> smap;
  // T pointer tracking: cur[l] = buffer holding T_s
  std::vector<float*> cur(n), other(n);
  for (int l = 0; l < n; ++l) {
    const LayerGeom& g = P->layers[l];
    SweepDesc& s = sd[l];
    memset(&s, 0, sizeof s);
    s.W = W ? W[l] : nullptr; s.U = U ? U[l] : nullptr; s.Z = Z ? Z[l] : nullptr;
    s.T0 = (float*)dev(tb0[l]);
    s.O = g.O; s.I = g.I; s.K2 = g.K2;
    s.numel = g.numel;
    if (g.K2 > 1) {
      int ich = std::min(g.I, 512);
      while ((int64_t)g.K2 * (ich + 4) > 12288 && ich > 1) ich /= 2;
      s.ichunk = ich;
      s.nchunk = (g.I + ich - 1) / ich;
      s.nblk = g.O * s.nchunk;
    } else {
      s.ichunk = 8192;
      s.nchunk = (int)((g.numel + s.ichunk - 1) / s.ichunk);
      s.nblk = s.nchunk;
    }
    s.blk_begin = (int)smap.size();
    for (int b = 0; b < s.nblk; ++b) smap.push_back(BlockRef{l, b});
    cur[l] = (float*)dev(tb0[l]);
    other[l] = (float*)dev(tb1[l]);
  }
  P->total_sweep_blocks = (int)smap.size();
  P->resid_partial_off = ar.take((size_t)smap.size() * 8);

  // ---- TT steps ----
  // Levels: layer l runs its TT step s at level s + lvl_off[l].  The longest chains fix the number of
  // levels; shorter chains are shifted so that their eigen-solves share a level with problems of at
  // least their size (a level costs ~ (nb_max - 1) ticks per sweep whatever its population), on the
  // least populated such level -- e.g. ResNet-50: the 1x1 convs' single N=256/512 solve runs beside the
  // 3x3 convs' N=480/512 steps instead of beside their N<=32 first step.
  int maxsteps = 0;
  for (const LayerGeom& g : P->layers) maxsteps = std::max(maxsteps, (int)g.steps.size());
  std::vector<int> lvl_off(n, 0);
  {
    std::vector<int> lvl_nb(maxsteps, 0);
    std::vector<long> lvl_wgs(maxsteps, 0);
    std::vector<int> order(n);
    for (int l = 0; l < n; ++l) order[l] = l;
    auto heavy = [&](int l) { int m = 0; for (const StepGeom& st : P->layers[l].steps) if (!st.skip) m = std::max(m, st.nb); return m; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      const size_t ca = P->layers[a].steps.size(), cb = P->layers[b].steps.size();
      if (ca != cb) return ca > cb;
      return heavy(a) > heavy(b);
    });
    for (int l : order) {
      const LayerGeom& g = P->layers[l];
      const int len = (int)g.steps.size();
      int best = 0; double best_cost = 1e300;
      for (int off = 0; off + len <= maxsteps; ++off) {
        // cost = added ticks (levels whose nb_max grows) first, then population of the touched levels
        double cost = 0;
        for (int s = 0; s < len; ++s) {
          if (g.steps[s].skip) continue;
          const int lv = off + s;
          cost += 1e6 * std::max(0, g.steps[s].nb - lvl_nb[lv]) + (double)lvl_wgs[lv] * g.steps[s].nb;
        }
        if (cost < best_cost) { best_cost = cost; best = off; }
      }
      lvl_off[l] = best;
      for (int s = 0; s < len; ++s) {
        if (g.steps[s].skip) continue;
        lvl_nb[best + s] = std::max(lvl_nb[best + s], g.steps[s].nb);
        lvl_wgs[best + s] += g.steps[s].nb / 2;
      }
    }
  }
  P->steps.assign(maxsteps, StepPlan());
  // recon chain bookkeeping: for each layer the list of (core ptr, m, r, cols) of non-skipped steps
  struct RecOp { const float* core; int m, r; int64_t cols; };
  std::vector<std::vector<RecOp>> recops(n);
  std::vector<const float*> lastT(n, nullptr);

  for (int lev = 0; lev < maxsteps; ++lev) {
    StepPlan& sp = P->steps[lev];
    std::vector<GramDesc> gd;
    std::vector<EigDesc> ed;
    std::vector<GemmDesc> pd;
    std::vector<BlockRef> m_gp, m_gr, m_proj;
    std::vector<int> gp_cost;     // per problem: K chunk of its Gram workgroups (longest first in the block map)
    std::vector<int> layer_of;
    for (int l = 0; l < n; ++l) {
      const LayerGeom& g = P->layers[l];
      const int s = lev - lvl_off[l];
      if (s < 0 || s >= (int)g.steps.size()) continue;
      const StepGeom& st = g.steps[s];
      if (st.skip) {   // identity: T_{s+1} aliases T_s, no core
        if (cores && cores[l]) CTX_FAIL(h, TADMM_ERR_INVALID, "TADMM_FLAG_SKIP_ROTATIONS is incompatible with cores output");
        continue;
      }
      layer_of.push_back(l);
    }
    sp.neig = (int)layer_of.size();
    sp.layer_of = layer_of;
    sp.off_off = ar.take((size_t)std::max(1, sp.neig) * 3 * 8);
    sp.done_off = ar.take((size_t)std::max(1, sp.neig) * 4);
    sp.prev_off_dev = ar.take((size_t)std::max(1, sp.neig) * 8);
    sp.skip_off = ar.take((size_t)std::max(1, sp.neig) * 4);
    sp.gsteps = 0;
    std::vector<FilterSpec> fspecs;
    sp.filt_of.clear();
    for (int p = 0; p < sp.neig; ++p) {
      const int l = layer_of[p];
      const LayerGeom& g = P->layers[l];
      const int s = lev - lvl_off[l];
      const StepGeom& st = g.steps[s];
      const bool last_step = (s + 2 == g.d);
      // core storage
      float* core_dev;
      float* tnext;
      if (cores && cores[l]) {
        size_t off = 0;
        for (int j = 0; j < s; ++j) off += (size_t)g.steps[j].m * g.steps[j].r;
        core_dev = cores[l] + off;
        tnext = last_step ? (cores[l] + off + (size_t)st.m * st.r) : other[l];
      } else {
        size_t off = 0;
        for (int j = 0; j < s; ++j) if (!g.steps[j].skip) off += align_up((size_t)g.steps[j].m * g.steps[j].r * 4, 256);
        core_dev = (float*)dev(cw[l] + off);
        tnext = other[l];
      }
      const float* Tin = cur[l];
      // Gram
      GramDesc gdsc;
      memset(&gdsc, 0, sizeof gdsc);
      gdsc.A = Tin; gdsc.m = st.m; gdsc.n = (int)st.cols; gdsc.trans = st.trans ? 1 : 0;
      gdsc.N = st.N; gdsc.K = st.trans ? st.m : (int)st.cols;
      gdsc.nt = st.nt; gdsc.ksplit = st.ksplit; gdsc.kchunk = st.kchunk;
      gdsc.partial = (double*)dev(gp[l]);
      gdsc.G = (double*)dev(xt[l]);
      gdsc.Npad = st.Npad; gdsc.ld = st.ld;
      const int ntp = st.nt * (st.nt + 1) / 2;
      for (int b = 0; b < st.ksplit * ntp; ++b) m_gp.push_back(BlockRef{p, b});
      gp_cost.push_back(st.kchunk);
      const int64_t gtot = (int64_t)st.Npad * st.ld;
      if (st.ksplit > 1)
        for (int b = 0; b < (int)((gtot + 1023) / 1024); ++b) m_gr.push_back(BlockRef{p, b});
      gd.push_back(gdsc);
      // eig
      EigDesc e;
      memset(&e, 0, sizeof e);
      e.XT = (double*)dev(xt[l]);
      e.N = st.N; e.Npad = st.Npad; e.ld = st.ld; e.nb = st.nb;
      e.off = (double*)dev(sp.off_off) + 3 * p;
      e.done = (int32_t*)dev(sp.done_off) + p;
      const size_t lam_off = ar.take((size_t)st.Npad * 8);
      const size_t ord_off = ar.take((size_t)st.Npad * 4);
      const size_t sig_off = ar.take((size_t)st.r * 8);
      const size_t sblk_off = ar.take((size_t)(st.Npad / 16) * 256 * 8);
      P->sigma_off[l][s] = sig_off;
      e.lam = (double*)dev(lam_off);
      e.order = (int32_t*)dev(ord_off);
      e.sigma = (double*)dev(sig_off);
      e.r = st.r;
      e.mode = st.trans ? 1 : 0;
      e.out_a = st.trans ? (float*)dev(vs[l]) : core_dev;
      e.out_b = st.trans ? tnext : nullptr;
      e.evec_out = nullptr;
      e.sblk = (double*)dev(sblk_off);
      ed.push_back(e);
      const bool zonly = (g.desc.flags & TADMM_FLAG_SKIP_ROTATIONS) && !(cores && cores[l]);
      if (const int rp = filter_block_size(st.N, st.r)) {
        FilterSpec fs;
        fs.N = st.N; fs.Npad = st.Npad; fs.ldg = st.ld; fs.r = st.r; fs.rp = rp;
        fs.G = e.XT; fs.mode = e.mode; fs.ldo = e.ldo; fs.out_a = e.out_a; fs.out_b = e.out_b; fs.sigma = e.sigma;
        fs.skip_slot = (int32_t*)dev(sp.skip_off) + p;
        fspecs.push_back(fs);
        sp.filt_of.push_back(p);
      } else if (const int rpc = complement_block_size(st.N, st.r, st.trans, zonly)) {
        // complement route (complement.hip): the filter runs on the reflected image G' for the N - r DISCARDED vectors
        FilterSpec fs;
        const int kdis = st.N - st.r;
        fs.N = st.N; fs.Npad = st.Npad; fs.ldg = st.ld; fs.r = kdis; fs.rp = rpc;
        fs.G = (const double*)dev(ar.take((size_t)st.Npad * st.ld * 8));
        fs.g_orig = e.XT; fs.comp_r = st.r; fs.sigma_layer = e.sigma;
        fs.mode = 4; fs.ldo = e.ldo; fs.out_a = e.out_a; fs.out_b = nullptr;
        fs.sigma = (double*)dev(ar.take((size_t)align_up(kdis, 32) * 8));      // (Ritz values of G': scratch)
        fs.skip_slot = (int32_t*)dev(sp.skip_off) + p;
        fspecs.push_back(fs);
        sp.filt_of.push_back(p);
      }
      // projection GEMM
      GemmDesc pg;
      memset(&pg, 0, sizeof pg);
      pg.alpha = 1.f; pg.beta = 0.f;
      if (!st.trans) {   // T_{s+1}[r x cols] = Uf^T[r x m] * A[m x cols]
        pg.A = core_dev; pg.a_rs = 1; pg.a_cs = st.r;
        pg.B = Tin; pg.b_rs = st.cols; pg.b_cs = 1;
        pg.C = tnext; pg.c_rs = st.cols; pg.c_cs = 1;
        pg.M = st.r; pg.N = (int)st.cols; pg.K = st.m;
      } else {           // core[m x r] = A[m x n] * Vs[n x r]
        pg.A = Tin; pg.a_rs = st.cols; pg.a_cs = 1;
        pg.B = (const float*)dev(vs[l]); pg.b_rs = st.r; pg.b_cs = 1;
        pg.C = core_dev; pg.c_rs = st.r; pg.c_cs = 1;
        pg.M = st.m; pg.N = st.r; pg.K = (int)st.cols;
      }
      pg.tiles_m = (pg.M + kGemmBM - 1) / kGemmBM;
      pg.tiles_n = (pg.N + kGemmBN - 1) / kGemmBN;
      for (int b = 0; b < pg.tiles_m * pg.tiles_n; ++b) m_proj.push_back(BlockRef{p, b});
      pd.push_back(pg);
      recops[l].push_back(RecOp{core_dev, st.m, st.r, st.cols});
      // advance the T chain
      if (tnext == other[l]) std::swap(cur[l], other[l]);
      else { cur[l] = tnext; }
    }
    auto place = [&](Phase& ph, const void* descs, size_t dbytes, int nprob, const std::vector<BlockRef>& map) {
      ph.nprob = nprob;
      ph.nblocks = (int)map.size();
      ph.desc_off = da.take(std::max<size_t>(dbytes, 16));
      ph.map_off = da.take(std::max<size_t>(map.size() * sizeof(BlockRef), 16));
      if (img) {
        if (dbytes) img->put(ph.desc_off, descs, dbytes);
        if (!map.empty()) img->put(ph.map_off, map.data(), map.size() * sizeof(BlockRef));
      }
    };
    std::stable_sort(m_gp.begin(), m_gp.end(),
                     [&](const BlockRef& a, const BlockRef& b) { return gp_cost[a.prob] > gp_cost[b.prob]; });
    place(sp.gram_p, gd.data(), gd.size() * sizeof(GramDesc), sp.neig, m_gp);
    sp.gram_r = sp.gram_p;
    sp.gram_r.map_off = da.take(std::max<size_t>(m_gr.size() * sizeof(BlockRef), 16));
    sp.gram_r.nblocks = (int)m_gr.size();
    if (img && !m_gr.empty()) img->put(sp.gram_r.map_off, m_gr.data(), m_gr.size() * sizeof(BlockRef));
    // ---- filtered problems: Rayleigh-Ritz problem in the eig group, full problem in the fallback group ----
    std::vector<EigDesc> ed_fb;
    {
      const int nf = (int)fspecs.size();
      sp.fb_skip_off = ar.take((size_t)std::max(1, nf) * 4);
      for (int i = 0; i < nf; ++i) fspecs[i].fb_skip = (int32_t*)dev(sp.fb_skip_off) + i;
      std::vector<FilterRR> rr;
      filter_layout(sp.fg, fspecs, da, ar, dev, img, rr);
      for (int i = 0; i < nf; ++i) {
        const int p = sp.filt_of[i];
        ed_fb.push_back(ed[p]);
        // the fallback problem needs convergence words of its own (the group's are used by the Rayleigh-Ritz solve)
        ed_fb.back().off = (double*)dev(ar.take(3 * 8));
        ed_fb.back().done = (int32_t*)dev(ar.take(4));
        EigDesc e = rr[i].desc;
        e.off = ed[p].off; e.done = ed[p].done;
        ed[p] = e;
      }
    }
    struct EigMaps { std::vector<BlockRef> tick, self, norm, ext; std::vector<int> nb, row_len, mid; int gsteps = 0, mode = 0, ld_max = 0, npad_max = 0; size_t tick_lds = 0; };
    auto build_maps = [&](const std::vector<EigDesc>& descs) {
      EigMaps m;
      for (const EigDesc& e : descs) { m.ld_max = std::max(m.ld_max, e.ld); m.npad_max = std::max(m.npad_max, e.Npad); }
      // tick shape of the group: LDS-resident super-pairs when every problem fits, else plain pairs
      m.mode = descs.empty() ? 0 : choose_jacobi_mode(m.ld_max);
      const bool super = m.mode >= 1;
      m.tick_lds = m.mode == 1 ? jacobi_tick2_lds_bytes(m.ld_max) : jacobi_tick_lds_bytes(m.ld_max);
      for (int pq = 0; pq < (int)descs.size(); ++pq) {
        const EigDesc& e = descs[pq];
        const int units = super ? e.nb / 2 : e.nb;       // players of the tournament
        m.nb.push_back(units);
        m.row_len.push_back(e.ld);
        m.mid.push_back((e.scratch && eig_mid_direct_size(e.N) && e.N == e.Npad) ? e.N : 0);
        m.gsteps = std::max(m.gsteps, units - 1);
        for (int b = 0; b < units / 2; ++b) m.tick.push_back(BlockRef{pq, b});
        if (m.mode >= 2) for (int b = 0; b < units; ++b) m.self.push_back(BlockRef{pq, b});
        for (int b = 0; b < (e.Npad + 3) / 4; ++b) m.norm.push_back(BlockRef{pq, b});
        for (int b = 0; b < (e.r + 3) / 4; ++b) m.ext.push_back(BlockRef{pq, b});
      }
      xcd_group(m.tick);
      xcd_group(m.self);
      return m;
    };
    auto place_map = [&](Phase& ph, const Phase& like, const std::vector<BlockRef>& map) {
      ph = like;
      ph.map_off = da.take(std::max<size_t>(map.size() * sizeof(BlockRef), 16));
      ph.nblocks = (int)map.size();
      if (img && !map.empty()) img->put(ph.map_off, map.data(), map.size() * sizeof(BlockRef));
    };
    const bool align_sweeps = align_sweeps_on();
    {
      EigMaps m = build_maps(ed);
      if (m.mode >= 2 && align_sweeps) for (EigDesc& e : ed) e.period = m.gsteps;
      sp.mode = m.mode; sp.super = m.mode >= 1; sp.tick_lds = m.tick_lds; sp.gsteps = m.gsteps;
      sp.ld_max = m.ld_max; sp.npad_max = m.npad_max; sp.nb = m.nb; sp.row_len = m.row_len; sp.mid = m.mid;
      place(sp.eig_tick, ed.data(), ed.size() * sizeof(EigDesc), sp.neig, m.tick);
      sp.eig_desc_off = sp.eig_tick.desc_off;
      place_map(sp.eig_self, sp.eig_tick, m.self);
      place_map(sp.eig_norm, sp.eig_tick, m.norm);
      place_map(sp.eig_ext, sp.eig_tick, m.ext);
    }
    {
      EigMaps m = build_maps(ed_fb);
      if (m.mode >= 2 && align_sweeps) for (EigDesc& e : ed_fb) e.period = m.gsteps;
      StepPlan::Fallback& fb = sp.fb;
      fb.neig = (int)ed_fb.size();
      fb.mode = m.mode; fb.tick_lds = m.tick_lds; fb.gsteps = m.gsteps; fb.ld_max = m.ld_max; fb.npad_max = m.npad_max;
      fb.nb = m.nb; fb.row_len = m.row_len;
      fb.prev_off_dev = ar.take((size_t)std::max(1, fb.neig) * 8);
      place(fb.tick, ed_fb.data(), ed_fb.size() * sizeof(EigDesc), fb.neig, m.tick);
      place_map(fb.self, fb.tick, m.self);
      place_map(fb.norm, fb.tick, m.norm);
      place_map(fb.ext, fb.tick, m.ext);
    }
    place(sp.proj, pd.data(), pd.size() * sizeof(GemmDesc), sp.neig, m_proj);
  }

  // ---- reconstruction chain, right to left:  R = T_{d-1};  R <- core_s * R ----
  for (int l = 0; l < n; ++l) lastT[l] = cur[l];
  size_t maxchain = 0;
  for (int l = 0; l < n; ++l) maxchain = std::max(maxchain, recops[l].size());
  P->recon.assign(maxchain, Phase());
  std::vector<const float*> rcur(n);
  std::vector<float*> rfree(n);
  for (int l = 0; l < n; ++l) {
    rcur[l] = lastT[l];
    // the "other" ping-pong buffer is free; if T_{d-1} lives in the user's cores buffer both are free
    float* b0 = (float*)dev(tb0[l]);
    float* b1 = (float*)dev(tb1[l]);
    rfree[l] = (rcur[l] == b0) ? b1 : b0;
  }
  for (size_t lev = 0; lev < maxchain; ++lev) {
    std::vector<GemmDesc> rd;
    std::vector<BlockRef> rmap;
    for (int l = 0; l < n; ++l) {
      if (lev >= recops[l].size()) continue;
      const RecOp& op = recops[l][recops[l].size() - 1 - lev];
      GemmDesc g;
      memset(&g, 0, sizeof g);
      g.alpha = 1.f;
      g.A = op.core; g.a_rs = op.r; g.a_cs = 1;
      g.B = rcur[l]; g.b_rs = op.cols; g.b_cs = 1;
      g.C = rfree[l]; g.c_rs = op.cols; g.c_cs = 1;
      g.M = op.m; g.N = (int)op.cols; g.K = op.r;
      g.tiles_m = (g.M + kGemmBM - 1) / kGemmBM;
      g.tiles_n = (g.N + kGemmBN - 1) / kGemmBN;
      const int p = (int)rd.size();
      for (int b = 0; b < g.tiles_m * g.tiles_n; ++b) rmap.push_back(BlockRef{p, b});
      rd.push_back(g);
      // ping-pong: the old input buffer becomes free unless it is a user buffer
      float* b0 = (float*)dev(tb0[l]);
      float* b1 = (float*)dev(tb1[l]);
      const float* produced = rfree[l];
      rfree[l] = (produced == b0) ? b1 : b0;
      rcur[l] = produced;
    }
    Phase& ph = P->recon[lev];
    ph.nprob = (int)rd.size();
    ph.nblocks = (int)rmap.size();
    ph.desc_off = da.take(std::max<size_t>(rd.size() * sizeof(GemmDesc), 16));
    ph.map_off = da.take(std::max<size_t>(rmap.size() * sizeof(BlockRef), 16));
    if (img) {
      if (!rd.empty()) img->put(ph.desc_off, rd.data(), rd.size() * sizeof(GemmDesc));
      if (!rmap.empty()) img->put(ph.map_off, rmap.data(), rmap.size() * sizeof(BlockRef));
    }
  }
  for (int l = 0; l < n; ++l) sd[l].Zmat = rcur[l];

  // ---- sweep phase descriptors ----
  P->sweep_desc_off = da.take(sd.size() * sizeof(SweepDesc));
  P->unfold.desc_off = P->sweep_desc_off;
  P->unfold.nprob = n;
  P->unfold.nblocks = (int)smap.size();
  P->unfold.map_off = da.take(smap.size() * sizeof(BlockRef));
  P->fold = P->unfold;
  if (img) {
    img->put(P->sweep_desc_off, sd.data(), sd.size() * sizeof(SweepDesc));
    img->put(P->unfold.map_off, smap.data(), smap.size() * sizeof(BlockRef));
  }
  *desc_bytes = align_up(da.off, 4096);
  *total_bytes = align_up(ar.off, 256);   // when desc_region == 0 this is the data size alone
  return TADMM_OK;
}

__global__ void square_copy_kernel(const double* __restrict__ in, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * in[i];
}
static void tadmm_square_copy(const double* in, double* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(square_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int tadmm_version(void) { return 100; }

int tadmm_abi_sizes(int* layer_desc_bytes, int* gemm_desc_bytes) {
  if (layer_desc_bytes) *layer_desc_bytes = (int)sizeof(tadmm_layer_desc);
  if (gemm_desc_bytes) *gemm_desc_bytes = (int)sizeof(tadmm_gemm_desc);
  return TADMM_OK;
}

int tadmm_chain_desc_bytes(void) { return (int)sizeof(tadmm_chain_desc); }

int tadmm_create(int device, tadmm_handle* out) {
  if (!out) return TADMM_ERR_INVALID;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  tadmm_handle h = new tadmm_ctx_s();
  h->device = device;
  *out = h;
  if (e != hipSuccess || device < 0 || device >= count) {
    h->err = std::string("no usable HIP device ") + std::to_string(device) + " (" + hipGetErrorString(e) + ")";
    return TADMM_ERR_HIP;
  }
  return TADMM_OK;
}

int tadmm_destroy(tadmm_handle h) {
  delete h;
  return TADMM_OK;
}

const char* tadmm_last_error(tadmm_handle h) { return h ? h->err.c_str() : "null handle"; }

int tadmm_tt_clamp_ranks(tadmm_layer_desc* desc) {
  if (!desc || desc->d < 2 || desc->d > TADMM_MAX_MODES) return TADMM_ERR_INVALID;
  return clamp_ranks(desc);
}

static int single_workspace_bytes(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, size_t* bytes) {
  if (!h || !descs || !bytes || n_layers <= 0) return TADMM_ERR_INVALID;
  tadmm_plan_s P;
  P.h = h;
  P.n = n_layers;
  P.layers.resize(n_layers);
  for (int l = 0; l < n_layers; ++l) {
    int rc = build_geom(h, descs[l], P.layers[l]);
    if (rc) return rc;
  }
  P.ws = nullptr;
  size_t db = 0, data = 0;
  int rc = layout_plan(&P, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &db, &data);
  if (rc) return rc;
  *bytes = db + data + 4096;
  return TADMM_OK;
}

static int single_create(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, const float* const* W,
                         float* const* U, float* const* Z, float* const* cores, void* workspace, size_t workspace_bytes,
                         tadmm_plan* out) {
  DeviceGuard device_guard(h);
  if (!h || !descs || !W || !U || !Z || !workspace || !out || n_layers <= 0) return TADMM_ERR_INVALID;
  tadmm_plan_s* P = new tadmm_plan_s();
  P->h = h;
  P->n = n_layers;
  P->layers.resize(n_layers);
  for (int l = 0; l < n_layers; ++l) {
    int rc = build_geom(h, descs[l], P->layers[l]);
    if (rc) { delete P; return rc; }
  }
  P->ws = (char*)workspace;
  P->ws_bytes = workspace_bytes;
  HostImage img;
  size_t need = 0, db = 0, data0 = 0;
  int rc = layout_plan(P, W, U, Z, cores, nullptr, 0, &db, &data0);      // pass 1: size of the descriptor region
  if (rc) { delete P; return rc; }
  rc = layout_plan(P, W, U, Z, cores, &img, db, &db, &need);             // pass 2: real offsets
  if (rc) { delete P; return rc; }
  if (need > workspace_bytes) {
    delete P;
    CTX_FAIL(h, TADMM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  }
  // descriptors were written at their arena offsets in the host image; upload the covered range
  if (!img.bytes.empty()) {
    hipError_t e = hipMemcpy(P->ws, img.bytes.data(), img.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete P; CTX_FAIL(h, TADMM_ERR_HIP, "descriptor upload failed: %s", hipGetErrorString(e)); }
  }
  for (const StepPlan& sp : P->steps) {      // Rayleigh-Ritz images / skip words start from zero
    hipError_t e = hipSuccess;
    if (sp.fg.zero_bytes) e = hipMemset(P->ws + sp.fg.zero_off, 0, sp.fg.zero_bytes);
    if (e == hipSuccess) e = hipMemset(P->ws + sp.skip_off, 0, (size_t)std::max(1, sp.neig) * 4);
    if (e == hipSuccess) e = hipMemset(P->ws + sp.fb_skip_off, 0, (size_t)std::max(1, sp.fg.nf) * 4);
    if (e != hipSuccess) { delete P; CTX_FAIL(h, TADMM_ERR_HIP, "workspace clear failed: %s", hipGetErrorString(e)); }
  }
  if (const char* e = getenv("TADMM_JACOBI_TOL")) P->tol = atof(e);
  if (getenv("TADMM_DEBUG")) P->debug = true;
  size_t maxe = 1;
  for (const StepPlan& sp : P->steps) maxe = std::max<size_t>(maxe, sp.neig);
  {
    const hipError_t e = P->poll.create(maxe);
    if (e != hipSuccess) { delete P; CTX_FAIL(h, TADMM_ERR_HIP, "poll buffers: %s", hipGetErrorString(e)); }
  }
  *out = P;
  return TADMM_OK;
}

static int single_run(tadmm_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream_);

// ---- lanes: split rule, creation, the worker of lane 1 ----
// modelled latency (us) of one eigen-problem alone on the device; mirrors tadmm/sched.py problem_latency_us
static double step_latency_us_full_or_filtered(int N, int r) {
  const int npad = (int)align_up(N, 32);
  if (npad <= 64) return 100.0;
  const int rp = filter_block_size(N, r);
  if (rp) return 7.0 * (8 * 35.0 + 170.0) + (rp / 16 - 1) * 7.0 * (9.0 + 0.02 * (double)align_up(rp, 128));
  return (npad / 16 - 1) * 11.0 * (9.0 + 0.02 * (double)align_up(N, 128));
}

// lane 0 = layers whose chain of eigen-solves is at least `thr` of the longest; lane 1 = the rest.  One lane (return
// false) when either side would be (nearly) empty, when asked to (TADMM_LANES=1) or for small tables.
static bool lane_split(tadmm_handle h, int n, const tadmm_layer_desc* descs, std::vector<int>& lane_of) {
  lane_of.assign(n, 0);
  if (const char* e = getenv("TADMM_LANES")) if (atoi(e) == 1) return false;
  if (n < 4) return false;
  double thr = 0.6;
  if (const char* e = getenv("TADMM_LANE_THRESHOLD")) thr = atof(e);
  std::vector<double> lat(n, 0.0);
  std::vector<int> nfilt(n, 0);
  double lmax = 0.0;
  for (int l = 0; l < n; ++l) {
    LayerGeom g;
    if (build_geom(h, descs[l], g)) return false;
    for (const StepGeom& st : g.steps)
      if (!st.skip) {
        const bool zonly = (descs[l].flags & TADMM_FLAG_SKIP_ROTATIONS) != 0;
        const bool comp = !filter_block_size(st.N, st.r) && complement_block_size(st.N, st.r, st.trans, zonly) > 0;
        lat[l] += comp ? 1500.0 : step_latency_us_full_or_filtered(st.N, st.r);
        nfilt[l] += (filter_block_size(st.N, st.r) > 0 || comp) ? 1 : 0;
      }
    lmax = std::max(lmax, lat[l]);
  }
  int n0 = 0;
  double t1 = 0.0;
  // A level runs its filter stages BEFORE the Jacobi group that holds the Rayleigh-Ritz problems AND the level's full
  // solves (single_run), so a long chain of full solves sharing a plan with filtered chains waits for their filters at
  // every level (ResNet-18: layer4.0.conv1, N = 480 keep 236, beside the three filtered layer4 chains).  When the long
  // chains are of both kinds, the unfiltered ones go to lane 1: their tournaments then run beside the filter stages.
  bool long_filt = false, long_full = false;
  for (int l = 0; l < n; ++l)
    if (lat[l] >= thr * lmax) { if (nfilt[l] > 0) long_filt = true; else long_full = true; }
  const bool mixed = long_filt && long_full;
  // Long chains of FULL solves beside filtered / complement chains of any length (DeiT-small: the N = 384 solves of qkv /
  // fc1 beside the complement route of proj / fc2): a filtered chain is a few hundred SMALL dependent launches, a full
  // tournament a few hundred launches that fill the chip.  The small ones go to lane 0 -- the high-priority stream --
  // where each of them finds CUs at once and delays the big launches by next to nothing; as the low-priority lane they
  // waited for a whole tick launch every time (measured: 11.3 ms instead of 10.0 ms without the route).
  bool any_filt = false;
  for (int l = 0; l < n; ++l) any_filt = any_filt || nfilt[l] > 0;
  const bool by_kind = long_full && any_filt && !long_filt;
  for (int l = 0; l < n; ++l) {
    if (by_kind) lane_of[l] = nfilt[l] > 0 ? 0 : 1;
    else lane_of[l] = (lat[l] >= thr * lmax && !(mixed && nfilt[l] == 0)) ? 0 : 1;
  }
  if (by_kind) {
    // a tournament launch of lane 1 holds Npad / 32 workgroups per problem, one per CU (128 KiB of LDS each): one more
    // than the device has CUs and every tick of the lane takes two rounds.  The excess (shortest chains first) rides in
    // lane 0 behind that lane's filter stages.
    int ncu = 256;
    if (h) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, h->device) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
    std::vector<int> wgs(n, 0);
    long total = 0;
    for (int l = 0; l < n; ++l) {
      if (lane_of[l] != 1) continue;
      LayerGeom g;
      if (build_geom(h, descs[l], g)) return false;
      for (const StepGeom& st : g.steps) if (!st.skip) wgs[l] = std::max(wgs[l], st.Npad / 32);
      total += wgs[l];
    }
    std::vector<int> order;
    for (int l = 0; l < n; ++l) if (lane_of[l] == 1 && wgs[l] >= 4) order.push_back(l);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lat[a] < lat[b]; });
    for (int l : order) {
      if (total <= ncu) break;
      lane_of[l] = 0;
      total -= wgs[l];
    }
  }
  for (int l = 0; l < n; ++l) { if (lane_of[l] == 0) ++n0; else t1 += lat[l]; }
  if (n0 == 0 || n - n0 < 2 || t1 < 0.2 * lmax) {
    // no short side to hide behind the long chains.  A big table of like layers (DeiT-small: 48 layers, every chain
    // within 0.6 of the longest) still gains from two half tables on two streams -- one lane's GEMM phases fill the gaps
    // of the other's eigen-solves (12.3 -> 10.9 ms); small or short tables stay in one lane.
    if (n >= 16 && n0 >= n - 1 && lmax >= 2000.0) {
      for (int l = 0; l < n; ++l) lane_of[l] = l < n / 2 ? 0 : 1;
      return true;
    }
    lane_of.assign(n, 0);
    return false;
  }
  return true;
}

static void lane_worker(tadmm_plan_s* parent) {
  Lanes* L = parent->lanes;
  (void)hipSetDevice(parent->h->device);
  for (;;) {
    std::unique_lock<std::mutex> lk(L->mu);
    L->cv.wait(lk, [&] { return L->job != 0; });
    if (L->job == 2) return;
    const int uu = L->a_update_u, us = L->a_use_u;
    double* resid = L->a_resid;
    lk.unlock();
    const int rc = single_run(L->sub[1], uu, us, resid, L->st[1]);
    lk.lock();
    L->rc = rc;
    L->job = 0;
    L->done = true;
    lk.unlock();
    L->cv.notify_all();
  }
}

static void lanes_destroy(tadmm_plan_s* p);

int tadmm_plan_workspace_bytes(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, size_t* bytes) {
  if (!h || !descs || !bytes || n_layers <= 0) return TADMM_ERR_INVALID;
  std::vector<int> lane_of;
  if (!lane_split(h, n_layers, descs, lane_of)) return single_workspace_bytes(h, n_layers, descs, bytes);
  size_t total = 0;
  for (int lane = 0; lane < 2; ++lane) {
    std::vector<tadmm_layer_desc> sub;
    for (int l = 0; l < n_layers; ++l) if (lane_of[l] == lane) sub.push_back(descs[l]);
    size_t b = 0;
    const int rc = single_workspace_bytes(h, (int)sub.size(), sub.data(), &b);
    if (rc) return rc;
    total += align_up(b, 4096);
  }
  *bytes = total + align_up((size_t)n_layers * 4, 4096);
  return TADMM_OK;
}

int tadmm_plan_create(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, const float* const* W,
                      float* const* U, float* const* Z, float* const* cores, void* workspace, size_t workspace_bytes,
                      tadmm_plan* out) {
  DeviceGuard device_guard(h);
  if (!h || !descs || !W || !U || !Z || !workspace || !out || n_layers <= 0) return TADMM_ERR_INVALID;
  std::vector<int> lane_of;
  if (!lane_split(h, n_layers, descs, lane_of))
    return single_create(h, n_layers, descs, W, U, Z, cores, workspace, workspace_bytes, out);
  tadmm_plan_s* P = new tadmm_plan_s();
  P->h = h;
  P->n = n_layers;
  P->layers.resize(n_layers);
  for (int l = 0; l < n_layers; ++l) {
    const int rc = build_geom(h, descs[l], P->layers[l]);
    if (rc) { delete P; return rc; }
  }
  Lanes* L = new Lanes();
  P->lanes = L;
  L->lane_of = lane_of;
  L->local_of.assign(n_layers, 0);
  char* ws = (char*)workspace;
  size_t off = 0;
  std::vector<int32_t> index_host;                                   // [lane 0 layers | lane 1 layers] -> global slot
  size_t index_off[2] = {0, 0};
  for (int lane = 0; lane < 2; ++lane) {
    index_off[lane] = index_host.size();
    for (int l = 0; l < n_layers; ++l) if (lane_of[l] == lane) { L->local_of[l] = (int)(index_host.size() - index_off[lane]); index_host.push_back(l); }
  }
  int rc = TADMM_OK;
  size_t sub_bytes[2] = {0, 0};
  for (int lane = 0; lane < 2 && rc == TADMM_OK; ++lane) {
    std::vector<tadmm_layer_desc> sd;
    std::vector<const float*> sw;
    std::vector<float*> su, sz, sc;
    for (int l = 0; l < n_layers; ++l) if (lane_of[l] == lane) {
      sd.push_back(descs[l]); sw.push_back(W[l]); su.push_back(U[l]); sz.push_back(Z[l]); sc.push_back(cores ? cores[l] : nullptr);
    }
    rc = single_workspace_bytes(h, (int)sd.size(), sd.data(), &sub_bytes[lane]);
    if (rc) break;
    const size_t need = align_up(sub_bytes[lane], 4096);
    if (off + need > workspace_bytes) { rc = TADMM_ERR_WORKSPACE; h->err = "workspace too small for the two lanes"; break; }
    rc = single_create(h, (int)sd.size(), sd.data(), sw.data(), su.data(), sz.data(), cores ? sc.data() : nullptr, ws + off,
                       need, &L->sub[lane]);
    off += need;
  }
  if (rc == TADMM_OK && off + (size_t)n_layers * 4 > workspace_bytes) { rc = TADMM_ERR_WORKSPACE; h->err = "workspace too small for the lane index"; }
  if (rc == TADMM_OK) {
    if (hipMemcpy(ws + off, index_host.data(), index_host.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = TADMM_ERR_HIP;
    for (int lane = 0; lane < 2; ++lane) L->sub[lane]->resid_index = (const int32_t*)(ws + off) + index_off[lane];
  }
  if (rc == TADMM_OK) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                 // hi is the numerically smaller, more urgent one
    hipError_t e = hipStreamCreateWithPriority(&L->st[0], hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&L->st[1], hipStreamNonBlocking, lo);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L->ev_begin, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&L->ev_end[i], hipEventDisableTiming);
    if (e != hipSuccess) { rc = TADMM_ERR_HIP; h->err = std::string("lane streams: ") + hipGetErrorString(e); }
  }
  if (rc != TADMM_OK) { lanes_destroy(P); delete P; return rc; }
  L->worker = std::thread(lane_worker, P);
  *out = P;
  return TADMM_OK;
}

int tadmm_plan_run(tadmm_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream_) {
  if (!p) return TADMM_ERR_INVALID;
  if (!p->lanes) return single_run(p, update_u, use_u, resid_sq_dev, stream_);
  DeviceGuard device_guard(p->h);
  Lanes* L = p->lanes;
  hipStream_t s = (hipStream_t)stream_;
  HIP_OK(p->h, hipEventRecord(L->ev_begin, s));                      // both lanes start behind the caller's stream
  for (int i = 0; i < 2; ++i) HIP_OK(p->h, hipStreamWaitEvent(L->st[i], L->ev_begin, 0));
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->a_update_u = update_u; L->a_use_u = use_u; L->a_resid = resid_sq_dev;
    L->done = false;
    L->job = 1;
  }
  L->cv.notify_all();
  const int rc0 = single_run(L->sub[0], update_u, use_u, resid_sq_dev, L->st[0]);
  int rc1;
  {
    std::unique_lock<std::mutex> lk(L->mu);
    L->cv.wait(lk, [&] { return L->done; });
    rc1 = L->rc;
  }
  for (int i = 0; i < 2; ++i) {                                      // ... and the caller's stream continues behind both
    HIP_OK(p->h, hipEventRecord(L->ev_end[i], L->st[i]));
    HIP_OK(p->h, hipStreamWaitEvent(s, L->ev_end[i], 0));
  }
  for (int i = 0; i < 8; ++i) p->last_ms[i] = L->sub[0]->last_ms[i] + L->sub[1]->last_ms[i];
  p->last_sweeps = L->sub[0]->last_sweeps + L->sub[1]->last_sweeps;
  return rc0 != TADMM_OK ? rc0 : rc1;
}

static void lanes_destroy(tadmm_plan_s* p) {
  Lanes* L = p->lanes;
  if (!L) return;
  if (L->worker.joinable()) {
    { std::lock_guard<std::mutex> lk(L->mu); L->job = 2; }
    L->cv.notify_all();
    L->worker.join();
  }
  for (int i = 0; i < 2; ++i) {
    if (L->sub[i]) {
      if (L->sub[i]->ev_made) for (auto& e : L->sub[i]->ev) (void)hipEventDestroy(e);
      L->sub[i]->poll.destroy();
      delete L->sub[i];
    }
    if (L->st[i]) { (void)hipStreamSynchronize(L->st[i]); (void)hipStreamDestroy(L->st[i]); }
    if (L->ev_end[i]) (void)hipEventDestroy(L->ev_end[i]);
  }
  if (L->ev_begin) (void)hipEventDestroy(L->ev_begin);
  delete L;
  p->lanes = nullptr;
}

int tadmm_lane_split(int n_layers, const tadmm_layer_desc* descs, int32_t* lane_of_out) {
  if (!descs || n_layers <= 0) return TADMM_ERR_INVALID;
  std::vector<int> lane_of;
  const bool two = lane_split(nullptr, n_layers, descs, lane_of);
  if (lane_of_out) for (int l = 0; l < n_layers; ++l) lane_of_out[l] = lane_of[l];
  return two ? 2 : 1;
}

int tadmm_plan_lanes(tadmm_plan p, int32_t* lane_of_out) {
  if (!p) return TADMM_ERR_INVALID;
  if (!p->lanes) { if (lane_of_out) for (int l = 0; l < p->n; ++l) lane_of_out[l] = 0; return 1; }
  if (lane_of_out) for (int l = 0; l < p->n; ++l) lane_of_out[l] = p->lanes->lane_of[l];
  return 2;
}

int tadmm_plan_enable_timing(tadmm_plan p, int on) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_ERR_INVALID;
  if (p->lanes) {
    for (int i = 0; i < 2; ++i) { const int rc = tadmm_plan_enable_timing(p->lanes->sub[i], on); if (rc) return rc; }
    p->timing = on != 0;
    return TADMM_OK;
  }
  if (on && !p->ev_made) {
    for (auto& e : p->ev) if (hipEventCreate(&e) != hipSuccess) return TADMM_ERR_HIP;
    p->ev_made = true;
  }
  p->timing = on != 0;
  return TADMM_OK;
}

int tadmm_plan_last_timing(tadmm_plan p, double out_ms[8]) {
  if (!p || !out_ms) return TADMM_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out_ms[i] = p->last_ms[i];
  out_ms[6] = p->last_sweeps;
  return TADMM_OK;
}

static int single_run(tadmm_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream_) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_ERR_INVALID;
  tadmm_handle h = p->h;
  hipStream_t s = (hipStream_t)stream_;
  char* ws = p->ws;
  auto D = [&](size_t off) { return ws + off; };
  double acc_ms[8] = {0};
  int total_sweeps = 0;
  p->filt_problems = p->filt_fallbacks = p->filt_stages = 0;
  p->ftm.on = p->timing;
  p->ftm.gemm_ms = 0.0; p->ftm.gemm_launches = 0; p->ftm.gemm_flops = 0.0;
  p->ftm.fast_ms = 0.0; p->ftm.fast_launches = 0; p->ftm.fast_flops = 0.0;
  if (p->timing) { p->ftm.a = p->ev[14]; p->ftm.b = p->ev[15]; }
  p->jtm.on = p->timing;
  p->jtm.tick_ms = 0.0; p->jtm.tick_launches = 0; p->jtm.tick_flops = 0.0; p->jtm.tick_wgs = 0.0;
  if (p->timing) { p->jtm.a = p->ev[14]; p->jtm.b = p->ev[15]; }
  // timing helper: record a pair of events around a phase and accumulate after a sync
  auto tic = [&](int i) { if (p->timing) (void)hipEventRecord(p->ev[i], s); };
  auto toc = [&](int i, int slot) {
    if (!p->timing) return;
    (void)hipEventRecord(p->ev[i + 1], s);
    (void)hipEventSynchronize(p->ev[i + 1]);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
    acc_ms[slot] += ms;
  };

  tic(0);
  launch_unfold((const SweepDesc*)D(p->sweep_desc_off), (const BlockRef*)D(p->unfold.map_off), p->unfold.nblocks,
                use_u, s);
  toc(0, 0);

  for (StepPlan& sp : p->steps) {
    if (sp.neig == 0) continue;
    tic(0);
    launch_gram_partial((const GramDesc*)D(sp.gram_p.desc_off), (const BlockRef*)D(sp.gram_p.map_off),
                        sp.gram_p.nblocks, s);
    launch_gram_reduce((const GramDesc*)D(sp.gram_r.desc_off), (const BlockRef*)D(sp.gram_r.map_off),
                       sp.gram_r.nblocks, s);
    toc(0, 1);
    tic(0);
    const bool filtered = sp.fg.nf > 0;
    if (filtered) {
      const int rc = filter_run_pre(h, sp.fg, ws, p->poll, s, p->debug, &p->ftm);
      if (rc != TADMM_OK) return rc;
    }
    const EigDesc* ed = (const EigDesc*)D(sp.eig_desc_off);
    const int32_t* skip = filtered ? (const int32_t*)D(sp.skip_off) : nullptr;
    EigGroup eg;
    eg.ed = ed; eg.neig = sp.neig; eg.players = sp.nb.data(); eg.gsteps = sp.gsteps; eg.mode = sp.mode; eg.aligned = sp.mode >= 2 && align_sweeps_on(); eg.row_len = sp.row_len.data();
    eg.mid_sizes = sp.mid.empty() ? nullptr : sp.mid.data();
    eg.ld_max = sp.ld_max; eg.tick_lds = sp.tick_lds;
    eg.tick_map = (const BlockRef*)D(sp.eig_tick.map_off); eg.tick_blocks = sp.eig_tick.nblocks;
    eg.self_map = (const BlockRef*)D(sp.eig_self.map_off); eg.self_blocks = sp.eig_self.nblocks;
    eg.prev_dev = (double*)D(sp.prev_off_dev);
    eg.off_dev = (const double*)D(sp.off_off); eg.done_dev = (const int*)D(sp.done_off);
    eg.npad_max = sp.npad_max;
    eg.expected = sp.last_sweeps;
    eg.skip = skip;
    int gs = 0;
    bool small_pending = false;
    {
      const int rc = run_eig_group(h, eg, p->poll, p->tol, p->inner_sweeps, p->max_global_sweeps, p->debug, s, &gs,
                                   &small_pending, &p->jtm);
      if (rc != TADMM_OK) return rc;
    }
    sp.last_sweeps = gs;
    total_sweeps += gs;
    launch_eig_norms(ed, (const BlockRef*)D(sp.eig_norm.map_off), sp.eig_norm.nblocks, s, skip);
    launch_eig_sort(ed, sp.neig, s, skip, sp.npad_max);
    launch_eig_extract(ed, (const BlockRef*)D(sp.eig_ext.map_off), sp.eig_ext.nblocks, s, skip);
    if (filtered) {
      if (small_pending) {
        const int rc = check_small_group(h, eg, p->poll);
        if (rc != TADMM_OK) return rc;
        small_pending = false;
      }
      int nbad = 0;
      int rc = filter_run_post(h, sp.fg, ws, p->poll, s, p->debug, &nbad, &p->ftm);
      if (rc != TADMM_OK) return rc;
      p->filt_problems += sp.fg.nf;
      p->filt_fallbacks += nbad;
      p->filt_stages = std::max(p->filt_stages, sp.fg.last_stages);
      if (nbad > 0) {       // the full Jacobi solve for the problems the filter could not certify
        StepPlan::Fallback& fb = sp.fb;
        const EigDesc* fd = (const EigDesc*)D(fb.tick.desc_off);
        const int32_t* fskip = (const int32_t*)D(sp.fb_skip_off);
        EigGroup fgp;
        fgp.ed = fd; fgp.neig = fb.neig; fgp.players = fb.nb.data(); fgp.gsteps = fb.gsteps; fgp.mode = fb.mode; fgp.aligned = fb.mode >= 2 && align_sweeps_on(); fgp.row_len = fb.row_len.data();
        fgp.ld_max = fb.ld_max; fgp.tick_lds = fb.tick_lds;
        fgp.tick_map = (const BlockRef*)D(fb.tick.map_off); fgp.tick_blocks = fb.tick.nblocks;
        fgp.self_map = (const BlockRef*)D(fb.self.map_off); fgp.self_blocks = fb.self.nblocks;
        fgp.prev_dev = (double*)D(fb.prev_off_dev);
        fgp.npad_max = fb.npad_max;
        fgp.expected = 0;
        fgp.skip = fskip;
        int fgs = 0;
        bool fsmall = false;
        rc = run_eig_group(h, fgp, p->poll, p->tol, p->inner_sweeps, p->max_global_sweeps, p->debug, s, &fgs, &fsmall, &p->jtm);
        if (rc != TADMM_OK) return rc;
        total_sweeps += fgs;
        launch_eig_norms(fd, (const BlockRef*)D(fb.norm.map_off), fb.norm.nblocks, s, fskip);
        launch_eig_sort(fd, fb.neig, s, fskip, fb.npad_max);
        launch_eig_extract(fd, (const BlockRef*)D(fb.ext.map_off), fb.ext.nblocks, s, fskip);
        if (fsmall) {
          rc = check_small_group(h, fgp, p->poll);
          if (rc != TADMM_OK) return rc;
        }
      }
    }
    toc(0, 2);
    tic(0);
    launch_gemm((const GemmDesc*)D(sp.proj.desc_off), (const BlockRef*)D(sp.proj.map_off), sp.proj.nblocks, s);
    toc(0, 3);
    if (small_pending) {      // the finalize + projection launches above are already queued behind it
      const int rc = check_small_group(h, eg, p->poll);
      if (rc != TADMM_OK) return rc;
    }
  }
  tic(0);
  for (Phase& ph : p->recon)
    launch_gemm((const GemmDesc*)D(ph.desc_off), (const BlockRef*)D(ph.map_off), ph.nblocks, s);
  toc(0, 4);
  tic(0);
  double* partial = (double*)D(p->resid_partial_off);
  launch_fold_update((const SweepDesc*)D(p->sweep_desc_off), (const BlockRef*)D(p->fold.map_off), p->fold.nblocks,
                     update_u, partial, s);
  if (resid_sq_dev)
    launch_resid_reduce((const SweepDesc*)D(p->sweep_desc_off), p->n, partial, resid_sq_dev, s, p->resid_index);
  toc(0, 5);
  HIP_OK(h, hipGetLastError());
  for (int i = 0; i < 8; ++i) p->last_ms[i] = acc_ms[i];
  p->last_sweeps = total_sweeps;
  if (p->debug) { (void)hipStreamSynchronize(s); dump_stamps(); }
  return TADMM_OK;
}

int tadmm_plan_singular_values(tadmm_plan p, int layer, int step, double* out_host, void* stream_) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p || !out_host || layer < 0 || layer >= p->n) return TADMM_ERR_INVALID;
  if (p->lanes)
    return tadmm_plan_singular_values(p->lanes->sub[p->lanes->lane_of[layer]], p->lanes->local_of[layer], step, out_host, stream_);
  const LayerGeom& g = p->layers[layer];
  if (step < 0 || step >= (int)g.steps.size()) return TADMM_ERR_INVALID;
  const size_t off = p->sigma_off[layer][step];
  if (off == (size_t)-1) CTX_FAIL(p->h, TADMM_ERR_INVALID, "step %d of layer %d was skipped (identity)", step, layer);
  hipStream_t s = (hipStream_t)stream_;
  HIP_OK(p->h, hipMemcpyAsync(out_host, p->ws + off, (size_t)g.steps[step].r * 8, hipMemcpyDeviceToHost, s));
  HIP_OK(p->h, hipStreamSynchronize(s));
  return TADMM_OK;
}

int tadmm_plan_filter_timing(tadmm_plan p, double out[4]) {
  if (!p || !out) return TADMM_ERR_INVALID;
  if (p->lanes) {
    double a[4], b[4];
    tadmm_plan_filter_timing(p->lanes->sub[0], a);
    tadmm_plan_filter_timing(p->lanes->sub[1], b);
    for (int i = 0; i < 4; ++i) out[i] = a[i] + b[i];
    return TADMM_OK;
  }
  out[0] = p->ftm.gemm_ms; out[1] = p->ftm.gemm_launches; out[2] = p->ftm.gemm_flops; out[3] = 0.0;
  return TADMM_OK;
}

int tadmm_plan_jacobi_timing(tadmm_plan p, double out[4]) {
  if (!p || !out) return TADMM_ERR_INVALID;
  if (p->lanes) {
    double a[4], b[4];
    tadmm_plan_jacobi_timing(p->lanes->sub[0], a);
    tadmm_plan_jacobi_timing(p->lanes->sub[1], b);
    for (int i = 0; i < 4; ++i) out[i] = a[i] + b[i];
    return TADMM_OK;
  }
  out[0] = p->jtm.tick_ms; out[1] = p->jtm.tick_launches; out[2] = p->jtm.tick_flops; out[3] = p->jtm.tick_wgs;
  return TADMM_OK;
}

int tadmm_plan_filter_timing_fast(tadmm_plan p, double out[4]) {
  if (!p || !out) return TADMM_ERR_INVALID;
  if (p->lanes) {
    double a[4], b[4];
    tadmm_plan_filter_timing_fast(p->lanes->sub[0], a);
    tadmm_plan_filter_timing_fast(p->lanes->sub[1], b);
    for (int i = 0; i < 4; ++i) out[i] = a[i] + b[i];
    return TADMM_OK;
  }
  out[0] = p->ftm.fast_ms; out[1] = p->ftm.fast_launches; out[2] = p->ftm.fast_flops; out[3] = 0.0;
  return TADMM_OK;
}

int tadmm_plan_filter_stats(tadmm_plan p, int32_t out[4]) {
  if (!p || !out) return TADMM_ERR_INVALID;
  if (p->lanes) {
    int32_t a[4], b[4];
    tadmm_plan_filter_stats(p->lanes->sub[0], a);
    tadmm_plan_filter_stats(p->lanes->sub[1], b);
    for (int i = 0; i < 3; ++i) out[i] = a[i] + b[i];
    out[3] = std::max(a[3], b[3]);
    return TADMM_OK;
  }
  int eligible = 0;
  for (const StepPlan& sp : p->steps) eligible += sp.fg.nf;
  out[0] = eligible; out[1] = p->filt_problems; out[2] = p->filt_fallbacks; out[3] = p->filt_stages;
  return TADMM_OK;
}

int tadmm_plan_destroy(tadmm_plan p) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_OK;
  lanes_destroy(p);
  if (p->ev_made) for (auto& e : p->ev) (void)hipEventDestroy(e);
  p->poll.destroy();
  delete p;
  return TADMM_OK;
}

int tadmm_plan_set_jacobi(tadmm_plan p, double tol, int inner_sweeps, int max_sweeps) {
  if (!p) return TADMM_ERR_INVALID;
  if (p->lanes) for (int i = 0; i < 2; ++i) tadmm_plan_set_jacobi(p->lanes->sub[i], tol, inner_sweeps, max_sweeps);
  if (tol > 0) p->tol = tol;
  if (inner_sweeps > 0) p->inner_sweeps = inner_sweeps;
  if (max_sweeps > 0) p->max_global_sweeps = max_sweeps;
  return TADMM_OK;
}

int tadmm_plan_ranks(tadmm_plan p, int layer, int32_t* ranks_out) {
  if (!p || !ranks_out || layer < 0 || layer >= p->n) return TADMM_ERR_INVALID;
  const tadmm_layer_desc& d = p->layers[layer].desc;
  for (int i = 0; i <= d.d; ++i) ranks_out[i] = d.ranks[i];
  return d.d + 1;
}

// ---- penalty ----
int tadmm_penalty_scratch_doubles(void) { return kPenaltyBlocks; }

int tadmm_penalty(tadmm_handle h, int n, const void* const* ptrs_dev, const int64_t* numel_dev, int64_t total_numel,
                  float rho, float grad_scale, double* loss_dev, double* partial_dev, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || n <= 0 || !ptrs_dev || !numel_dev || !loss_dev || !partial_dev) return TADMM_ERR_INVALID;
  launch_penalty(n, ptrs_dev, numel_dev, total_numel, rho, grad_scale, loss_dev, partial_dev, (hipStream_t)stream_);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

// ---- grouped GEMM ----
size_t tadmm_gemm_pack_bytes(int n, const tadmm_gemm_desc* descs) {
  if (n <= 0 || !descs) return 0;
  size_t blocks = 0;
  for (int i = 0; i < n; ++i)
    blocks += (size_t)((descs[i].M + kGemmBM - 1) / kGemmBM) * ((descs[i].N + kGemmBN - 1) / kGemmBN);
  return align_up((size_t)n * sizeof(GemmDesc), 256) + blocks * sizeof(BlockRef);
}

int tadmm_gemm_pack(int n, const tadmm_gemm_desc* descs, void* blob_host, size_t blob_bytes, int* nblocks_out) {
  if (n <= 0 || !descs || !blob_host || !nblocks_out) return TADMM_ERR_INVALID;
  if (blob_bytes < tadmm_gemm_pack_bytes(n, descs)) return TADMM_ERR_WORKSPACE;
  GemmDesc* gd = (GemmDesc*)blob_host;
  BlockRef* map = (BlockRef*)((char*)blob_host + align_up((size_t)n * sizeof(GemmDesc), 256));
  int nb = 0;
  for (int i = 0; i < n; ++i) {
    const tadmm_gemm_desc& s = descs[i];
    if (s.M <= 0 || s.N <= 0 || s.K <= 0) return TADMM_ERR_INVALID;
    if (!((s.a_rs == 1) || (s.a_cs == 1)) || !((s.b_rs == 1) || (s.b_cs == 1))) return TADMM_ERR_INVALID;
    GemmDesc& g = gd[i];
    memset(&g, 0, sizeof g);
    g.A = s.A; g.B = s.B; g.C = s.C; g.M = s.M; g.N = s.N; g.K = s.K;
    g.a_rs = s.a_rs; g.a_cs = s.a_cs; g.b_rs = s.b_rs; g.b_cs = s.b_cs; g.c_rs = s.c_rs; g.c_cs = s.c_cs;
    g.alpha = s.alpha; g.beta = s.beta; g.bias_n = s.bias_n; g.bias_m = s.bias_m;
    g.tiles_m = (s.M + kGemmBM - 1) / kGemmBM;
    g.tiles_n = (s.N + kGemmBN - 1) / kGemmBN;
    for (int b = 0; b < g.tiles_m * g.tiles_n; ++b) map[nb++] = BlockRef{i, b};
  }
  *nblocks_out = nb;
  return TADMM_OK;
}

int tadmm_gemm_run(tadmm_handle h, const void* blob_dev, int n, int nblocks, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !blob_dev || n <= 0 || nblocks <= 0) return TADMM_ERR_INVALID;
  const GemmDesc* gd = (const GemmDesc*)blob_dev;
  const BlockRef* map = (const BlockRef*)((const char*)blob_dev + align_up((size_t)n * sizeof(GemmDesc), 256));
  launch_gemm(gd, map, nblocks, (hipStream_t)stream_);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

int tadmm_gemm(tadmm_handle h, const tadmm_gemm_desc* sdesc, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !sdesc) return TADMM_ERR_INVALID;
  const tadmm_gemm_desc& s = *sdesc;
  if (s.M <= 0 || s.N <= 0 || s.K <= 0 || !s.A || !s.B || !s.C) CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_gemm: empty operand");
  if (!((s.a_rs == 1) || (s.a_cs == 1)) || !((s.b_rs == 1) || (s.b_cs == 1)))
    CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_gemm: each operand needs one unit stride");
  GemmDesc g;
  memset(&g, 0, sizeof g);
  g.A = s.A; g.B = s.B; g.C = s.C; g.M = s.M; g.N = s.N; g.K = s.K;
  g.a_rs = s.a_rs; g.a_cs = s.a_cs; g.b_rs = s.b_rs; g.b_cs = s.b_cs; g.c_rs = s.c_rs; g.c_cs = s.c_cs;
  g.alpha = s.alpha; g.beta = s.beta; g.bias_n = s.bias_n; g.bias_m = s.bias_m;
  g.tiles_m = (s.M + kGemmBM - 1) / kGemmBM;
  g.tiles_n = (s.N + kGemmBN - 1) / kGemmBN;
  launch_gemm_one(g, (hipStream_t)stream_);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

int tadmm_gemm_bf16_nt(tadmm_handle h, const void* A, const void* Bt, void* C, int M, int N, int K, int64_t lda,
                       int64_t ldb, int64_t ldc, const float* bias_n, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !A || !Bt || !C) return TADMM_ERR_INVALID;
  if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_gemm_bf16_nt: bad shape");
  launch_gemm_bf16_nt(A, Bt, C, M, N, K, lda, ldb, ldc, bias_n, (hipStream_t)stream_);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

// ---- forward chains of the factorised layers (chain.hip) ----
static int chain_entry(tadmm_handle h, const tadmm_chain_desc* c, int fused, const char* who, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !c) return TADMM_ERR_INVALID;
  if (!c->X || !c->Y || !c->Win || (fused && !c->Wout)) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: null operand");
  if (c->T < 0 || c->Kin <= 0 || c->R <= 0 || (fused && c->Nout <= 0)) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: bad shape");
  if (c->dtype != TADMM_CHAIN_F32 && c->dtype != TADMM_CHAIN_BF16) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: bad dtype");
  const int epl = c->dtype == TADMM_CHAIN_F32 ? 4 : 8;          // elements per 16-byte load of X
  const int64_t ks1 = (c->Kin + 31) / 32, nt1 = (c->R + 15) / 16;
  if ((((uintptr_t)c->Win) & 15) || c->win_plane < nt1 * ks1 * 512 || (c->win_plane & 7))
    CTX_FAIL(h, TADMM_ERR_INVALID, "chain: Win planes must be 16-byte aligned fragment-major images of ceil(R/16) x ceil(Kin/32) KiB blocks");
  if (c->x_hw == 0 && (c->ldx < c->Kin || c->Kin % epl || c->ldx % epl || (((uintptr_t)c->X) & 15)))
    CTX_FAIL(h, TADMM_ERR_INVALID, "chain: X rows must be 16-byte aligned with Kin a whole number of 16-byte vectors");
  if (c->y_hw == 0 && c->ldy < (fused ? c->Nout : c->R)) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: ldy too small");
  if (c->x_hw < 0 || c->y_hw < 0) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: negative image size");
  if (((uintptr_t)c->bias) & 15) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: bias must be 16-byte aligned");
  if (fused) {
    if (c->R % 64 || c->R > 256) CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "chain: fused middle rank must be a multiple of 64, at most 256");
    const int64_t nt2 = (c->Nout + 15) / 16;
    if ((((uintptr_t)c->Wout) & 15) || c->wout_plane < nt2 * (c->R / 32) * 512 || (c->wout_plane & 7))
      CTX_FAIL(h, TADMM_ERR_INVALID, "chain: Wout planes must be 16-byte aligned fragment-major images of ceil(Nout/16) x R/32 KiB blocks");
  }
  if (c->tile_tokens != 0 && c->tile_tokens != 32 && c->tile_tokens != 64) CTX_FAIL(h, TADMM_ERR_INVALID, "chain: tile_tokens");
  ChainDesc d;
  memset(&d, 0, sizeof d);
  d.X = c->X; d.Y = c->Y; d.Win = (const uint16_t*)c->Win; d.Wout = (const uint16_t*)c->Wout; d.bias = c->bias;
  d.T = c->T; d.Kin = c->Kin; d.R = c->R; d.Nout = c->Nout;
  d.ldx = c->ldx; d.ldy = c->ldy;
  d.win_plane = c->win_plane; d.wout_plane = c->wout_plane;
  d.x_hw = c->x_hw; d.y_hw = c->y_hw; d.fused = fused;
  d.x_vec = (c->x_hw > 0 && c->x_hw % epl == 0 && (((uintptr_t)c->X) & 15) == 0) ? 1 : 0;
  const int nfeat = fused ? c->Nout : c->R;
  if (c->y_hw > 0) d.y_vec = (c->y_hw % epl == 0 && (((uintptr_t)c->Y) & 15) == 0) ? 1 : 0;
  else d.y_vec = (c->ldy % epl == 0 && nfeat % epl == 0 && (((uintptr_t)c->Y) & 15) == 0) ? 1 : 0;
  if (launch_tt_chain(d, c->dtype, c->tile_tokens, (hipStream_t)stream_) != 0)
    CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "chain: token tile does not fit the LDS");
  HIP_OK(h, hipGetLastError());
  (void)who;
  return TADMM_OK;
}
int tadmm_ttlinear_fwd(tadmm_handle h, const tadmm_chain_desc* d, void* s) { return chain_entry(h, d, 1, "ttlinear_fwd", s); }
int tadmm_ttlinear_bwd(tadmm_handle h, const tadmm_chain_desc* d, void* s) { return chain_entry(h, d, 1, "ttlinear_bwd", s); }
int tadmm_ttconv_chain_in(tadmm_handle h, const tadmm_chain_desc* d, void* s) { return chain_entry(h, d, 0, "ttconv_chain_in", s); }
int tadmm_ttconv_chain_out(tadmm_handle h, const tadmm_chain_desc* d, void* s) { return chain_entry(h, d, 0, "ttconv_chain_out", s); }
int tadmm_tucker_1x1(tadmm_handle h, const tadmm_chain_desc* d, void* s) { return chain_entry(h, d, 0, "tucker_1x1", s); }

int tadmm_conv_chain_desc_bytes(void) { return (int)sizeof(tadmm_conv_chain_desc); }

int tadmm_ttconv_fused(tadmm_handle h, const tadmm_conv_chain_desc* c, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !c) return TADMM_ERR_INVALID;
  if (!c->X || !c->Y || !c->W1 || !c->W2 || !c->W3) CTX_FAIL(h, TADMM_ERR_INVALID, "conv chain: null operand");
  if (c->dtype != TADMM_CHAIN_F32 && c->dtype != TADMM_CHAIN_BF16) CTX_FAIL(h, TADMM_ERR_INVALID, "conv chain: bad dtype");
  if (c->B < 0 || c->C <= 0 || c->Nout <= 0 || c->H <= 0 || c->W <= 0 || c->kh <= 0 || c->kw <= 0 || c->stride_h <= 0 ||
      c->stride_w <= 0 || c->dil_h <= 0 || c->dil_w <= 0 || c->pad_h < 0 || c->pad_w < 0)
    CTX_FAIL(h, TADMM_ERR_INVALID, "conv chain: bad geometry");
  const int ho = (c->H + 2 * c->pad_h - c->dil_h * (c->kh - 1) - 1) / c->stride_h + 1;
  const int wo = (c->W + 2 * c->pad_w - c->dil_w * (c->kw - 1) - 1) / c->stride_w + 1;
  if (ho != c->Ho || wo != c->Wo) CTX_FAIL(h, TADMM_ERR_INVALID, "conv chain: output size does not match the geometry");
  if (ho <= 0 || wo <= 0 || wo > 64) CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "conv chain: output rows of more than 64 pixels take the three-launch path");
  if (c->R1 <= 0 || c->R2 <= 0 || c->R1 % 32 || c->R2 % 32 || c->R1 > 256 || c->R2 > 256)
    CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "conv chain: ranks must be padded to multiples of 32 and at most 256");
  const int64_t taps = (int64_t)c->kh * c->kw;
  if ((((uintptr_t)c->W1) & 15) || (((uintptr_t)c->W2) & 15) || (((uintptr_t)c->W3) & 15) || (((uintptr_t)c->bias) & 15) ||
      c->w1_plane < (int64_t)(c->R1 / 16) * ((c->C + 31) / 32) * 512 || c->w2_plane < (int64_t)(c->R2 / 16) * taps * (c->R1 / 32) * 512 ||
      c->w3_plane < (int64_t)((c->Nout + 15) / 16) * (c->R2 / 32) * 512)
    CTX_FAIL(h, TADMM_ERR_INVALID, "conv chain: weight planes too small or misaligned");
  ConvChainDesc d;
  memset(&d, 0, sizeof d);
  d.X = c->X; d.Y = c->Y; d.W1 = (const uint16_t*)c->W1; d.W2 = (const uint16_t*)c->W2; d.W3 = (const uint16_t*)c->W3;
  d.bias = c->bias; d.w1_plane = c->w1_plane; d.w2_plane = c->w2_plane; d.w3_plane = c->w3_plane;
  d.B = c->B; d.C = c->C; d.R1 = c->R1; d.R2 = c->R2; d.Nout = c->Nout;
  d.H = c->H; d.W = c->W; d.Ho = ho; d.Wo = wo; d.kh = c->kh; d.kw = c->kw; d.sh = c->stride_h; d.sw = c->stride_w;
  d.ph = c->pad_h; d.pw = c->pad_w; d.dh = c->dil_h; d.dw = c->dil_w;
  {
    // pixels per workgroup: 64, or 32 when the intermediates of 64 do not fit the LDS; output rows per workgroup: as many
    // as give <= TM output pixels and a halo of <= 3 TM input pixels
    const int planes = c->dtype == TADMM_CHAIN_F32 ? 3 : 1, kc = c->dtype == TADMM_CHAIN_F32 ? 64 : 128;
    bool found = false;
    for (int tmx = 64; tmx >= 32 && !found; tmx /= 2) {
      if (wo > tmx) continue;
      int tr = std::min(ho, tmx / wo), nt = 0;
      for (; tr >= 1; --tr) {
        const int irows = std::min(c->H, (tr - 1) * c->stride_h + (c->kh - 1) * c->dil_h + 1);
        nt = (irows * c->W + tmx - 1) / tmx;
        if (nt <= 3) break;
      }
      if (tr < 1) continue;
      const size_t lds = ((size_t)2 * planes * tmx * (kc + 8) + (size_t)planes * tmx * nt * (c->R1 + 8) +
                          (size_t)planes * tmx * (c->R2 + 8)) * 2;
      if (lds > 160 * 1024) continue;
      d.TM = tmx; d.TR = tr; d.tiles = (ho + tr - 1) / tr; d.NT = nt;
      found = true;
    }
    if (!found) CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "conv chain: halo or intermediates do not fit the LDS");
  }
  const int epl = c->dtype == TADMM_CHAIN_F32 ? 4 : 8;
  d.x_vec = ((c->H * c->W) % epl == 0 && (((uintptr_t)c->X) & 15) == 0) ? 1 : 0;
  if (launch_tt_conv(d, c->dtype, (hipStream_t)stream_) != 0)
    CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "conv chain: the intermediates do not fit the LDS");
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

// ---- standalone Gram / eigh (tests, Tucker path) ----
static void gram_geom(int m, int n, StepGeom& st) {
  st.m = m; st.cols = n; st.trans = m > n;
  st.N = std::min(m, n);
  st.Npad = (int)align_up(st.N, 4 * kJB);
  st.nb = st.Npad / kJB;
  st.ld = eig_ld(st.N);
  st.nt = (st.N + 31) / 32;
  const int64_t K = st.trans ? m : n;
  const int ntp = st.nt * (st.nt + 1) / 2;
  int ks = (256 + ntp - 1) / ntp;   // ~256 workgroups per problem; levels batch 15-30 problems
  const int maxks = (int)std::max<int64_t>(1, (K + 255) / 256);
  ks = std::max(1, std::min(ks, maxks));
  st.kchunk = (int)align_up((K + ks - 1) / ks, 64);
  st.ksplit = (int)((K + st.kchunk - 1) / st.kchunk);
}

size_t tadmm_gram_scratch_bytes(int m, int n) {
  StepGeom st;
  gram_geom(m, n, st);
  const size_t ntp = (size_t)st.nt * (st.nt + 1) / 2;
  const size_t nblk_p = (size_t)st.ksplit * ntp;
  const size_t nblk_r = ((size_t)st.Npad * st.ld + 1023) / 1024;
  return align_up(st.ksplit * ntp * 1024 * 8, 256) + align_up(sizeof(GramDesc), 256) +
         align_up(nblk_p * sizeof(BlockRef), 256) + align_up(nblk_r * sizeof(BlockRef), 256);
}

int tadmm_gram_ld(int m, int n, int* Npad, int* ld) {
  StepGeom st;
  gram_geom(m, n, st);
  if (Npad) *Npad = st.Npad;
  if (ld) *ld = st.ld;
  return st.N;
}

int tadmm_gram_f64(tadmm_handle h, const float* A, int m, int n, double* G, int ldg, void* scratch, size_t scratch_bytes,
                   void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !A || !G || !scratch || m <= 0 || n <= 0) return TADMM_ERR_INVALID;
  StepGeom st;
  gram_geom(m, n, st);
  if (ldg != st.ld) CTX_FAIL(h, TADMM_ERR_INVALID, "ldg must be %d (tadmm_gram_ld)", st.ld);
  if (scratch_bytes < tadmm_gram_scratch_bytes(m, n)) CTX_FAIL(h, TADMM_ERR_WORKSPACE, "gram scratch too small");
  hipStream_t s = (hipStream_t)stream_;
  const size_t ntp = (size_t)st.nt * (st.nt + 1) / 2;
  char* base = (char*)scratch;
  size_t off = 0;
  double* partial = (double*)(base + off); off += align_up(st.ksplit * ntp * 1024 * 8, 256);
  GramDesc* gdev = (GramDesc*)(base + off); off += align_up(sizeof(GramDesc), 256);
  std::vector<BlockRef> mp, mr;
  for (int b = 0; b < (int)(st.ksplit * ntp); ++b) mp.push_back(BlockRef{0, b});
  if (st.ksplit > 1)   // ksplit == 1: the product kernel writes G itself
    for (int b = 0; b < (int)(((size_t)st.Npad * st.ld + 1023) / 1024); ++b) mr.push_back(BlockRef{0, b});
  BlockRef* mpd = (BlockRef*)(base + off); off += align_up(mp.size() * sizeof(BlockRef), 256);
  BlockRef* mrd = (BlockRef*)(base + off);
  GramDesc gd;
  memset(&gd, 0, sizeof gd);
  gd.A = A; gd.m = m; gd.n = n; gd.trans = st.trans; gd.N = st.N; gd.K = st.trans ? m : n; gd.nt = st.nt;
  gd.ksplit = st.ksplit; gd.kchunk = st.kchunk; gd.partial = partial; gd.G = G; gd.Npad = st.Npad; gd.ld = st.ld;
  HIP_OK(h, hipMemcpyAsync(gdev, &gd, sizeof gd, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(mpd, mp.data(), mp.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(mrd, mr.data(), mr.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipStreamSynchronize(s));   // host vectors die at return
  launch_gram_partial(gdev, mpd, (int)mp.size(), s);
  launch_gram_reduce(gdev, mrd, (int)mr.size(), s);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

size_t tadmm_eigh_scratch_bytes(int N) {
  const size_t Npad = align_up(N, 4 * kJB), ld = (size_t)eig_ld(N);
  return align_up(Npad * ld * 8, 256) + align_up(sizeof(EigDesc), 256) + 4 * align_up(Npad * sizeof(BlockRef), 256) +
         align_up(Npad * 8, 256) * 2 + align_up(Npad * 4, 256) + 1024 + align_up((Npad / 16) * 256 * 8, 256);
}

int tadmm_eigh_f64(tadmm_handle h, const double* G, int N, double* evals_out, double* evecs_out, void* scratch,
                   size_t scratch_bytes, int* sweeps_out, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !G || !evals_out || !evecs_out || !scratch || N <= 0) return TADMM_ERR_INVALID;
  if (scratch_bytes < tadmm_eigh_scratch_bytes(N)) CTX_FAIL(h, TADMM_ERR_WORKSPACE, "eigh scratch too small");
  hipStream_t s = (hipStream_t)stream_;
  const int Npad = (int)align_up(N, 4 * kJB), ld = eig_ld(N),
            nb = Npad / kJB;
  int mode = getenv("TADMM_JACOBI_MODE") ? atoi(getenv("TADMM_JACOBI_MODE")) : 3;
  if (mode == 3 && (!jacobi_tick3_fits(ld) || ld % 64)) mode = 1;
  if (mode == 1 && !jacobi_tick2_fits(ld)) mode = 0;
  const bool super = mode >= 1;
  const int units = super ? nb / 2 : nb;
  const size_t tick_lds = mode == 1 ? jacobi_tick2_lds_bytes(ld) : jacobi_tick_lds_bytes(ld);
  char* base = (char*)scratch;
  size_t off = 0;
  double* XT = (double*)(base + off); off += align_up((size_t)Npad * ld * 8, 256);
  EigDesc* edev = (EigDesc*)(base + off); off += align_up(sizeof(EigDesc), 256);
  BlockRef* m_tick = (BlockRef*)(base + off); off += align_up(Npad * sizeof(BlockRef), 256);
  BlockRef* m_norm = (BlockRef*)(base + off); off += align_up(Npad * sizeof(BlockRef), 256);
  BlockRef* m_ext = (BlockRef*)(base + off); off += align_up(Npad * sizeof(BlockRef), 256);
  BlockRef* m_self = (BlockRef*)(base + off); off += align_up(Npad * sizeof(BlockRef), 256);
  double* lam = (double*)(base + off); off += align_up((size_t)Npad * 8, 256);
  double* sigma = (double*)(base + off); off += align_up((size_t)Npad * 8, 256);
  int32_t* order = (int32_t*)(base + off); off += align_up((size_t)Npad * 4, 256);
  double* offs = (double*)(base + off); off += 64;
  int32_t* done = (int32_t*)(base + off); off += 64;
  double* sblk = (double*)(base + off);
  HIP_OK(h, hipMemsetAsync(XT, 0, (size_t)Npad * ld * 8, s));
  HIP_OK(h, hipMemcpy2DAsync(XT, (size_t)ld * 8, G, (size_t)N * 8, (size_t)N * 8, N, hipMemcpyDeviceToDevice, s));
  EigDesc e;
  memset(&e, 0, sizeof e);
  e.XT = XT; e.N = N; e.Npad = Npad; e.ld = ld; e.nb = nb; e.off = offs; e.done = done; e.lam = lam; e.order = order;
  e.sigma = sigma; e.r = N; e.mode = 2; e.out_a = nullptr; e.out_b = nullptr; e.evec_out = evecs_out;
  e.sblk = sblk;
  std::vector<BlockRef> vt, vn, ve;
  for (int b = 0; b < units / 2; ++b) vt.push_back(BlockRef{0, b});
  for (int b = 0; b < (Npad + 3) / 4; ++b) vn.push_back(BlockRef{0, b});
  for (int b = 0; b < (N + 3) / 4; ++b) ve.push_back(BlockRef{0, b});
  HIP_OK(h, hipMemcpyAsync(edev, &e, sizeof e, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(m_tick, vt.data(), vt.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(m_norm, vn.data(), vn.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(m_ext, ve.data(), ve.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  std::vector<BlockRef> vs2;
  for (int b = 0; b < nb / 2; ++b) vs2.push_back(BlockRef{0, b});
  HIP_OK(h, hipMemcpyAsync(m_self, vs2.data(), vs2.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipStreamSynchronize(s));
  const double tol = 1e-9;
  int tick = 0, gs = 0;
  bool conv = false;
  // problems of at most 64 columns take the route the plans take (run_eig_group): the direct solver of tridiag.hip, then
  // jacobi_small_kernel for whatever that one did not certify.  TADMM_EIGH_TICK=1 keeps them on the tournament kernels.
  if (jacobi_small_fits(Npad) && !(getenv("TADMM_EIGH_TICK") && atoi(getenv("TADMM_EIGH_TICK")))) {
    int32_t* fast = done + 4;                 // device words inside the 64-byte `done` slot: [4] direct-solver flag,
    int* verdict = (int*)(done + 8);          // [8..9] verdict of the single-launch solvers
    HIP_OK(h, hipMemsetAsync(done, 0, 64, s));
    const bool direct = eig_small_direct_on();
    if (direct) launch_eig_small_direct(edev, 1, nullptr, fast, verdict, s);
    launch_jacobi_small(edev, 1, Npad, tol, 60, nullptr, verdict, s, false, direct ? fast : nullptr);
    int hv[6] = {0, 0, 0, 0, 0, 0};
    HIP_OK(h, hipMemcpyAsync(hv, done + 4, sizeof hv, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipStreamSynchronize(s));
    conv = hv[5] != 0;                        // verdict[1]
    gs = hv[0] ? 0 : 1;                       // 0 sweeps: solved by the direct route
    if (sweeps_out) *sweeps_out = gs;
    if (!conv) CTX_FAIL(h, TADMM_ERR_NOCONVERGE, "small eigen-solve did not converge");
  } else {
  launch_jacobi_init(edev, 1, s);
  double hoff[3];
  int hdone = 0;
  for (; gs < 40 && !conv; ++gs) {
    for (int t = 0; t < units - 1; ++t, ++tick) {
      if (mode >= 2) {
        if (t == 0) launch_jacobi_self(edev, m_self, units, tick, tol, 1, ld, s);
        launch_jacobi_tick3(edev, m_tick, (int)vt.size(), tick, tol, ld, s);
      } else {
        launch_jacobi_tick(edev, m_tick, (int)vt.size(), tick, tol, 1, tick_lds, mode == 1, s);
      }
    }
    HIP_OK(h, hipMemcpyAsync(hoff, offs, 24, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipMemcpyAsync(&hdone, done, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(h, hipStreamSynchronize(s));
    conv = hdone || hoff[gs & 1] < tol;
  }
  if (sweeps_out) *sweeps_out = gs;
  if (getenv("TADMM_STAMPS_DUMP")) { (void)hipStreamSynchronize(s); dump_stamps(); }   // -DTADMM_STAMPS builds only
  if (!conv) CTX_FAIL(h, TADMM_ERR_NOCONVERGE, "Jacobi did not converge in 40 sweeps");
  }
  launch_eig_norms(edev, m_norm, (int)vn.size(), s);
  launch_eig_sort(edev, 1, s, nullptr, Npad);
  launch_eig_extract(edev, m_ext, (int)ve.size(), s);
  // eigenvalues in descending order = sigma^2
  HIP_OK(h, hipGetLastError());
  // sigma holds sqrt(lambda); square it on the host side of the caller? keep device-only: reuse lam/order
  // -> evals_out[c] = lam[order[c]] via a tiny gather done with the extract's sigma: sigma^2
  //    (done by the caller-visible helper below to stay allocation-free)
  tadmm_square_copy(sigma, evals_out, N, s);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

// ---- building blocks of the filtered eigen-solver, exposed for tests ----
size_t tadmm_dgemm_scratch_bytes(int M, int N) {
  return align_up(sizeof(DgemmDesc), 256) + align_up((size_t)(M / 32) * (N / 32) * sizeof(BlockRef), 256);
}

int tadmm_dgemm_f64(tadmm_handle h, const double* A, const double* B, double* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int b_transposed, void* scratch, size_t scratch_bytes, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !A || !B || !C || !scratch) return TADMM_ERR_INVALID;
  if (M <= 0 || N <= 0 || K <= 0 || M % 32 || N % 32 || K % 16 || (lda & 1) || (ldb & 1) || (ldc & 1))
    CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_dgemm_f64: M, N multiples of 32, K of 16, even leading dimensions");
  if (scratch_bytes < tadmm_dgemm_scratch_bytes(M, N)) CTX_FAIL(h, TADMM_ERR_WORKSPACE, "dgemm scratch too small");
  hipStream_t s = (hipStream_t)stream_;
  DgemmDesc g;
  memset(&g, 0, sizeof g);
  g.A = A; g.B = B; g.C = C; g.selA = g.selB = g.selC = g.selP = g.selQ = -1;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.tiles_m = M / 32; g.tiles_n = N / 32;
  std::vector<BlockRef> map;
  for (int b = 0; b < g.tiles_m * g.tiles_n; ++b) map.push_back(BlockRef{0, b});
  DgemmDesc* gd = (DgemmDesc*)scratch;
  BlockRef* md = (BlockRef*)((char*)scratch + align_up(sizeof(DgemmDesc), 256));
  HIP_OK(h, hipMemcpyAsync(gd, &g, sizeof g, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(md, map.data(), map.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipStreamSynchronize(s));
  if (b_transposed && K % 32 == 0 && !getenv("TADMM_DGEMM_OLD")) {        // the 64x64 LDS-staged kernel (what the filter uses)
    std::vector<BlockRef> m64;
    for (int b = 0; b < ((M + 63) / 64) * ((N + 63) / 64); ++b) m64.push_back(BlockRef{0, b});
    HIP_OK(h, hipMemcpyAsync(md, m64.data(), m64.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
    HIP_OK(h, hipStreamSynchronize(s));
    launch_dgemm_nt64(gd, md, (int)m64.size(), s);
  } else {
    launch_dgemm(gd, md, (int)map.size(), b_transposed != 0, s);
  }
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

// C[M][N] = A[M][K] * G[N][K]^T at fp32 accuracy on the bf16 matrix cores (dgemm3.hip), the products the filter's early
// stages use: packs G into its three fragment-major planes, then one launch.  M % 32 == 0, N = K, N % 32 == 0.
size_t tadmm_dgemm3_scratch_bytes(int M, int N) {
  const size_t planes = (size_t)3 * (N / 16) * (N / 32) * 512 * 2;
  return align_up(sizeof(DgemmDesc), 256) + align_up(sizeof(GPlaneDesc), 256) + align_up(planes, 256) +
         2 * align_up(((size_t)(M / 32) * ((N + 63) / 64) + (size_t)(N / 16) * (N / 32) / 4 + 8) * sizeof(BlockRef), 256);
}

int tadmm_dgemm3_f64(tadmm_handle h, const double* A, const double* Gm, double* C, int M, int N, int lda, int ldg, int ldc,
                     int repeats, void* scratch, size_t scratch_bytes, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !A || !Gm || !C || !scratch) return TADMM_ERR_INVALID;
  if (M <= 0 || N <= 0 || M % 32 || N % 32 || lda < N || ldg < N || ldc < N || (lda & 1) || (ldg & 1) || (ldc & 1))
    CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_dgemm3_f64: M, N multiples of 32, even leading dimensions >= N");
  if (scratch_bytes < tadmm_dgemm3_scratch_bytes(M, N)) CTX_FAIL(h, TADMM_ERR_WORKSPACE, "dgemm3 scratch too small");
  hipStream_t s = (hipStream_t)stream_;
  char* base = (char*)scratch;
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
  const size_t o_g = take(sizeof(DgemmDesc)), o_p = take(sizeof(GPlaneDesc));
  const int nt = N / 16, ks = N / 32;
  const int64_t plane = (int64_t)nt * ks * 512;
  const size_t o_planes = take((size_t)3 * plane * 2);
  std::vector<BlockRef> m_gp, m_fast;
  for (int b = 0; b < (nt * ks + 3) / 4; ++b) m_gp.push_back(BlockRef{0, b});
  const int tn = (N + 63) / 64;
  for (int b = 0; b < (M / 32) * tn; ++b) m_fast.push_back(BlockRef{0, b});
  xcd_by_key(m_fast, [&](const BlockRef& b) { return b.local % tn; });
  const size_t o_mgp = take(m_gp.size() * sizeof(BlockRef)), o_mf = take(m_fast.size() * sizeof(BlockRef));
  GPlaneDesc gd;
  memset(&gd, 0, sizeof gd);
  gd.Gm = Gm; gd.ldg = ldg; gd.nt = nt; gd.ks = ks; gd.out = (uint16_t*)(base + o_planes); gd.plane = plane;
  DgemmDesc g;
  memset(&g, 0, sizeof g);
  g.A = A; g.C = C; g.selA = g.selB = g.selC = g.selP = g.selQ = -1;
  g.M = M; g.N = N; g.K = N; g.lda = lda; g.ldb = ldg; g.ldc = ldc; g.tiles_m = M / 32; g.tiles_n = N / 32;
  g.Gp = (const uint16_t*)(base + o_planes); g.g_plane = plane;
  HIP_OK(h, hipMemcpyAsync(base + o_g, &g, sizeof g, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(base + o_p, &gd, sizeof gd, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(base + o_mgp, m_gp.data(), m_gp.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(base + o_mf, m_fast.data(), m_fast.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipStreamSynchronize(s));
  launch_gplanes((const GPlaneDesc*)(base + o_p), (const BlockRef*)(base + o_mgp), (int)m_gp.size(), s);
  for (int i = 0; i < std::max(1, repeats); ++i)
    launch_dgemm3((const DgemmDesc*)(base + o_g), (const BlockRef*)(base + o_mf), (int)m_fast.size(), s);
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

size_t tadmm_cholqr_scratch_bytes(int n, int ncols) {
  const size_t cols64 = align_up(ncols, 64);
  return align_up(sizeof(DgemmDesc), 256) + align_up(sizeof(CholDesc), 256) + 2 * align_up((size_t)n * n * 8, 256) +
         align_up((size_t)n * 16 * 8, 256) + align_up((size_t)(n / 32) * (n / 32) * sizeof(BlockRef), 256) +
         align_up((cols64 / 64) * sizeof(BlockRef), 256) + 256;
}

int tadmm_cholqr_f64(tadmm_handle h, double* YT, int n, int ncols, int ldy, void* scratch, size_t scratch_bytes,
                     int* bad_out_host, void* stream_) {
  DeviceGuard device_guard(h);
  if (!h || !YT || !scratch || !bad_out_host) return TADMM_ERR_INVALID;
  if (n <= 0 || n > 256 || n % 32 || ncols <= 0 || ncols % 64 || ldy < ncols || (ldy & 1))
    CTX_FAIL(h, TADMM_ERR_INVALID, "tadmm_cholqr_f64: n multiple of 32 (<= 256), ncols multiple of 64, ldy >= ncols even");
  if (scratch_bytes < tadmm_cholqr_scratch_bytes(n, ncols)) CTX_FAIL(h, TADMM_ERR_WORKSPACE, "cholqr scratch too small");
  hipStream_t s = (hipStream_t)stream_;
  char* base = (char*)scratch;
  size_t off = 0;
  DgemmDesc* gd = (DgemmDesc*)(base + off); off += align_up(sizeof(DgemmDesc), 256);
  CholDesc* cd = (CholDesc*)(base + off); off += align_up(sizeof(CholDesc), 256);
  double* Cm = (double*)(base + off); off += align_up((size_t)n * n * 8, 256);
  double* Rm = (double*)(base + off); off += align_up((size_t)n * n * 8, 256);
  double* Wd = (double*)(base + off); off += align_up((size_t)n * 16 * 8, 256);
  BlockRef* mg = (BlockRef*)(base + off); off += align_up((size_t)(n / 32) * (n / 32) * sizeof(BlockRef), 256);
  BlockRef* ms = (BlockRef*)(base + off); off += align_up((size_t)(ncols / 64) * sizeof(BlockRef), 256);
  int32_t* bad = (int32_t*)(base + off);
  DgemmDesc g;
  memset(&g, 0, sizeof g);
  g.A = YT; g.B = YT; g.C = Cm; g.selA = g.selB = g.selC = g.selP = g.selQ = -1;
  g.M = n; g.N = n; g.K = ncols; g.lda = ldy; g.ldb = ldy; g.ldc = n; g.tiles_m = n / 32; g.tiles_n = n / 32;
  CholDesc c;
  memset(&c, 0, sizeof c);
  c.C = Cm; c.ldc = n; c.n = n; c.R = Rm; c.ldr = n; c.Wd = Wd;
  c.ring[0] = YT; c.ring[1] = YT; c.ring[2] = YT; c.rot = nullptr; c.sel = 0; c.ldy = ldy; c.ncols = ncols; c.bad = bad;
  std::vector<BlockRef> vg, vs;
  for (int b = 0; b < g.tiles_m * g.tiles_n; ++b) vg.push_back(BlockRef{0, b});
  for (int b = 0; b < ncols / 64; ++b) vs.push_back(BlockRef{0, b});
  HIP_OK(h, hipMemsetAsync(bad, 0, 4, s));
  HIP_OK(h, hipMemcpyAsync(gd, &g, sizeof g, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(cd, &c, sizeof c, hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(mg, vg.data(), vg.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipMemcpyAsync(ms, vs.data(), vs.size() * sizeof(BlockRef), hipMemcpyHostToDevice, s));
  HIP_OK(h, hipStreamSynchronize(s));
  launch_dgemm(gd, mg, (int)vg.size(), true, s);     // (32x32 kernel: ncols is only required to be a multiple of 16)
  launch_chol_factor(cd, 1, s);
  launch_chol_solve(cd, ms, (int)vs.size(), s);
  HIP_OK(h, hipMemcpyAsync(bad_out_host, bad, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(h, hipStreamSynchronize(s));
  HIP_OK(h, hipGetLastError());
  return TADMM_OK;
}

}  // extern "C"
