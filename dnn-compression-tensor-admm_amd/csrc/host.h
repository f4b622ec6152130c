// Host-side pieces shared by the plan builders (plan.hip: TT / SVD projection, tucker_plan.hip: Tucker-2):
// the context object behind tadmm_handle, error macros, the two-pass workspace arena, the XCD-aware block
// order of the Jacobi tournament, and the driver loop of one grouped eigen-solve.
#pragma once
#include "common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

struct tadmm_ctx_s {
  int device = 0;
  std::string err;
};

#define CTX_FAIL(h, code, ...)                                   \
  do {                                                           \
    char _b[512];                                                \
    snprintf(_b, sizeof _b, __VA_ARGS__);                        \
    if (h) (h)->err = _b;                                        \
    return (code);                                               \
  } while (0)

#define HIP_OK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t _e = (call);                                                                          \
    if (_e != hipSuccess) CTX_FAIL(h, TADMM_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)

namespace tadmm {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every C-ABI entry that launches or allocates runs on the handle's device, whatever device the calling thread
// has current (one process may drive several GPUs); the caller's choice is restored on exit.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const tadmm_ctx_s* h) {
    if (!h) return;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct Phase {  // one grouped launch: descriptor array + block map inside the device arena
  size_t desc_off = 0, map_off = 0;
  int nprob = 0, nblocks = 0;
};

// A simple bump allocator that is run twice: once with base==nullptr to size the workspace, once for real.
struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(char* b) : base(b) {}
  size_t take(size_t bytes, size_t align = 256) {
    off = align_up(off, align);
    const size_t o = off;
    off += bytes;
    return o;
  }
};

struct HostImage {   // host copy of the descriptor part of the arena
  std::vector<char> bytes;
  void put(size_t off, const void* src, size_t n) {
    if (bytes.size() < off + n) bytes.resize(off + n);
    memcpy(bytes.data() + off, src, n);
  }
};

// descriptor array + block map of one grouped launch -> descriptor arena (and the host image when given)
static inline void place_phase(Phase& ph, Arena& da, HostImage* img, const void* descs, size_t dbytes, int nprob,
                               const std::vector<BlockRef>& map) {
  ph.nprob = nprob;
  ph.nblocks = (int)map.size();
  ph.desc_off = da.take(std::max<size_t>(dbytes, 16));
  ph.map_off = da.take(std::max<size_t>(map.size() * sizeof(BlockRef), 16));
  if (img) {
    if (dbytes) img->put(ph.desc_off, descs, dbytes);
    if (!map.empty()) img->put(ph.map_off, map.data(), map.size() * sizeof(BlockRef));
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, MI355X_MICROARCH "Workgroup
// dispatch").  Re-order a block map so that CONSECUTIVE logical blocks land on the same XCD: the tournament
// hands a super-block from workgroup q to q+-1 between launches, so the next launch finds it in that XCD's L2.
static inline void xcd_group(std::vector<BlockRef>& m) {
  const char* e = getenv("TADMM_XCD_MAP");   // default on; 0 = plain order (for A/B measurements)
  if (e && !atoi(e)) return;
  const int G = (int)m.size();
  if (G < 16) return;
  std::vector<BlockRef> out(G);
  int i = 0;
  for (int x = 0; x < 8; ++x)
    for (int b = x; b < G; b += 8) out[b] = m[i++];
  m.swap(out);
}

// Grouped launches whose problems are re-read many times by their own blocks (the block products of the filtered
// eigen-solver: every 32x32 tile of a problem streams the problem's whole G and block image): give each PROBLEM to
// one XCD, so that its operands stay in that XCD's 4 MiB L2 instead of being fetched by all eight.  Blocks i, i+8, ..
// are observed to share an XCD; problems are dealt to the eight bins longest first, and a bin that runs dry steals
// from the fullest one (balance over locality).  Placement is a speed matter only.
static inline void xcd_by_problem(std::vector<BlockRef>& m, const std::vector<double>& weight) {
  const char* e = getenv("TADMM_XCD_MAP");
  if (e && !atoi(e)) return;
  const int G = (int)m.size();
  if (G < 64) return;
  const int np = (int)weight.size();
  std::vector<std::vector<BlockRef>> per(np);
  for (const BlockRef& b : m) per[b.prob].push_back(b);
  std::vector<int> order(np);
  for (int i = 0; i < np; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return weight[a] * per[a].size() > weight[b] * per[b].size();
  });
  std::vector<std::vector<BlockRef>> bin(8);
  double load[8] = {0};
  for (int p : order) {
    int best = 0;
    for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
    load[best] += weight[p] * per[p].size();
    bin[best].insert(bin[best].end(), per[p].begin(), per[p].end());
  }
  size_t head[8] = {0};
  std::vector<BlockRef> out;
  out.reserve(G);
  for (int i = 0; (int)out.size() < G; ++i) {
    int x = i & 7;
    if (head[x] >= bin[x].size()) {         // dry: steal from the bin with the most blocks left
      int full = -1; size_t left = 0;
      for (int y = 0; y < 8; ++y) if (bin[y].size() - head[y] > left) { left = bin[y].size() - head[y]; full = y; }
      if (full < 0) break;
      x = full;
    }
    out.push_back(bin[x][head[x]++]);
  }
  m.swap(out);
}

// The same dealing with an arbitrary affinity key per block (blocks with equal key share an XCD and hence an L2): dgemm3
// keys its tiles by (problem, column tile), so the slice of G a column tile reads is fetched into ONE L2.
template <class KeyFn>
static inline void xcd_by_key(std::vector<BlockRef>& m, KeyFn key) {
  const char* e = getenv("TADMM_XCD_MAP");
  if (e && !atoi(e)) return;
  const int G = (int)m.size();
  if (G < 64) return;
  std::vector<std::vector<BlockRef>> bin(8);
  for (const BlockRef& b : m) bin[(size_t)(key(b) & 7)].push_back(b);
  size_t head[8] = {0};
  std::vector<BlockRef> out;
  out.reserve(G);
  for (int i = 0; (int)out.size() < G; ++i) {
    int x = i & 7;
    if (head[x] >= bin[x].size()) {
      int full = -1; size_t left = 0;
      for (int y = 0; y < 8; ++y) if (bin[y].size() - head[y] > left) { left = bin[y].size() - head[y]; full = y; }
      if (full < 0) break;
      x = full;
    }
    out.push_back(bin[x][head[x]++]);
  }
  m.swap(out);
}

// Which Jacobi kernel a group of problems whose longest row is ld_max uses:
// 3 = tick3 (carried self-Grams) + self pass, 1 = LDS super-pair, 0 = plain pair kernel.
// TADMM_JACOBI_MODE (0 | 1 | 3) overrides the preference, never the capacity checks.
static inline int choose_jacobi_mode(int ld_max) {
  const char* em = getenv("TADMM_JACOBI_MODE");
  int want = em ? atoi(em) : 3;
  if (want != 0 && want != 1) want = 3;
  if (want == 3 && (!jacobi_tick3_fits(ld_max) || ld_max % 64)) want = 1;
  if (want == 1 && !jacobi_tick2_fits(ld_max)) want = 0;
  return want;
}

// One sweep period for every problem of a tick3 group (EigDesc::period); TADMM_JACOBI_ALIGN=0 restores per-problem periods.
static inline bool align_sweeps_on() {
  static const bool on = !(getenv("TADMM_JACOBI_ALIGN") && !atoi(getenv("TADMM_JACOBI_ALIGN")));
  return on;
}

// Pinned verdict slots + events of the pipelined convergence poll (one per plan).
struct PollCtx {
  hipEvent_t ev[2];
  int* host = nullptr;      // 2 slots of `stride` ints, written by jacobi_conv_kernel
  size_t stride = 0;
  bool made = false;
  hipError_t create(size_t max_problems) {
    stride = (max_problems + 1 + 15) & ~(size_t)15;
    hipError_t e = hipHostMalloc((void**)&host, 2 * stride * sizeof(int), hipHostMallocDefault);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    made = e == hipSuccess;
    return e;
  }
  void destroy() {
    if (!made) return;
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipHostFree(host);
    made = false;
  }
};

// instrumented runs only (tadmm_plan_enable_timing): every jacobi_tick3 launch timed on its own
struct JacobiTiming {
  bool on = false;
  hipEvent_t a{}, b{};
  double tick_ms = 0.0; int tick_launches = 0; double tick_flops = 0.0; double tick_wgs = 0.0;
  double small_ms = 0.0; int small_launches = 0; double small_flops = 0.0;    // single-launch solver (<= 64 columns)
  const int* small_n = nullptr;       // per problem: N of its eigen-problem (8 N^3 model of the launch's work)
};

// One grouped eigen-solve: `neig` symmetric problems whose descriptors / block maps already sit on the device.
struct EigGroup {
  const EigDesc* ed = nullptr;
  int neig = 0;
  const int* players = nullptr;     // per problem: tournament players (super-blocks or blocks)
  int gsteps = 0;                   // ticks per global sweep = max(players - 1)
  int mode = 0;
  int ld_max = 0;
  size_t tick_lds = 0;
  const BlockRef* tick_map = nullptr; int tick_blocks = 0;
  const BlockRef* self_map = nullptr; int self_blocks = 0;
  double* prev_dev = nullptr;       // [neig] scratch of the convergence kernel
  const int32_t* skip = nullptr;    // optional per-problem predicate (non-zero: the problem is dropped)
  int npad_max = 0;                 // largest padded problem size (<= 64: single-launch solver)
  int expected = 0;                 // sweeps the previous run of this group needed (0: unknown)
  bool aligned = false;             // every EigDesc carries period = gsteps: all sweeps start at the same tick
  bool warm = false;                // some EigDesc of the group carries a warm-start image (jacobi_small: second LDS image)
  const int* row_len = nullptr;     // per problem: ld of its X image (timing only: executed flops of a tick)
  const int* mid_sizes = nullptr;   // per problem: 128 / 192 when it may take the direct route (EigDesc::scratch set), else 0
  std::function<void(hipStream_t)> after_init;   // tick path only: queued right after jacobi_init (which takes the scale
                                                 // of a problem from X = G), e.g. the warm start X <- V G of big problems
  // debug only
  const double* off_dev = nullptr; const int* done_dev = nullptr;
};

// Runs jacobi_init + sweeps until every problem of the group has its `done` flag.
// Convergence is decided on the device after every sweep (jacobi_conv_kernel sets the sticky per-problem flags,
// so finished problems cost nothing in later launches).  The host only needs "all finished?" and reads that
// verdict one sweep late: the verdict of sweep g (written by the kernel straight into pinned host memory -- no
// copy in the stream) is consumed after sweep g+1 has been queued, so the GPU never idles on a poll; the price
// is one sweep of empty launches at the end.  The per-problem flags let the host stop launching the self pass
// for small problems that are long finished.
// Small groups (every problem <= 64 columns) run in ONE launch (jacobi_small_kernel decides convergence itself);
// *small_pending is set and the caller, after queueing the group's finalize launches, calls check_small_group.
static inline int run_eig_group(tadmm_handle h, const EigGroup& g, PollCtx& poll, double tol, int inner_sweeps,
                                int max_sweeps, bool debug, hipStream_t s, int* sweeps_out, bool* small_pending,
                                JacobiTiming* jt = nullptr) {
  *sweeps_out = 0;
  *small_pending = false;
  if (g.neig == 0) return TADMM_OK;
  {
    const char* e = getenv("TADMM_JACOBI_SMALL");     // 0: always use the tick kernels (A/B measurements)
    if (g.npad_max > 0 && jacobi_small_fits(g.npad_max) && !(e && !atoi(e))) {
      const bool timed = jt && jt->on;
      if (timed) (void)hipEventRecord(jt->a, s);
      // direct route first (tridiag.hip); the Jacobi launch behind it skips what that one solved and verified.  The flag
      // words live in the group's `prev` scratch (doubles of the tick path's convergence kernel, unused here).
      int32_t* fast = (g.prev_dev && eig_small_direct_on()) ? reinterpret_cast<int32_t*>(g.prev_dev) : nullptr;
      if (fast) launch_eig_small_direct(g.ed, g.neig, g.skip, fast, poll.host, s);
      launch_jacobi_small(g.ed, g.neig, g.npad_max, tol, std::max(max_sweeps, 60), g.skip, poll.host, s, g.warm, fast);
      if (timed) {
        float ms = 0.f;
        (void)hipEventRecord(jt->b, s);
        (void)hipEventSynchronize(jt->b);
        (void)hipEventElapsedTime(&ms, jt->a, jt->b);
        jt->small_ms += ms; jt->small_launches += 1;
        for (int q = 0; q < g.neig && jt->small_n; ++q) { const double nn = jt->small_n[q]; jt->small_flops += 8.0 * nn * nn * nn; }
      }
      HIP_OK(h, hipEventRecord(poll.ev[0], s));
      *small_pending = true;
      return TADMM_OK;
    }
  }
  launch_jacobi_init(g.ed, g.neig, s, g.skip, g.prev_dev);
  if (g.after_init) g.after_init(s);
  if (g.gsteps == 0) return TADMM_OK;      // every problem is a single block: nothing to rotate
  bool all_done = false;
  int tick = 0, gs = 0, pending = -1, needed = 0;
  std::vector<char> known_done(g.neig, 0);   // what the host has learnt so far (lags the device by a sweep)
  // Rayleigh-Ritz problems of 128 / 192 columns (EigDesc::scratch set by the filter layout): direct route first
  // (tridiag_mid.hip), ONE launch per size instead of ~80; what it solves and verifies has its `done` word set and the
  // tournament below skips it -- if that is every problem of the group the tournament is not queued at all.  The host has
  // to know, so this costs one stream round trip per group.
  if (g.mode >= 2 && g.mid_sizes && eig_mid_direct_on()) {
    bool any192 = false, any128 = false;
    for (int q = 0; q < g.neig; ++q) { any192 = any192 || g.mid_sizes[q] == 192; any128 = any128 || g.mid_sizes[q] == 128; }
    if (any192 || any128) {
      for (int q = 0; q <= g.neig; ++q) poll.host[q] = 0;
      if (any192) launch_eig_mid_direct(g.ed, g.neig, 192, g.skip, poll.host, s);
      if (any128) launch_eig_mid_direct(g.ed, g.neig, 128, g.skip, poll.host, s);
      HIP_OK(h, hipEventRecord(poll.ev[0], s));
      if (hipEventSynchronize(poll.ev[0]) != hipSuccess) CTX_FAIL(h, TADMM_ERR_HIP, "poll event failed");
      bool every = true;
      for (int q = 0; q < g.neig; ++q) {
        known_done[q] = poll.host[1 + q] != 0;
        every = every && (known_done[q] || g.players[q] < 2);
      }
      if (getenv("TADMM_MID_DEBUG")) {
        int ok = 0, cand = 0;
        for (int q = 0; q < g.neig; ++q) { ok += known_done[q] ? 1 : 0; cand += g.mid_sizes[q] ? 1 : 0; }
        fprintf(stderr, "[tadmm] direct RR route: %d of %d candidates solved (%d problems in the group)\n", ok, cand, g.neig);
      }
      if (every) { *sweeps_out = 0; return TADMM_OK; }
    }
  }
  std::vector<double> h_off;
  std::vector<int> h_done;
  auto consume = [&]() -> int {
    if (pending < 0) return TADMM_OK;
    if (hipEventSynchronize(poll.ev[pending & 1]) != hipSuccess) return TADMM_ERR_HIP;
    const int* v = poll.host + (size_t)(pending & 1) * poll.stride;
    if (v[0]) { all_done = true; needed = pending + 1; }
    for (int q = 0; q < g.neig; ++q) known_done[q] = v[1 + q] != 0;
    pending = -1;
    return TADMM_OK;
  };
  for (; gs < max_sweeps && !all_done; ++gs) {
    for (int t = 0; t < g.gsteps; ++t, ++tick) {
      if (g.mode >= 2) {
        bool any_first = false;     // does any unfinished problem start a sweep of its own at this tick?
        for (int q = 0; q < g.neig && !any_first; ++q)
          any_first = !known_done[q] && g.players[q] > 1 && (tick % (g.aligned ? g.gsteps : g.players[q] - 1)) == 0;
        if (any_first) launch_jacobi_self(g.ed, g.self_map, g.self_blocks, tick, tol, inner_sweeps, g.ld_max, s);
        const bool timed = jt && jt->on;
        if (timed) (void)hipEventRecord(jt->a, s);
        launch_jacobi_tick3(g.ed, g.tick_map, g.tick_blocks, tick, tol, g.ld_max, s);
        if (timed) {
          float ms = 0.f;
          (void)hipEventRecord(jt->b, s);
          (void)hipEventSynchronize(jt->b);
          (void)hipEventElapsedTime(&ms, jt->a, jt->b);
          jt->tick_ms += ms; jt->tick_launches += 1;
          // executed MFMA work of a workgroup (two 16-column super-blocks, rows of length ld): cross Gram 2*16*16*ld, two
          // rounds of column updates 2 * (2*16*16*ld) each -> 2560*ld; problems the host knows to be finished are left out,
          // workgroups of a padded schedule's idle ticks too
          for (int q = 0; q < g.neig; ++q) {
            if (known_done[q] || g.players[q] < 2) continue;
            const int own = g.players[q] - 1;
            if (g.aligned && (tick % g.gsteps) >= own) continue;
            const double ld = g.row_len ? g.row_len[q] : g.ld_max;
            jt->tick_flops += 2560.0 * ld * (g.players[q] / 2);
            jt->tick_wgs += g.players[q] / 2;
          }
        }
      } else {
        launch_jacobi_tick(g.ed, g.tick_map, g.tick_blocks, tick, tol, inner_sweeps, g.tick_lds, g.mode == 1, s);
      }
    }
    launch_jacobi_conv(g.ed, g.neig, tick, tol, g.mode >= 1, g.prev_dev, poll.host + (size_t)(gs & 1) * poll.stride, s);
    {   // a failed launch (LDS attribute not set on this device, bad configuration) must surface, not spin to max_sweeps
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) CTX_FAIL(h, TADMM_ERR_HIP, "Jacobi launch failed: %s", hipGetErrorString(le));
    }
    if (debug && g.off_dev) {
      h_off.resize((size_t)g.neig * 3); h_done.resize(g.neig);
      HIP_OK(h, hipMemcpyAsync(h_off.data(), g.off_dev, (size_t)g.neig * 3 * 8, hipMemcpyDeviceToHost, s));
      HIP_OK(h, hipMemcpyAsync(h_done.data(), g.done_dev, (size_t)g.neig * 4, hipMemcpyDeviceToHost, s));
      HIP_OK(h, hipStreamSynchronize(s));
      double mxo = 0; int nd = 0;
      for (int q = 0; q < g.neig; ++q) {
        const int steps = g.players[q] - 1;
        if (h_done[q]) ++nd;
        if (steps > 0 && tick % steps == 0) mxo = std::max(mxo, h_off[3 * q + ((tick / steps - 1) & 1)]);
      }
      fprintf(stderr, "[tadmm]   sweep %d: max observed off %.3e, done %d/%d\n", gs, mxo, nd, g.neig);
    }
    int rc = consume();                     // verdict of the previous sweep (long since on the host)
    if (rc != TADMM_OK) CTX_FAIL(h, rc, "poll event failed");
    if (all_done) break;
    // The verdict only matters near the end: with a sweep count to go by (the previous run's), the sweeps well before it
    // queue no event (a marker packet costs the chain ~5 us a sweep) and are not polled; a solve that finishes much
    // earlier than last time runs a few sweeps of gated (empty) launches before the host notices.
    static const bool lazy_poll = !(getenv("TADMM_POLL_EVERY_SWEEP") && atoi(getenv("TADMM_POLL_EVERY_SWEEP")));
    // (not in instrumented runs: their flop count leaves out the problems the host KNOWS to be finished)
    if (lazy_poll && !(jt && jt->on) && g.expected > 0 && gs + 3 < g.expected) continue;
    HIP_OK(h, hipEventRecord(poll.ev[gs & 1], s));
    pending = gs;
    if (g.expected > 0 && gs + 1 >= g.expected) {
      // Jacobi needs about the same number of sweeps from one ADMM iteration to the next: from the sweep that
      // was the last one last time, wait for the verdict (one short stall) instead of queueing a sweep of
      // launches that would most likely find every problem finished.
      rc = consume();
      if (rc != TADMM_OK) CTX_FAIL(h, rc, "poll event failed");
    }
  }
  if (!all_done) {
    const int rc = consume();
    if (rc != TADMM_OK) CTX_FAIL(h, rc, "poll event failed");
  }
  if (debug) {
    int pmax = 0;
    for (int q = 0; q < g.neig; ++q) pmax = std::max(pmax, g.players[q]);
    fprintf(stderr, "[tadmm] eig group: neig=%d players_max=%d ticks/sweep=%d wgs/tick=%d sweeps=%d ticks=%d mode=%d\n",
            g.neig, pmax, g.gsteps, g.tick_blocks, all_done ? needed : gs, tick, g.mode);
  }
  if (!all_done) CTX_FAIL(h, TADMM_ERR_NOCONVERGE, "Jacobi did not converge in %d sweeps", max_sweeps);
  *sweeps_out = needed;
  return TADMM_OK;
}

// The single-launch solver reports per-problem convergence through pinned memory: fail loudly if one hit the cap.
static inline int check_small_group(tadmm_handle h, const EigGroup& g, PollCtx& poll) {
  if (hipEventSynchronize(poll.ev[0]) != hipSuccess) CTX_FAIL(h, TADMM_ERR_HIP, "poll event failed");
  for (int q = 0; q < g.neig; ++q)
    if (!poll.host[1 + q]) CTX_FAIL(h, TADMM_ERR_NOCONVERGE, "Jacobi (small problems) did not converge, problem %d", q);
  return TADMM_OK;
}

}  // namespace tadmm
