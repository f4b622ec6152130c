// Batched Tucker-2 projection (reference admm.py:113-127 -> tensorly `partial_tucker(modes=[0,1], init='svd')` +
// `tucker_to_tensor`): HOSVD initialisation followed by HOOI sweeps, for ALL Tucker layers of a model at once.
//
// PARITY UNPINNED: tensorly is un-vendored and absent; this restates the published
// tensorly<=0.7 algorithm -- n_iter_max = 100, tol = 1e-4 on successive relative reconstruction errors
// sqrt(|‖X‖²-‖core‖²|)/‖X‖, checked from the third sweep.
//
// Layout: T = unfold(W+U) as (O, K2, I) (the same LDS-staged sweep as the TT path), so
//   mode-0 unfolding = T viewed (O x K2*I),       mode-1 unfolding = (T viewed (O*K2 x I))^T.
// One HOOI sweep for every layer still iterating is a fixed sequence of grouped launches:
//   P  = T x_1 U_in^T      (O*K2 x I)(I x r_in)                 fp32 MFMA GEMM
//   U_out <- leading r_out LEFT singular vectors of P viewed (O x K2*r_in)      Gram (fp64 MFMA) + Jacobi + extract
//   P  = T x_0 U_out^T     (r_out x O)(O x K2*I)                 GEMM
//   U_in  <- leading r_in RIGHT singular vectors of P viewed (r_out*K2 x I)
//   C  = P x_1 U_in^T      (r_out*K2 x I)(I x r_in)              GEMM   (the core, (r_out, K2, r_in))
//   ‖C‖², stopping rule                                          on the device: finished layers get a `skip` flag
// that turns their blocks of every later launch into no-ops; the host reads "all finished?" one sweep late from
// pinned memory (same scheme as the Jacobi convergence poll).  Then Z = C x_0 U_out x_1 U_in (two GEMMs) and the
// fused fold / U += W-Z / ‖W-Z‖² sweep.
#include "host.h"

#include <cmath>

using namespace tadmm;

namespace {

struct TkArena {
  size_t off = 0;
  size_t take(size_t bytes, size_t align = 256) {
    off = align_up(off, align);
    const size_t o = off;
    off += bytes;
    return o;
  }
};

struct OpGeom {   // eigen-problem geometry of one "leading singular vectors" request
  int m = 0, n = 0, N = 0, Npad = 0, ld = 0, nb = 0, nt = 0, ksplit = 1, kchunk = 0, r_eff = 0;
  bool trans = false;
};

OpGeom op_geom(int m, int n, int r) {
  OpGeom g;
  g.m = m; g.n = n; g.trans = m > n;
  g.N = std::min(m, n);
  g.Npad = (int)align_up(g.N, 4 * kJB);
  g.nb = g.Npad / kJB;
  g.ld = eig_ld(g.N);
  g.nt = (g.N + 31) / 32;
  const int64_t K = g.trans ? m : n;
  const int ntp = g.nt * (g.nt + 1) / 2;
  int ks = (64 + ntp - 1) / ntp;
  const int maxks = (int)std::max<int64_t>(1, (K + 255) / 256);
  ks = std::max(1, std::min(ks, maxks));
  ks = std::max(ks, (int)((K + 2047) / 2048));
  g.kchunk = (int)align_up((K + ks - 1) / ks, 64);
  g.ksplit = (int)((K + g.kchunk - 1) / g.kchunk);
  g.r_eff = std::min(r, g.N);
  return g;
}

struct Group {   // descriptors + block map of one grouped launch (offsets into the workspace)
  size_t desc_off = 0, map_off = 0;
  int nblocks = 0;
};

// Warm start of a streamed (N > 1152) HOOI eigen-solve: X0 = V G with V the eigenvectors of the same mode's previous solve.
// The solve's scale is taken from X = G first (jacobi_init), then G is copied aside and one gated fp64 tile GEMM
// rewrites the image; after the solve the normalised columns of X become the next V (bigwarm_* kernels).
struct BigWarm {
  int layer = 0, Npad = 0, ld = 0, nblocks = 0;
  bool apply = true;                          // false: the HOSVD start of the same mode only SAVES its eigenvectors
                                              // (first HOOI sweep: the projected Gram has nearly the same leading ones)
  size_t V = 0, Gt = 0, XT = 0, ok = 0;      // workspace offsets: [Npad][ld] eigenvectors, G copy, the X image, int flag
  size_t desc_off = 0, map_off = 0;           // one DgemmDesc + its 64 x 64 tile map
};

struct Lsv {     // one grouped "leading singular vectors" phase over all layers
  std::vector<BigWarm> big;
  Group gram_p, gram_r, tick, self, norm, ext, xg;
  size_t eig_desc_off = 0;
  std::vector<int> players;
  std::vector<int> players_n;     // per problem: N of the eigen-problem (timing: 8 N^3 model)
  int gsteps = 0, mode = 0, ld_max = 0, npad_max = 0;
  bool warm = false;              // some problem of the group has a warm-start image
  size_t tick_lds = 0;
};

struct TLayer {
  int O = 0, I = 0, K2 = 1, ro = 0, ri = 0;
  int64_t numel = 0;
  size_t T = 0, P = 0, C = 0, Uo = 0, Ui = 0, Vs = 0, XT = 0, lam = 0, order = 0, sigma = 0, sblk = 0, gpart = 0;
  size_t warm[2] = {0, 0};        // eigenvectors of the previous HOOI solve of modes 0 / 1 (problems of <= 64 columns)
  size_t bigV[2] = {0, 0}, bigG = 0;   // the same for streamed problems (N > kLdResidentMax): [Npad][ld] + a G copy
  int reff[4] = {0, 0, 0, 0};   // vectors each of the four singular-vector requests can deliver
};

// ‖x‖² of one fp32 buffer per workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* const* __restrict__ ptrs, const int64_t* __restrict__ numel,
                                                    double* __restrict__ out, const int32_t* __restrict__ skip) {
  const int l = blockIdx.x;
  if (skip && skip[l]) return;
  __shared__ double red[256];
  const float* __restrict__ x = ptrs[l];
  const int64_t n = numel[l];
  double acc = 0.0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // 16-byte loads, four in flight per thread (one workgroup reads the whole tensor: latency, not bandwidth, is the limit)
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    const int64_t n4 = n >> 2;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
      const float4 u0 = x4[i], u1 = x4[i + 256], u2 = x4[i + 512], u3 = x4[i + 768];
      a0 += (double)u0.x * u0.x + (double)u0.y * u0.y + (double)u0.z * u0.z + (double)u0.w * u0.w;
      a1 += (double)u1.x * u1.x + (double)u1.y * u1.y + (double)u1.z * u1.z + (double)u1.w * u1.w;
      a2 += (double)u2.x * u2.x + (double)u2.y * u2.y + (double)u2.z * u2.z + (double)u2.w * u2.w;
      a3 += (double)u3.x * u3.x + (double)u3.y * u3.y + (double)u3.z * u3.z + (double)u3.w * u3.w;
    }
    for (; i < n4; i += 256) {
      const float4 u0 = x4[i];
      a0 += (double)u0.x * u0.x + (double)u0.y * u0.y + (double)u0.z * u0.z + (double)u0.w * u0.w;
    }
    acc = (a0 + a1) + (a2 + a3);
    for (int64_t j = (n4 << 2) + threadIdx.x; j < n; j += 256) { const double v = x[j]; acc += v * v; }
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 256) { const double v = x[i]; acc += v * v; }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[l] = red[0];
}

// verdict on a finished streamed solve: may its columns serve as the next start?  (converged, every genuine column
// normalisable: norm above 1e-8 of the largest -- below that the direction is rounding residue)
__global__ __launch_bounds__(256) void bigwarm_check_kernel(const EigDesc* __restrict__ descs, int prob,
                                                            const int32_t* __restrict__ skip, int32_t* __restrict__ ok) {
  if (skip && skip[prob]) return;
  const EigDesc d = descs[prob];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = *d.done ? 0 : 1;
  __syncthreads();
  const double floor_ = 1e-8 * d.lam[d.order[0]];
  int b = 0;
  for (int j = threadIdx.x; j < d.N; j += 256) b |= !(d.lam[j] > floor_ && d.lam[j] > 0.0);
  if (b) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) *ok = bad ? 0 : 1;
}
// V[j][:] = X[j][:] / |X[j]| (one wave per row; rows beyond N and the padding stay zero)
__global__ __launch_bounds__(256) void bigwarm_save_kernel(const EigDesc* __restrict__ descs, int prob,
                                                           const int32_t* __restrict__ skip,
                                                           const int32_t* __restrict__ ok, double* __restrict__ V) {
  if ((skip && skip[prob]) || !*ok) return;
  const EigDesc d = descs[prob];
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= d.Npad) return;
  const double inv = j < d.N ? 1.0 / d.lam[j] : 0.0;
  const double* __restrict__ x = d.XT + (int64_t)j * d.ld;
  double* __restrict__ v = V + (int64_t)j * d.ld;
  for (int i = lane; i < d.ld; i += 64) v[i] = x[i] * inv;
}

// tensorly's stopping rule after HOOI sweep `it` (0-based), per layer; err: [2][n] ping-pong of the relative errors
__global__ __launch_bounds__(256) void hooi_conv_kernel(int n, int it, const float* __restrict__ tol,
                                                        const double* __restrict__ nT, const double* __restrict__ nC,
                                                        double* __restrict__ err, int32_t* __restrict__ skip,
                                                        int32_t* __restrict__ iters, int* __restrict__ verdict) {
  __shared__ int open_layers;
  if (threadIdx.x == 0) open_layers = 0;
  __syncthreads();
  for (int l = threadIdx.x; l < n; l += 256) {
    if (skip[l]) { verdict[1 + l] = 1; continue; }
    const double t = nT[l];
    const double e = t > 0.0 ? sqrt(fabs(t - nC[l])) / sqrt(t) : 0.0;
    const double prev = err[((it + 1) & 1) * n + l];
    err[(it & 1) * n + l] = e;
    iters[l] = it + 1;
    const bool conv = it > 1 && fabs(e - prev) < (double)tol[l];
    if (conv) skip[l] = 1;
    else open_layers = 1;
    verdict[1 + l] = conv ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) verdict[0] = open_layers ? 0 : 1;
}

}  // namespace

struct tadmm_tucker_plan_s {
  tadmm_handle h = nullptr;
  int n = 0;
  std::vector<tadmm_layer_desc> descs;
  std::vector<TLayer> L;
  char* ws = nullptr;
  size_t ws_bytes = 0, desc_bytes = 0;
  Lsv lsv[4];            // 0: init U_out, 1: init U_in, 2: HOOI U_out, 3: HOOI U_in
  Group gemm[5];         // 0: P = T x1 U_in, 1: P = T x0 U_out, 2: C, 3: Z' = C x1 U_in, 4: Zmat = Z' x0 U_out
  Group unfold, fold;
  size_t sweep_desc_off = 0, resid_partial_off = 0, fac_begin = 0, fac_end = 0;
  size_t warm_ok_off = 0;
  size_t off_off = 0, done_off = 0, prev_off = 0, skip_off = 0, nT_off = 0, nC_off = 0, err_off = 0, iters_off = 0,
         tol_off = 0, ptrT_off = 0, ptrC_off = 0, numT_off = 0, numC_off = 0;
  PollCtx poll;          // Jacobi verdicts
  PollCtx hooi;          // HOOI verdicts
  double jtol = 1e-9;
  int inner = 1, max_sweeps = 40, n_iter_max = 100;
  bool debug = false;
  int last_hooi = 0, last_jacobi_sweeps = 0;
  // instrumented runs (tadmm_tucker_enable_timing): every launch of the eigen-solver timed on the launch stream
  JacobiTiming jtm;
  bool ev_made = false;
  hipEvent_t tev[4];
  double last_total_ms = 0.0;
};

// Lays the plan out in `base` (nullptr: sizes only).  `img` receives the host copy of the descriptor region.
// HOOI solves of a mode start from the eigenvectors of that mode's previous solve (TADMM_TUCKER_WARM=0: always cold)
static bool warm_start_on() {      // read at every plan creation (tests switch it inside one process)
  const char* e = getenv("TADMM_TUCKER_WARM");
  return !(e && !atoi(e));
}

static int tucker_layout(tadmm_tucker_plan_s* P, char* base, const float* const* W, float* const* U, float* const* Z,
                         std::vector<char>* img, size_t desc_region, size_t* desc_bytes, size_t* total_bytes) {
  tadmm_handle h = P->h;
  const int n = P->n;
  TkArena da, ar;
  ar.off = desc_region;
  auto dev = [&](size_t off) -> char* { return base ? base + off : reinterpret_cast<char*>((uintptr_t)off); };
  auto put = [&](size_t off, const void* src, size_t bytes) {
    if (!img || bytes == 0) return;
    if (img->size() < off + bytes) img->resize(off + bytes);
    memcpy(img->data() + off, src, bytes);
  };
  auto place = [&](Group& g, const void* descs, size_t dbytes, const std::vector<BlockRef>& map) {
    g.desc_off = da.take(std::max<size_t>(dbytes, 16));
    g.map_off = da.take(std::max<size_t>(map.size() * sizeof(BlockRef), 16));
    g.nblocks = (int)map.size();
    put(g.desc_off, descs, dbytes);
    put(g.map_off, map.data(), map.size() * sizeof(BlockRef));
  };

  // ---- per-layer data buffers ----
  // the factors of all layers sit back to back so that one memset clears them at the start of a run
  P->fac_begin = ar.take(0);
  for (int l = 0; l < n; ++l) {
    TLayer& t = P->L[l];
    t.Uo = ar.take((size_t)t.O * t.ro * 4);
    t.Ui = ar.take((size_t)t.I * t.ri * 4);
  }
  P->fac_end = ar.off;
  std::vector<OpGeom> og(4 * n);
  for (int l = 0; l < n; ++l) {
    TLayer& t = P->L[l];
    og[0 * n + l] = op_geom(t.O, t.K2 * t.I, t.ro);        // init  : LEFT  of T (O x K2 I)
    og[1 * n + l] = op_geom(t.O * t.K2, t.I, t.ri);        // init  : RIGHT of T (O K2 x I)
    og[2 * n + l] = op_geom(t.O, t.K2 * t.ri, t.ro);       // HOOI  : LEFT  of P (O x K2 r_in)
    og[3 * n + l] = op_geom(t.ro * t.K2, t.I, t.ri);       // HOOI  : RIGHT of P (r_out K2 x I)
    size_t xtb = 0, gpb = 0, vsb = 0, npad = 0;
    for (int k = 0; k < 4; ++k) {
      const OpGeom& g = og[k * n + l];
      t.reff[k] = g.r_eff;
      if (!jacobi_size_supported(g.N))
        CTX_FAIL(h, TADMM_ERR_UNSUPPORTED, "Tucker layer %d: eigen-problem of size %d exceeds the Jacobi kernels (max %d)", l, g.N, kJacobiMaxN);
      xtb = std::max(xtb, (size_t)g.Npad * g.ld * 8);
      gpb = std::max(gpb, (size_t)g.ksplit * (g.nt * (g.nt + 1) / 2) * 1024 * 8);
      vsb = std::max(vsb, (size_t)g.N * std::max(1, g.r_eff) * 4);
      npad = std::max<size_t>(npad, g.Npad);
    }
    t.T = ar.take((size_t)t.numel * 4);
    t.P = ar.take((size_t)std::max((int64_t)t.O * t.K2 * t.ri, (int64_t)t.ro * t.K2 * t.I) * 4);
    t.C = ar.take((size_t)t.ro * t.K2 * t.ri * 4);
    t.Vs = ar.take(vsb);
    t.XT = ar.take(xtb);
    t.gpart = ar.take(gpb);
    t.lam = ar.take(npad * 8);
    t.sigma = ar.take(npad * 8);
    t.order = ar.take(npad * 4);
    t.sblk = ar.take((npad / 16) * 256 * 8);
    t.bigG = 0;          // (the layout runs twice on the same TLayer: sizes, then addresses)
    for (int k = 0; k < 2; ++k) {
      t.bigV[k] = 0;
      const int np = og[(2 + k) * n + l].Npad;
      t.warm[k] = jacobi_small_fits(np) ? ar.take((size_t)np * np * 8) : 0;
      const OpGeom& gk = og[(2 + k) * n + l];
      if (gk.ld > kLdResidentMax && warm_start_on()) {
        t.bigV[k] = ar.take((size_t)gk.Npad * gk.ld * 8);
        if (!t.bigG) t.bigG = ar.take(xtb);
      }
    }
  }
  P->off_off = ar.take((size_t)n * 3 * 8);
  P->done_off = ar.take((size_t)n * 4);
  P->prev_off = ar.take((size_t)n * 8);
  P->skip_off = ar.take((size_t)n * 4);
  P->warm_ok_off = ar.take((size_t)n * 2 * 4);
  P->nT_off = ar.take((size_t)n * 8);
  P->nC_off = ar.take((size_t)n * 8);
  P->err_off = ar.take((size_t)n * 2 * 8);
  P->iters_off = ar.take((size_t)n * 4);

  // ---- sweep descriptors (unfold / fold_update), same rules as the TT plan ----
  std::vector<SweepDesc> sd(n);
  std::vector<BlockRef> smap;
  for (int l = 0; l < n; ++l) {
    const TLayer& t = P->L[l];
    SweepDesc& s = sd[l];
    memset(&s, 0, sizeof s);
    s.W = W ? W[l] : nullptr; s.U = U ? U[l] : nullptr; s.Z = Z ? Z[l] : nullptr;
    s.T0 = (float*)dev(t.T);
    s.Zmat = (const float*)dev(t.T);          // the reconstruction overwrites T (dead after the HOOI loop)
    s.O = t.O; s.I = t.I; s.K2 = t.K2;
    s.numel = t.numel;
    if (t.K2 > 1) {
      int ich = std::min(t.I, 256);
      while ((int64_t)t.K2 * (ich + 4) > 12288 && ich > 1) ich /= 2;
      s.ichunk = ich;
      s.nchunk = (t.I + ich - 1) / ich;
      s.nblk = t.O * s.nchunk;
    } else {
      s.ichunk = 8192;
      s.nchunk = (int)((t.numel + s.ichunk - 1) / s.ichunk);
      s.nblk = s.nchunk;
    }
    s.blk_begin = (int)smap.size();
    for (int b = 0; b < s.nblk; ++b) smap.push_back(BlockRef{l, b});
  }
  P->resid_partial_off = ar.take(smap.size() * 8);
  P->sweep_desc_off = da.take(sd.size() * sizeof(SweepDesc));
  put(P->sweep_desc_off, sd.data(), sd.size() * sizeof(SweepDesc));
  P->unfold.map_off = da.take(smap.size() * sizeof(BlockRef));
  P->unfold.nblocks = (int)smap.size();
  put(P->unfold.map_off, smap.data(), smap.size() * sizeof(BlockRef));
  P->fold = P->unfold;

  // ---- small per-layer tables: tolerances, pointer / length lists of the norm kernel ----
  {
    std::vector<float> tol(n);
    std::vector<const float*> pT(n), pC(n);
    std::vector<int64_t> nT(n), nC(n);
    for (int l = 0; l < n; ++l) {
      const TLayer& t = P->L[l];
      tol[l] = P->descs[l].hooi_tol > 0.f ? P->descs[l].hooi_tol : 1e-4f;
      pT[l] = (const float*)dev(t.T); nT[l] = t.numel;
      pC[l] = (const float*)dev(t.C); nC[l] = (int64_t)t.ro * t.K2 * t.ri;
    }
    P->tol_off = da.take((size_t)n * 4);   put(P->tol_off, tol.data(), (size_t)n * 4);
    P->ptrT_off = da.take((size_t)n * 8);  put(P->ptrT_off, pT.data(), (size_t)n * 8);
    P->ptrC_off = da.take((size_t)n * 8);  put(P->ptrC_off, pC.data(), (size_t)n * 8);
    P->numT_off = da.take((size_t)n * 8);  put(P->numT_off, nT.data(), (size_t)n * 8);
    P->numC_off = da.take((size_t)n * 8);  put(P->numC_off, nC.data(), (size_t)n * 8);
  }

  // ---- the four "leading singular vectors" phases ----
  for (int k = 0; k < 4; ++k) {
    Lsv& v = P->lsv[k];
    const bool want_left = (k == 0 || k == 2);
    std::vector<GramDesc> gd(n);
    std::vector<EigDesc> ed(n);
    std::vector<GemmDesc> xg(n);
    std::vector<BlockRef> m_gp, m_gr, m_tick, m_self, m_norm, m_ext, m_xg;
    std::vector<int> gp_cost(n);
    v.players.assign(n, 0);
    v.players_n.assign(n, 0);
    v.ld_max = 0;
    v.npad_max = 0;
    v.warm = false;
    v.big.clear();
    for (int l = 0; l < n; ++l) {
      v.ld_max = std::max(v.ld_max, og[k * n + l].ld);
      v.npad_max = std::max(v.npad_max, og[k * n + l].Npad);
    }
    v.mode = choose_jacobi_mode(v.ld_max);
    v.tick_lds = v.mode == 1 ? jacobi_tick2_lds_bytes(v.ld_max) : jacobi_tick_lds_bytes(v.ld_max);
    v.gsteps = 0;
    for (int l = 0; l < n; ++l) {
      const TLayer& t = P->L[l];
      const OpGeom& g = og[k * n + l];
      const float* A = (k < 2) ? (const float*)dev(t.T) : (const float*)dev(t.P);
      float* factor = want_left ? (float*)dev(t.Uo) : (float*)dev(t.Ui);
      const int r_full = want_left ? t.ro : t.ri;
      GramDesc& q = gd[l];
      memset(&q, 0, sizeof q);
      q.A = A; q.m = g.m; q.n = g.n; q.trans = g.trans ? 1 : 0; q.N = g.N; q.K = g.trans ? g.m : g.n;
      q.nt = g.nt; q.ksplit = g.ksplit; q.kchunk = g.kchunk;
      q.partial = (double*)dev(t.gpart); q.G = (double*)dev(t.XT); q.Npad = g.Npad; q.ld = g.ld;
      const int ntp = g.nt * (g.nt + 1) / 2;
      for (int b = 0; b < g.ksplit * ntp; ++b) m_gp.push_back(BlockRef{l, b});
      gp_cost[l] = g.kchunk;
      if (g.ksplit > 1)
        for (int b = 0; b < (int)(((int64_t)g.Npad * g.ld + 1023) / 1024); ++b) m_gr.push_back(BlockRef{l, b});
      // the Gram's eigenvectors live on the SMALL side of the matrix: LEFT when m <= n, RIGHT when m > n
      const bool have_left = !g.trans;
      const bool direct = (have_left == want_left);
      EigDesc& e = ed[l];
      memset(&e, 0, sizeof e);
      e.XT = (double*)dev(t.XT); e.N = g.N; e.Npad = g.Npad; e.ld = g.ld; e.nb = g.nb;
      e.off = (double*)dev(P->off_off) + 3 * l;
      e.done = (int32_t*)dev(P->done_off) + l;
      e.lam = (double*)dev(t.lam); e.order = (int32_t*)dev(t.order); e.sigma = (double*)dev(t.sigma);
      e.r = g.r_eff; e.sblk = (double*)dev(t.sblk);
      if (k >= 2 && t.warm[k - 2] && warm_start_on()) {
        e.warm = (double*)dev(t.warm[k - 2]);
        e.warm_ok = (int32_t*)dev(P->warm_ok_off) + 2 * l + (k - 2);
        v.warm = true;
      }
      const bool big_apply = k >= 2 && t.bigV[k - 2];
      const bool big_save_only = k < 2 && t.bigV[k] && og[(k + 2) * n + l].N == g.N && og[(k + 2) * n + l].ld == g.ld;
      if (big_apply || big_save_only) {
        BigWarm bw;
        bw.layer = l; bw.Npad = g.Npad; bw.ld = g.ld; bw.apply = big_apply;
        bw.V = t.bigV[k & 1]; bw.Gt = t.bigG; bw.XT = t.XT;
        bw.ok = P->warm_ok_off + (size_t)(2 * l + (k & 1)) * 4;
        DgemmDesc dg;
        memset(&dg, 0, sizeof dg);
        dg.A = (const double*)dev(bw.V); dg.B = (const double*)dev(bw.Gt); dg.C = (const double*)dev(bw.XT);
        dg.selA = dg.selB = dg.selC = dg.selP = dg.selQ = -1;
        dg.M = dg.N = dg.K = g.Npad; dg.lda = dg.ldb = dg.ldc = g.ld;
        dg.tiles_m = g.Npad / 32; dg.tiles_n = g.Npad / 32;
        dg.gate = (const int32_t*)dev(bw.ok); dg.gate_min = 1;      // no-op while the previous solve left no usable V
        std::vector<BlockRef> m64;
        const int t64 = (g.Npad + 63) / 64;
        for (int b = 0; b < t64 * t64; ++b) m64.push_back(BlockRef{0, b});
        bw.nblocks = (int)m64.size();
        bw.desc_off = da.take(sizeof(DgemmDesc));
        bw.map_off = da.take(m64.size() * sizeof(BlockRef));
        put(bw.desc_off, &dg, sizeof dg);
        put(bw.map_off, m64.data(), m64.size() * sizeof(BlockRef));
        v.big.push_back(bw);
      }
      if (direct) { e.mode = 0; e.out_a = factor; e.ldo = r_full; }
      else { e.mode = 3; e.out_a = (float*)dev(t.Vs); e.ldo = 0; }
      const int units = v.mode >= 1 ? g.nb / 2 : g.nb;
      v.players[l] = units;
      v.players_n[l] = g.N;
      v.gsteps = std::max(v.gsteps, units - 1);
      for (int b = 0; b < units / 2; ++b) m_tick.push_back(BlockRef{l, b});
      if (v.mode >= 2) for (int b = 0; b < units; ++b) m_self.push_back(BlockRef{l, b});
      for (int b = 0; b < (g.Npad + 3) / 4; ++b) m_norm.push_back(BlockRef{l, b});
      for (int b = 0; b < (g.r_eff + 3) / 4; ++b) m_ext.push_back(BlockRef{l, b});
      // the other side: factor = A * (V / sigma)  |  A^T * (U / sigma)
      GemmDesc& x = xg[l];
      memset(&x, 0, sizeof x);
      if (!direct) {
        x.B = (const float*)dev(t.Vs); x.C = factor;
        x.N = g.r_eff; x.b_rs = g.r_eff; x.b_cs = 1; x.c_rs = r_full; x.c_cs = 1;
        x.alpha = 1.f; x.beta = 0.f;
        x.A = A;
        if (want_left) { x.M = g.m; x.K = g.n; x.a_rs = g.n; x.a_cs = 1; }     // (m x n)(n x r)
        else { x.M = g.n; x.K = g.m; x.a_rs = 1; x.a_cs = g.n; }                // (n x m)(m x r)
        x.tiles_m = (x.M + kGemmBM - 1) / kGemmBM; x.tiles_n = (x.N + kGemmBN - 1) / kGemmBN;
        for (int b = 0; b < x.tiles_m * x.tiles_n; ++b) m_xg.push_back(BlockRef{l, b});
      }
    }
    std::stable_sort(m_gp.begin(), m_gp.end(),
                     [&](const BlockRef& a, const BlockRef& b) { return gp_cost[a.prob] > gp_cost[b.prob]; });
    xcd_group(m_tick);
    xcd_group(m_self);
    place(v.gram_p, gd.data(), gd.size() * sizeof(GramDesc), m_gp);
    v.gram_r = v.gram_p;
    v.gram_r.map_off = da.take(std::max<size_t>(m_gr.size() * sizeof(BlockRef), 16));
    v.gram_r.nblocks = (int)m_gr.size();
    put(v.gram_r.map_off, m_gr.data(), m_gr.size() * sizeof(BlockRef));
    place(v.tick, ed.data(), ed.size() * sizeof(EigDesc), m_tick);
    v.eig_desc_off = v.tick.desc_off;
    auto extra = [&](Group& g, const std::vector<BlockRef>& m) {
      g = v.tick;
      g.map_off = da.take(std::max<size_t>(m.size() * sizeof(BlockRef), 16));
      g.nblocks = (int)m.size();
      put(g.map_off, m.data(), m.size() * sizeof(BlockRef));
    };
    extra(v.self, m_self);
    extra(v.norm, m_norm);
    extra(v.ext, m_ext);
    place(v.xg, xg.data(), xg.size() * sizeof(GemmDesc), m_xg);
  }

  // ---- the five GEMM phases ----
  for (int k = 0; k < 5; ++k) {
    std::vector<GemmDesc> g(n);
    std::vector<BlockRef> m;
    for (int l = 0; l < n; ++l) {
      const TLayer& t = P->L[l];
      GemmDesc& x = g[l];
      memset(&x, 0, sizeof x);
      x.alpha = 1.f; x.beta = 0.f;
      float* T = (float*)dev(t.T); float* Pb = (float*)dev(t.P); float* C = (float*)dev(t.C);
      float* Uo = (float*)dev(t.Uo); float* Ui = (float*)dev(t.Ui);
      const int64_t KI = (int64_t)t.K2 * t.I;
      switch (k) {
        case 0: x.A = T; x.M = t.O * t.K2; x.K = t.I; x.a_rs = t.I; x.a_cs = 1;
                x.B = Ui; x.N = t.ri; x.b_rs = t.ri; x.b_cs = 1; x.C = Pb; x.c_rs = t.ri; x.c_cs = 1; break;
        case 1: x.A = Uo; x.M = t.ro; x.K = t.O; x.a_rs = 1; x.a_cs = t.ro;
                x.B = T; x.N = (int)KI; x.b_rs = KI; x.b_cs = 1; x.C = Pb; x.c_rs = KI; x.c_cs = 1; break;
        case 2: x.A = Pb; x.M = t.ro * t.K2; x.K = t.I; x.a_rs = t.I; x.a_cs = 1;
                x.B = Ui; x.N = t.ri; x.b_rs = t.ri; x.b_cs = 1; x.C = C; x.c_rs = t.ri; x.c_cs = 1; break;
        case 3: x.A = C; x.M = t.ro * t.K2; x.K = t.ri; x.a_rs = t.ri; x.a_cs = 1;
                x.B = Ui; x.N = t.I; x.b_rs = 1; x.b_cs = t.ri; x.C = Pb; x.c_rs = t.I; x.c_cs = 1; break;
        default: x.A = Uo; x.M = t.O; x.K = t.ro; x.a_rs = t.ro; x.a_cs = 1;
                x.B = Pb; x.N = (int)KI; x.b_rs = KI; x.b_cs = 1; x.C = T; x.c_rs = KI; x.c_cs = 1; break;
      }
      x.tiles_m = (x.M + kGemmBM - 1) / kGemmBM; x.tiles_n = (x.N + kGemmBN - 1) / kGemmBN;
      for (int b = 0; b < x.tiles_m * x.tiles_n; ++b) m.push_back(BlockRef{l, b});
    }
    place(P->gemm[k], g.data(), g.size() * sizeof(GemmDesc), m);
  }
  *desc_bytes = align_up(da.off, 4096);
  *total_bytes = ar.off;
  if (desc_region && da.off > desc_region) CTX_FAIL(h, TADMM_ERR_INVALID, "internal: descriptor region overflow");
  return TADMM_OK;
}

static int tucker_geom(tadmm_handle h, int n, const tadmm_layer_desc* descs, tadmm_tucker_plan_s* P) {
  P->h = h; P->n = n;
  P->descs.assign(descs, descs + n);
  P->L.resize(n);
  for (int l = 0; l < n; ++l) {
    const tadmm_layer_desc& d = descs[l];
    TLayer& t = P->L[l];
    if (d.kind != TADMM_KIND_TUCKER2) CTX_FAIL(h, TADMM_ERR_INVALID, "layer %d: kind must be TADMM_KIND_TUCKER2", l);
    if (d.ndim != 2 && d.ndim != 4) CTX_FAIL(h, TADMM_ERR_INVALID, "layer %d: ndim must be 2 or 4", l);
    t.numel = 1;
    for (int i = 0; i < d.ndim; ++i) {
      if (d.dims[i] <= 0) CTX_FAIL(h, TADMM_ERR_INVALID, "layer %d: non-positive dim", l);
      t.numel *= d.dims[i];
    }
    t.O = (int)d.dims[0]; t.I = (int)d.dims[1];
    t.K2 = d.ndim == 4 ? (int)(d.dims[2] * d.dims[3]) : 1;
    t.ro = d.ranks[0]; t.ri = d.ranks[1];
    if (t.ro <= 0 || t.ri <= 0) CTX_FAIL(h, TADMM_ERR_INVALID, "layer %d: Tucker ranks must be positive", l);
  }
  return TADMM_OK;
}

extern "C" {

int tadmm_tucker_workspace_bytes(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, size_t* bytes) {
  if (!h || !descs || !bytes || n_layers < 0) return TADMM_ERR_INVALID;
  tadmm_tucker_plan_s P;
  int rc = tucker_geom(h, n_layers, descs, &P);
  if (rc != TADMM_OK) return rc;
  size_t db = 0, tb = 0;
  rc = tucker_layout(&P, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &db, &tb);
  if (rc != TADMM_OK) return rc;
  rc = tucker_layout(&P, nullptr, nullptr, nullptr, nullptr, nullptr, db, &db, &tb);
  if (rc != TADMM_OK) return rc;
  *bytes = align_up(tb, 4096);
  return TADMM_OK;
}

int tadmm_tucker_create(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, const float* const* W,
                        float* const* U, float* const* Z, void* workspace, size_t workspace_bytes,
                        tadmm_tucker_plan* out) {
  DeviceGuard device_guard(h);
  if (!h || !descs || !out || !W || !U || !Z || n_layers < 0) return TADMM_ERR_INVALID;
  tadmm_tucker_plan_s* P = new tadmm_tucker_plan_s;
  int rc = tucker_geom(h, n_layers, descs, P);
  if (rc != TADMM_OK) { delete P; return rc; }
  size_t db = 0, tb = 0;
  rc = tucker_layout(P, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &db, &tb);
  if (rc == TADMM_OK) rc = tucker_layout(P, nullptr, nullptr, nullptr, nullptr, nullptr, db, &db, &tb);
  if (rc != TADMM_OK) { delete P; return rc; }
  if (!workspace || workspace_bytes < tb) { delete P; CTX_FAIL(h, TADMM_ERR_WORKSPACE, "Tucker workspace too small: need %zu bytes", tb); }
  P->ws = (char*)workspace; P->ws_bytes = workspace_bytes; P->desc_bytes = db;
  std::vector<char> img;
  size_t db2 = 0, tb2 = 0;
  rc = tucker_layout(P, P->ws, W, U, Z, &img, db, &db2, &tb2);
  if (rc != TADMM_OK) { delete P; return rc; }
  hipError_t e = hipMemcpy(P->ws, img.data(), img.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = P->poll.create((size_t)std::max(1, n_layers));
  if (e == hipSuccess) e = P->hooi.create((size_t)std::max(1, n_layers));
  if (e != hipSuccess) {
    P->poll.destroy(); P->hooi.destroy();
    delete P;
    CTX_FAIL(h, TADMM_ERR_HIP, "Tucker plan setup failed: %s", hipGetErrorString(e));
  }
  if (const char* t = getenv("TADMM_JACOBI_TOL")) P->jtol = atof(t);
  if (getenv("TADMM_DEBUG")) P->debug = true;
  for (int l = 0; l < n_layers; ++l)
    if (descs[l].hooi_max_iter > 0) P->n_iter_max = descs[l].hooi_max_iter;
  *out = P;
  return TADMM_OK;
}

int tadmm_tucker_run(tadmm_tucker_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream_) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_ERR_INVALID;
  tadmm_handle h = p->h;
  hipStream_t s = (hipStream_t)stream_;
  const int n = p->n;
  if (n == 0) return TADMM_OK;
  char* ws = p->ws;
  auto D = [&](size_t off) { return ws + off; };
  int32_t* skip = (int32_t*)D(p->skip_off);
  int jac_sweeps = 0;

  auto run_lsv = [&](Lsv& v, const int32_t* sk) -> int {
    launch_gram_partial((const GramDesc*)D(v.gram_p.desc_off), (const BlockRef*)D(v.gram_p.map_off), v.gram_p.nblocks, s, sk);
    launch_gram_reduce((const GramDesc*)D(v.gram_r.desc_off), (const BlockRef*)D(v.gram_r.map_off), v.gram_r.nblocks, s, sk);
    const EigDesc* ed = (const EigDesc*)D(v.eig_desc_off);
    EigGroup eg;
    eg.ed = ed; eg.neig = n; eg.players = v.players.data(); eg.gsteps = v.gsteps; eg.mode = v.mode;
    eg.ld_max = v.ld_max; eg.tick_lds = v.tick_lds;
    eg.tick_map = (const BlockRef*)D(v.tick.map_off); eg.tick_blocks = v.tick.nblocks;
    eg.self_map = (const BlockRef*)D(v.self.map_off); eg.self_blocks = v.self.nblocks;
    eg.prev_dev = (double*)D(p->prev_off);
    eg.off_dev = (const double*)D(p->off_off); eg.done_dev = (const int*)D(p->done_off);
    eg.skip = sk;
    eg.npad_max = v.npad_max;
    eg.warm = v.warm;
    if (!v.big.empty())
      eg.after_init = [&](hipStream_t st) {
        for (const BigWarm& bw : v.big) {
          if (!bw.apply) continue;
          const hipError_t e0 = hipGetLastError();
          const hipError_t e1 = hipMemcpyAsync(D(bw.Gt), D(bw.XT), (size_t)bw.Npad * bw.ld * 8, hipMemcpyDeviceToDevice, st);
          launch_dgemm_nt64((const DgemmDesc*)D(bw.desc_off), (const BlockRef*)D(bw.map_off), bw.nblocks, st);
          const hipError_t e2 = hipPeekAtLastError();
          if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess)
            fprintf(stderr, "[tadmm] big warm start (layer %d, N %d): before %s, copy %s, product %s\n", bw.layer, bw.Npad,
                    hipGetErrorString(e0), hipGetErrorString(e1), hipGetErrorString(e2));
        }
      };
    int gs = 0;
    bool small_pending = false;
    p->jtm.small_n = v.players_n.empty() ? nullptr : v.players_n.data();
    const int rc = run_eig_group(h, eg, p->poll, p->jtol, p->inner, p->max_sweeps, p->debug, s, &gs, &small_pending, &p->jtm);
    if (rc != TADMM_OK) return rc;
    jac_sweeps += gs;
    launch_eig_norms(ed, (const BlockRef*)D(v.norm.map_off), v.norm.nblocks, s, sk);
    launch_eig_sort(ed, n, s, sk, v.npad_max);
    for (const BigWarm& bw : v.big) {
      hipLaunchKernelGGL(bigwarm_check_kernel, dim3(1), dim3(256), 0, s, ed, bw.layer, sk, (int32_t*)D(bw.ok));
      hipLaunchKernelGGL(bigwarm_save_kernel, dim3((bw.Npad + 3) / 4), dim3(256), 0, s, ed, bw.layer, sk,
                         (const int32_t*)D(bw.ok), (double*)D(bw.V));
    }
    launch_eig_extract(ed, (const BlockRef*)D(v.ext.map_off), v.ext.nblocks, s, sk);
    launch_gemm((const GemmDesc*)D(v.xg.desc_off), (const BlockRef*)D(v.xg.map_off), v.xg.nblocks, s, sk);
    return small_pending ? check_small_group(h, eg, p->poll) : TADMM_OK;
  };
  auto gemm = [&](int k, const int32_t* sk) {
    launch_gemm((const GemmDesc*)D(p->gemm[k].desc_off), (const BlockRef*)D(p->gemm[k].map_off), p->gemm[k].nblocks, s, sk);
  };

  p->jtm.tick_ms = p->jtm.small_ms = 0.0; p->jtm.tick_launches = p->jtm.small_launches = 0;
  p->jtm.tick_flops = p->jtm.small_flops = p->jtm.tick_wgs = 0.0;
  if (p->jtm.on) (void)hipEventRecord(p->tev[2], s);
  HIP_OK(h, hipMemsetAsync(skip, 0, (size_t)n * 4, s));
  HIP_OK(h, hipMemsetAsync(D(p->warm_ok_off), 0, (size_t)n * 2 * 4, s));     // a new tensor: every HOOI solve starts cold once
  HIP_OK(h, hipMemsetAsync(D(p->err_off), 0, (size_t)n * 16, s));
  HIP_OK(h, hipMemsetAsync(D(p->iters_off), 0, (size_t)n * 4, s));
  // factors carry zero columns beyond the number of singular values of their unfolding (never written)
  HIP_OK(h, hipMemsetAsync(D(p->fac_begin), 0, p->fac_end - p->fac_begin, s));
  launch_unfold((const SweepDesc*)D(p->sweep_desc_off), (const BlockRef*)D(p->unfold.map_off), p->unfold.nblocks, use_u, s);
  hipLaunchKernelGGL(sumsq_kernel, dim3(n), dim3(256), 0, s, (const float* const*)D(p->ptrT_off),
                     (const int64_t*)D(p->numT_off), (double*)D(p->nT_off), (const int32_t*)nullptr);
  // HOSVD initialisation
  int rc = run_lsv(p->lsv[0], nullptr);
  if (rc != TADMM_OK) return rc;
  rc = run_lsv(p->lsv[1], nullptr);
  if (rc != TADMM_OK) return rc;
  // A HOOI unfolding can have fewer singular values than the HOSVD one (e.g. O x K2*r_in with K2*r_in < r_out):
  // tensorly pads such a factor with an arbitrary completion that only meets zero rows of the core; here the
  // columns the HOOI requests never write are zeroed once, so the factor stays orthonormal-or-zero.
  for (const TLayer& t : p->L) {
    if (t.reff[2] < t.ro)
      HIP_OK(h, hipMemset2DAsync(D(t.Uo) + (size_t)t.reff[2] * 4, (size_t)t.ro * 4, 0, (size_t)(t.ro - t.reff[2]) * 4, t.O, s));
    if (t.reff[3] < t.ri)
      HIP_OK(h, hipMemset2DAsync(D(t.Ui) + (size_t)t.reff[3] * 4, (size_t)t.ri * 4, 0, (size_t)(t.ri - t.reff[3]) * 4, t.I, s));
  }
  // HOOI
  bool all_done = false;
  int it = 0, pending = -1, needed = 0;
  auto consume = [&]() -> int {
    if (pending < 0) return TADMM_OK;
    if (hipEventSynchronize(p->hooi.ev[pending & 1]) != hipSuccess) return TADMM_ERR_HIP;
    if (p->hooi.host[(size_t)(pending & 1) * p->hooi.stride]) { all_done = true; needed = pending + 1; }
    pending = -1;
    return TADMM_OK;
  };
  for (; it < p->n_iter_max && !all_done; ++it) {
    gemm(0, skip);
    rc = run_lsv(p->lsv[2], skip);
    if (rc != TADMM_OK) return rc;
    gemm(1, skip);
    rc = run_lsv(p->lsv[3], skip);
    if (rc != TADMM_OK) return rc;
    gemm(2, skip);
    hipLaunchKernelGGL(sumsq_kernel, dim3(n), dim3(256), 0, s, (const float* const*)D(p->ptrC_off),
                       (const int64_t*)D(p->numC_off), (double*)D(p->nC_off), (const int32_t*)skip);
    hipLaunchKernelGGL(hooi_conv_kernel, dim3(1), dim3(256), 0, s, n, it, (const float*)D(p->tol_off),
                       (const double*)D(p->nT_off), (const double*)D(p->nC_off), (double*)D(p->err_off), skip,
                       (int32_t*)D(p->iters_off), p->hooi.host + (size_t)(it & 1) * p->hooi.stride);
    rc = consume();
    if (rc != TADMM_OK) CTX_FAIL(h, rc, "HOOI poll event failed");
    if (all_done) break;
    HIP_OK(h, hipEventRecord(p->hooi.ev[it & 1], s));
    pending = it;
  }
  if (!all_done) {
    rc = consume();
    if (rc != TADMM_OK) CTX_FAIL(h, rc, "HOOI poll event failed");
  }
  p->last_hooi = all_done ? needed : it;
  p->last_jacobi_sweeps = jac_sweeps;
  // Z = C x_0 U_out x_1 U_in, fold, dual update, residual
  gemm(3, nullptr);
  gemm(4, nullptr);
  double* partial = (double*)D(p->resid_partial_off);
  launch_fold_update((const SweepDesc*)D(p->sweep_desc_off), (const BlockRef*)D(p->fold.map_off), p->fold.nblocks,
                     update_u, partial, s);
  if (resid_sq_dev) launch_resid_reduce((const SweepDesc*)D(p->sweep_desc_off), n, partial, resid_sq_dev, s);
  HIP_OK(h, hipGetLastError());
  if (p->jtm.on) {
    float ms = 0.f;
    (void)hipEventRecord(p->tev[3], s);
    (void)hipEventSynchronize(p->tev[3]);
    (void)hipEventElapsedTime(&ms, p->tev[2], p->tev[3]);
    p->last_total_ms = ms;
  }
  if (p->debug) fprintf(stderr, "[tadmm] tucker: %d layers, HOOI sweeps (max over layers) %d, Jacobi sweeps %d\n", n,
                        p->last_hooi, jac_sweeps);
  return TADMM_OK;
}

int tadmm_tucker_factors(tadmm_tucker_plan p, int layer, const float** core, const float** u_out, const float** u_in) {
  if (!p || layer < 0 || layer >= p->n) return TADMM_ERR_INVALID;
  const TLayer& t = p->L[layer];
  if (core) *core = (const float*)(p->ws + t.C);
  if (u_out) *u_out = (const float*)(p->ws + t.Uo);
  if (u_in) *u_in = (const float*)(p->ws + t.Ui);
  return TADMM_OK;
}

int tadmm_tucker_jacobi_sweeps(tadmm_tucker_plan p) { return p ? p->last_jacobi_sweeps : TADMM_ERR_INVALID; }

int tadmm_tucker_enable_timing(tadmm_tucker_plan p, int on) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_ERR_INVALID;
  if (on && !p->ev_made) {
    for (auto& e : p->tev) if (hipEventCreate(&e) != hipSuccess) return TADMM_ERR_HIP;
    p->ev_made = true;
    p->jtm.a = p->tev[0]; p->jtm.b = p->tev[1];
  }
  p->jtm.on = on != 0;
  return TADMM_OK;
}

// out[0] = summed ms of the eigen-solver launches of the last run, [1] = their number, [2] = 8 N^3 model flops they stand
// for, [3] = ms of the whole run, [4] = HOOI sweeps (max over layers), [5..7] reserved
int tadmm_tucker_last_timing(tadmm_tucker_plan p, double out[8]) {
  if (!p || !out) return TADMM_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  out[0] = p->jtm.small_ms + p->jtm.tick_ms; out[1] = p->jtm.small_launches + p->jtm.tick_launches;
  out[2] = p->jtm.small_flops; out[3] = p->last_total_ms; out[4] = p->last_hooi;
  return TADMM_OK;
}

int tadmm_tucker_iterations(tadmm_tucker_plan p, int32_t* iters_out_host, double* errors_out_host, void* stream_) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream_;
  HIP_OK(p->h, hipStreamSynchronize(s));
  std::vector<int32_t> it(p->n);
  std::vector<double> err((size_t)2 * p->n);
  if (p->n) {
    HIP_OK(p->h, hipMemcpy(it.data(), p->ws + p->iters_off, (size_t)p->n * 4, hipMemcpyDeviceToHost));
    HIP_OK(p->h, hipMemcpy(err.data(), p->ws + p->err_off, (size_t)p->n * 16, hipMemcpyDeviceToHost));
  }
  for (int l = 0; l < p->n; ++l) {
    if (iters_out_host) iters_out_host[l] = it[l];
    if (errors_out_host) errors_out_host[l] = it[l] > 0 ? err[(size_t)((it[l] - 1) & 1) * p->n + l] : 0.0;
  }
  return TADMM_OK;
}

int tadmm_tucker_destroy(tadmm_tucker_plan p) {
  DeviceGuard device_guard(p ? p->h : nullptr);
  if (!p) return TADMM_OK;
  p->poll.destroy();
  p->hooi.destroy();
  if (p->ev_made) for (auto& e : p->tev) (void)hipEventDestroy(e);
  delete p;
  return TADMM_OK;
}

}  // extern "C"
