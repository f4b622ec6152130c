// Filtered eigen-solver: the leading r eigenvectors of a Gram matrix G (N x N, fp64) without a full
// eigen-decomposition.  Replaces, for unfoldings whose kept rank is a fraction of N, the full block-Jacobi solve
// behind numpy.linalg.svd (reference ttd.py:17) by
//
//   Chebyshev-filtered subspace iteration on a block of r' ~ 1.45 r vectors   (dgemm.hip: products with G)
//   + Cholesky QR between filter stages                                        (chol.hip)
//   + ONE Rayleigh-Ritz solve of the r' x r' projection  H = Q^T G Q           (jacobi.hip, the same kernels)
//   + a posteriori verification of the Ritz pairs                              (residuals over spectral gaps)
//
// so that the latency-bound Jacobi tournament runs over r'/16 instead of N/16 super-blocks.  Everything that
// decides the course of the iteration lives on the device:
//   * filter bounds come from the sorted Rayleigh quotients of the current (Cholesky-ordered) basis columns:
//     damped interval [0, b], b = smallest quotient; the r-th largest estimates lambda_r;
//   * the degree of a stage is limited by the growth it causes in the block's condition number (Cholesky QR
//     must stay safe) and by what is still needed: log-amplification of the boundary vector is tracked and the
//     filter stops at ln(2/eps);
//   * any anomaly -- pivot breakdown (numerically rank-deficient block, e.g. an exactly low-rank input),
//     degenerate bounds, or a failed verification -- marks the problem `bad`; the host then runs the full
//     Jacobi solve for it (the path of round 1), so results never depend on the filter being applicable.
// The host only launches; it reads one word per problem and stage ("more filtering wanted?") and one after
// the verification.
#include "common.h"

namespace tadmm {

typedef double double2v_t __attribute__((ext_vector_type(2)));
constexpr int kMomParts = 64;      // workgroups per problem of the moments guard

__device__ __forceinline__ double hash_unit(uint32_t a, uint32_t b) {
  // splitmix-style integer hash -> uniform in (-1, 1); fixed function of (problem, element): deterministic runs
  uint64_t x = ((uint64_t)a << 32) ^ (uint64_t)b;
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(int64_t)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;    // 53 bits -> [0,2) - 1
}

// start block: ring[0][j][i] random for j < rp, i < N (zero in the padding); state reset
__global__ __launch_bounds__(256) void filt_init_kernel(const FiltProb* __restrict__ probs,
                                                        const BlockRef* __restrict__ map, int test_zero_col) {
  const BlockRef br = map[blockIdx.x];
  const FiltProb p = probs[br.prob];
  if (br.local == 0 && threadIdx.x == 0) {
    FiltState* st = p.st;
    st->base = 0; st->res = 1; st->nsteps = 0; st->active = 1; st->alive = 1; st->bad = 0; st->stage = 0;
    st->products = 2;       // stage-0 product + the product in front of the Rayleigh-Ritz projection
    st->products_fast = 0; st->precise_stages = 0; st->logamp_precise = 0.0;
    st->logamp = 0.0; st->crit = 0.0; st->b = st->lr = st->l1 = 0.0; st->guard = 0.0; st->mom_on = 0;
    *p.skip_slot = 0;
    *p.fb_skip = 0;
  }
  const int64_t total = (int64_t)p.rp * p.ldy;
  const int64_t i0 = ((int64_t)br.local * 256 + threadIdx.x) * 4;
  double* Y = p.ring[0];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i0 + u;
    if (i < total) {
      const int col = (int)(i % p.ldy);
      // test_zero_col (TADMM_FILTER_TEST_ZERO_COL, tests only): a start block with NO component along one coordinate
      // axis -- when that axis is an exact eigenvector the block can never acquire it (tests/test_gpu_filter.py)
      Y[i] = (col < p.N && col != test_zero_col) ? hash_unit((uint32_t)br.prob * 2654435761u + 17u, (uint32_t)i) : 0.0;
    }
  }
}

__device__ __forceinline__ double acosh_pos(double x) { return log(x + sqrt(fmax(x * x - 1.0, 0.0))); }

// One workgroup per problem, after the product T = G Q of a stage (whose epilogue left the Rayleigh-quotient
// partials): bounds, degree and scalars of the stage.
__global__ __launch_bounds__(256) void filt_plan_kernel(const FiltProb* __restrict__ probs, FiltParams prm,
                                                        int last_stage, int stage_fast, int* __restrict__ verdict) {
  __shared__ double rho[256], sorted[256];
  const FiltProb p = probs[blockIdx.x];
  FiltState* st = p.st;
  const int tid = threadIdx.x;
  if (st->bad || !st->active) {              // uniform: read before anybody writes
    __syncthreads();
    if (tid == 0) {
      st->nsteps = 0; st->res = st->base; st->active = 0;
      if (st->bad) { st->alive = 0; *p.skip_slot = 1; }
      verdict[1 + blockIdx.x] = 0;
    }
    return;
  }
  const int rp = p.rp;
  for (int j = tid; j < rp; j += 256) {
    double v = 0.0;
    for (int t = 0; t < p.rq_tiles; ++t) v += p.rqpart[(int64_t)t * rp + j];
    rho[j] = v;
  }
  __syncthreads();
  for (int j = tid; j < rp; j += 256) {
    const double v = rho[j];
    int rank = 0;
    for (int i = 0; i < rp; ++i) {
      const double w = rho[i];
      rank += (w > v) || (w == v && i < j);
    }
    sorted[rank] = v;
  }
  __syncthreads();
  if (tid == 0) {
    const double l1 = sorted[0], lr = sorted[p.r - 1], b = sorted[rp - 1];
    bool ok = isfinite(l1) && isfinite(b) && b > 0.0 && lr > b * (1.0 + 1e-9) && l1 >= lr;
    int m = 0;
    bool more = false;
    if (ok) {
      const double e = 0.5 * b, c = 0.5 * b;             // damped interval [0, b]
      const double xr = (lr - c) / e, x1 = (l1 - c) / e;
      const double ar = acosh_pos(xr), a1 = acosh_pos(x1);
      // Stages that ran at fp32 accuracy (dgemm3.hip) leave a noise floor of ~1e-7 in the block; it takes a few units of
      // log-amplification IN FP64 to push it below what the verification accepts.  A problem therefore only finishes
      // once its fp64 stages have contributed prm.log_precise on their own.
      double need = prm.log_target - st->logamp;
      if (st->products_fast > 0 || (stage_fast & 1)) need = fmax(need, prm.log_precise - st->logamp_precise);
      if (need > 0.0) {
        const double ln2 = 0.6931471805599453;
        // the first stage only has the Rayleigh quotients of one power step to go by, which under-estimate lambda_1
        // (and with it the growth of the block's condition number): keep it short
        const double cmax = st->stage == 0 ? fmin(prm.cond_max, prm.cond_first) : prm.cond_max;
        int mc = (int)floor(log(2.0 * cmax) / a1);
        mc = max(1, mc);
        int mn = (int)ceil((need + ln2) / ar);
        mn = max(1, mn);
        m = min(prm.max_degree, min(mc, mn));
        st->logamp += m * ar - ln2;
        st->coef1[0] = 1.0 / e; st->coef1[1] = -c / e;
        st->coefk[0] = 2.0 / e; st->coefk[1] = -2.0 * c / e; st->coefk[2] = -1.0;
        if (!(stage_fast & 1)) { st->precise_stages += 1; st->logamp_precise += m * ar - ln2; }
        // what is still missing AFTER this stage is known now (the amplification is accounted from the bounds, not
        // measured): the problem drops out of the next stage's launches here, not after another product + plan
        double after = prm.log_target - st->logamp;
        if (st->products_fast > 0 || (stage_fast & 1)) after = fmax(after, prm.log_precise - st->logamp_precise);
        more = after > 0.0;
      }
      st->b = b; st->lr = lr; st->l1 = l1;
    } else {
      st->bad = 1; st->alive = 0; *p.skip_slot = 1;
    }
    st->nsteps = m;
    st->products += (m > 0) ? m : 1;     // the stage's first product was launched before its degree was known
    if (stage_fast & 1) st->products_fast += (m > 0) ? m : 1;
    if ((stage_fast & 2) && st->stage == 0) st->products_fast += 1;      // the stage-0 product
    int res = st->base + m;
    res -= (res >= 3) ? 3 : 0; res -= (res >= 3) ? 3 : 0; res -= (res >= 3) ? 3 : 0;
    st->res = res % 3;
    st->active = (m > 0 && more && !last_stage) ? 1 : 0;
    st->stage += 1;
    verdict[1 + blockIdx.x] = (m > 0 ? 1 : 0) | (st->active ? 2 : 0);     // bit 0: this stage filters, bit 1: wants another
  }
}

// after the polishing CholQR: fold late failures into the gates (alive / skip) before the Rayleigh-Ritz solve
__global__ void filt_flags_kernel(const FiltProb* __restrict__ probs, int nprob) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nprob) return;
  const FiltProb p = probs[q];
  if (p.st->bad) { p.st->alive = 0; *p.skip_slot = 1; }
}

__global__ __launch_bounds__(256) void filt_theta_kernel(const FiltProb* __restrict__ probs) {
  const FiltProb p = probs[blockIdx.x];
  if (p.st->bad) return;
  for (int c = threadIdx.x; c < p.r32; c += 256) {
    const double s = c < p.r ? p.sigma[c] : 0.0;
    p.theta[c] = s * s;
  }
}

// Acceptance test.  For a Ritz pair (theta_c, u_c) with residual r_c the part of u_c outside the invariant subspace
// of the eigenvalues above lambda_{r+1} is at most ||r_c|| / (theta_c - lambda_{r+1}); summed over the kept pairs
// this bounds ||sin Theta||_F of the accepted subspace.  lambda_{r+1} is taken from the (r+1)-th Ritz value.
__global__ __launch_bounds__(256) void filt_verdict_kernel(const FiltProb* __restrict__ probs, FiltParams prm,
                                                           int* __restrict__ verdict) {
  __shared__ double part[256];
  const FiltProb p = probs[blockIdx.x];
  FiltState* st = p.st;
  const int tid = threadIdx.x;
  if (st->bad) {
    if (tid == 0) { verdict[1 + blockIdx.x] = 0; *p.fb_skip = 0; }
    return;
  }
  const double th_next = p.lam[p.order[p.r]];
  double acc = 0.0;
  for (int c = tid; c < p.r; c += 256) {
    double rn = 0.0;
    for (int t = 0; t < p.v_tiles; ++t) rn += p.vpart[(int64_t)t * p.r32 + c];
    const double gap = p.theta[c] - th_next;
    acc += (gap > 0.0) ? rn / (gap * gap) : INFINITY;
  }
  part[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) part[tid] += part[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double crit = sqrt(part[0]);
    st->crit = crit;
    // (guard: filt_guard_kernel; 0 when the guard did not run.  NaN compares false -> rejected by the negation)
    const double th_r = p.theta[p.r - 1];
    bool guard_ok = !(st->guard > th_r * (1.0 + 1e-9)) && st->guard == st->guard;
    if (st->mom_on) {                        // moments guard (filt_moments_kernel): ||D||_F^2 / tr D <= lambda_max(D) must stay below theta_r
      double m[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      for (int w = 0; w < kMomParts; ++w)
        for (int q = 0; q < 5; ++q) m[q] += st->mom[w * 5 + q];
      const double trd = m[1] - m[4], d2 = m[0] - 2.0 * m[2] + m[3];
      const double bound = trd > 0.0 ? d2 / trd : 0.0;
      if (bound > th_r * (1.0 + 1e-6) || bound != bound) guard_ok = false;
      st->guard = fmax(st->guard, bound);
    }
    const bool ok = crit <= prm.sin_tol && guard_ok;     // false for NaN
    if (!ok) { st->bad = 1; st->alive = 0; }
    *p.fb_skip = ok ? 1 : 0;
    verdict[1 + blockIdx.x] = ok ? 1 : 0;
  }
}

// Moments guard: a lower bound of the largest eigenvalue the block does NOT contain, from quantities that are lying around.
// With P = I - Q Q^T (Q orthonormal), T = G Q and H = Q^T G Q the deflated operator D = P G P has
//      tr D = tr G - tr H,        ||D||_F^2 = ||G||_F^2 - 2 ||T||_F^2 + ||H||_F^2,
// and ||D||_F^2 / tr D is a weighted mean of its (non-negative) eigenvalues, hence <= lambda_max(D): a healthy block leaves
// only eigenvalues below the r-th Ritz value out there and the ratio stays below it whatever the spectrum; an eigenvector
// that dominates what is left and that the block never acquired pushes it above.  One pass over G, T and H by 64
// workgroups per problem, in line behind the product that forms H (~3 us); partials are summed in a fixed order by the
// verdict kernel.  Weaker than power steps when the missed eigenvalue is one among many of similar size -- neither catches
// that within a few steps -- and two orders of magnitude cheaper: the power-step guard below costs 0.35 ms of a 6.8 ms
// iteration even on a side stream (measured in separate processes on one box: 6.84 / 7.19 / 7.28 ms for off / side
// stream / in line), so it is the opt-in (TADMM_FILTER_GUARD=n) and this one the default.
__global__ __launch_bounds__(256) void filt_moments_kernel(const FiltProb* __restrict__ probs) {
  __shared__ double red[5][4];
  const FiltProb p = probs[blockIdx.x];
  FiltState* st = p.st;
  if (st->bad) return;
  const int part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tb = st->base + 1;
  tb -= tb >= 3 ? 3 : 0;
  // the three images are zero in their padding, so the squared sums run over the flat arrays: each workgroup one
  // contiguous slice, eight 16-byte loads in flight per thread
  auto sumsq = [&](const double* __restrict__ a, int64_t total) {
    const int64_t per = ((total / kMomParts + 511) / 512) * 512;          // multiple of 2 * 256
    const int64_t lo = (int64_t)part * per, hi = lo + per < total ? lo + per : total;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t i = lo + 2 * tid; i < hi; i += 2048) {
      double2v_t v0 = {0, 0}, v1 = {0, 0}, v2 = {0, 0}, v3 = {0, 0};
      v0 = *reinterpret_cast<const double2v_t*>(a + i);
      if (i + 512 < hi) v1 = *reinterpret_cast<const double2v_t*>(a + i + 512);
      if (i + 1024 < hi) v2 = *reinterpret_cast<const double2v_t*>(a + i + 1024);
      if (i + 1536 < hi) v3 = *reinterpret_cast<const double2v_t*>(a + i + 1536);
      s0 += v0.x * v0.x + v0.y * v0.y; s1 += v1.x * v1.x + v1.y * v1.y;
      s2 += v2.x * v2.x + v2.y * v2.y; s3 += v3.x * v3.x + v3.y * v3.y;
    }
    return (s0 + s1) + (s2 + s3);
  };
  const double g2 = sumsq(p.G, (int64_t)p.Npad * p.ldg);
  const double t2 = sumsq(p.ring[tb], (int64_t)p.rp * p.ldy);
  const double h2 = sumsq(p.H, (int64_t)p.rp * p.ldh);
  double tg = 0.0, th = 0.0;
  if (part == 0) {
    for (int i = tid; i < p.N; i += 256) tg += p.G[(int64_t)i * p.ldg + i];
    for (int j = tid; j < p.rp; j += 256) th += p.H[(int64_t)j * p.ldh + j];
  }
  double v[5] = {g2, tg, t2, h2, th};
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if (lane == 0) red[q][wave] = v[q];
  }
  __syncthreads();
  if (tid < 5) st->mom[part * 5 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
  if (part == 0 && tid == 0) st->mom_on = 1;
}

// Independent guard of the filtered solve.  The residual-based acceptance test bounds the error of the Ritz pairs it is
// shown; it cannot see an eigenpair the block never contained (a start block without a component along a wanted
// eigenvector: probability zero for the hashed start on generic weights, certain for a stale warm start).  This kernel
// looks OUTSIDE the block: a few power steps with the deflated operator P G P, P = I - Q Q^T (Q = the whole orthonormal
// r'-block), from a start vector hashed with a different seed.  The Rayleigh quotient of a unit vector orthogonal to Q is
// a LOWER bound of the largest eigenvalue the block does not contain; a healthy solve leaves ~lambda_{r'+1} out there,
// well below the r-th Ritz value, so `guard > theta_r` sends the problem to the full solve (filt_verdict_kernel).
// One 1024-thread workgroup per problem; z, w and the coefficients live in LDS.
__global__ __launch_bounds__(1024) void filt_guard_kernel(const FiltProb* __restrict__ probs, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) double gsm[];
  __shared__ double red[16];
  const FiltProb p = probs[blockIdx.x];
  FiltState* st = p.st;
  if (st->bad) return;                                   // uniform: nobody has written yet
  const int N = p.N, Npad = p.Npad, rp = p.rp, ldy = p.ldy, ldg = p.ldg;
  const double* __restrict__ Q = p.ring[st->base];
  const double* __restrict__ G = p.G;
  double* z = gsm;
  double* w = z + Npad;
  double* c = w + Npad;
  double* pacc = c + rp;                                 // [nparts][Npad] partial sums of the projection
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto block_sum = [&](double v) {                       // every thread gets the sum
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
  };
  // y[j] = <rows[j], x> for j < nrows (rows contiguous, leading dimension ld): a wave takes four rows at a time so that
  // 4 x N/64 loads are in flight per lane -- the phase is bound by load latency, not by bytes
  auto rows_dot = [&](const double* __restrict__ rows, int ld, int nrows, const double* x, double* y) {
    for (int j0 = wave * 4; j0 < nrows; j0 += 64) {
      const double* r0 = rows + (int64_t)j0 * ld;
      const bool h1 = j0 + 1 < nrows, h2 = j0 + 2 < nrows, h3 = j0 + 3 < nrows;
      const double* r1 = h1 ? r0 + ld : r0;
      const double* r2 = h2 ? r0 + 2 * (int64_t)ld : r0;
      const double* r3 = h3 ? r0 + 3 * (int64_t)ld : r0;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int i = lane; i < N; i += 64) {
        const double xv = x[i];
        a0 += r0[i] * xv; a1 += r1[i] * xv; a2 += r2[i] * xv; a3 += r3[i] * xv;
      }
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
      if (lane == 0) {
        y[j0] = a0;
        if (h1) y[j0 + 1] = a1;
        if (h2) y[j0 + 2] = a2;
        if (h3) y[j0 + 3] = a3;
      }
    }
  };
  // v <- (I - Q Q^T) v.  The update v[i] -= sum_j c_j Q[j][i] is a chain of rp loads per element: the j range is split over
  // `nparts` thread groups (all 1024 threads busy at N <= 512) with 8 loads in flight each; partial sums meet in LDS.
  const int Nc = (N + 63) & ~63;
  const int nparts = max(1, 1024 / Nc);
  auto project = [&](double* v) {
    rows_dot(Q, ldy, rp, v, c);
    __syncthreads();
    const int part = tid / Nc, i = tid - part * Nc;
    if (nparts > 1) {
      if (part < nparts && i < N) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int per = (rp / nparts + 7) & ~7, j0 = part * per, j1 = min(rp, j0 + per);
        int j = j0;
        for (; j + 7 < j1; j += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] += c[j + u] * Q[(int64_t)(j + u) * ldy + i];
        }
        for (; j < j1; ++j) a[0] += c[j] * Q[(int64_t)j * ldy + i];
        pacc[part * Npad + i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
      __syncthreads();
      for (int k = tid; k < N; k += 1024) {
        double t = 0.0;
        for (int q = 0; q < nparts; ++q) t += pacc[q * Npad + k];
        v[k] -= t;
      }
    } else {
      for (int k = tid; k < N; k += 1024) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int j = 0;
        for (; j + 7 < rp; j += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] += c[j + u] * Q[(int64_t)(j + u) * ldy + k];
        }
        for (; j < rp; ++j) a[0] += c[j] * Q[(int64_t)j * ldy + k];
        v[k] -= ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
    }
    __syncthreads();
  };
  for (int i = tid; i < Npad; i += 1024)
    z[i] = i < N ? hash_unit((uint32_t)blockIdx.x * 2246822519u + 0x5bd1e995u, (uint32_t)i) : 0.0;
  __syncthreads();
  project(z);
  double rho = 0.0;
  {
    double a = 0.0;
    for (int i = tid; i < N; i += 1024) a += z[i] * z[i];
    const double n2 = block_sum(a);
    const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
    for (int i = tid; i < N; i += 1024) z[i] *= inv;
    __syncthreads();
  }
  for (int s = 0; s < nsteps; ++s) {
    rows_dot(G, ldg, N, z, w);                           // w = G z  (G symmetric: row i is contiguous)
    __syncthreads();
    project(w);
    double a = 0.0, b = 0.0;
    for (int i = tid; i < N; i += 1024) { a += z[i] * w[i]; b += w[i] * w[i]; }
    rho = fmax(rho, block_sum(a));                       // z is a unit vector orthogonal to Q
    const double n2 = block_sum(b);
    if (!(n2 > 0.0)) break;                              // uniform
    const double inv = 1.0 / sqrt(n2);
    for (int i = tid; i < N; i += 1024) z[i] = w[i] * inv;
    __syncthreads();
  }
  if (tid == 0) st->guard = rho;
}

// Ritz vectors -> the outputs the projection GEMMs read (same conventions and sign rule as eig_extract_kernel)
__global__ __launch_bounds__(256) void filt_emit_kernel(const FiltProb* __restrict__ probs,
                                                        const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const FiltProb p = probs[br.prob];
  if (p.st->bad) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = br.local * 4 + wave;
  if (c >= p.r) return;
  const double* row = p.UT + (int64_t)c * p.ldy;
  const int N = p.N;
  double best = -1.0; int besti = 0;
  double nrm = 0.0;
  for (int i = lane; i < N; i += 64) {
    const double v = row[i], a = fabs(v);
    nrm += v * v;
    if (a > best) { best = a; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_down(best, o, 64);
    const int oi = __shfl_down(besti, o, 64);
    nrm += __shfl_xor(nrm, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  besti = __shfl(besti, 0, 64);
  const double sgn = (row[besti] < 0.0) ? -1.0 : 1.0;
  const double inv = nrm > 0.0 ? sgn / sqrt(nrm) : 0.0;
  const double sig = p.sigma[c];
  const double isig = sig > 0.0 ? 1.0 / sig : 0.0;
  const int R = p.ldo ? p.ldo : p.r;
  for (int i = lane; i < N; i += 64) {
    const double v = row[i] * inv;
    if (p.mode == 0) {
      p.out_a[(int64_t)i * R + c] = (float)v;
    } else if (p.mode == 1) {
      p.out_a[(int64_t)i * R + c] = (float)(v * isig);
      p.out_b[(int64_t)c * N + i] = (float)(v * sig);
    } else if (p.mode == 3) {
      p.out_a[(int64_t)i * R + c] = (float)(v * isig);
    }
  }
}

void launch_filt_init(const FiltProb* probs_dev, const BlockRef* map_dev, int nblocks, int nprob, hipStream_t s) {
  if (nblocks <= 0) return;
  int zero_col = -1;
  if (const char* e = getenv("TADMM_FILTER_TEST_ZERO_COL")) zero_col = atoi(e);
  hipLaunchKernelGGL(filt_init_kernel, dim3(nblocks), dim3(256), 0, s, probs_dev, map_dev, zero_col);
}
void launch_filt_moments(const FiltProb* probs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(filt_moments_kernel, dim3(nprob, kMomParts), dim3(256), 0, s, probs_dev);
}
void launch_filt_guard(const FiltProb* probs_dev, int nprob, int npad_max, int rp_max, int steps, hipStream_t s) {
  if (nprob <= 0 || steps <= 0) return;
  const size_t lds = ((size_t)2 * npad_max + rp_max + (size_t)std::max(1, 1024 / std::max(64, npad_max)) * 1024 + 1024) * sizeof(double);
  hipLaunchKernelGGL(filt_guard_kernel, dim3(nprob), dim3(1024), lds, s, probs_dev, steps);
}
void launch_filt_plan(const FiltProb* probs_dev, int nprob, FiltParams prm, int last_stage, int stage_fast, int* verdict_pinned,
                      hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(filt_plan_kernel, dim3(nprob), dim3(256), 0, s, probs_dev, prm, last_stage, stage_fast, verdict_pinned);
}
void launch_filt_flags(const FiltProb* probs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(filt_flags_kernel, dim3((nprob + 63) / 64), dim3(64), 0, s, probs_dev, nprob);
}
void launch_filt_theta(const FiltProb* probs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(filt_theta_kernel, dim3(nprob), dim3(256), 0, s, probs_dev);
}
void launch_filt_verdict(const FiltProb* probs_dev, int nprob, FiltParams prm, int* verdict_pinned, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(filt_verdict_kernel, dim3(nprob), dim3(256), 0, s, probs_dev, prm, verdict_pinned);
}
void launch_filt_emit(const FiltProb* probs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(filt_emit_kernel, dim3(nblocks), dim3(256), 0, s, probs_dev, map_dev);
}

}  // namespace tadmm
