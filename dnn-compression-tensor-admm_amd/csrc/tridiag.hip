// Single-launch symmetric eigen-solver for small problems (N <= 64) by the direct route
//
//   Householder tridiagonalisation  ->  bisection (Sturm counts) for the leading eigenvalues
//   ->  inverse iteration on the tridiagonal  ->  back-transformation  ->  verification,
//
// one 256-thread workgroup per problem, everything in registers / LDS.  It replaces, whenever its own checks pass,
// the block-Jacobi solve of `jacobi_small_kernel` behind numpy.linalg.svd (reference ttd.py:17) / tensorly's
// partial_svd (reference admm.py:116,124) for the <= 64-column Gram matrices: the first and last TT steps of every
// conv layer, and EVERY eigen-problem of the Tucker tables (BASELINE config 2).
//
// Why: the Jacobi solve is a chain of ~(sweeps x (N-1)) dependent rotation steps of ~0.4 us each (the 16x16 inner
// solve is a dependent chain of ~50 instructions per step): 90 us at N = 32, 200-300 us at N = 64, whatever the
// hardware does around it.  The direct route has O(N) dependent steps per phase: ~N Householder steps, ~N-step Sturm
// recurrences (one FMA per step on the critical path: product form with rescaling), two ~N-step tridiagonal solves per
// vector, ~N reflector applications.
//
// Safety net, not trust: inverse iteration can lose orthogonality inside tight clusters of eigenvalues and the whole
// route is younger than the Jacobi path, so the kernel VERIFIES what it computed (orthogonality against the
// neighbouring vectors, residual of every pair, cluster size) and only then overwrites the problem's image and sets its
// `fast_done` word; otherwise the image is left untouched and `jacobi_small_kernel`, launched right behind it, solves
// the problem as before.  Exactly rank-deficient or constant inputs (large clusters at zero) take that path by design.
//
// Output convention = jacobi_small_kernel's: row j of XT = lambda_j * v_j for the computed pairs (the leading
// min(N, r + 2)), zero rows for the rest (they sort last in eig_sort_kernel and are never extracted).
#include "tridiag_common.h"

namespace tadmm {

// Dynamic LDS (doubles): Hv[64][66] | Z[64][66] | LUa[64][64] | LUb[64][64] | vectors
constexpr int kTLdv = kTN + 2;       // even leading dimension: rows are 16-byte aligned (b128 reads of row pieces)
constexpr int kTVecDoubles = 12 * kTN + 64;
constexpr size_t kTLdsBytes = ((size_t)2 * kTN * kTLdv + 2 * kTN * kTN + kTVecDoubles) * sizeof(double);

// NS: padded size of the tridiagonal phases (32 or 64).  Beyond n the tridiagonal is padded with diagonal entries far
// below the (scaled) spectrum and zero couplings: a decoupled block whose eigenvalues are never among the leading ones
// and whose eigenvector entries stay exactly zero -- every loop of phases 2-4 then has a compile-time trip count and
// the vectors live in registers with static indices.
template <int NS>
__device__ __forceinline__ void eig_small_direct_body(const EigDesc& d, int p, double* __restrict__ tsm,
                                                      int32_t* __restrict__ fast_done, int* __restrict__ verdict) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = d.N, ld = d.ld;
  double (*Hv)[kTLdv] = reinterpret_cast<double (*)[kTLdv]>(tsm);
  double (*Z)[kTLdv] = Hv + kTN;
  double* LUa = tsm + 2 * kTN * kTLdv;               // [k][j]: reciprocal pivots
  double* LUb = LUa + kTN * kTN;                     // [k][j]: super-diagonal after elimination
  double* vec = LUb + kTN * kTN;
  double* dd = vec;            // [64] diagonal of T
  double* ee = dd + kTN;       // [64] sub-diagonal
  double* ds = ee + kTN;       // [64] scaled diagonal (padded)
  double* es = ds + kTN;       // [64] scaled sub-diagonal (zero from n-1 on)
  double* tauv = es + kTN;     // [64]
  double* xs = tauv + kTN;     // [64] column being eliminated
  double* vs = xs + kTN;       // [64] Householder vector
  double* ps = vs + kTN;       // [64] A v
  double* lam = ps + kTN;      // [64] eigenvalues (descending, scaled)
  double* theta = lam + kTN;   // [64] Rayleigh quotients
  double* red = theta + kTN;   // [64] scratch
  int* ivec = reinterpret_cast<int*>(red + kTN);     // [128] ints: cluster start, flags
  double* __restrict__ XT = d.XT;

  TSTAMP(0);
  // ---- 0. the matrix: 4x4 tile per thread, in registers ----
  const int R = tid >> 4, Cc = tid & 15;             // a 16-lane DPP row shares R
  double a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = 4 * R + r, j = 4 * Cc + c;
      a[r][c] = (i < n && j < n) ? XT[(int64_t)i * ld + j] : 0.0;
    }
  if (tid < kTN) { dd[tid] = 0.0; ee[tid] = 0.0; tauv[tid] = 0.0; }
  for (int idx = tid; idx < kTN * kTLdv; idx += 256) (&Hv[0][0])[idx] = 0.0;
  __syncthreads();

  TSTAMP(1);
#ifdef TADMM_TRI_STAMPS
  long long tsub = (long long)__builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 8; q < 16; ++q) g_tri_stamps[q] = 0;
#endif
  // ---- 1. Householder tridiagonalisation (dsytd2, lower) ----
  for (int kb = 0; kb < 16; ++kb) {
    if (4 * kb >= n - 2) break;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kb + kk;
      if (k >= n - 2) break;                          // uniform
      if (Cc == kb) {
        *reinterpret_cast<double2_t*>(&xs[4 * R]) = double2_t{a[0][kk], a[1][kk]};
        *reinterpret_cast<double2_t*>(&xs[4 * R + 2]) = double2_t{a[2][kk], a[3][kk]};
        if (R == kb) dd[k] = a[kk][kk];
      }
      __syncthreads();
      TSUB(0);
      // every wave forms the reflector redundantly (lane l <-> row l)
      const double x = (lane > k) ? xs[lane] : 0.0;
      const double alpha = xs[k + 1];
      // the entries of the column this thread's tile rows / columns will need, requested together with the above: the
      // reflector's entries are then formed from registers (no LDS round trip of v behind the reduction)
      const double2_t xr0 = *reinterpret_cast<const double2_t*>(&xs[4 * R]), xr1 = *reinterpret_cast<const double2_t*>(&xs[4 * R + 2]);
      const double2_t xc0 = *reinterpret_cast<const double2_t*>(&xs[4 * Cc]), xc1 = *reinterpret_cast<const double2_t*>(&xs[4 * Cc + 2]);
      const double xn2 = wave_sum((lane > k + 1) ? x * x : 0.0);
      double tau = 0.0, beta = alpha, sc = 0.0;
      if (xn2 > 0.0) {                                // uniform
        beta = -copysign(fast_sqrt(fma(alpha, alpha, xn2)), alpha);
        tau = (beta - alpha) * fast_rcp(beta);
        sc = fast_rcp(alpha - beta);
      }
      auto vof = [&](int i, double xi) { return i == k + 1 ? 1.0 : (i > k + 1 ? xi * sc : 0.0); };
      if (wave == 0) {
        Hv[k][lane] = vof(lane, x);
        if (lane == 0) { ee[k] = beta; tauv[k] = tau; }
      }
      TSUB(1);
      if (tau != 0.0) {                               // uniform
        double vr[4], vc[4], pr[4];
        vr[0] = vof(4 * R, xr0.x); vr[1] = vof(4 * R + 1, xr0.y); vr[2] = vof(4 * R + 2, xr1.x); vr[3] = vof(4 * R + 3, xr1.y);
        vc[0] = vof(4 * Cc, xc0.x); vc[1] = vof(4 * Cc + 1, xc0.y); vc[2] = vof(4 * Cc + 2, xc1.x); vc[3] = vof(4 * Cc + 3, xc1.y);
        double t4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] = (a[r][0] * vc[0] + a[r][1] * vc[1]) + (a[r][2] * vc[2] + a[r][3] * vc[3]);
        // four independent reductions over the 16 lanes of the row group, stage by stage (the DPP latencies overlap)
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] += tdpp<0xB1>(t4[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] += tdpp<0x4E>(t4[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] += tdpp<0x141>(t4[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] += tdpp<0x140>(t4[r]);
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = (4 * R + r > k) ? t4[r] : 0.0;      // rows <= k are finished: keep them out of the update
          s += vr[r] * pr[r];
        }
        // v^T A v: one value per 16-lane row group -> per wave -> LDS
        const double sw = (readlane_f64(s, 0) + readlane_f64(s, 16)) + (readlane_f64(s, 32) + readlane_f64(s, 48));
        if (Cc == 0) {
          *reinterpret_cast<double2_t*>(&ps[4 * R]) = double2_t{pr[0], pr[1]};
          *reinterpret_cast<double2_t*>(&ps[4 * R + 2]) = double2_t{pr[2], pr[3]};
        }
        if (lane == 0) red[wave] = sw;
        TSUB(2);
        __syncthreads();
        TSUB(3);
        const double2_t q0 = *reinterpret_cast<const double2_t*>(&red[0]), q1 = *reinterpret_cast<const double2_t*>(&red[2]);
        const double K = 0.5 * tau * tau * ((q0.x + q0.y) + (q1.x + q1.y));
        const double2_t p0 = *reinterpret_cast<const double2_t*>(&ps[4 * Cc]), p1 = *reinterpret_cast<const double2_t*>(&ps[4 * Cc + 2]);
        const double pc[4] = {p0.x, p0.y, p1.x, p1.y};
        double wr[4], wc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wr[r] = tau * pr[r] - K * vr[r];
          wc[r] = tau * pc[r] - K * vc[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[r][c] -= vr[r] * wc[c] + wr[r] * vc[c];
        TSUB(4);
      } else {
        __syncthreads();                              // keep two barriers per step: a wave must not start step k+1 (new xs /
      }                                               // vs) while another one still reads this step's
    }
  }
  __syncthreads();
  // the last 2x2 block: d[n-2], d[n-1], e[n-2] (dynamic position -> through LDS, aliasing the LU area)
  {
    double (*Af)[kTLd] = reinterpret_cast<double (*)[kTLd]>(LUa);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Af[4 * R + r][4 * Cc + c] = a[r][c];
    __syncthreads();
    if (tid == 0) {
      dd[n - 2] = Af[n - 2][n - 2];
      dd[n - 1] = Af[n - 1][n - 1];
      ee[n - 2] = Af[n - 1][n - 2];
    }
    __syncthreads();
  }

  TSTAMP(2);
  // ---- 2. scale + pad, bisection for the leading rw eigenvalues ----
  const int rw = min(n, max(1, d.r) + 2);
  double tn;
  {
    double g = 0.0;
    if (lane < n) g = fabs(dd[lane]) + (lane > 0 ? fabs(ee[lane - 1]) : 0.0) + (lane < n - 1 ? fabs(ee[lane]) : 0.0);
    g = fmax(g, tdpp<0xB1>(g)); g = fmax(g, tdpp<0x4E>(g)); g = fmax(g, tdpp<0x141>(g)); g = fmax(g, tdpp<0x140>(g));
    tn = fmax(fmax(readlane_f64(g, 0), readlane_f64(g, 16)), fmax(readlane_f64(g, 32), readlane_f64(g, 48)));
  }
  if (!(tn > 0.0) || !(tn < 1e300)) return;           // zero / non-finite matrix: the Jacobi path decides (uniform)
  const double itn = 1.0 / tn;
  double* e2p = xs;            // [64] squared coupling in FRONT of row i: (e_{i-1}/tn)^2, 0 for i = 0 (xs is free now)
  if (tid < kTN) {
    ds[tid] = tid < n ? dd[tid] * itn : -8.0;         // padding: decoupled entries far below the spectrum [-1, 1]
    es[tid] = tid < n - 1 ? ee[tid] * itn : 0.0;
    const double ep = (tid > 0 && tid < n) ? ee[tid - 1] * itn : 0.0;
    e2p[tid] = ep * ep;
  }
  __syncthreads();
  // the scaled tridiagonal is read from LDS eight entries at a time (every lane the same address: broadcast reads), the
  // next block requested before the current one is consumed: the dependent chains below never wait for a load
  auto load8 = [&](const double* src, double* dst) {
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      const double2_t t = *reinterpret_cast<const double2_t*>(&src[u]);
      dst[u] = t.x; dst[u + 1] = t.y;
    }
  };
  {
    // number of eigenvalues of the padded tridiagonal below x: sign changes of p_0 = 1, p_i = det(T_i - x I).  Product
    // form -- ONE FMA per step on the dependent chain instead of a division -- with the pair (p_{i-1}, p_i) rescaled
    // every 8 steps; an exact zero counts as a sign change (the sign opposite to its predecessor).
    auto sturm = [&](double x) -> int {
      double dv[8], ev[8], dn[8], en[8];
      load8(ds, dv); load8(e2p, ev);
      double p0 = 1.0, p1 = 1.0;
      int cnt = 0;
#pragma unroll
      for (int b = 0; b < NS / 8; ++b) {
        if (b + 1 < NS / 8) { load8(ds + 8 * (b + 1), dn); load8(e2p + 8 * (b + 1), en); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          // (block 0, u = 0: p_1 = (d_0 - x) * 1 - 0.)  An exact zero needs no special case: its successor is -e^2 p_{i-1},
          // of the sign opposite to its predecessor, and the two comparisons below count exactly one change for the pair
          const double p2 = fma(dv[u] - x, p1, -ev[u] * p0);
          cnt += ((p2 < 0.0) != (p1 < 0.0)) ? 1 : 0;
          p0 = p1; p1 = p2;
        }
        const int ex = __builtin_amdgcn_frexp_exp(fmax(fabs(p0), fabs(p1)));
        p0 = __builtin_amdgcn_ldexp(p0, -ex);
        p1 = __builtin_amdgcn_ldexp(p1, -ex);
        if (p1 == 0.0) p1 = (p0 < 0.0) ? 1e-300 : -1e-300;       // (underflow of the smaller one)
#pragma unroll
        for (int u = 0; u < 8; ++u) { dv[u] = dn[u]; ev[u] = en[u]; }
      }
      return cnt;
    };
    // P probes per eigenvalue, the P lanes of an eigenvalue adjacent; eigenvalue j (descending) has kth = NS-1-j below it
    const int P = rw <= 16 ? 16 : (rw <= 32 ? 8 : 4);
    const int j = tid / P, pi = tid - j * P;
    const bool act = j < rw;
    const int kth = NS - 1 - j;
    double lo = -1.0 - 1e-12, hi = 1.0 + 1e-12;      // Gershgorin bounds of the scaled matrix
    const int rounds = P == 16 ? 11 : (P == 8 ? 14 : 19);      // >= 44 bits
    const double step = 1.0 / (P + 1);
    for (int it = 0; it < rounds; ++it) {
      const double xq = lo + (hi - lo) * ((pi + 1) * step);
      const int c = sturm(xq);
      const bool le = c <= kth;                       // the probe is still <= lambda_j
      const unsigned long long m = __ballot(le);
      const int base = (lane / P) * P;
      const unsigned long long grp = (m >> base) & ((1ull << P) - 1ull);
      const int cntle = __popcll(grp);                // counts are monotone in x: the first cntle probes are <= lambda_j
      const double w = hi - lo;
      const double nlo = cntle > 0 ? lo + w * (cntle * step) : lo;
      const double nhi = cntle < P ? lo + w * ((cntle + 1) * step) : hi;
      lo = nlo; hi = nhi;
    }
    if (act && pi == 0) lam[j] = 0.5 * (lo + hi);     // scaled eigenvalue
  }
  __syncthreads();

  TSTAMP(3);
  // ---- 3. clusters, inverse iteration (one thread per eigenvector, the vector in registers) ----
  int* cstart = ivec;              // [64]
  int* flags = ivec + 64;          // [0] max cluster position, [1] failure
  if (tid == 0) {
    int mx = 0;
    cstart[0] = 0;
    for (int j = 1; j < rw; ++j) {
      cstart[j] = (lam[j - 1] - lam[j] <= kTClusterTol) ? cstart[j - 1] : j;
      mx = max(mx, j - cstart[j]);
    }
    flags[0] = mx; flags[1] = 0;
  }
  __syncthreads();
  const int maxpos = flags[0];
  if (maxpos >= kTMaxCluster) return;                 // a large cluster (rank-deficient / constant input): Jacobi path
  {
    const int j = tid;
    const bool act = j < rw;
    const int st = act ? cstart[j] : 0, pos = j - st;
    double shift = act ? lam[j] : 0.0;
    if (act && pos > 0) {                             // keep the shifts of a cluster apart (dstein)
      const double sep = 10.0 * 2.220446049250313e-16;
      double prev = lam[st];
      for (int i = st + 1; i <= j; ++i) prev = fmin(lam[i], prev - sep);
      shift = prev;
    }
    const double tol = 2.220446049250313e-16;         // pivot floor (scaled matrix: ||T|| ~ 1)
    double x[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
      x[i] = (act && i < n) ? hash_pm1((uint32_t)p * 2654435761u + 101u, (uint32_t)(j * kTN + i)) : 0.0;
    for (int iter = 0; iter < 2; ++iter) {
      if (act) {
        // forward elimination with partial pivoting on rows (k, k+1); the right-hand side rides along in x
        double db[8], eb[8], dnx[8], enx[8];
        load8(ds, db); load8(es, eb);                 // db[u] = d_{8b+u}, eb[u] = e_{8b+u} (coupling of rows 8b+u, 8b+u+1)
        double ak = db[0] - shift;                    // current diagonal entry of row k
        double bk = eb[0];                            // current super-diagonal entry of row k
        double xk = x[0];
        unsigned long long swapped = 0ull;
#pragma unroll
        for (int b = 0; b < NS / 8; ++b) {
          if (b + 1 < NS / 8) { load8(ds + 8 * (b + 1), dnx); load8(es + 8 * (b + 1), enx); }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = 8 * b + u;
            if (k < NS - 1) {
              const double ck = eb[u];                                       // sub-diagonal entry of row k+1
              const double ak1 = ((u < 7) ? db[u + 1] : dnx[0]) - shift;
              const double bk1 = (k + 1 < NS - 1) ? ((u < 7) ? eb[u + 1] : enx[0]) : 0.0;
              const double xk1 = x[k + 1];
              // branch-free (the lanes of a wave disagree about interchanges: both sides of a branch would run)
              const bool sw = fabs(ak) < fabs(ck);    // interchange rows k and k+1
              double piv = sw ? ck : ak;
              if (!(fabs(piv) > tol)) piv = (piv < 0.0) ? -tol : tol;
              const double ip = fast_rcp(piv);
              const double m = (sw ? ak : ck) * ip;
              LUa[k * kTN + j] = ip; LUb[k * kTN + j] = sw ? ak1 : bk;
              swapped |= sw ? (1ull << k) : 0ull;
              x[k] = sw ? xk1 : xk;
              const double na = fma(-m, sw ? ak1 : bk, sw ? bk : ak1);
              const double nx = fma(-m, sw ? xk1 : xk, sw ? xk : xk1);
              bk = sw ? -m * bk1 : bk1;
              ak = na; xk = nx;
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { db[u] = dnx[u]; eb[u] = enx[u]; }
        }
        double piv = ak;
        if (!(fabs(piv) > tol)) piv = (piv < 0.0) ? -tol : tol;
        // back substitution: row k = (piv_k, LUb_k, swapped_k ? e_{k+1} : 0); the stored factors come back in blocks of 8
        double x2 = 0.0, x1 = xk * fast_rcp(piv);
        x[NS - 1] = x1;
        double mx = fabs(x1);
#pragma unroll
        for (int b = NS / 8 - 1; b >= 0; --b) {
          double la[8], lb[8], e1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = 8 * b + u;
            la[u] = (k < NS - 1) ? LUa[k * kTN + j] : 0.0;
            lb[u] = (k < NS - 1) ? LUb[k * kTN + j] : 0.0;
            e1[u] = (k + 1 < NS - 1) ? es[k + 1] : 0.0;
          }
#pragma unroll
          for (int u = 7; u >= 0; --u) {
            const int k = 8 * b + u;
            if (k < NS - 1) {
              const double s2 = ((swapped >> k) & 1ull) ? e1[u] : 0.0;
              const double xv = (x[k] - lb[u] * x1 - s2 * x2) * la[u];
              x[k] = xv;
              x2 = x1; x1 = xv;
              mx = fmax(mx, fabs(xv));
            }
          }
        }
        // overflow guard + normalisation (cluster members are re-normalised after their orthogonalisation)
        const double sc = (mx > 0.0 && mx < 1e300) ? fast_rcp(mx) : 0.0;
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
#pragma unroll
        for (int i = 0; i < NS; i += 4) {
          const double t0 = x[i] * sc, t1 = x[i + 1] * sc, t2 = x[i + 2] * sc, t3 = x[i + 3] * sc;
          n0 = fma(t0, t0, n0); n1 = fma(t1, t1, n1); n2 = fma(t2, t2, n2); n3 = fma(t3, t3, n3);
        }
        const double nn = (n0 + n1) + (n2 + n3);
        const double inv = nn > 0.0 ? sc * fast_rsqrt(nn) : 0.0;
        if (!(inv > 0.0)) flags[1] = 1;
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] *= inv;
#pragma unroll
        for (int i = 0; i < NS; i += 2) *reinterpret_cast<double2_t*>(&Z[j][i]) = double2_t{x[i], x[i + 1]};
      }
      __syncthreads();
      for (int q = 1; q <= maxpos; ++q) {             // modified Gram-Schmidt inside clusters, in order (rare, short)
        if (act && pos == q) {
          for (int i = st; i < j; ++i) {
            double dot = 0.0;
#pragma unroll
            for (int t = 0; t < NS; ++t) dot = fma(Z[i][t], x[t], dot);
#pragma unroll
            for (int t = 0; t < NS; ++t) x[t] = fma(-dot, Z[i][t], x[t]);
          }
          double nn = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) nn = fma(x[t], x[t], nn);
          const double inv = nn > 1e-20 ? fast_rsqrt(nn) : 0.0;
          if (!(inv > 0.0)) flags[1] = 1;
#pragma unroll
          for (int t = 0; t < NS; ++t) { x[t] *= inv; Z[j][t] = x[t]; }
        }
        __syncthreads();
      }
    }
    TSTAMP(4);
    // ---- 4. Rayleigh quotients (unscaled) and residuals ----
    if (act) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int b = 0; b < NS / 8; ++b) {
        double db[8], eb[8];
        load8(ds + 8 * b, db); load8(es + 8 * b, eb);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = 8 * b + u;
          t0 = fma(db[u] * x[i], x[i], t0);
          if (i + 1 < NS) t1 = fma(eb[u] * x[i], x[i + 1], t1);
        }
      }
      const double th = t0 + 2.0 * t1;
      double rmax = 0.0;
#pragma unroll
      for (int b = 0; b < NS / 8; ++b) {
        double db[8], eb[8];
        load8(ds + 8 * b, db); load8(es + 8 * b, eb);
        const double eprev = b > 0 ? es[8 * b - 1] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = 8 * b + u;
          double t = (db[u] - th) * x[i];
          if (i > 0) t = fma(u > 0 ? eb[u - 1] : eprev, x[i - 1], t);
          if (i + 1 < NS) t = fma(eb[u], x[i + 1], t);
          rmax = fmax(rmax, fabs(t));
        }
      }
      if (!(rmax <= 1e-13) || th != th) flags[1] = 1;
      theta[j] = th * tn;
    }
  }
  __syncthreads();
  TSTAMP(5);
  {   // orthogonality of ALL computed pairs (inverse iteration gives no guarantee): thread (j, quarter) takes i = quarter, +4, ...
    const int j = tid & 63, q4 = tid >> 6;
    if (j < rw) {
      double zj[NS];
#pragma unroll
      for (int t = 0; t < NS; t += 2) {
        const double2_t v2 = *reinterpret_cast<const double2_t*>(&Z[j][t]);
        zj[t] = v2.x; zj[t + 1] = v2.y;
      }
      double worst = 0.0;
      for (int i = q4; i < j; i += 4) {
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int t = 0; t < NS; t += 4) {
          const double2_t v2 = *reinterpret_cast<const double2_t*>(&Z[i][t]), v3 = *reinterpret_cast<const double2_t*>(&Z[i][t + 2]);
          d0 = fma(v2.x, zj[t], d0); d1 = fma(v2.y, zj[t + 1], d1); d2 = fma(v3.x, zj[t + 2], d2); d3 = fma(v3.y, zj[t + 3], d3);
        }
        worst = fmax(worst, fabs((d0 + d1) + (d2 + d3)));
      }
      if (!(worst <= 1e-11)) flags[1] = 1;
    }
  }
  __syncthreads();
  if (flags[1]) return;                               // uniform: something failed its check -> Jacobi path, image untouched

  TSTAMP(6);
  // ---- 5. back-transformation u = H_0 H_1 ... H_{n-3} z and output: four threads per vector, 16 entries each ----
  {
    const int j = tid >> 2, part = tid & 3;
    const bool act = j < rw;
    double u[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const double2_t v2 = *reinterpret_cast<const double2_t*>(&Z[j][16 * part + c]);
      u[c] = (act && 16 * part + c < NS) ? v2.x : 0.0;          // (Z rows hold NS entries)
      u[c + 1] = (act && 16 * part + c + 1 < NS) ? v2.y : 0.0;
    }
    double hv[16];
    auto load_hv = [&](int k) {
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        const double2_t v2 = *reinterpret_cast<const double2_t*>(&Hv[k][16 * part + c]);
        hv[c] = v2.x; hv[c + 1] = v2.y;
      }
    };
    double taun = 0.0;
    if (n >= 3) { load_hv(n - 3); taun = tauv[n - 3]; }
    for (int k = n - 3; k >= 0; --k) {
      const double tau = taun;
      double h[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) h[c] = hv[c];
      if (k > 0) { load_hv(k - 1); taun = tauv[k - 1]; }     // next reflector on its way while this one is applied
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
      for (int c = 0; c < 16; c += 4) {
        d0 = fma(h[c], u[c], d0); d1 = fma(h[c + 1], u[c + 1], d1); d2 = fma(h[c + 2], u[c + 2], d2); d3 = fma(h[c + 3], u[c + 3], d3);
      }
      const double dot = quad_sum((d0 + d1) + (d2 + d3)) * tau;
#pragma unroll
      for (int c = 0; c < 16; ++c) u[c] = fma(-dot, h[c], u[c]);
    }
    {   // last check before the image is overwritten: the back-transformed vectors are unit vectors (finite!)
      double nn = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) nn = fma(u[c], u[c], nn);
      nn = quad_sum(nn);
      if (act && !(fabs(nn - 1.0) <= 1e-9)) flags[1] = 1;
    }
    __syncthreads();                                  // (also: every read of the input image happened long ago)
    if (flags[1]) return;                             // uniform
    const int Npad = d.Npad;
    if (j < Npad) {
      const double sc = act ? theta[j] : 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int i = 16 * part + c;
        if (i < Npad) XT[(int64_t)j * ld + i] = (act && i < n) ? sc * u[c] : 0.0;
      }
    }
  }
  TSTAMP(7);
  if (tid == 0) {
    *d.done = 1;
    if (d.warm_ok) *d.warm_ok = 0;                    // no Jacobi state to continue from
    verdict[1 + p] = 1;
    __threadfence();
    fast_done[p] = 1;
  }
}

__global__ __launch_bounds__(256) void eig_small_direct_kernel(const EigDesc* __restrict__ descs,
                                                               const int32_t* __restrict__ skip,
                                                               int32_t* __restrict__ fast_done, int* __restrict__ verdict) {
  extern __shared__ __attribute__((aligned(16))) double tsm[];
  const int p = blockIdx.x;
  if (threadIdx.x == 0) fast_done[p] = 0;
  if (skip && skip[p]) return;                       // jacobi_small_kernel, launched behind this one, does the bookkeeping
  const EigDesc d = descs[p];
  if (d.N < 3 || d.N > kTN) return;
  if (d.N <= 32) eig_small_direct_body<32>(d, p, tsm, fast_done, verdict);
  else eig_small_direct_body<64>(d, p, tsm, fast_done, verdict);
}

bool eig_small_direct_on() {
  const char* e = getenv("TADMM_SMALL_DIRECT");       // 0: always the Jacobi path (A/B measurements, tests)
  return !(e && !atoi(e));
}

void launch_eig_small_direct(const EigDesc* descs_dev, int nprob, const int32_t* skip, int32_t* fast_done_dev,
                             int* verdict_pinned, hipStream_t s) {
  if (nprob <= 0) return;
  static bool attr_done[64] = {false};
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (!attr_done[devi & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eig_small_direct_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done[devi & 63] = true;
  }
  hipLaunchKernelGGL(eig_small_direct_kernel, dim3(nprob), dim3(256), kTLdsBytes, s, descs_dev, skip, fast_done_dev,
                     verdict_pinned);
#ifdef TADMM_TRI_STAMPS
  if (getenv("TADMM_TRI_STAMPS_DUMP")) {
    long long h[16];
    (void)hipStreamSynchronize(s);
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tri_stamps), sizeof h) == hipSuccess)
      fprintf(stderr, "[tri stamps] cycles: load=%lld tridiag=%lld bisect=%lld invit=%lld rq=%lld orth=%lld back=%lld total=%lld | tridiag: xs+barrier=%lld reflector=%lld matvec=%lld barrier=%lld update=%lld\n",
              h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0],
              h[8], h[9], h[10], h[11], h[12]);
  }
#endif
}

}  // namespace tadmm
