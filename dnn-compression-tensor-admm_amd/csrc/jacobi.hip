// Symmetric eigen-solver for the Gram matrices of the TT-SVD steps: one-sided block Jacobi in fp64.
//
// The truncated SVD of an unfolding A (reference: numpy.linalg.svd inside ttd.py:17) is obtained from
// the eigen-decomposition of its Gram matrix G (N x N, N = min(m,n) <= ~1.2k).  We run Hestenes'
// one-sided Jacobi on X = G: X <- X*Q with plane rotations that orthogonalise the columns of X; at
// convergence X = G*V has orthogonal columns whose norms are the eigenvalues and whose directions are
// the eigenvectors (x_j = lambda_j v_j).
//
// Blocking: columns are grouped in blocks of kJB = 8; a *pair* of blocks = 16 columns = exactly one
// 16x16 fp64 MFMA tile.  One launch ("tick") processes nb/2 disjoint block pairs, one workgroup each,
// following a round-robin tournament so that after nb-1 ticks every pair of blocks has met once (one
// sweep).  Per pair and tick:
//   1. H = Xp^T Xp  (16x16 Gram of the pair's columns over all N rows)        -- v_mfma_f64_16x16x4
//   2. a few cyclic two-sided Jacobi sweeps on H accumulate the 16x16 rotation Q (one wave, LDS)
//   3. Xp <- Xp * Q                                                             -- v_mfma_f64_16x16x4
// Pairs never share columns, so a tick needs no inter-workgroup communication; the kernel boundary is
// the only global synchronisation.  X is stored transposed (XT[j][:] = column j, contiguous) so every
// access is a coalesced row segment.
//
// Convergence: every pair visit records max |h_ab|/sqrt(h_aa h_bb) *before* rotating into a per-problem
// slot (double compared as integer, values are >= 0).  A problem is converged when a whole sweep saw
// nothing above `tol`; Jacobi converges quadratically, so tol = 1e-9 leaves ~1e-16 after that sweep.
#include "common.h"

namespace tadmm {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kPair = 2 * kJB;  // 16

__global__ __launch_bounds__(256) void jacobi_init_kernel(const EigDesc* __restrict__ descs) {
  const EigDesc d = descs[blockIdx.x];
  if (threadIdx.x == 0) {
    d.off[0] = 1.0; d.off[1] = 1.0;
    *d.done = 0;
  }
  // scale reference for "numerically null column": max squared column norm of the initial X = G
  __shared__ double red[4];
  double mx = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < d.N; j += 4) {
    double s = 0.0;
    const double* row = d.XT + (int64_t)j * d.ld;
    for (int i = lane; i < d.N; i += 64) s += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    s = __shfl(s, 0, 64);
    mx = fmax(mx, s);
  }
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) d.off[2] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__device__ __forceinline__ void rr_pair(int nb, int step, int q, int& a, int& b) {
  // circle-method round robin on nb players (nb even): player nb-1 is fixed
  const int m = nb - 1;
  if (q == 0) { a = m; b = step % m; }
  else { a = (step + q) % m; b = (step - q + m) % m; }
}

__global__ __launch_bounds__(256) void jacobi_tick_kernel(const EigDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map, int tick, double tol,
                                                          int inner_sweeps) {
  __shared__ double red[4][256];
  __shared__ double Hs[kPair][kPair + 1];
  __shared__ double Qs[kPair][kPair + 1];
  __shared__ double coefA[kPair], coefB[kPair];
  __shared__ int partner[kPair];
  __shared__ int rotated;

  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  if (*d.done) return;
  const int nb = d.nb;
  const int steps = nb - 1;
  const int sweep = tick / steps;
  const int step = tick - sweep * steps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (step == 0 && sweep > 0) {
    if (d.off[(sweep - 1) & 1] < tol) {           // previous sweep saw nothing left to rotate
      if (br.local == 0 && tid == 0) *d.done = 1;
      return;
    }
  }
  int ba, bb;
  rr_pair(nb, step, br.local, ba, bb);
  const int r = lane & 15, q = lane >> 4;
  const int ld = d.ld;
  // MFMA row r of the pair -> row of XT
  const int myrow = (r < kJB) ? (ba * kJB + r) : (bb * kJB + (r - kJB));
  double* __restrict__ XT = d.XT;

  // ---- 1. H = Xp^T Xp : each wave reduces a quarter of the rows of X (= columns of XT) ----
  {
    const int per = ld >> 2;                       // ld is a multiple of 32 -> per % 8 == 0
    const int i0 = wave * per, i1 = i0 + per;
    const double* row = XT + (int64_t)myrow * ld;
    double4_t acc = {0, 0, 0, 0};
    for (int i = i0; i < i1; i += 8) {
      const double2_t v = *reinterpret_cast<const double2_t*>(row + i + 2 * q);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.y, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
  }
  if (tid == 0) rotated = 0;
  __syncthreads();
  // every thread has taken its convergence decision by now: safe to clear the slot of the NEXT sweep
  if (step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;
  {
    const int l = tid >> 2, reg = tid & 3;
    const int hr = (l >> 4) + 4 * reg, hc = l & 15;
    Hs[hr][hc] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    Qs[hr][hc] = (hr == hc) ? 1.0 : 0.0;
  }
  __syncthreads();

  // ---- 2. inner two-sided Jacobi on the 16x16 H, one wave, LDS resident ----
  if (wave == 0) {
    volatile double (*H)[kPair + 1] = Hs;
    volatile double (*Q)[kPair + 1] = Qs;
    volatile double* cA = coefA;
    volatile double* cB = coefB;
    volatile int* pt = partner;
    const double nullfloor = d.off[2] * 1e-26;
    // convergence measure before rotating
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = lane * 4 + k;
      const int i = e >> 4, j = e & 15;
      if (i < j) {
        const double hii = H[i][i], hjj = H[j][j];
        if (hii > nullfloor && hjj > nullfloor) mx = fmax(mx, fabs(H[i][j]) / sqrt(hii * hjj));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if (lane == 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]),
                (unsigned long long)__double_as_longlong(mx));
    }
    mx = __shfl(mx, 0, 64);
    if (mx > 1e-15) {
      int did = 0;
      for (int isw = 0; isw < inner_sweeps; ++isw) {
        for (int st = 0; st < kPair - 1; ++st) {
          if (lane < kPair / 2) {
            int p, qq;
            rr_pair(kPair, st, lane, p, qq);
            if (p > qq) { const int t = p; p = qq; qq = t; }
            const double hpp = H[p][p], hqq = H[qq][qq], hpq = H[p][qq];
            double c = 1.0, s = 0.0;
            if (fabs(hpq) > 1e-18 * sqrt(fabs(hpp * hqq)) && fabs(hpq) > 1e-300) {
              const double tau = (hqq - hpp) / (2.0 * hpq);
              const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
              c = 1.0 / sqrt(1.0 + t * t);
              s = t * c;
              did = 1;
            }
            // new_p = c*old_p - s*old_q ; new_q = s*old_p + c*old_q
            cA[p] = c; cB[p] = -s; pt[p] = qq;
            cA[qq] = c; cB[qq] = s; pt[qq] = p;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          double nh[4], nq[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = lane * 4 + k;
            const int i = e >> 4, j = e & 15;
            const int is = pt[i], js = pt[j];
            const double ai = cA[i], bi = cB[i], aj = cA[j], bj = cB[j];
            nh[k] = ai * (aj * H[i][j] + bj * H[i][js]) + bi * (aj * H[is][j] + bj * H[is][js]);
            nq[k] = aj * Q[i][j] + bj * Q[i][js];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = lane * 4 + k;
            H[e >> 4][e & 15] = nh[k];
            Q[e >> 4][e & 15] = nq[k];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      did = __any(did);
      if (lane == 0) rotated = did;
    }
  }
  __syncthreads();
  if (!rotated) return;

  // ---- 3. Xp <- Xp * Q, i.e. rows of XT:  Y[a][:] = sum_b Q[b][a] * XT[row(b)][:]  ----
  {
    double qa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qa[t] = Qs[4 * t + q][r];      // A operand: A[m=a][k=b] = Q[b][a]
    int rowk[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int b = 4 * t + q;
      rowk[t] = (b < kJB) ? (ba * kJB + b) : (bb * kJB + (b - kJB));
    }
    const int ntile = ld >> 4;
    for (int it = wave; it < ntile; it += 4) {
      const int col = it * 16 + r;
      double4_t acc = {0, 0, 0, 0};
      double bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = XT[(int64_t)rowk[t] * ld + col];   // B[k=b][n=i]
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], bv[t], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int a = q + 4 * e;                                           // D row = (l>>4) + 4*reg
        const int orow = (a < kJB) ? (ba * kJB + a) : (bb * kJB + (a - kJB));
        XT[(int64_t)orow * ld + col] = acc[e];
      }
    }
  }
}

// ---- finalize: eigenvalues = column norms, descending order, scaled eigenvectors ----
__global__ __launch_bounds__(256) void eig_norms_kernel(const EigDesc* __restrict__ descs,
                                                        const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = br.local * 4 + wave;
  if (j >= d.Npad) return;
  double s = 0.0;
  if (j < d.N) {
    const double* row = d.XT + (int64_t)j * d.ld;
    for (int i = lane; i < d.N; i += 64) s += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  }
  if (lane == 0) d.lam[j] = (j < d.N) ? sqrt(s) : -1.0;
}

__global__ __launch_bounds__(256) void eig_sort_kernel(const EigDesc* __restrict__ descs) {
  extern __shared__ double slam[];
  const EigDesc d = descs[blockIdx.x];
  for (int j = threadIdx.x; j < d.Npad; j += 256) slam[j] = d.lam[j];
  __syncthreads();
  for (int j = threadIdx.x; j < d.Npad; j += 256) {
    const double lj = slam[j];
    int rank = 0;
    for (int i = 0; i < d.Npad; ++i) {
      const double li = slam[i];
      rank += (li > lj) || (li == lj && i < j);
    }
    d.order[rank] = j;
    if (rank < d.r) d.sigma[rank] = sqrt(fmax(lj, 0.0));
  }
}

__global__ __launch_bounds__(256) void eig_extract_kernel(const EigDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = br.local * 4 + wave;
  if (c >= d.r) return;
  const int j = d.order[c];
  const double lam = d.lam[j];
  const double* row = d.XT + (int64_t)j * d.ld;
  // deterministic sign: the entry of largest magnitude (first on ties) is made positive
  double best = -1.0; int besti = 0;
  for (int i = lane; i < d.N; i += 64) {
    const double a = fabs(row[i]);
    if (a > best) { best = a; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_down(best, o, 64);
    const int oi = __shfl_down(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  besti = __shfl(besti, 0, 64);
  const double sgn = (row[besti] < 0.0) ? -1.0 : 1.0;
  const double inv = (lam > 0.0) ? sgn / lam : 0.0;
  const double sig = sqrt(fmax(lam, 0.0));
  const double isig = (sig > 0.0) ? 1.0 / sig : 0.0;
  const int r = d.r, N = d.N;
  for (int i = lane; i < N; i += 64) {
    const double v = row[i] * inv;
    if (d.mode == 0) {
      d.out_a[(int64_t)i * r + c] = (float)v;
    } else if (d.mode == 1) {
      d.out_a[(int64_t)i * r + c] = (float)(v * isig);
      d.out_b[(int64_t)c * N + i] = (float)(v * sig);
    }
    if (d.evec_out) d.evec_out[(int64_t)c * N + i] = v;
  }
}

void launch_jacobi_init(const EigDesc* descs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(jacobi_init_kernel, dim3(nprob), dim3(256), 0, s, descs_dev);
}
void launch_jacobi_tick(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                        int inner_sweeps, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(jacobi_tick_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, tick, tol, inner_sweeps);
}
void launch_eig_norms(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(eig_norms_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}
void launch_eig_sort(const EigDesc* descs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(eig_sort_kernel, dim3(nprob), dim3(256), 16384, s, descs_dev);
}
void launch_eig_extract(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(eig_extract_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
