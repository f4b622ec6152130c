// Symmetric eigen-solver for the Gram matrices of the TT-SVD steps: one-sided block Jacobi in fp64.
//
// The truncated SVD of an unfolding A (reference: numpy.linalg.svd inside ttd.py:17) is obtained from
// the eigen-decomposition of its Gram matrix G (N x N, N = min(m,n) <= ~1.2k).  We run Hestenes'
// one-sided Jacobi on X = G: X <- X*Q with plane rotations that orthogonalise the columns of X; at
// convergence X = G*V has orthogonal columns whose norms are the eigenvalues and whose directions are
// the eigenvectors (x_j = lambda_j v_j).
//
// Blocking.  Columns are grouped in blocks of kJB = 8; a *pair* of blocks = 16 columns = exactly one
// 16x16 fp64 MFMA tile.  A pair visit does
//   1. H = Xp^T Xp  (16x16 Gram of the pair's columns over all rows)           -- v_mfma_f64_16x16x4
//   2. one cyclic sweep of two-sided Jacobi on H accumulating the 16x16 rotation Q (one wave, LDS)
//   3. Xp <- Xp * Q                                                              -- v_mfma_f64_16x16x4
// X is stored transposed (XT[j][:] = column j, contiguous) so every access is a row segment.
//
// Launch shapes ("ticks"), all following a round-robin tournament so that after a sweep every pair of blocks
// has met exactly once; the kernel boundary is the only global synchronisation:
//   tick3 (default, ld <= 512): a workgroup owns a *super-pair* of two 16-column super-blocks (32 columns in LDS),
//          carries the two self-Grams from launch to launch, computes ONE cross-Gram tile, and runs 2 rounds of
//          2 concurrent 16x16 sub-problems (see the header of jacobi_tick3_kernel); the within-super-block
//          pairs are rotated once per sweep by tick1 in "self mode", which also refreshes the carried Grams;
//   tick2 (fallback, ld <= ~590): the same super-pair shape, every Gram recomputed from the columns;
//   tick1 (any ld that fits 16 columns in LDS): a workgroup owns one pair (16 columns);
//   small (Npad <= 64): the whole eigen-solve in one launch, one workgroup per problem (jacobi_small_kernel).
// Index pairs inside a block (tick1) / inside a super-block (tick2) are rotated on the first tick of a
// sweep only, when every (super-)block is in exactly one pair.
//
// Convergence.  Every pair visit records, *before* rotating, max_ij |h_ij| / (sqrt(h_ii h_jj) * w_ij) with
// w_ij = max(1, 1e-14/tol * lambda_max/min(lambda_i, lambda_j)): columns with small eigenvalues carry fp64
// rounding noise of relative size ~eps*lambda_max/lambda and cannot be orthogonalised beyond it, so their
// target scales accordingly (exactly low-rank inputs would otherwise never terminate).  A problem is
// converged when a whole sweep saw nothing above `tol` (Jacobi converges quadratically, so tol = 1e-9 leaves
// ~1e-16 after that sweep) or when the quadratic-phase prediction of jacobi_conv_kernel says so.
#include "common.h"
#include <algorithm>
#include <cstdio>

namespace tadmm {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kPair = 2 * kJB;  // 16
constexpr int kHP = kPair + 2;  // padded leading dimension of the 16x16 LDS images (even: 16-byte aligned rows)

__global__ __launch_bounds__(256) void jacobi_init_kernel(const EigDesc* __restrict__ descs,
                                                          const int32_t* __restrict__ skip, double* __restrict__ prev) {
  const EigDesc d = descs[blockIdx.x];
  if (prev && threadIdx.x == 0) prev[blockIdx.x] = 0.0;     // history of the convergence kernel (was a memset launch)
  if (skip && skip[blockIdx.x]) {           // dropped problem: finished before it starts
    if (threadIdx.x == 0) *d.done = 1;
    return;
  }
  // scale reference: squared column norms of X = G are >= G_jj^2, so the largest diagonal entry squared
  // is a cheap lower bound of the largest squared column norm (= lambda_max^2 at convergence)
  __shared__ double red[4];
  double mx = 0.0;
  for (int j = threadIdx.x; j < d.N; j += 256) mx = fmax(mx, fabs(d.XT[(int64_t)j * d.ld + j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double g = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    d.off[0] = 1.0; d.off[1] = 1.0; d.off[2] = g * g;
    *d.done = 0;
  }
}

__device__ __forceinline__ void rr_pair(int nb, int step, int q, int& a, int& b) {
  // circle-method round robin on nb players (nb even): player nb-1 is fixed
  const int m = nb - 1;
  if (q == 0) { a = m; b = step % m; }
  else { a = (step + q) % m; b = (step - q + m) % m; }
}

// LDS ordering inside ONE wave: the hardware executes a wave's LDS instructions in order, so data
// written by some lanes is visible to the other lanes of the same wave at the next instruction; the
// fence only stops the compiler from caching LDS values in registers / reordering across it.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Scratch of one pair visit (all in LDS)
struct PairScratch {
  double* red;            // [4][256] cross-wave partials of H
  double (*H)[kHP];       // [16][18]: H on entry, Y = Q^T H while solving; column 16 carries h_qq of the row
  double (*Q)[kHP];       // [16][18]: Q^T while solving
  int* rotated;           // [1]
};
constexpr int kPairScratchDoubles = 4 * 256 + 2 * kPair * kHP + 2;

__device__ __forceinline__ PairScratch carve_scratch(double* base) {
  PairScratch s;
  s.red = base;
  s.H = reinterpret_cast<double (*)[kHP]>(base + 4 * 256);
  s.Q = s.H + kPair;
  s.rotated = reinterpret_cast<int*>(s.Q + kPair);
  return s;
}

// H partial of one wave: rows `lrow(r)` of the LDS slab, columns [i0, i1) (multiple of 8 long)
__device__ __forceinline__ void pair_gram_partial(const double* __restrict__ Xs, int ldp, int lrow_r, int i0, int i1,
                                                  int q, double* __restrict__ red_w, int lane) {
  const double* row = Xs + lrow_r * ldp;
  double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int i = i0; i < i1; i += 8) {
    const double2_t v = *reinterpret_cast<const double2_t*>(row + i + 2 * q);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.y, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red_w[lane * 4 + e] = acc0[e] + acc1[e];
}

// 256 threads (t = 0..255) fold the 4 wave partials into H and reset Q = I
__device__ __forceinline__ void pair_gram_reduce(const PairScratch& S, int t) {
  const int l = t >> 2, reg = t & 3;
  const int hr = (l >> 4) + 4 * reg, hc = l & 15;
  S.H[hr][hc] = (S.red[t] + S.red[256 + t]) + (S.red[512 + t] + S.red[768 + t]);
  S.Q[hr][hc] = (hr == hc) ? 1.0 : 0.0;
}

// ---- DPP helpers (row = 16 lanes) ----
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  // every lane has a valid source for the controls used here, so the "old" operand is left undefined (mov_dpp)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over each aligned group of 8 lanes, result in all 8 (xor 1, xor 2, then mirror inside the group)
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);   // row_half_mirror
  return v;
}

// max over the 64 lanes of a wave, result in every lane: DPP inside the 16-lane rows, readlane across them
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_mov_f32<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov_f32<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov_f32<0x141>(v));   // row_half_mirror
  v = fmaxf(v, dpp_mov_f32<0x140>(v));   // row_mirror
  const int b = __float_as_int(v);
  const float m0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), m1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float m2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), m3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

#ifdef TADMM_STAMPS
__device__ unsigned long long g_stamps[32];
#define SSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { const unsigned long long tn = clock64(); g_stamps[i] += tn - ts; ts = tn; } } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif
// Inner solve, row formulation: two-sided cyclic Jacobi on the 16x16 symmetric H, carried as the pair
//   Y = Q^T H0   (rows rotated only)      and      Qt = Q^T,
// so that the current two-sided matrix is H_cur = Y Qt^T and its entries are 16-term dot products
//   h_pp = <Y_p,Qt_p>,  h_qq = <Y_q,Qt_q>,  h_pq = <Y_p,Qt_q>.
// A rotation then only mixes ROWS p,q of Y and of Qt.  8 lanes per index pair, 2 entries per lane and row;
// the three dot products are reduced on the DPP crossbar inside the 8-lane group, so every lane derives (c, s)
// itself: no coefficient broadcast, no partner table, one LDS round trip per step.  Exactly the rotations of
// two-sided Jacobi on H (same accuracy for small eigen-components), at less than half the latency of the
// element-wise formulation `pair_inner_solve`.
// Q is accumulated TRANSPOSED in S.Q during the pass (row p = column p of Q) and transposed back at the end.
// kKeepQt: leave Q^T in S.Q (the caller's MFMA operand reads take the transposed index order for free).
// kMeasure = false: skip the convergence measure of the incoming H (tick3 lets an otherwise idle wave compute it
// from the 32x32 image, `pair_measure_h32`) and rotate unconditionally; returns 0.
template <bool kKeepQt = false, bool kMeasure = true>
__device__ __forceinline__ double pair_inner_solve_fast(const PairScratch& S, int lane, double hmax, double tol,
                                                        bool within, double* __restrict__ hcur_out = nullptr) {
  double (*Hs)[kHP] = S.H;
  double (*Qs)[kHP] = S.Q;
#ifdef TADMM_STAMPS
  unsigned long long ts = clock64();
#endif
  const double inv_hmax = hmax > 0.0 ? 1.0 / hmax : 1.0;
  double mx = 0.0;
  if (kMeasure) {
    const float wscale = (float)(1e-14 / tol);
    float mxf = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = lane * 4 + k;
      const int i = e >> 4, j = e & 15;
      if (i < j) {
        const float ri = (float)(Hs[i][i] * inv_hmax), rj = (float)(Hs[j][j] * inv_hmax);
        const float rmin = fminf(ri, rj);
        if (rmin > 1e-28f) {
          const float hij = fabsf((float)(Hs[i][j] * inv_hmax));
          const float w = fmaxf(1.0f, wscale * __builtin_amdgcn_rsqf(rmin));
          mxf = fmaxf(mxf, hij * __builtin_amdgcn_rsqf(ri) * __builtin_amdgcn_rsqf(rj) * __builtin_amdgcn_rcpf(w));
        }
      }
    }
    mx = (double)wave_max_f32(mxf);
  }
  SSTAMP(10);
  int did = 0;
  if (!kMeasure || mx > 1e-15) {
    const int g8 = lane >> 3, l8 = lane & 7;
    // rotation of rows (p,q) given the three entries a = h_pp, b = h_qq, g = h_pq of the current matrix
    auto rotation = [&](double a, double b, double g, double& c, double& s, double& t) -> bool {
      if (!(g * g > 1e-36 * fabs(a * b) && fabs(g) > 1e-300)) return false;   // uniform inside the 8-lane group
      // (b - a) and 2g are brought to a common binary exponent before they are narrowed to fp32: at the rounding floor of a
      // rank-deficient block (columns annihilated to ~1e-16 of the others, entries ~1e-32 hmax) both used to underflow to
      // zero in fp32 and the angle became 0 * inf = NaN -- eight (nearly) identical columns in one block were enough
      const double zd = b - a, wd = 2.0 * g;
      const int ew = __builtin_amdgcn_frexp_exp(wd);
      const int ex = zd == 0.0 ? ew : max(__builtin_amdgcn_frexp_exp(zd), ew);
      const float zf = (float)__builtin_amdgcn_ldexp(zd, -ex), wf = (float)__builtin_amdgcn_ldexp(wd, -ex);
      // tan of the rotation angle, t = sign(z) w / (|z| + sqrt(z^2 + w^2)): after the scaling the larger of |z|, |w| lies in
      // [0.5, 1), so the sum of squares neither overflows nor underflows and ONE branch-free expression serves both
      // regimes (the |z| >= |w| / |z| < |w| pair it replaces executed both of its sides whenever the eight-lane groups of
      // a wave disagreed: a reciprocal, a square root and a reciprocal each)
      const float az = fabsf(zf);
      float tf = wf * __builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(__builtin_fmaf(az, az, wf * wf)));
      if (zf < 0.0f) tf = -tf;
      t = (double)tf;
      const double x = 1.0 + t * t;
      c = __builtin_amdgcn_rsq(x);            // ~2^-26 relative; one Newton step squares that
      c = c * (1.5 - 0.5 * x * c * c);
      s = t * c;
      return true;
    };
    // cross pairs: row p = g8 stays with this 8-lane group for all kJB steps -> carried in registers; only the
    // q rows (which move from group to group) make the LDS round trip.  The diagonal entries are not recomputed
    // as dot products: a rotation changes exactly h_pp -> h_pp - t*g and h_qq -> h_qq + t*g (t = tan), so h_pp
    // lives in a register and h_qq travels with its row in the padding column of H; only g = h_pq is a reduction.
    {
      const int p = g8;
      if (lane < kPair) Hs[lane][kPair] = Hs[lane][lane];        // Q = I on entry: h_ii = H[i][i]
      wave_lds_fence();
      double2_t* yp = reinterpret_cast<double2_t*>(&Hs[p][2 * l8]);
      double2_t* tp = reinterpret_cast<double2_t*>(&Qs[p][2 * l8]);
      double2_t vp = *yp, up = *tp;
      double a = Hs[p][kPair];
      for (int st = 0; st < kJB; ++st) {
        const int q = kJB + ((g8 + st) & (kJB - 1));
        double2_t* yq = reinterpret_cast<double2_t*>(&Hs[q][2 * l8]);
        double2_t* tq = reinterpret_cast<double2_t*>(&Qs[q][2 * l8]);
        const double2_t vq = *yq, uq = *tq;
        const double b = Hs[q][kPair];
        const double g = group8_sum(vp.x * uq.x + vp.y * uq.y);
        double c, s, t;
        if (rotation(a, b, g, c, s, t)) {
          // new_p = c*old_p - s*old_q ; new_q = s*old_p + c*old_q
          *yq = double2_t{s * vp.x + c * vq.x, s * vp.y + c * vq.y};
          *tq = double2_t{s * up.x + c * uq.x, s * up.y + c * uq.y};
          vp = double2_t{c * vp.x - s * vq.x, c * vp.y - s * vq.y};
          up = double2_t{c * up.x - s * uq.x, c * up.y - s * uq.y};
          if (l8 == 0) Hs[q][kPair] = b + t * g;
          a -= t * g;
          did = 1;
        }
        wave_lds_fence();
      }
      *yp = vp; *tp = up;
      wave_lds_fence();
    }
    SSTAMP(11);
    // within-block pairs (self pass only)
    for (int st = 0; within && st < kJB - 1; ++st) {
      int a2, b2;
      rr_pair(kJB, st, g8 & 3, a2, b2);
      const int base = (g8 >> 2) * kJB;
      const int p = base + min(a2, b2), q = base + max(a2, b2);
      double2_t* yp = reinterpret_cast<double2_t*>(&Hs[p][2 * l8]);
      double2_t* yq = reinterpret_cast<double2_t*>(&Hs[q][2 * l8]);
      double2_t* tp = reinterpret_cast<double2_t*>(&Qs[p][2 * l8]);
      double2_t* tq = reinterpret_cast<double2_t*>(&Qs[q][2 * l8]);
      const double2_t vp = *yp, vq = *yq, up = *tp, uq = *tq;
      const double a = group8_sum(vp.x * up.x + vp.y * up.y);
      const double b = group8_sum(vq.x * uq.x + vq.y * uq.y);
      const double g = group8_sum(vp.x * uq.x + vp.y * uq.y);
      double c, s, t;
      if (rotation(a, b, g, c, s, t)) {
        *yp = double2_t{c * vp.x - s * vq.x, c * vp.y - s * vq.y};
        *yq = double2_t{s * vp.x + c * vq.x, s * vp.y + c * vq.y};
        *tp = double2_t{c * up.x - s * uq.x, c * up.y - s * uq.y};
        *tq = double2_t{s * up.x + c * uq.x, s * up.y + c * uq.y};
        did = 1;
      }
      wave_lds_fence();
    }
    did = __any(did);
    if (hcur_out) {     // Gram of the rotated columns: H_cur[a][b] = <Y_a, Qt_b>  (16-term dot products)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = lane * 4 + k;
        const int a = e >> 4, b = e & 15;
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < kPair; ++t) acc += Hs[a][t] * Qs[b][t];
        hcur_out[e] = acc;
      }
    }
    if (!kKeepQt) {   // S.Q holds Q^T: transpose in place (read everything, fence, write)
      double qt[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int e = lane * 4 + k; qt[k] = Qs[e & 15][e >> 4]; }
      wave_lds_fence();
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int e = lane * 4 + k; Qs[e >> 4][e & 15] = qt[k]; }
      wave_lds_fence();
    }
  } else if (hcur_out) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int e = lane * 4 + k; hcur_out[e] = Hs[e >> 4][e & 15]; }
  }
  if (lane == 0) *S.rotated = did;
  SSTAMP(12);
  return mx;
}

// Dynamic LDS layout of tick1 (doubles): X[16][ldp] | PairScratch
__global__ __launch_bounds__(256) void jacobi_tick_kernel(const EigDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map, int tick, double tol,
                                                          int inner_sweeps, int self_mode) {
  // self_mode: companion of jacobi_tick3_kernel.  The tournament runs over nb/2 super-blocks of 16
  // columns; this kernel acts only on the first tick of a sweep, one workgroup per super-block, and
  // rotates all 120 index pairs inside it (blocks 2*local and 2*local+1, `within` rotations included).
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  if (*d.done) return;
  const int nb = self_mode ? (d.nb >> 1) : d.nb;
  const int steps = nb - 1;
  const int period = (self_mode && d.period > 0) ? d.period : steps;
  const int sweep = tick / period;
  const int step = tick - sweep * period;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (step == 0 && sweep > 0) {
    if (d.off[(sweep - 1) & 1] < tol) {           // previous sweep saw nothing left to rotate
      if (br.local == 0 && tid == 0) *d.done = 1;
      return;
    }
  }
  if (self_mode && step != 0) return;
  const int ld = d.ld, ldp = ld + 2;
  double* Xs = smem;
  const PairScratch S = carve_scratch(Xs + kPair * ldp);

  int ba, bb;
  if (self_mode) { ba = 2 * br.local; bb = ba + 1; }
  else rr_pair(nb, step, br.local, ba, bb);
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ XT = d.XT;

  // ---- 0. stage the pair's 16 columns (rows of XT) in LDS ----
  if ((ld & 127) == 0) {
    // LDS-DMA, 1 KiB chunks, every chunk of a wave in flight at once (see jacobi_tick3_kernel)
    const int cpr = ld >> 7;
    const int nchunk = kPair * cpr;
    for (int c = wave; c < nchunk; c += 4) {
      const int row = c / cpr, ch = c - row * cpr;
      const int grow = (row < kJB) ? (ba * kJB + row) : (bb * kJB + (row - kJB));
      const double* src = XT + (int64_t)grow * ld + ch * 128 + lane * 2;
      double* dst = Xs + row * ldp + ch * 128;        // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
  } else {
    const int c2n = ld >> 1;                       // double2 chunks per row
    const int total = kPair * c2n;
    constexpr int kBatch = 8;                      // 16-byte loads in flight per thread (latency-bound phase)
    for (int base = tid; base < total; base += 256 * kBatch) {
      double2_t v[kBatch];
      int dst[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int idx = base + k * 256;
        const bool ok = idx < total;
        const int row = ok ? idx / c2n : 0, c2 = ok ? idx - row * c2n : 0;
        const int grow = (row < kJB) ? (ba * kJB + row) : (bb * kJB + (row - kJB));
        dst[k] = ok ? row * ldp + 2 * c2 : -1;
        v[k] = *reinterpret_cast<const double2_t*>(XT + (int64_t)grow * ld + 2 * c2);
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        if (dst[k] >= 0) *reinterpret_cast<double2_t*>(Xs + dst[k]) = v[k];
    }
  }
  __syncthreads();
  // every thread has taken its convergence decision by now: safe to clear the slot of the NEXT sweep
  // (in self mode the tick3 launch of the same tick does it)
  if (!self_mode && step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;

  const int per = ld >> 2;                         // ld is a multiple of 32 -> per % 8 == 0
  pair_gram_partial(Xs, ldp, r, wave * per, (wave + 1) * per, q, S.red + wave * 256, lane);
  __syncthreads();
  pair_gram_reduce(S, tid);
  __syncthreads();
  if (wave == 0) {
    double* hcur = (self_mode && d.sblk) ? d.sblk + (int64_t)br.local * (kPair * kPair) : nullptr;
    const double mx = pair_inner_solve_fast(S, lane, d.off[2], tol, step == 0, hcur);
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]), (unsigned long long)__double_as_longlong(mx));
  }
  __syncthreads();
  if (!*S.rotated) return;

  // ---- 3. Xp <- Xp * Q, i.e. rows of XT:  Y[a][:] = sum_b Q[b][a] * X[b][:]  ----
  {
    double qa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qa[t] = S.Q[4 * t + q][r];     // A operand: A[m=a][k=b] = Q[b][a]
    int64_t orow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = q + 4 * e;                                  // D row = (l>>4) + 4*reg
      orow[e] = (int64_t)((a < kJB) ? (ba * kJB + a) : (bb * kJB + (a - kJB))) * ld;
    }
    const int ntile = ld >> 4;
    for (int it = wave; it < ntile; it += 4) {
      const int col = it * 16 + r;
      double4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], Xs[(4 * t + q) * ldp + col], acc, 0, 0, 0);   // B[k=b][n=i]
#pragma unroll
      for (int e = 0; e < 4; ++e) XT[orow[e] + col] = acc[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tick1, streamed: the pair kernel for problems whose 16 columns do not fit the LDS (ld > 1152: the
// 4096-wide classifier layers of the VGG tables).  Same tournament, same inner solve, same descriptors
// and block map as jacobi_tick_kernel (self_mode = 0); the columns pass through the LDS in chunks of
// kStreamChunk rows twice -- once for H = Xp^T Xp (the four waves keep their partial tiles in
// registers across chunks), once for Xp <- Xp Q (second read mostly from L2 / Infinity Cache).
// Dynamic LDS (doubles): X[16][kStreamChunk + 2] | PairScratch
// ------------------------------------------------------------------------------------------------
constexpr int kStreamChunk = 1024;

// LDS-DMA, 1 KiB pieces, every piece of a wave in flight at once (128 KiB per workgroup): the pass is bound by
// what one CU pulls from L2 / Infinity Cache, not by a handful of register loads per thread
__device__ __forceinline__ void stream_load_chunk(double* __restrict__ Xs, const double* __restrict__ XT, int ld,
                                                  int ba, int bb, int c0, int len, int wave, int lane) {
  constexpr int ldp = kStreamChunk + 2;
  const int cpr = len >> 7;                        // 1 KiB pieces per row of the chunk (len % 128 == 0)
  const int npiece = kPair * cpr;
  for (int c = wave; c < npiece; c += 4) {
    const int row = c / cpr, ch = c - row * cpr;
    const int grow = (row < kJB) ? (ba * kJB + row) : (bb * kJB + (row - kJB));
    const double* src = XT + (int64_t)grow * ld + c0 + ch * 128 + lane * 2;
    double* dst = Xs + row * ldp + ch * 128;       // wave-uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
}

__global__ __launch_bounds__(256) void jacobi_tick_stream_kernel(const EigDesc* __restrict__ descs,
                                                                 const BlockRef* __restrict__ map, int tick,
                                                                 double tol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  if (*d.done) return;
  const int nb = d.nb;
  const int steps = nb - 1;
  const int sweep = tick / steps;
  const int step = tick - sweep * steps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (step == 0 && sweep > 0) {
    if (d.off[(sweep - 1) & 1] < tol) {
      if (br.local == 0 && tid == 0) *d.done = 1;
      return;
    }
  }
  constexpr int ldp = kStreamChunk + 2;
  const int ld = d.ld;
  double* Xs = smem;
  const PairScratch S = carve_scratch(Xs + kPair * ldp);
  int ba, bb;
  rr_pair(nb, step, br.local, ba, bb);
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ XT = d.XT;

  // ---- 1. H = Xp^T Xp over all chunks ----
  double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int c0 = 0; c0 < ld; c0 += kStreamChunk) {
    const int len = min(kStreamChunk, ld - c0);    // multiple of 128 (eig_ld)
    if (c0) __syncthreads();                       // the previous chunk has been consumed
    stream_load_chunk(Xs, XT, ld, ba, bb, c0, len, wave, lane);
    __syncthreads();
    const int per = len >> 2;                      // rows of the chunk per wave (multiple of 8)
    const double* row = Xs + r * ldp;
    for (int i = wave * per; i < (wave + 1) * per; i += 8) {
      const double2_t v = *reinterpret_cast<const double2_t*>(row + i + 2 * q);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.y, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) S.red[wave * 256 + lane * 4 + e] = acc0[e] + acc1[e];
  if (step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;
  __syncthreads();
  pair_gram_reduce(S, tid);
  __syncthreads();
  if (wave == 0) {
    const double mx = pair_inner_solve_fast(S, lane, d.off[2], tol, step == 0, nullptr);
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]), (unsigned long long)__double_as_longlong(mx));
  }
  __syncthreads();
  if (!*S.rotated) return;

  // ---- 2. Xp <- Xp * Q chunk by chunk ----
  double qa[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qa[t] = S.Q[4 * t + q][r];
  int64_t orow[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int a = q + 4 * e;
    orow[e] = (int64_t)((a < kJB) ? (ba * kJB + a) : (bb * kJB + (a - kJB))) * ld;
  }
  const int last0 = ((ld - 1) / kStreamChunk) * kStreamChunk;     // the chunk still in the LDS
  for (int c0 = last0; c0 >= 0; c0 -= kStreamChunk) {             // backwards: the resident chunk first
    const int len = min(kStreamChunk, ld - c0);
    if (c0 != last0) {
      __syncthreads();
      stream_load_chunk(Xs, XT, ld, ba, bb, c0, len, wave, lane);
      __syncthreads();
    }
    const int ntile = len >> 4;
    for (int it = wave; it < ntile; it += 4) {
      const int col = it * 16 + r;
      double4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], Xs[(4 * t + q) * ldp + col], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) XT[orow[e] + c0 + col] = acc[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tick2: one workgroup (512 threads) = one super-pair of 2 x 16 columns, LDS resident.
// Dynamic LDS (doubles): X[32][ldp] | PairScratch[2]
// Sub-blocks of 8 rows in the slab: 0,1 = super-block A ; 2,3 = super-block B.
// ------------------------------------------------------------------------------------------------
constexpr int kSuper = 2 * kPair;   // 32 rows in the slab

#ifdef TADMM_STAMPS
#define STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && tid == 0) g_stamps[i] += clock64() - t0; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(512) void jacobi_tick2_kernel(const EigDesc* __restrict__ descs,
                                                           const BlockRef* __restrict__ map, int tick, double tol,
                                                           int inner_sweeps) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  if (*d.done) return;
  const int nbs = d.nb >> 1;                       // super-blocks of 16 columns
  const int steps = nbs - 1;
  const int sweep = tick / steps;
  const int step = tick - sweep * steps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave >> 2, wv = wave & 3, th = tid & 255;
  if (step == 0 && sweep > 0) {
    if (d.off[(sweep - 1) & 1] < tol) {
      if (br.local == 0 && tid == 0) *d.done = 1;
      return;
    }
  }
  const int ld = d.ld, ldp = ld + 2;
#ifdef TADMM_STAMPS
  const unsigned long long t0 = clock64();
  if (blockIdx.x == gridDim.x / 2 && tid == 0) g_stamps[31] += 1;
#endif
  double* Xs = smem;
  const PairScratch S = carve_scratch(Xs + kSuper * ldp + half * kPairScratchDoubles);

  int sa, sb;
  rr_pair(nbs, step, br.local, sa, sb);
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ XT = d.XT;

  // ---- load the 32 columns once ----
  {
    const int c2n = ld >> 1;
    const int total = kSuper * c2n;
#pragma unroll 4
    for (int idx = tid; idx < total; idx += 512) {
      const int row = idx / c2n, c2 = idx - row * c2n;
      const int grow = (row < kPair) ? (sa * kPair + row) : (sb * kPair + (row - kPair));
      const double2_t v = *reinterpret_cast<const double2_t*>(XT + (int64_t)grow * ld + 2 * c2);
      *reinterpret_cast<double2_t*>(Xs + row * ldp + 2 * c2) = v;
    }
  }
  __syncthreads();
  STAMP(0);
  if (step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;

  const double hmax = d.off[2];
  const int per = ld >> 2;
  double mxall = 0.0;
  int any_rot = 0;
  // rounds: (first tick of a sweep only) inside the super-blocks: (0,1) | (2,3) incl. the pairs inside
  //         each block of 8;   round 1: (0,2) | (1,3);   round 2: (0,3) | (1,2)
  const int first_round = (step == 0) ? 0 : 1;
  for (int round = first_round; round < 3; ++round) {
    int u, v;
    if (round == 0) { u = half ? 2 : 0; v = half ? 3 : 1; }
    else if (round == 1) { u = half ? 1 : 0; v = half ? 3 : 2; }
    else { u = half ? 1 : 0; v = half ? 2 : 3; }
    const int lrow_r = (r < kJB) ? (u * kJB + r) : (v * kJB + (r - kJB));
    pair_gram_partial(Xs, ldp, lrow_r, wv * per, (wv + 1) * per, q, S.red + wv * 256, lane);
    __syncthreads();
    STAMP(1 + 4 * round);
    pair_gram_reduce(S, th);
    __syncthreads();
    STAMP(2 + 4 * round);
    if (wv == 0) {
      const double mx = pair_inner_solve_fast(S, lane, hmax, tol, round == 0);
      mxall = fmax(mxall, mx);
    }
    __syncthreads();
    STAMP(3 + 4 * round);
    if (*S.rotated) {
      any_rot = 1;
      double qa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qa[t] = S.Q[4 * t + q][r];
      int rowk[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {       // k = 4t+q (B operand row) and D row q+4e share this map
        const int b = 4 * t + q;
        rowk[t] = ((b < kJB) ? (u * kJB + b) : (v * kJB + (b - kJB))) * ldp;
      }
      const int ntile = ld >> 4;
      for (int it = wv; it < ntile; it += 4) {
        const int col = it * 16 + r;
        double4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], Xs[rowk[t] + col], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) Xs[rowk[e] + col] = acc[e];   // in place: this tile is touched by this wave only
      }
    }
    __syncthreads();
    STAMP(4 + 4 * round);
  }
  if (wv == 0 && lane == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]), (unsigned long long)__double_as_longlong(mxall));
  // ---- store back (skipped when nothing rotated in this workgroup) ----
  any_rot = __syncthreads_or(any_rot);
  if (!any_rot) return;
  {
    const int c2n = ld >> 1;
    const int total = kSuper * c2n;
#pragma unroll 4
    for (int idx = tid; idx < total; idx += 512) {
      const int row = idx / c2n, c2 = idx - row * c2n;
      const int grow = (row < kPair) ? (sa * kPair + row) : (sb * kPair + (row - kPair));
      *reinterpret_cast<double2_t*>(XT + (int64_t)grow * ld + 2 * c2) =
          *reinterpret_cast<const double2_t*>(Xs + row * ldp + 2 * c2);
    }
  }
  STAMP(20);
}

// ------------------------------------------------------------------------------------------------
// tick3: super-pair kernel with carried self-Grams.
//
// Every 16-column super-block b carries its 16x16 self-Gram S_b = X_b^T X_b in global memory (EigDesc::sblk,
// refreshed exactly once per sweep by the self kernel).  A launch on the super-pair (A, B) then needs only the
// CROSS Gram C = X_A^T X_B -- one 16x16 MFMA tile accumulated over all rows (128 MFMAs instead of the 512 of
// four sub-pair Grams) -- to assemble the full 32x32 Gram  H32 = [S_A C; C^T S_B]  in LDS.  Both rounds of
// sub-pair solves work on 16x16 sub-blocks of H32; between the rounds H32 is transformed by the round-1
// rotations (H32 <- Q1^T H32 Q1, small LDS arithmetic), so the round-2 solves do NOT need the updated columns:
// they run on waves 0 and 4 WHILE the other six waves apply Q1 to the 32 columns on the matrix cores.
// Critical path per launch: load -> cross Gram -> solve 1 -> solve 2 (|| apply 1) -> apply 2 -> store.
//
// Dynamic LDS (doubles): X[32][ldp] | R[2048] (cross-Gram partials, then H32[32][34]) |
//                        per half: H[16][18] Q1[16][18] Q2[16][18] | flags
// ------------------------------------------------------------------------------------------------
constexpr int kH32 = kSuper + 2;    // leading dimension of the 32x32 LDS image

__device__ __forceinline__ int sub_index(int u, int v, int k) { return (k < kJB) ? (u * kJB + k) : (v * kJB + (k - kJB)); }

// (Q is passed TRANSPOSED, as the inner solve leaves it.)
// X[:, sub-pair (u,v)] <- X[:, (u,v)] * Q  for the tiles it = wi, wi+nw, ...   (rows of the transposed slab)
__device__ __forceinline__ void apply_q_tiles(double* __restrict__ Xs, int ldp, int ld, const double (*Q)[kHP], int u,
                                              int v, int wi, int nw, int lane) {
  const int r = lane & 15, q = lane >> 4;
  double qa[4];
  int rowk[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qa[t] = Q[r][4 * t + q];                                  // A operand: A[m=a][k=b] = Q[b][a] = Qt[a][b]
    rowk[t] = sub_index(u, v, 4 * t + q) * ldp;               // k = 4t+q (B operand row) and D row q+4e share this map
  }
  const int ntile = ld >> 4;
  for (int it = wi; it < ntile; it += nw) {
    const int col = it * 16 + r;
    double4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], Xs[rowk[t] + col], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) Xs[rowk[e] + col] = acc[e];   // in place: a tile is touched by one wave only
  }
}

// Same product, result written to the global column image (XT[j][:] = column j) instead of LDS: 16 lanes write
// 128 contiguous bytes of one column.
__device__ __forceinline__ void apply_q_tiles_store(const double* __restrict__ Xs, int ldp, int ld, const double (*Q)[kHP],
                                                    int u, int v, int wi, int nw, int lane, double* __restrict__ XT, int sa,
                                                    int sb) {
  const int r = lane & 15, q = lane >> 4;
  double qa[4];
  int rowk[4];
  int64_t growk[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qa[t] = Q[r][4 * t + q];
    const int row = sub_index(u, v, 4 * t + q);
    rowk[t] = row * ldp;
    growk[t] = (int64_t)((row < kPair) ? (sa * kPair + row) : (sb * kPair + (row - kPair))) * ld;
  }
  const int ntile = ld >> 4;
  for (int it = wi; it < ntile; it += nw) {
    const int col = it * 16 + r;
    double4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], Xs[rowk[t] + col], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) __hip_atomic_store(XT + growk[e] + col, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Convergence measure of the 16x16 subproblem (u,v) read from the 32x32 image (same formula as the preamble of
// `pair_inner_solve_fast`); executed by a wave that is not solving, off the critical path.
__device__ __forceinline__ double pair_measure_h32(const double (*H32)[kH32], int u, int v, int lane, double hmax,
                                                   double tol) {
  const double inv_hmax = hmax > 0.0 ? 1.0 / hmax : 1.0;
  const float wscale = (float)(1e-14 / tol);
  float mxf = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = lane * 4 + k;
    const int i = e >> 4, j = e & 15;
    if (i < j) {
      const int gi = sub_index(u, v, i), gj = sub_index(u, v, j);
      const float ri = (float)(H32[gi][gi] * inv_hmax), rj = (float)(H32[gj][gj] * inv_hmax);
      const float rmin = fminf(ri, rj);
      if (rmin > 1e-28f) {
        const float hij = fabsf((float)(H32[gi][gj] * inv_hmax));
        const float w = fmaxf(1.0f, wscale * __builtin_amdgcn_rsqf(rmin));
        mxf = fmaxf(mxf, hij * __builtin_amdgcn_rsqf(ri) * __builtin_amdgcn_rsqf(rj) * __builtin_amdgcn_rcpf(w));
      }
    }
  }
  return (double)wave_max_f32(mxf);
}

// H32 <- Q^T H32 Q with Q = the two 16x16 rotations Qa (on the index set of sub-pair (ua,va)) and Qb (on (ub,vb)).
// In the permuted coordinates [idx_a | idx_b] Q is block diagonal, so each of the two products is four
// independent 16x16x16 tile products: waves 0-3 take one tile each (4 MFMAs), operands straight from LDS.
// Qa / Qb are passed TRANSPOSED (Qt[a][b] = Q[b][a]).
template <int ua, int va, int ub, int vb>
__device__ __forceinline__ void transform_h32(double (*H32)[kH32], const double (*Qa)[kHP], const double (*Qb)[kHP],
                                              int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int I = (wave >> 1) & 1, J = wave & 1;                     // tile (I,J) for waves 0..3
  const double (*QI)[kHP] = I ? Qb : Qa;
  const double (*QJ)[kHP] = J ? Qb : Qa;
  auto idx = [](int blk, int k) { return blk ? sub_index(ub, vb, k) : sub_index(ua, va, k); };
  double4_t acc = {0, 0, 0, 0}, out = {0, 0, 0, 0};
  if (wave < 4) {
    // T = H32[I][J] * Q_J, then H32'[I][J] = Q_I^T * T.  The D layout of the first product (lane (r,kq) holds
    // T[kq+4e][r] in register e) IS the B-operand layout of the second (k-step t wants T[4t+kq][r]): e = t, so
    // the chained product needs no LDS round trip.
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(H32[idx(I, r)][idx(J, 4 * t + kq)], QJ[r][4 * t + kq], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) out = __builtin_amdgcn_mfma_f64_16x16x4f64(QI[r][4 * t + kq], acc[t], out, 0, 0, 0);
  }
  __syncthreads();                   // every tile has been read before any is overwritten
  if (wave < 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) H32[idx(I, kq + 4 * e)][idx(J, r)] = out[e];
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void jacobi_tick3_kernel(const EigDesc* __restrict__ descs,
                                                           const BlockRef* __restrict__ map, int tick, double tol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  const int nbs = d.nb >> 1;
  const int steps = nbs - 1;
  const int period = d.period > 0 ? d.period : steps;
  const int sweep = tick / period;
  const int step = tick - sweep * period;
  if (step >= steps) return;                     // idle tick of the group's schedule (this problem has fewer players)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave >> 2, wv = wave & 3, th = tid & 255;
  const int ld = d.ld, ldp = ld + 2;
#ifdef TADMM_STAMPS
  const unsigned long long t0 = clock64();
  if (blockIdx.x == gridDim.x / 2 && tid == 0) g_stamps[31] += 1;
#endif
  double* Xs = smem;
  double* R = Xs + kSuper * ldp;                                   // 2048 doubles
  double (*H32)[kH32] = reinterpret_cast<double (*)[kH32]>(R);     // aliases R after the reduction
  double* hb = R + 2048 + half * (3 * kPair * kHP);
  double (*Hs)[kHP] = reinterpret_cast<double (*)[kHP]>(hb);
  double (*Q1)[kHP] = Hs + kPair;
  double (*Q2)[kHP] = Q1 + kPair;
  double (*Q1o)[kHP] = reinterpret_cast<double (*)[kHP]>(R + 2048 + (1 - half) * (3 * kPair * kHP)) + kPair;
  double (*Q2o)[kHP] = Q1o + kPair;
  int* flags = reinterpret_cast<int*>(R + 2048 + 2 * (3 * kPair * kHP));   // [0..1] rotated round 1, [2..3] round 2

  int sa, sb;
  rr_pair(nbs, step, br.local, sa, sb);
  const int r = lane & 15, q = lane >> 4;
  double* __restrict__ XT = d.XT;

  // ---- load the 32 columns ----
  const bool dma = (ld & 127) == 0;
  if (dma) {
    // LDS-DMA (global_load_lds_dwordx4): a wave moves 1 KiB chunks of a column straight into LDS -- 64 lanes x
    // 16 bytes land at consecutive LDS addresses behind a wave-uniform base, no staging registers, so all 16
    // chunks of a wave are in flight at once (the phase is latency-, not bandwidth-bound).  Rows are whole
    // 1 KiB chunks (ld % 128 == 0) and an LDS row starts 16 bytes after the previous one ends (ldp = ld + 2).
    // Issued before the convergence flags are even read: their load latency hides behind the columns'.
    const int cpr = ld >> 7;                          // chunks per column
    const int nchunk = kSuper * cpr;
    for (int c = wave; c < nchunk; c += 8) {
      const int row = c / cpr, ch = c - row * cpr;
      const int grow = (row < kPair) ? (sa * kPair + row) : (sb * kPair + (row - kPair));
      const double* src = XT + (int64_t)grow * ld + ch * 128 + lane * 2;
      double* dst = Xs + row * ldp + ch * 128;        // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  {
    bool quit = *d.done != 0;
    if (!quit && step == 0 && sweep > 0 && d.off[(sweep - 1) & 1] < tol) {
      if (br.local == 0 && tid == 0) *d.done = 1;
      quit = true;
    }
    if (quit) {                                       // uniform over the workgroup
      if (dma) __builtin_amdgcn_s_waitcnt(0);         // no DMA may still target this workgroup's LDS at exit
      return;
    }
  }
  // the two carried self-Grams go to registers
  const double sreg = d.sblk[(int64_t)(tid < 256 ? sa : sb) * (kPair * kPair) + th];
  if (dma) {
    __builtin_amdgcn_s_waitcnt(0);
  } else {
    constexpr int kLoadBatch = 8;
    // register-staged path: keep kLoadBatch x 16 bytes per thread in flight
    const int c2n = ld >> 1;
    const int total = kSuper * c2n;
    for (int base = tid; base < total; base += 512 * kLoadBatch) {
      double2_t v[kLoadBatch];
      int dst[kLoadBatch];
#pragma unroll
      for (int k = 0; k < kLoadBatch; ++k) {
        const int idx = base + k * 512;
        const bool ok = idx < total;
        const int row = ok ? idx / c2n : 0, c2 = ok ? idx - row * c2n : 0;
        const int grow = (row < kPair) ? (sa * kPair + row) : (sb * kPair + (row - kPair));
        dst[k] = ok ? row * ldp + 2 * c2 : -1;
        v[k] = *reinterpret_cast<const double2_t*>(XT + (int64_t)grow * ld + 2 * c2);
      }
#pragma unroll
      for (int k = 0; k < kLoadBatch; ++k)
        if (dst[k] >= 0) *reinterpret_cast<double2_t*>(Xs + dst[k]) = v[k];
    }
  }
  __syncthreads();
  STAMP(0);
  if (step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;

  // ---- cross Gram C = X_A^T X_B : every wave reduces ld/8 rows ----
  {
    const int per = ld >> 3;                                       // ld % 128 == 0 -> per % 16 == 0
    const double* ra = Xs + r * ldp + 2 * q;
    const double* rb = Xs + (kPair + r) * ldp + 2 * q;
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int i = wave * per; i < (wave + 1) * per; i += 8) {
      const double2_t va = *reinterpret_cast<const double2_t*>(ra + i);
      const double2_t vb = *reinterpret_cast<const double2_t*>(rb + i);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va.x, vb.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(va.y, vb.y, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) R[wave * 256 + lane * 4 + e] = acc0[e] + acc1[e];
  }
  __syncthreads();
  double cval = 0.0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < 8; ++w) cval += R[w * 256 + tid];
  }
  __syncthreads();                                                  // R is dead from here: H32 takes its place
  if (tid < 256) {
    const int l = tid >> 2, reg = tid & 3;
    const int hr = (l >> 4) + 4 * reg, hc = l & 15;                 // D layout: C[a = hr][b = hc]
    H32[hr][kPair + hc] = cval;
    H32[kPair + hc][hr] = cval;
    H32[th >> 4][th & 15] = sreg;                                   // S_A
  } else {
    H32[kPair + (th >> 4)][kPair + (th & 15)] = sreg;               // S_B
  }
  __syncthreads();
  STAMP(1);

  const double hmax = d.off[2];
  double mxall = 0.0;
  const int mwave = (half + 2) & 3;                                 // the measuring wave of this half
  // ---- round 1: (0,2) | (1,3) ----
  {
    const int u = half ? 1 : 0, v = half ? 3 : 2;
    Hs[th >> 4][th & 15] = H32[sub_index(u, v, th >> 4)][sub_index(u, v, th & 15)];
    Q1[th >> 4][th & 15] = ((th >> 4) == (th & 15)) ? 1.0 : 0.0;
    __syncthreads();
    if (wv == half) {      // waves 0 and 5: different SIMDs (waves 0 and 4 would share one)
      PairScratch S;
      S.H = Hs; S.Q = Q1; S.rotated = flags + half;
      pair_inner_solve_fast<true, false>(S, lane, hmax, tol, false);
    } else if (wv == mwave) {   // waves 2 and 7 (SIMDs 2 and 3): the convergence measure, off the critical path
      mxall = pair_measure_h32(H32, u, v, lane, hmax, tol);
    }
    __syncthreads();
  }
  STAMP(2);
  // ---- H32 <- Q1^T H32 Q1 ; extract round 2: (0,3) | (1,2) ----
  {
    double (*Qa)[kHP] = half ? Q1o : Q1;       // rotation of sub-pair (0,2)
    double (*Qb)[kHP] = half ? Q1 : Q1o;       // rotation of sub-pair (1,3)
    transform_h32<0, 2, 1, 3>(H32, Qa, Qb, tid);
    const int u = half ? 1 : 0, v = half ? 2 : 3;
    Hs[th >> 4][th & 15] = H32[sub_index(u, v, th >> 4)][sub_index(u, v, th & 15)];
    Q2[th >> 4][th & 15] = ((th >> 4) == (th & 15)) ? 1.0 : 0.0;
    __syncthreads();
  }
  STAMP(3);
  // ---- round-2 solves on waves 0 and 4, round-1 column update on the other six waves ----
  if (wv == half) {
    PairScratch S;
    S.H = Hs; S.Q = Q2; S.rotated = flags + 2 + half;
    __builtin_amdgcn_s_setprio(3);        // the solve is the critical path; the updater waves have slack
    pair_inner_solve_fast<true, false>(S, lane, hmax, tol, false);
    __builtin_amdgcn_s_setprio(0);
  } else {
    if (wv == mwave) mxall = fmax(mxall, pair_measure_h32(H32, half ? 1 : 0, half ? 2 : 3, lane, hmax, tol));
    const int u = half ? 1 : 0, v = half ? 3 : 2;                   // this half's round-1 sub-pair
    const int wi = (wv > half) ? wv - 1 : wv;                       // 0..2 among the three non-solver waves
    if (flags[half]) apply_q_tiles(Xs, ldp, ld, Q1, u, v, wi, 3, lane);
  }
  __syncthreads();
  STAMP(4);
  // ---- final self-Grams, then the round-2 column update straight from the MFMA accumulators to HBM ----
  if (wv == mwave && lane == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]), (unsigned long long)__double_as_longlong(mxall));
  const int any_rot = flags[0] | flags[1] | flags[2] | flags[3];
  if (!any_rot) return;
  {
    double (*Qa)[kHP] = half ? Q2o : Q2;       // (0,3)
    double (*Qb)[kHP] = half ? Q2 : Q2o;       // (1,2)
    transform_h32<0, 3, 1, 2>(H32, Qa, Qb, tid);
  }
  STAMP(5);
  d.sblk[(int64_t)(tid < 256 ? sa : sb) * (kPair * kPair) + th] =
      (tid < 256) ? H32[th >> 4][th & 15] : H32[kPair + (th >> 4)][kPair + (th & 15)];
  {
    // the two halves' round-2 sub-pairs (0,3) and (1,2) cover all 32 columns; a sub-pair that was not rotated
    // in round 2 has Q2 = I and the product is an exact copy of what round 1 left in LDS
    const int u = half ? 1 : 0, v = half ? 2 : 3;
    apply_q_tiles_store(Xs, ldp, ld, Q2, u, v, wv, 4, lane, XT, sa, sb);
  }
  STAMP(20);
}

// ------------------------------------------------------------------------------------------------
// Small problems (Npad <= 64, i.e. at most 8 blocks of 8 columns): the WHOLE eigen-solve in one launch.
// One 4-wave workgroup per problem keeps all columns in LDS; a round of the tournament has at most four
// block pairs, one per wave (Gram tile, 16x16 inner solve, column update -- no cross-wave traffic inside a
// round, one barrier between rounds), and the convergence rule of `jacobi_conv_kernel` is evaluated by the
// workgroup itself, so there is no host poll and no kernel boundary per round.  Used for the first / last TT
// steps (N <= 32 on the ResNet tables) and for every eigen-problem of the CIFAR Tucker tables.
// verdict: pinned host ints, [1 + p] = problem p converged.
// ------------------------------------------------------------------------------------------------
constexpr int kSmallNpad = 64;
constexpr int kSmallWaveScratch = 2 * kPair * kHP;     // H and Q of one wave

__global__ __launch_bounds__(256) void jacobi_small_kernel(const EigDesc* __restrict__ descs,
                                                           const int32_t* __restrict__ skip, double tol,
                                                           int max_sweeps, int* __restrict__ verdict,
                                                           const int32_t* __restrict__ fast_done) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int p = blockIdx.x;
  if (fast_done && fast_done[p]) return;     // solved (and verified) by eig_small_direct_kernel (tridiag.hip)
  const EigDesc d = descs[p];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (skip && skip[p]) {
    if (tid == 0) { *d.done = 1; verdict[1 + p] = 1; }
    return;
  }
  const int Npad = d.Npad, ld = d.ld, ldp = Npad + 2, nb = d.nb;
  const int pairs = nb >> 1, steps = nb - 1;
  double* Xs = smem;                                                   // [Npad][ldp], row j = column j of X
  double* wbase = Xs + Npad * ldp + wave * kSmallWaveScratch;
  double (*Hs)[kHP] = reinterpret_cast<double (*)[kHP]>(wbase);
  double (*Qs)[kHP] = Hs + kPair;
  double* red = Xs + Npad * ldp + 4 * kSmallWaveScratch;               // [8]
  int* flags = reinterpret_cast<int*>(red + 8);                        // [4]
  double* __restrict__ XT = d.XT;
  {
    const int c2n = Npad >> 1;
    for (int idx = tid; idx < Npad * c2n; idx += 256) {
      const int row = idx / c2n, c2 = idx - row * c2n;
      *reinterpret_cast<double2_t*>(Xs + row * ldp + 2 * c2) =
          *reinterpret_cast<const double2_t*>(XT + (int64_t)row * ld + 2 * c2);
    }
  }
  __syncthreads();
  // scale reference, as jacobi_init_kernel: (largest diagonal entry of G)^2 bounds the largest squared column norm
  {
    float g = 0.0f;
    for (int j = tid; j < d.N; j += 256) g = fmaxf(g, fabsf((float)Xs[j * ldp + j]));
    g = wave_max_f32(g);
    if (lane == 0) red[wave] = (double)g;
  }
  __syncthreads();
  const double g0 = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double hmax = g0 * g0;
  __syncthreads();
  // Warm start (HOOI: the same mode's Gram changes little from one sweep to the next).  One-sided Jacobi on X = G V0 with
  // V0 the eigenvectors of the previous solve ends in G V0 V = G (V0 V): the same eigenpairs, but X starts out with
  // nearly orthogonal columns and the solve is in its quadratic phase from the first sweep.  X0 = G V0 is formed here
  // in LDS (row j of X0 = sum_k V0[j][k] * row k of G; G is symmetric), into the second image behind the scratch.
  if (d.warm && *d.warm_ok) {
    double* X0 = red + 16;                                             // [Npad][ldp]
    const int j = tid >> 2, per = Npad >> 2, i0 = (tid & 3) * per;     // per = 8 (Npad 32) or 16 (Npad 64)
    if (j < Npad) {
      double acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.0;
      const double* __restrict__ vj = d.warm + (int64_t)j * Npad;
      for (int k = 0; k < Npad; ++k) {
        const double a = vj[k];
        const double* g = Xs + k * ldp + i0;
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < per) acc[c] += a * g[c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < per) X0[j * ldp + i0 + c] = acc[c];
    }
    __syncthreads();
    for (int idx = tid; idx < Npad * ldp; idx += 256) Xs[idx] = X0[idx];
    __syncthreads();
  }
  const int r = lane & 15, q = lane >> 4;
  double prev_m = 0.0, m = 1.0;
  bool conv = false;
  int sweep = 0;
  for (; sweep < max_sweeps && !conv; ++sweep) {
    double mx_sweep = 0.0;
    for (int step = 0; step < steps; ++step) {
      if (wave < pairs) {
        int a, b;
        rr_pair(nb, step, wave, a, b);
        const double* row = Xs + sub_index(a, b, r) * ldp;
        double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int i = 0; i < Npad; i += 8) {
          const double2_t v = *reinterpret_cast<const double2_t*>(row + i + 2 * q);
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.y, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Hs[q + 4 * e][r] = acc0[e] + acc1[e];
          Qs[q + 4 * e][r] = (q + 4 * e == r) ? 1.0 : 0.0;
        }
        wave_lds_fence();
        PairScratch S;
        S.H = Hs; S.Q = Qs; S.rotated = flags + wave;
        mx_sweep = fmax(mx_sweep, pair_inner_solve_fast<true, true>(S, lane, hmax, tol, step == 0));
        wave_lds_fence();
        if (flags[wave]) apply_q_tiles(Xs, ldp, Npad, Qs, a, b, 0, 1, lane);
      }
      __syncthreads();
    }
    if (lane == 0) red[wave] = mx_sweep;
    __syncthreads();
    m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    conv = m < tol;
    if (!conv && prev_m > 0.0 && prev_m < 1e-1 && m < 1e-3) {          // quadratic phase: see jacobi_conv_kernel
      const double C = 10.0 * fmax(1.0, m / (prev_m * prev_m));
      conv = C * m * m < 10.0 * tol;
    }
    prev_m = m;
    __syncthreads();
  }
  {
    const int c2n = Npad >> 1;
    for (int idx = tid; idx < Npad * c2n; idx += 256) {
      const int row = idx / c2n, c2 = idx - row * c2n;
      *reinterpret_cast<double2_t*>(XT + (int64_t)row * ld + 2 * c2) =
          *reinterpret_cast<const double2_t*>(Xs + row * ldp + 2 * c2);
    }
  }
  if (d.warm) {
    // eigenvectors for the next solve of this problem: the normalised columns of the converged X.  One-sided Jacobi leaves
    // them orthogonal to ~tol RELATIVE to their own norms, so small columns are as good as large ones until rounding takes
    // over (column norm below 1e-8 of the largest): then, or without convergence, the next solve starts cold.
    __shared__ int warm_bad;
    if (tid == 0) warm_bad = conv ? 0 : 1;
    __syncthreads();
    const double floor2 = 1e-16 * hmax;
    for (int jrow = wave; jrow < Npad; jrow += 4) {
      const double* row = Xs + jrow * ldp;
      double s = 0.0;
      for (int i = lane; i < Npad; i += 64) s += row[i] * row[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      double inv = 0.0;
      if (jrow < d.N) {
        if (s > floor2 && s > 0.0) inv = 1.0 / sqrt(s);
        else if (lane == 0) warm_bad = 1;
      }
      for (int i = lane; i < Npad; i += 64) d.warm[(int64_t)jrow * Npad + i] = row[i] * inv;
    }
    __syncthreads();
    if (tid == 0) *d.warm_ok = warm_bad ? 0 : 1;
  }
  if (tid == 0) {
    *d.done = conv ? 1 : 0;
    d.off[0] = m; d.off[1] = (double)sweep; d.off[2] = hmax;
    verdict[1 + p] = conv ? 1 : 0;
  }
}

bool jacobi_small_fits(int npad_max) { return npad_max <= kSmallNpad; }

void launch_jacobi_small(const EigDesc* descs_dev, int nprob, int npad_max, double tol, int max_sweeps,
                         const int32_t* skip, int* verdict_pinned, hipStream_t s, bool warm, const int32_t* fast_done) {
  if (nprob <= 0) return;
  // X image | 4 wave scratches | red[8] + flags (8 doubles) | second image (warm start only)
  const size_t lds = ((size_t)npad_max * (npad_max + 2) * (warm ? 2 : 1) + 4 * kSmallWaveScratch + 16) * 8;
  if (lds > 64 * 1024) {                     // two images of a 64-column problem: above the default dynamic-LDS cap
    static bool attr_done[64] = {false};     // per device: the attribute belongs to the device's code object
    int devi = 0;
    (void)hipGetDevice(&devi);
    if (!attr_done[devi & 63]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_small_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();
      attr_done[devi & 63] = true;
    }
  }
  hipLaunchKernelGGL(jacobi_small_kernel, dim3(nprob), dim3(256), lds, s, descs_dev, skip, tol, max_sweeps,
                     verdict_pinned, fast_done);
}

#ifdef TADMM_STAMPS
void dump_stamps() {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) != hipSuccess) return;
  const double n = h[31] ? (double)h[31] : 1.0;
  fprintf(stderr, "[stampsS] per launch (2 solves): preamble=%.0f loop=%.0f epilogue=%.0f\n", h[10] / n, h[11] / n, h[12] / n);
  fprintf(stderr, "[stamps3] launches=%llu cumulative: load=%.0f gram=%.0f solve1=%.0f xform=%.0f solve2||apply1=%.0f apply2+xform=%.0f end=%.0f\n",
          h[31], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[20] / n);
  fprintf(stderr, "[stamps] launches=%llu  cumulative cycles/launch: load=%.0f", h[31], h[0] / n);
  for (int r = 0; r < 3; ++r)
    fprintf(stderr, " | r%d gram=%.0f red=%.0f inner=%.0f upd=%.0f", r, h[1 + 4 * r] / n, h[2 + 4 * r] / n, h[3 + 4 * r] / n, h[4 + 4 * r] / n);
  fprintf(stderr, " | end=%.0f\n", h[20] / n);
}
#else
void dump_stamps() {}
#endif

// ---- finalize: eigenvalues = column norms, descending order, scaled eigenvectors ----
__global__ __launch_bounds__(256) void eig_norms_kernel(const EigDesc* __restrict__ descs,
                                                        const BlockRef* __restrict__ map,
                                                        const int32_t* __restrict__ skip) {
  const BlockRef br = map[blockIdx.x];
  if (skip && skip[br.prob]) return;
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

__global__ __launch_bounds__(256) void eig_sort_kernel(const EigDesc* __restrict__ descs,
                                                       const int32_t* __restrict__ skip) {
  extern __shared__ double slam[];
  if (skip && skip[blockIdx.x]) return;
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

#ifndef TADMM_RESIDUE_CUT
#define TADMM_RESIDUE_CUT 1e-12
#endif
constexpr double kResidueCut = TADMM_RESIDUE_CUT;     // eigenvalue (= sigma^2) below this fraction of the largest: residue
__global__ __launch_bounds__(256) void eig_extract_kernel(const EigDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map,
                                                          const int32_t* __restrict__ skip) {
  const BlockRef br = map[blockIdx.x];
  if (skip && skip[br.prob]) return;
  const EigDesc d = descs[br.prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = br.local * 4 + wave;
  if (c >= d.r) return;
  const int j = d.order[c];
  double lam = d.lam[j];
  // Numerically rank-deficient input: a column whose eigenvalue is at the rounding floor of the largest one is
  // rounding residue -- its direction is NOT orthogonal to the genuine eigenvectors, so keeping it would count
  // their energy twice.  Such directions carry nothing (sigma <= 1e-6 sigma_max): emit a zero vector instead
  // (the reference's LAPACK pads with an arbitrary orthonormal completion; Z is unaffected).
  if (lam <= kResidueCut * d.lam[d.order[0]]) lam = 0.0;
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
  const int r = d.ldo ? d.ldo : d.r, N = d.N;
  for (int i = lane; i < N; i += 64) {
    const double v = row[i] * inv;
    if (d.mode == 0) {
      d.out_a[(int64_t)i * r + c] = (float)v;
    } else if (d.mode == 1) {
      d.out_a[(int64_t)i * r + c] = (float)(v * isig);
      d.out_b[(int64_t)c * N + i] = (float)(v * sig);
    } else if (d.mode == 3) {
      d.out_a[(int64_t)i * r + c] = (float)(v * isig);
    }
    if (d.evec_out) d.evec_out[(int64_t)c * N + i] = v;
  }
}

// Convergence decision after a global sweep, on the device (one workgroup): a problem whose own sweep just ended
// (tick % steps == 0) is finished when that sweep saw nothing above tol BEFORE rotating, or -- quadratic phase --
// when the C*m^2 it leaves behind (C estimated from the last two sweeps, x10 safety) is below 10*tol, so no
// sweep is spent just to observe it (post-sweep target 1e-8: eigenvector errors stay << 1e-5).  Finished
// problems get their sticky `done` flag, which every later tick launch honours; *all_done tells the host.
__global__ __launch_bounds__(256) void jacobi_conv_kernel(const EigDesc* __restrict__ descs, int nprob, int tick,
                                                          double tol, int super, double* __restrict__ prev,
                                                          int* __restrict__ verdict) {
  // verdict: pinned host memory, [0] = all finished, [1 + q] = problem q finished (visible to the host once the
  // event recorded behind this launch has completed)
  __shared__ int open_problems;
  if (threadIdx.x == 0) open_problems = 0;
  __syncthreads();
  for (int q = threadIdx.x; q < nprob; q += 256) {
    const EigDesc d = descs[q];
    if (*d.done) { verdict[1 + q] = 1; continue; }
    const int own = (super ? (d.nb >> 1) : d.nb) - 1;
    const int steps = (own > 0 && d.period > 0) ? d.period : own;
    bool conv = false;
    if (steps > 0 && tick % steps == 0) {
      const int swp = tick / steps - 1;
      const double m = d.off[swp & 1];
      conv = m < tol;
      const double mp = prev[q];
      if (!conv && mp > 0.0 && mp < 1e-1 && m < 1e-3) {
        const double C = 10.0 * fmax(1.0, m / (mp * mp));
        conv = C * m * m < 10.0 * tol;
      }
      prev[q] = m;
    }
    if (conv) *d.done = 1;
    else open_problems = 1;
    verdict[1 + q] = conv ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) verdict[0] = open_problems ? 0 : 1;
}

void launch_jacobi_conv(const EigDesc* descs_dev, int nprob, int tick, double tol, bool super, double* prev_dev,
                        int* verdict_pinned, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(jacobi_conv_kernel, dim3(1), dim3(256), 0, s, descs_dev, nprob, tick, tol, super ? 1 : 0, prev_dev,
                     verdict_pinned);
}

void launch_jacobi_init(const EigDesc* descs_dev, int nprob, hipStream_t s, const int32_t* skip, double* prev_dev) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(jacobi_init_kernel, dim3(nprob), dim3(256), 0, s, descs_dev, skip, prev_dev);
}

size_t jacobi_tick_stream_lds_bytes() { return ((size_t)kPair * (kStreamChunk + 2) + kPairScratchDoubles) * 8; }
bool jacobi_size_supported(int n) { return n >= 1 && n <= kJacobiMaxN; }
size_t jacobi_tick_lds_bytes(int ld_max) { return ((size_t)kPair * (ld_max + 2) + kPairScratchDoubles) * 8; }
size_t jacobi_tick3_lds_bytes(int ld_max) {
  return ((size_t)kSuper * (ld_max + 2) + 2048 + 2 * (3 * kPair * kHP)) * 8 + 16;
}
bool jacobi_tick3_fits(int ld_max) { return jacobi_tick3_lds_bytes(ld_max) <= 160 * 1024; }
void launch_jacobi_tick3(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                         int ld_max, hipStream_t s) {
  if (nblocks <= 0) return;
  static bool attr_done[64] = {false};       // per device: the attribute belongs to the device's code object
  int devi = 0;
  (void)hipGetDevice(&devi);
  bool& attr_set = attr_done[devi & 63];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_tick3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(jacobi_tick3_kernel, dim3(nblocks), dim3(512), jacobi_tick3_lds_bytes(ld_max), s, descs_dev, map_dev,
                     tick, tol);
  const hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess)
    fprintf(stderr, "[tadmm] jacobi tick3 launch failed: %s (blocks=%d lds=%zu)\n", hipGetErrorString(e), nblocks,
            jacobi_tick3_lds_bytes(ld_max));
}
size_t jacobi_tick2_lds_bytes(int ld_max) { return ((size_t)kSuper * (ld_max + 2) + 2 * kPairScratchDoubles) * 8; }
bool jacobi_tick2_fits(int ld_max) { return jacobi_tick2_lds_bytes(ld_max) <= 160 * 1024 - 256; }

void launch_jacobi_self(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                        int inner_sweeps, int ld_max, hipStream_t s) {
  if (nblocks <= 0) return;
  static bool attr_done[64] = {false};
  int devi = 0;
  (void)hipGetDevice(&devi);
  bool& attr_set = attr_done[devi & 63];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_tick_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(jacobi_tick_kernel, dim3(nblocks), dim3(256), jacobi_tick_lds_bytes(ld_max), s, descs_dev, map_dev,
                     tick, tol, inner_sweeps, 1);
}

void launch_jacobi_tick(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                        int inner_sweeps, size_t lds_bytes, bool super, hipStream_t s) {
  if (nblocks <= 0) return;
  static bool attr_done[64] = {false};
  int devi = 0;
  (void)hipGetDevice(&devi);
  bool& attr_set = attr_done[devi & 63];
  if (!attr_set) {   // allow the full 160 KiB of a CU as dynamic LDS (default cap is 64 KiB)
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_tick_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // tick2 carries 256 bytes of static LDS: asking for the full 160 KiB as dynamic LDS is refused (invalid argument)
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_tick2_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_tick_stream_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess || e2 != hipSuccess)
      fprintf(stderr, "[tadmm] hipFuncSetAttribute(max dynamic LDS): %s / %s\n", hipGetErrorString(e1),
              hipGetErrorString(e2));
    (void)hipGetLastError();
    attr_set = true;
  }
  if (super)
    hipLaunchKernelGGL(jacobi_tick2_kernel, dim3(nblocks), dim3(512), lds_bytes, s, descs_dev, map_dev, tick, tol,
                       inner_sweeps);
  else if (lds_bytes > 160 * 1024)      // rows too long for an LDS-resident pair: streamed pair kernel
    hipLaunchKernelGGL(jacobi_tick_stream_kernel, dim3(nblocks), dim3(256), jacobi_tick_stream_lds_bytes(), s, descs_dev,
                       map_dev, tick, tol);
  else
    hipLaunchKernelGGL(jacobi_tick_kernel, dim3(nblocks), dim3(256), lds_bytes, s, descs_dev, map_dev, tick, tol,
                       inner_sweeps, 0);
  hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess)
    fprintf(stderr, "[tadmm] jacobi tick launch failed: %s (blocks=%d lds=%zu super=%d)\n", hipGetErrorString(e),
            nblocks, lds_bytes, (int)super);
}
void launch_eig_norms(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                      const int32_t* skip) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(eig_norms_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
}
void launch_eig_sort(const EigDesc* descs_dev, int nprob, hipStream_t s, const int32_t* skip, int npad_max) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(eig_sort_kernel, dim3(nprob), dim3(256), (size_t)std::max(2048, (npad_max > 0 && npad_max <= kJacobiMaxN) ? npad_max : kJacobiMaxN) * 8, s, descs_dev, skip);
}
void launch_eig_extract(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                        const int32_t* skip) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(eig_extract_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
}

}  // namespace tadmm
