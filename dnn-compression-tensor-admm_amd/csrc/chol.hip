// Cholesky QR of a tall block on the fp64 matrix cores: the orthonormalisation step of the filtered
// eigen-solver (filter.hip).  Given the Gram matrix C = Y^T Y (n x n, n <= 256) of a block image
// YT[n][ldy] (row j = column j of Y):
//
//   chol_factor : C = R^T R.  One 512-thread workgroup per problem keeps all upper 16x16 tiles of C in
//                 registers (MFMA accumulator layout) and runs the right-looking block algorithm
//                    W_k = R_kk^{-T}  (16x16 Cholesky + triangular inverse, one wave, in LDS)
//                    R_kj = W_k C_kj                      (panel, one MFMA tile product each)
//                    C_ij -= R_ki^T R_kj   (k < i <= j)   (trailing update, operands from the LDS panel)
//                 A non-positive pivot (numerically rank deficient block) sets the problem's `bad` word.
//   chol_solve  : Q^T = R^{-T} Y^T in place, by block forward substitution.  A wave owns 16 columns of the
//                 image and keeps the solved 16x16 tiles in registers (the D layout of one product is the
//                 B-operand layout of the next), so it needs neither LDS nor any cross-wave traffic:
//                    X_kb = W_kb (Y_kb - sum_{j<kb} R_j,kb^T X_j).
// Used twice in a row ("CholQR2") the result is orthonormal to rounding for condition numbers up to ~1e7.
#include "common.h"

namespace tadmm {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kCT = 16;            // tile edge
constexpr int kCLd = kCT + 1;      // padded leading dimension of the LDS tiles
constexpr int kCMaxT = 16;         // n <= 256

__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);                    // ~2^-26 relative
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);                       // (second step kept: 16 pivots in a row feed one another)
  return y;
}

// W = L^-1 for the 16x16 SPD tile Dg = L L^T, one wave, blocked by 4 so that every rank-4 update is ONE
// v_mfma_f64_16x16x4 on a register-resident tile:
//   S (accumulator layout) starts as Dg; for block b: the 4x4 block S_bb is factored and inverted redundantly by
//   every lane (M = L_bb^-1, four dependent rsqrt), the block column P = S[:, b] M^T gives columns 4b..4b+3 of L,
//   and S -= P P^T eliminates them.  The inverse is carried along: R starts as I, X_b = M R[b, :] are rows 4b..4b+3 of
//   W, and R -= L[:, b] X_b -- again one MFMA, whose operands (P of this lane, X of this lane) are already in place.
// Returns false (uniformly) on a pivot <= tiny.
__device__ __forceinline__ bool diag_inverse(const double (*Dg)[kCLd], double (*Cb)[5], double (*Wt)[kCLd],
                                             double* __restrict__ Wg, int lane, double tiny) {
  const int r = lane & 15, q = lane >> 4;
  double4_t S, R;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    S[e] = Dg[q + 4 * e][r];
    R[e] = (q + 4 * e == r) ? 1.0 : 0.0;
  }
  bool ok = true;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    // block column 4b..4b+3 of S -> LDS (lanes whose column r lies in the block hold it, rows q + 4e)
    if ((r >> 2) == b) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Cb[q + 4 * e][r & 3] = S[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double s00 = Cb[4 * b][0], s10 = Cb[4 * b + 1][0], s20 = Cb[4 * b + 2][0], s30 = Cb[4 * b + 3][0];
    const double s11 = Cb[4 * b + 1][1], s21 = Cb[4 * b + 2][1], s31 = Cb[4 * b + 3][1];
    const double s22 = Cb[4 * b + 2][2], s32 = Cb[4 * b + 3][2], s33 = Cb[4 * b + 3][3];
    const double c0 = Cb[r][0], c1 = Cb[r][1], c2 = Cb[r][2], c3 = Cb[r][3];
    ok = ok && (s00 > tiny);
    const double i0 = rsqrt_f64(fmax(s00, tiny));
    const double l10 = s10 * i0, l20 = s20 * i0, l30 = s30 * i0;
    const double t11 = s11 - l10 * l10;
    ok = ok && (t11 > tiny);
    const double i1 = rsqrt_f64(fmax(t11, tiny));
    const double l21 = (s21 - l20 * l10) * i1, l31 = (s31 - l30 * l10) * i1;
    const double t22 = s22 - l20 * l20 - l21 * l21;
    ok = ok && (t22 > tiny);
    const double i2 = rsqrt_f64(fmax(t22, tiny));
    const double l32 = (s32 - l30 * l20 - l31 * l21) * i2;
    const double t33 = s33 - l30 * l30 - l31 * l31 - l32 * l32;
    ok = ok && (t33 > tiny);
    const double i3 = rsqrt_f64(fmax(t33, tiny));
    // M = L_bb^-1 (lower)
    const double m00 = i0, m11 = i1, m22 = i2, m33 = i3;
    const double m10 = -l10 * m00 * i1;
    const double m20 = -(l20 * m00 + l21 * m10) * i2, m21 = -l21 * m11 * i2;
    const double m30 = -(l30 * m00 + l31 * m10 + l32 * m20) * i3, m31 = -(l31 * m11 + l32 * m21) * i3, m32 = -l32 * m22 * i3;
    // P[r][q] = sum_{m <= q} S[r][4b+m] M[q][m]   (zero above the block: those rows are eliminated already)
    const double p0 = c0 * m00, p1 = c0 * m10 + c1 * m11, p2 = c0 * m20 + c1 * m21 + c2 * m22,
                 p3 = c0 * m30 + c1 * m31 + c2 * m32 + c3 * m33;
    double P = q == 0 ? p0 : (q == 1 ? p1 : (q == 2 ? p2 : p3));
    if (r < 4 * b) P = 0.0;
    S = __builtin_amdgcn_mfma_f64_16x16x4f64(-P, P, S, 0, 0, 0);
    // rows 4b..4b+3 of W: X[4b+q][r] = sum_m M[q][m] R[4b+m][r]; R[4b+m][r] is register b of lane (r, m)
    // (rows 4b..4b+3 of R are exchanged through the rows of Wt they are about to define)
    Wt[4 * b + q][r] = R[b];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double r0 = Wt[4 * b][r], r1 = Wt[4 * b + 1][r], r2 = Wt[4 * b + 2][r], r3 = Wt[4 * b + 3][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double x0 = m00 * r0, x1 = m10 * r0 + m11 * r1, x2 = m20 * r0 + m21 * r1 + m22 * r2,
                 x3 = m30 * r0 + m31 * r1 + m32 * r2 + m33 * r3;
    const double X = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
    Wt[4 * b + q][r] = X;
    Wg[(4 * b + q) * kCT + r] = X;
    R = __builtin_amdgcn_mfma_f64_16x16x4f64(-P, X, R, 0, 0, 0);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return ok;
}

#ifdef TADMM_CHOL_STAMPS
__device__ long long g_cstamps[128];
#define CSTAMP(i) do { if (blockIdx.x == 0 && lane == 0) g_cstamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif

// Barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores (vmcnt(0));
// the factor rows and inverse tiles this kernel writes to memory are never read back by it, and waiting for their
// write acknowledgements at three barriers per step cost ~3 k cycles of every 12 k-cycle step (scripts/stamp_chol.sh).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave 0 owns no tiles: it only inverts diagonal tiles, and does so for step k WHILE the other seven waves finish
// the trailing update of step k-1 (look-ahead: the owner of tile (k,k) updates it first and hands it over).
constexpr int kCTileWaves = 7;
constexpr int kCSlots7 = (kCMaxT * (kCMaxT + 1) / 2 + kCTileWaves - 1) / kCTileWaves;   // 20

__global__ __launch_bounds__(512) void chol_factor_kernel(const CholDesc* __restrict__ descs) {
  __shared__ double Dg[kCT][kCLd];
  __shared__ double Wt[kCT][kCLd];
  __shared__ double Cb[kCT][5];        // current 16x4 block column of the diagonal tile's Schur complement
  __shared__ double Pn[kCMaxT][kCT][kCLd];
  __shared__ double red[8];
  __shared__ unsigned char ti[kCMaxT * (kCMaxT + 1) / 2], tj[kCMaxT * (kCMaxT + 1) / 2];
  __shared__ int fail;
  const CholDesc d = descs[blockIdx.x];
  if (d.gate && *d.gate < d.gate_min) return;
  if (*d.bad) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: the per-slot tests below become branches
  const int r = lane & 15, q = lane >> 4;
  const int nbt = d.n / kCT;
  const int ntile = nbt * (nbt + 1) / 2;
  for (int t = tid; t < ntile; t += 512) {
    int i = 0, rem = t, rowlen = nbt;
    while (rem >= rowlen) { rem -= rowlen; ++i; --rowlen; }
    ti[t] = (unsigned char)i; tj[t] = (unsigned char)(i + rem);
  }
  if (tid == 0) fail = 0;
  double scale = 0.0;                   // largest diagonal entry of C: pivots are judged relative to it
  for (int i = tid; i < d.n; i += 512) scale = fmax(scale, d.C[(int64_t)i * d.ldc + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) scale = fmax(scale, __shfl_xor(scale, o, 64));
  if (lane == 0) red[wave] = scale;
  __syncthreads();
  scale = 0.0;
#pragma unroll
  for (int w = 0; w < 8; ++w) scale = fmax(scale, red[w]);
  const double tiny = scale * 1e-15;    // a pivot this small relative to the largest norm: numerically rank deficient

  if (wave == 0) {
    // ---------------- diagonal wave ----------------
    for (int k = 0; k < nbt; ++k) {
      lds_barrier();                                               // A_k: tile (k,k) is in Dg
      CSTAMP(3 * k);
      if (!diag_inverse(Dg, Cb, Wt, d.Wd + (int64_t)k * (kCT * kCT), lane, tiny) && lane == 0) fail = 1;
      CSTAMP(3 * k + 1);
      lds_barrier();                                               // B_k: W_k is in Wt (and trailing k-1 is complete)
      if (fail) { if (tid == 0) *d.bad = 1; return; }
      lds_barrier();                                               // C_k: panel k is in Pn
      CSTAMP(3 * k + 2);
    }
    return;
  }
  // ---------------- tile waves ----------------
  const int tw = wave - 1;
  double4_t acc[kCSlots7];
  // tile coordinates of this wave's slots, read from the LDS table ONCE (packed, one vector register per slot): looked
  // up per slot and step -- two dependent LDS reads in front of every branch -- they cost the panel phase ~3 k cycles of
  // a 12 k-cycle step.  WHICH slots take part in a step is scalar arithmetic on the tile index: tiles are numbered row
  // by row, so rows >= k are the indices >= tkk and row k is [tkk, tkk + nbt - k).
  int sij[kCSlots7];
  auto row_of = [](int v) { return v & 0xff; };
  auto col_of = [](int v) { return v >> 8; };
  // the LDS addresses derived from a slot's coordinates are loop invariant; hoisted out of the step loop for all 20 slots
  // they would cost 40 vector registers and spill.  The empty volatile asm makes the compiler rebuild them where used.
  auto fresh = [](int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; };
#pragma unroll
  for (int s = 0; s < kCSlots7; ++s) {
    const int t = kCTileWaves * s + tw;
    acc[s] = double4_t{0, 0, 0, 0};
    sij[s] = 0;
    if (t < ntile) {
      const int i = ti[t], j = tj[t];
      sij[s] = (j << 8) | i;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][e] = d.C[(int64_t)(kCT * i + q + 4 * e) * d.ldc + kCT * j + r];
    }
  }
  // trailing update of ONE slot with the panel of step k (tile (i,j), i > k)
  auto trail = [&](double4_t& a, int i, int j) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      a = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pn[i][q + 4 * e][r], Pn[j][q + 4 * e][r], a, 0, 0, 0);
  };
  // hand tile (k,k) to the diagonal wave
  auto give_diag = [&](int k) {
    const int tkk = k * nbt - (k * (k - 1)) / 2;
#pragma unroll
    for (int s = 0; s < kCSlots7; ++s)
      if (kCTileWaves * s + tw == tkk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Dg[q + 4 * e][r] = acc[s][e];
      }
  };
  give_diag(0);
  for (int k = 0; k < nbt; ++k) {
    lds_barrier();                                                 // A_k
    // trailing update of step k-1 for everything but tile (k,k), which was updated before it was handed over
    if (k > 0) {
      const int tkk = k * nbt - (k * (k - 1)) / 2;
#pragma unroll
      for (int s = 0; s < kCSlots7; ++s) {
        const int t = kCTileWaves * s + tw;
        if (t > tkk && t < ntile) { const int c = fresh(sij[s]); trail(acc[s], row_of(c), col_of(c)); }
        __builtin_amdgcn_sched_barrier(0);     // keep the operand loads of one slot from being hoisted over the others
      }
    }
    if (wave == 1) CSTAMP(64 + 2 * k);
    lds_barrier();                                                 // B_k
    if (fail) return;
    // ---- panel R_kj = W C_kj (j > k) ----
    {
      double wa[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) wa[e] = Wt[r][q + 4 * e];
#pragma unroll
      for (int s = 0; s < kCSlots7; ++s) {
        const int t = kCTileWaves * s + tw;
        const int tkk = k * nbt - (k * (k - 1)) / 2;
        if (t > tkk && t < tkk + nbt - k) {
          const int pj = col_of(fresh(sij[s]));
          double4_t o = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[e], acc[s][e], o, 0, 0, 0);
          acc[s] = o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            Pn[pj][q + 4 * e][r] = o[e];
            d.R[(int64_t)(kCT * k + q + 4 * e) * d.ldr + kCT * pj + r] = o[e];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (wave == 1) CSTAMP(65 + 2 * k);
    lds_barrier();                                                 // C_k
    // ---- look-ahead: bring tile (k+1,k+1) up to date with panel k and hand it over ----
    if (k + 1 < nbt) {
      const int tnn = (k + 1) * nbt - ((k + 1) * k) / 2;
#pragma unroll
      for (int s = 0; s < kCSlots7; ++s)
        if (kCTileWaves * s + tw == tnn) trail(acc[s], k + 1, k + 1);
      give_diag(k + 1);
    }
  }
}

void launch_chol_factor(const CholDesc* descs_dev, int nprob, hipStream_t s) {
  if (nprob <= 0) return;
#ifdef TADMM_CHOL_STAMPS
  if (getenv("TADMM_CHOL_STAMPS_DUMP")) {
    long long h[128];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cstamps), sizeof h);
    fprintf(stderr, "[chol stamps] step: diag | wait B | panel+C | (tile wave: trail done, panel done, relative to A_k)\n");
    for (int k = 0; k < 16 && h[3 * k + 2]; ++k)
      fprintf(stderr, "  k=%2d A_k@%6lld  diag %5lld  toC %5lld   trail@%5lld panel@%5lld\n", k, h[3 * k] - h[0], h[3 * k + 1] - h[3 * k],
              h[3 * k + 2] - h[3 * k + 1], h[64 + 2 * k] - h[3 * k], h[65 + 2 * k] - h[3 * k]);
  }
#endif
  hipLaunchKernelGGL(chol_factor_kernel, dim3(nprob), dim3(512), 0, s, descs_dev);
}

__global__ __launch_bounds__(256) void chol_solve_kernel(const CholDesc* __restrict__ descs,
                                                         const BlockRef* __restrict__ map) {
  // R's column block kb (tiles (j, kb), j < kb: the A operands of step kb) is the same for every wave of the problem.
  // Read straight from global memory by each MFMA it costs a wave 512 bytes per 64-cycle MFMA -- four waves saturate
  // what a CU gets from L2 -- so the workgroup stages it ONCE in LDS (double-buffered, fetched one step ahead).
  __shared__ double Rs[2][(kCMaxT - 1) * kCT * kCT];
  const BlockRef br = map[blockIdx.x];
  const CholDesc d = descs[br.prob];
  if (d.gate && *d.gate < d.gate_min) return;
  if (*d.bad) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nbt = d.n / kCT;
  int ring = d.rot ? *d.rot + d.sel : d.sel;
  ring -= (ring >= 3) ? 3 : 0;
  ring -= (ring >= 3) ? 3 : 0;
  if (d.rot_out && br.local == 0 && tid == 0) *d.rot_out = ring;
  G<double>* __restrict__ Y = gp(ring == 0 ? d.ring[0] : (ring == 1 ? d.ring[1] : d.ring[2]));
  const G<const double>* __restrict__ R = gp((const double*)d.R);
  const G<const double>* __restrict__ Wd = gp((const double*)d.Wd);
  const bool active = br.local * 64 + wave * 16 < d.ncols;   // ncols is a multiple of 16: whole waves idle, but they
  const int col = br.local * 64 + wave * 16 + r;             // still help staging R and keep the barriers
  const int64_t ldy = d.ldy, ldr = d.ldr;
  // element t + 256 i of column block kb: tile j = i (256 doubles per tile), row kk = t / 16, column m = t % 16
  const int srow = tid >> 4, scol = tid & 15;
  double pre[kCMaxT - 1];
  auto fetch = [&](int kb) {                                  // tiles (j, kb), j < kb -> registers
#pragma unroll
    for (int j = 0; j < kCMaxT - 1; ++j)
      if (j < kb && kb < nbt) pre[j] = R[(int64_t)(kCT * j + srow) * ldr + kCT * kb + scol];
  };
  auto stash = [&](int kb) {
#pragma unroll
    for (int j = 0; j < kCMaxT - 1; ++j)
      if (j < kb) Rs[kb & 1][j * (kCT * kCT) + tid] = pre[j];
  };
  double4_t X[kCMaxT];
  fetch(1);
#pragma unroll
  for (int kb = 0; kb < kCMaxT; ++kb) {
    if (kb < nbt) {
      if (kb >= 1) {
        stash(kb);
        __syncthreads();
      }
      fetch(kb + 1);
      if (active) {
        // two accumulator chains (even / odd j): a single chain of up to 60 dependent MFMAs was the latency floor of a step
        double4_t acc, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = Y[(int64_t)(kCT * kb + q + 4 * e) * ldy + col];
#pragma unroll
        for (int j = 0; j < kb; ++j) {
          // A[m][kk] = R[16j + kk][16kb + m], kk = q + 4e (the k order of the D-layout B operand X[j])
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double a = -Rs[kb & 1][j * (kCT * kCT) + (q + 4 * e) * kCT + r];
            if (j & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][e], acc1, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][e], acc, 0, 0, 0);
          }
        }
        acc += acc1;
        double4_t o = {0, 0, 0, 0};
        const G<const double>* W = Wd + (int64_t)kb * (kCT * kCT);
#pragma unroll
        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f64_16x16x4f64(W[r * kCT + q + 4 * e], acc[e], o, 0, 0, 0);
        X[kb] = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) Y[(int64_t)(kCT * kb + q + 4 * e) * ldy + col] = o[e];
      }
    }
  }
}

void launch_chol_solve(const CholDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(chol_solve_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
