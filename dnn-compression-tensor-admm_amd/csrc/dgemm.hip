// Grouped fp64 GEMM on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) for the filtered eigen-solver
// (filter.hip): block products with the Gram matrix G, Gram matrices of the block, Rayleigh-Ritz projections.
//
//   NT:  C[m][n] = sum_k A[m][k] * B[n][k]      (both operands contiguous along k: rows of the "transposed"
//                                                 block images YT[j][:] = column j and rows of the symmetric G)
//   NN:  C[m][n] = sum_k A[m][k] * B[k][n]      (B contiguous along n; used once per solve: U = Q * V_H)
//
// followed by an epilogue chosen per problem (`mode`):
//   0  C = acc
//   1  C = s0*acc + s1*P + s2*Q                 three-term Chebyshev recurrence, scalars read from device memory
//   2  C = acc ; rowpart[tn][m] = sum_n acc*P   Rayleigh quotients of the block columns (fixed-order partials)
//   3  rowpart[tn][m] = sum_n (acc - th[m]*P)^2 residual norms of Ritz pairs, nothing stored
//
// Operands may be named indirectly through a ring of three buffers + a device-side base index (`rot`), so that
// a data-dependent number of recurrence steps per problem needs no host round trip; a launch is a no-op for a
// problem whose gate word is below `gate_min`.
//
// Work decomposition (as gram.hip): one workgroup = one 32x32 output tile, 4 waves splitting K, each wave a
// 2x2 grid of 16x16 MFMA tiles with register double-buffered operand loads; fixed-order reduction through LDS.
// All dimensions are multiples of 32 (M, N) / 16 (K): the callers work on zero-padded images.
#include "common.h"

namespace tadmm {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// The descriptor as the kernels use it: every field copied by name from global memory (one batch of scalar loads) and
// the ring as three separate members.  `const DgemmDesc d = descs[i]` followed by the run-time indexed `d.ring[k]` made
// the compiler keep the whole 192-byte copy in SCRATCH (vector loads of the descriptor, a scratch store of all of it, a
// scratch load per field): that was a large part of the fixed cost of every launch of these kernels.
struct DgemmArgs {
  const double *A, *B, *C, *P, *Q;
  double *ring0, *ring1, *ring2;
  const int32_t* rot;
  int32_t selA, selB, selC, selP, selQ;
  int32_t M, N, K, lda, ldb, ldc, tiles_m, tiles_n, mode;
  const double* coef; const double* theta; double* rowpart;
  const int32_t* gate; int32_t gate_min;
};
__device__ __forceinline__ DgemmArgs dg_args(const DgemmDesc* __restrict__ g) {
  DgemmArgs d;
  d.A = g->A; d.B = g->B; d.C = g->C; d.P = g->P; d.Q = g->Q;
  d.ring0 = g->ring[0]; d.ring1 = g->ring[1]; d.ring2 = g->ring[2];
  d.rot = g->rot;
  d.selA = g->selA; d.selB = g->selB; d.selC = g->selC; d.selP = g->selP; d.selQ = g->selQ;
  d.M = g->M; d.N = g->N; d.K = g->K; d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc;
  d.tiles_m = g->tiles_m; d.tiles_n = g->tiles_n; d.mode = g->mode;
  d.coef = g->coef; d.theta = g->theta; d.rowpart = g->rowpart;
  d.gate = g->gate; d.gate_min = g->gate_min;
  return d;
}

__device__ __forceinline__ const double* dg_sel(const DgemmArgs& d, int sel, const double* explicit_ptr, int base) {
  if (sel < 0) return explicit_ptr;
  int i = base + sel;
  i -= (i >= 3) ? 3 : 0;
  i -= (i >= 3) ? 3 : 0;
  // prvalues (unary +): `c ? d.ring0 : d.ring1` on lvalues selects between ADDRESSES inside the struct, which is the
  // run-time indexing again
  const double* r0 = +d.ring0; const double* r1 = +d.ring1; const double* r2 = +d.ring2;
  return i == 0 ? +r0 : (i == 1 ? +r1 : +r2);
}

#ifdef TADMM_DGEMM_STAMPS
__device__ long long g_dstamps[64];
#define DSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dstamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

template <bool kBT>
__global__ __launch_bounds__(256) void dgemm_kernel(const DgemmDesc* __restrict__ descs,
                                                    const BlockRef* __restrict__ map) {
  __shared__ double red[4][4][64 * 4];   // [wave][tile][lane*4+reg]  32 KB
  __shared__ double rowred[32][33];
  const BlockRef br = map[blockIdx.x];
  const DgemmArgs d = dg_args(descs + br.prob);
  if (d.gate && *d.gate < d.gate_min) return;
  const int base = d.rot ? *d.rot : 0;
  const double* __restrict__ A = dg_sel(d, d.selA, d.A, base);
  const double* __restrict__ B = dg_sel(d, d.selB, d.B, base);
  double* __restrict__ C = const_cast<double*>(dg_sel(d, d.selC, d.C, base));
  const int tm = br.local / d.tiles_n, tn = br.local - tm * d.tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int K = d.K;
  // each wave takes a contiguous quarter (multiple of 16) of K
  const int per = (((K + 3) / 4) + 15) & ~15;
  const int wk0 = min(K, wave * per), wk1 = min(K, wk0 + per);

  double4_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const int64_t lda = d.lda, ldb = d.ldb;
  const double* pa0 = A + (int64_t)(tm * 32 + r) * lda;
  const double* pa1 = pa0 + 16 * lda;

  struct Chunk { double a0[4], a1[4], b0[4], b1[4]; };
  auto mma = [&](const Chunk& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a0[e], c.b0[e], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a0[e], c.b1[e], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a1[e], c.b0[e], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a1[e], c.b1[e], acc11, 0, 0, 0);
    }
  };
  // lane (r,q) supplies reduction index kk + 4q + e to MFMA e of a 16-wide chunk (a permutation of the hardware's
  // k order that both operands share)
  auto fetch = [&](int kk, Chunk& c) {
    const int k = kk + 4 * q;
    const double2_t t0 = *reinterpret_cast<const double2_t*>(pa0 + k), t1 = *reinterpret_cast<const double2_t*>(pa0 + k + 2);
    const double2_t t2 = *reinterpret_cast<const double2_t*>(pa1 + k), t3 = *reinterpret_cast<const double2_t*>(pa1 + k + 2);
    c.a0[0] = t0.x; c.a0[1] = t0.y; c.a0[2] = t1.x; c.a0[3] = t1.y;
    c.a1[0] = t2.x; c.a1[1] = t2.y; c.a1[2] = t3.x; c.a1[3] = t3.y;
    if (kBT) {
      const double* pb0 = B + (int64_t)(tn * 32 + r) * ldb + k;
      const double* pb1 = pb0 + 16 * ldb;
      const double2_t u0 = *reinterpret_cast<const double2_t*>(pb0), u1 = *reinterpret_cast<const double2_t*>(pb0 + 2);
      const double2_t u2 = *reinterpret_cast<const double2_t*>(pb1), u3 = *reinterpret_cast<const double2_t*>(pb1 + 2);
      c.b0[0] = u0.x; c.b0[1] = u0.y; c.b0[2] = u1.x; c.b0[3] = u1.y;
      c.b1[0] = u2.x; c.b1[1] = u2.y; c.b1[2] = u3.x; c.b1[3] = u3.y;
    } else {
      const double* pb = B + (int64_t)k * ldb + tn * 32 + r;    // 16 lanes read 128 contiguous bytes of a k-row
#pragma unroll
      for (int e = 0; e < 4; ++e) { c.b0[e] = pb[(int64_t)e * ldb]; c.b1[e] = pb[(int64_t)e * ldb + 16]; }
    }
  };
  if (wk0 < wk1) {
    Chunk cur, nxt;
    fetch(wk0, cur);
    for (int kk = wk0; kk < wk1; kk += 16) {
      const bool more = kk + 16 < wk1;
      if (more) fetch(kk + 16, nxt);
      mma(cur);
      if (more) cur = nxt;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][0][lane * 4 + e] = acc00[e];
    red[wave][1][lane * 4 + e] = acc01[e];
    red[wave][2][lane * 4 + e] = acc10[e];
    red[wave][3][lane * 4 + e] = acc11[e];
  }
  __syncthreads();
  const int mode = d.mode;
  double s0 = 1.0, s1 = 0.0, s2 = 0.0;
  if (mode == 1) { s0 = d.coef[0]; s1 = d.coef[1]; s2 = d.coef[2]; }
  const double* __restrict__ P = (mode >= 1) ? dg_sel(d, d.selP, d.P, base) : nullptr;
  const double* __restrict__ Q = (mode == 1 && s2 != 0.0) ? dg_sel(d, d.selQ, d.Q, base) : nullptr;
  const int64_t ldc = d.ldc;
  // element (row, col) of the tile <- thread t: 4 consecutive columns of one row (32-byte stores, 256 B per row)
  {
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = c4 + u;
      const int t = ((row >> 4) << 1) | (col >> 4);              // sub-tile: 0 = (0,0) 1 = (0,1) 2 = (1,0) 3 = (1,1)
      const int rr = row & 15, cc = col & 15;
      const int le = ((((rr & 3) << 4) | cc) << 2) | (rr >> 2);   // D layout: row = (l>>4) + 4*reg, col = l&15
      v[u] = (red[0][t][le] + red[1][t][le]) + (red[2][t][le] + red[3][t][le]);
    }
    const int64_t gi = (int64_t)(tm * 32 + row) * ldc + tn * 32 + c4;
    if (mode == 0) {
      *reinterpret_cast<double2_t*>(C + gi) = double2_t{v[0], v[1]};
      *reinterpret_cast<double2_t*>(C + gi + 2) = double2_t{v[2], v[3]};
    } else if (mode == 1) {
      const double2_t p0 = *reinterpret_cast<const double2_t*>(P + gi), p1 = *reinterpret_cast<const double2_t*>(P + gi + 2);
      double o[4] = {s0 * v[0] + s1 * p0.x, s0 * v[1] + s1 * p0.y, s0 * v[2] + s1 * p1.x, s0 * v[3] + s1 * p1.y};
      if (Q) {
        const double2_t q0 = *reinterpret_cast<const double2_t*>(Q + gi), q1 = *reinterpret_cast<const double2_t*>(Q + gi + 2);
        o[0] += s2 * q0.x; o[1] += s2 * q0.y; o[2] += s2 * q1.x; o[3] += s2 * q1.y;
      }
      *reinterpret_cast<double2_t*>(C + gi) = double2_t{o[0], o[1]};
      *reinterpret_cast<double2_t*>(C + gi + 2) = double2_t{o[2], o[3]};
    } else {
      const double2_t p0 = *reinterpret_cast<const double2_t*>(P + gi), p1 = *reinterpret_cast<const double2_t*>(P + gi + 2);
      double part;
      if (mode == 2) {
        *reinterpret_cast<double2_t*>(C + gi) = double2_t{v[0], v[1]};
        *reinterpret_cast<double2_t*>(C + gi + 2) = double2_t{v[2], v[3]};
        part = (v[0] * p0.x + v[1] * p0.y) + (v[2] * p1.x + v[3] * p1.y);
      } else {
        const double th = d.theta[tm * 32 + row];
        const double e0 = v[0] - th * p0.x, e1 = v[1] - th * p0.y, e2 = v[2] - th * p1.x, e3 = v[3] - th * p1.y;
        part = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
      }
      rowred[row][threadIdx.x & 7] = part;
    }
  }
  if (mode >= 2) {
    __syncthreads();
    if (threadIdx.x < 32) {
      const double* p = rowred[threadIdx.x];
      d.rowpart[(int64_t)tn * d.M + tm * 32 + threadIdx.x] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
  }
}

void launch_dgemm(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, bool b_transposed, hipStream_t s) {
  if (nblocks <= 0) return;
  if (b_transposed) hipLaunchKernelGGL(dgemm_kernel<true>, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
  else hipLaunchKernelGGL(dgemm_kernel<false>, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}


// ---------------------------------------------------------------------------------------------------------------
// NT product, 64x64 output tile per workgroup, operands staged through LDS (the workhorse of the filter: the
// 32x32/K-split kernel above re-reads every operand row from L2 once per 32 output columns and is bound by the
// CU's vector-memory path; here a row is fetched once per 64 and shared by two waves).
//   4 waves in a 2x2 grid, each a 32x32 block = 2x2 MFMA tiles, full K per workgroup, K chunks of 32:
//   global -> registers (next chunk, issued before the MFMAs of the current one) -> LDS -> MFMA operands.
//   LDS rows hold 32 k-values + 2 doubles of padding: lane (r, q) reads row r at k = 4s + q, i.e. 8-byte word
//   34 r + 4 s + q -- the 32 lanes of a ds_read_b64 group hit 32 distinct bank pairs.
// M, N multiples of 32 (edge tiles are masked by 32-row halves), K multiple of 32.
// Row partials of the epilogue modes 2/3 are per 64-column tile: rowpart[tn64][M].
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDT = 64, kDK = 32, kDLd = kDK + 2;

// TN = 64: 2x2 waves of 32x32;  TN = 32: 2x2 waves of 32x16 (half the work per workgroup: the grouped launches of the
// filter have only a few hundred 64x64 tiles, and a chip of 256 CUs is filled evenly only by tasks well below
// (total work / 256) in size).
// KW = 2: eight waves, the second four take the upper half of every K chunk (partial sums folded through LDS at the
// end).  A workgroup then keeps two waves on every SIMD of its CU and finishes in half the time: the grouped launches
// of the filter put about one workgroup on a CU, whose duration IS the launch's.
// TM = 32 (with KW = 4: the same eight waves, four K quarters of a 32 x 32 tile): half the work per workgroup again.  A
// workgroup's duration is its share of the fp64 matrix rate of ONE CU (a 64 x 32 x 480 tile is 2 MFLOP = 11 us at the
// ~176 GFLOP/s a CU issues), and a product launch of the critical chain has 45 - 135 such workgroups on 256 CUs: the
// smaller tile buys latency with operand traffic (each workgroup still streams its 32 rows of G).
template <int TN, int KW, int TM = kDT>
__global__ __launch_bounds__(128 * (TM / 32) * KW) void dgemm_nt_tile_kernel(const DgemmDesc* __restrict__ descs,
                                                                            const BlockRef* __restrict__ map) {
  constexpr int NT = 128 * (TM / 32) * KW;      // threads: (TM / 32) x 2 waves per K slice
  constexpr int WN = TN / 2;            // columns per wave
  constexpr int NB = WN / 16;           // B fragments per wave and k-step
  constexpr int AP = TM * 16 / NT;      // 16-byte pieces of A per thread and chunk (TM rows x 16 pieces / NT threads)
  constexpr int BP = TN * 16 / NT;      // same for B (TN rows)
  constexpr int RS = NT / 16;           // row stride of a thread's pieces
  constexpr int TPR = 256 / TM;         // threads per output row in the epilogue (256 threads)
  constexpr int CPT = TN / TPR;         // output columns per thread in the epilogue
  constexpr int kBuf = (TM + TN) * kDLd;           // doubles per LDS stage (A rows then B rows)
  static_assert(AP >= 1 && BP >= 1 && CPT >= 2, "tile / thread layout");
  __shared__ __attribute__((aligned(16))) double smem[2 * kBuf > TM * (TN + 1) ? 2 * kBuf : TM * (TN + 1)];
  __shared__ double rowred[TM][TPR];
  const BlockRef br = map[blockIdx.x];
  const DgemmArgs d = dg_args(descs + br.prob);
  if (d.gate && *d.gate < d.gate_min) return;
  const int base = d.rot ? *d.rot : 0;
  const G<const double>* __restrict__ A = gp(dg_sel(d, d.selA, d.A, base));
  const G<const double>* __restrict__ B = gp(dg_sel(d, d.selB, d.B, base));
  G<double>* __restrict__ C = gp(const_cast<double*>(dg_sel(d, d.selC, d.C, base)));
  const int tiles_n = (d.N + TN - 1) / TN;
  const int tm = br.local / tiles_n, tn = br.local - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kh = (TM == 64) ? wave >> 2 : wave >> 1, wm = (TM == 64) ? (wave >> 1) & 1 : 0, wn = wave & 1;
  const int K = d.K;
  const int64_t lda = d.lda, ldb = d.ldb;
  const bool live = (m0 + 32 * wm < d.M) && (n0 + WN * wn < d.N);      // this wave's block exists

  // Pipeline: two LDS stages (one barrier per K chunk) and two register sets, so that a chunk's global loads are
  // issued two chunks ahead of the LDS store that consumes them -- a workgroup's chunk (16-32 MFMAs per wave) is
  // shorter than an L2 / Infinity-Cache round trip, and the grouped launches of the filter put only one or two
  // workgroups on a CU.
  struct Regs { double2_t a[AP], b[BP]; };
  const int srow = tid >> 4, sc2 = tid & 15;                             // rows srow + RS i
  // Rows beyond an edge are clamped, not predicated: a row of A (B) only feeds its own output row (column), which
  // the epilogue never stores, and unconditional loads let the compiler count them (a branch per load makes it wait
  // for vmcnt(0) at every LDS store, which serialises the whole prefetch).
  const G<const double>* pa[AP]; const G<const double>* pb[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) pa[i] = A + (int64_t)min(m0 + srow + RS * i, d.M - 1) * lda + 2 * sc2;
#pragma unroll
  for (int i = 0; i < BP; ++i) pb[i] = B + (int64_t)min(n0 + srow + RS * i, d.N - 1) * ldb + 2 * sc2;
  auto gload = [&](Regs& R, int k0) {
#pragma unroll
    for (int i = 0; i < AP; ++i) R.a[i] = *reinterpret_cast<const G<const double2_t>*>(pa[i] + k0);
#pragma unroll
    for (int i = 0; i < BP; ++i) R.b[i] = *reinterpret_cast<const G<const double2_t>*>(pb[i] + k0);
  };
  auto sstore = [&](const Regs& R, int stage) {
    double (*As)[kDLd] = reinterpret_cast<double (*)[kDLd]>(smem + stage * kBuf);
    double (*Bs)[kDLd] = As + TM;
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<double2_t*>(&As[srow + RS * i][2 * sc2]) = R.a[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<double2_t*>(&Bs[srow + RS * i][2 * sc2]) = R.b[i];
  };
  double4_t acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = double4_t{0, 0, 0, 0};
  auto compute = [&](int stage) {
    if (!live) return;
    const double (*As)[kDLd] = reinterpret_cast<const double (*)[kDLd]>(smem + stage * kBuf);
    const double (*Bs)[kDLd] = As + TM;
    const int kof = kh * (kDK / KW);                                      // this wave's part of the chunk
    const double* a0p = &As[32 * wm + r][q + kof];
    const double* a1p = &As[32 * wm + 16 + r][q + kof];
    const double* b0p = &Bs[WN * wn + r][q + kof];
    // all fragments of the chunk first (one burst of LDS reads, one wait), then the MFMAs back to back: left to
    // itself the compiler waits for lgkmcnt(0) in front of every second k-step
    constexpr int KS = kDK / KW / 4;
    double fa0[KS], fa1[KS], fb[KS][NB];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      fa0[s] = a0p[4 * s]; fa1[s] = a1p[4 * s];
#pragma unroll
      for (int j = 0; j < NB; ++j) fb[s][j] = b0p[16 * j * kDLd + 4 * s];
    }
    // (two accumulators per wave; splitting them by k-step parity into four independent chains changes nothing: the
    // instruction issues at ~150 cycles at one wave per SIMD whatever the chain length, scripts/micro/mfma_f64_peak.hip)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[s], fb[s][j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[s], fb[s][j], acc[1][j], 0, 0, 0);
      }
    }
  };
  DSTAMP(0);
  const int nch = K / kDK;
  constexpr int NPF = 2;                 // register sets: a chunk's loads are issued NPF chunks before its LDS store (deeper: no gain)
  Regs R[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u)
    if (u < nch) gload(R[u], u * kDK);
  sstore(R[0], 0);
  if (NPF < nch) gload(R[0], NPF * kDK);
  __syncthreads();
  DSTAMP(1);
  for (int c = 0; c < nch; c += NPF) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int i = c + u;               // chunk i sits in stage u & 1 (NPF is even)
      if (i < nch) {
        if (i + 1 < nch) {
          sstore(R[(u + 1) % NPF], (u + 1) & 1);
          if (i + 1 + NPF < nch) gload(R[(u + 1) % NPF], (i + 1 + NPF) * kDK);
        }
        compute(u & 1);
        DSTAMP(2 + 2 * i);
        __syncthreads();
        DSTAMP(3 + 2 * i);
      }
    }
  }
  // accumulators -> LDS tile [64][TN+1] (aliases the staging buffers; the loop's last barrier has passed)
  double (*Ct)[TN + 1] = reinterpret_cast<double (*)[TN + 1]>(smem);
  // the K slices fold their sums through LDS: the last slice writes, the others add in turn, slice 0 last
#pragma unroll
  for (int h = KW - 1; h >= 0; --h) {
    if (kh == h) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            double* c = &Ct[32 * wm + 16 * i + q + 4 * e][WN * wn + 16 * j + r];
            *c = (h == KW - 1 ? 0.0 : *c) + acc[i][j][e];
          }
    }
    __syncthreads();
  }
  if (NT > 256 && tid >= 256) return;   // the epilogue is laid out for 256 threads
  const int mode = d.mode;
  double s0 = 1.0, s1 = 0.0, s2 = 0.0;
  if (mode == 1) { s0 = d.coef[0]; s1 = d.coef[1]; s2 = d.coef[2]; }
  const G<const double>* __restrict__ P = gp((mode >= 1) ? dg_sel(d, d.selP, d.P, base) : nullptr);
  const G<const double>* __restrict__ Q = gp((mode == 1 && s2 != 0.0) ? dg_sel(d, d.selQ, d.Q, base) : nullptr);
  const int64_t ldc = d.ldc;
  const int row = tid / TPR, c0 = (tid % TPR) * CPT;
  const bool rok = m0 + row < d.M;
  double part = 0.0;
  if (rok && n0 + c0 < d.N) {                                            // N % 16 == 0: a thread's columns are in or out together
    const double th = (mode == 3) ? d.theta[m0 + row] : 0.0;
#pragma unroll
    for (int u = 0; u < CPT; u += 2) {
      const int col = c0 + u;
      const int64_t gi = (int64_t)(m0 + row) * ldc + n0 + col;
      const double v0 = Ct[row][col], v1 = Ct[row][col + 1];
      if (mode == 0) {
        *reinterpret_cast<G<double2_t>*>(C + gi) = double2_t{v0, v1};
      } else {
        const double2_t p = *reinterpret_cast<const G<const double2_t>*>(P + gi);
        if (mode == 1) {
          double o0 = s0 * v0 + s1 * p.x, o1 = s0 * v1 + s1 * p.y;
          if (Q) {
            const double2_t qq = *reinterpret_cast<const G<const double2_t>*>(Q + gi);
            o0 += s2 * qq.x; o1 += s2 * qq.y;
          }
          *reinterpret_cast<G<double2_t>*>(C + gi) = double2_t{o0, o1};
        } else if (mode == 2) {
          *reinterpret_cast<G<double2_t>*>(C + gi) = double2_t{v0, v1};
          part += v0 * p.x + v1 * p.y;
        } else {
          const double e0 = v0 - th * p.x, e1 = v1 - th * p.y;
          part += e0 * e0 + e1 * e1;
        }
      }
    }
  }
  DSTAMP(40);
  if (mode >= 2) {
    rowred[row][tid % TPR] = part;
    __syncthreads();
    if (tid < TM && m0 + tid < d.M) {
      const double* pr = rowred[tid];
      double sum = (pr[0] + pr[1]) + (pr[2] + pr[3]);
      if (TPR == 8) sum += (pr[4] + pr[5]) + (pr[6] + pr[7]);
      d.rowpart[(int64_t)tn * d.M + m0 + tid] = sum;
    }
  }
}

void launch_dgemm_nt64(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s, int tile_n, int tile_m) {
  if (nblocks <= 0) return;
#ifdef TADMM_DGEMM_STAMPS
  if (getenv("TADMM_DGEMM_STAMPS_DUMP")) {
    long long h[64];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dstamps), sizeof h);
    fprintf(stderr, "[dgemm stamps]");
    for (int i = 1; i < 41; ++i) if (h[i]) fprintf(stderr, " %d:%lld", i, h[i] - h[0]);
    fprintf(stderr, "\n");
  }
#endif
  // eight waves (two K halves) by default: two waves per SIMD issue fp64 MFMAs at 45 instead of 33-35 TF/s chip-wide
  // (scripts/micro/mfma_f64_peak.hip); worth 1.6 % of the ResNet-50 iteration, neutral elsewhere.  TADMM_DGEMM_KW=1: four.
  static const int kw = getenv("TADMM_DGEMM_KW") ? atoi(getenv("TADMM_DGEMM_KW")) : 2;
  if (tile_m == 32) {                   // 32 x 32 tiles, four K quarters (filter products; filter_tile_m())
    hipLaunchKernelGGL((dgemm_nt_tile_kernel<32, 4, 32>), dim3(nblocks), dim3(512), 0, s, descs_dev, map_dev);
    return;
  }
  if (tile_n == 32) {
    if (kw == 2) hipLaunchKernelGGL((dgemm_nt_tile_kernel<32, 2>), dim3(nblocks), dim3(512), 0, s, descs_dev, map_dev);
    else hipLaunchKernelGGL((dgemm_nt_tile_kernel<32, 1>), dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
  } else {
    if (kw == 2) hipLaunchKernelGGL((dgemm_nt_tile_kernel<64, 2>), dim3(nblocks), dim3(512), 0, s, descs_dev, map_dev);
    else hipLaunchKernelGGL((dgemm_nt_tile_kernel<64, 1>), dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
  }
}

// Y <- s0*T + s1*Qb  (first Chebyshev step of a stage, in place of T); ring-addressed, gated like the products.
__global__ __launch_bounds__(256) void daxpby_kernel(const DgemmDesc* __restrict__ descs, const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const DgemmArgs d = dg_args(descs + br.prob);
  if (d.gate && *d.gate < d.gate_min) return;
  const int base = d.rot ? *d.rot : 0;
  double* __restrict__ C = const_cast<double*>(dg_sel(d, d.selC, d.C, base));
  const double* __restrict__ P = dg_sel(d, d.selP, d.P, base);
  const double s0 = d.coef[0], s1 = d.coef[1];
  const int64_t total = (int64_t)d.M * d.ldc;
  const int64_t i0 = ((int64_t)br.local * 256 + threadIdx.x) * 4;
  if (i0 + 3 < total) {
    const double2_t c0 = *reinterpret_cast<const double2_t*>(C + i0), c1 = *reinterpret_cast<const double2_t*>(C + i0 + 2);
    const double2_t p0 = *reinterpret_cast<const double2_t*>(P + i0), p1 = *reinterpret_cast<const double2_t*>(P + i0 + 2);
    *reinterpret_cast<double2_t*>(C + i0) = double2_t{s0 * c0.x + s1 * p0.x, s0 * c0.y + s1 * p0.y};
    *reinterpret_cast<double2_t*>(C + i0 + 2) = double2_t{s0 * c1.x + s1 * p1.x, s0 * c1.y + s1 * p1.y};
  } else {
    for (int64_t i = i0; i < total; ++i) C[i] = s0 * C[i] + s1 * P[i];
  }
}

void launch_daxpby(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(daxpby_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
