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

__device__ __forceinline__ const double* dg_sel(const DgemmDesc& d, int sel, const double* explicit_ptr, int base) {
  if (sel < 0) return explicit_ptr;
  int i = base + sel;
  i -= (i >= 3) ? 3 : 0;
  i -= (i >= 3) ? 3 : 0;
  return d.ring[i];
}

template <bool kBT>
__global__ __launch_bounds__(256) void dgemm_kernel(const DgemmDesc* __restrict__ descs,
                                                    const BlockRef* __restrict__ map) {
  __shared__ double red[4][4][64 * 4];   // [wave][tile][lane*4+reg]  32 KB
  __shared__ double rowred[32][33];
  const BlockRef br = map[blockIdx.x];
  const DgemmDesc d = descs[br.prob];
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

// Y <- s0*T + s1*Qb  (first Chebyshev step of a stage, in place of T); ring-addressed, gated like the products.
__global__ __launch_bounds__(256) void daxpby_kernel(const DgemmDesc* __restrict__ descs, const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const DgemmDesc d = descs[br.prob];
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
