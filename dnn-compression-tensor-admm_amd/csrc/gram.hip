// Gram matrix of a TT unfolding on the fp64 matrix cores.
//
//   G = A A^T  (m <= n, reduce over the n columns)      or      G = A^T A  (m > n, reduce over rows)
//
// A is float32; every product of two float32 values is exact in float64 (24+24 <= 53 mantissa
// bits), so v_mfma_f64_16x16x4_f64 gives an *exactly-rounded-per-add* fp64 accumulation of the
// Gram entries -- this is what keeps the Gram + eigen-solve route inside the 1e-5 parity bar
// (SURVEY.md section 7 "Accuracy of the Gram route").
//
// Work decomposition: 32x32 output tiles (upper triangle only, the matrix is symmetric) x split-K.
// One workgroup = 4 waves, each wave owns a quarter of the workgroup's K-chunk and a 2x2 grid of
// 16x16 MFMA tiles; partial tiles are reduced through LDS in a fixed order and written to a
// partial buffer; a second kernel sums the split-K partials (fixed order => deterministic),
// mirrors the triangle and zero-pads to the [Npad][ld] layout the Jacobi solver wants.
//
// MFMA operand maps (cdna_hip_programming.md section 3): A operand lane l holds A[i=l&15][k=l>>4],
// B operand lane l holds B[k=l>>4][j=l&15]; D: col = l&15, row = (l>>4) + 4*reg.
// For a Gram both operands are rows of the same matrix: B[k][j] = X[c0+j][k] has the *same* lane
// map as the A operand of row block c0, so one load serves both roles.
#include "common.h"

namespace tadmm {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// tile-pair index -> (ti, tj), ti <= tj, row-major over the upper triangle
__device__ __forceinline__ void tile_pair(int tp, int nt, int& ti, int& tj) {
  int a = 0, rowlen = nt;
  while (tp >= rowlen) { tp -= rowlen; ++a; --rowlen; }
  ti = a; tj = a + tp;
}

__global__ __launch_bounds__(256) void gram_partial_kernel(const GramDesc* __restrict__ descs,
                                                           const BlockRef* __restrict__ map) {
  __shared__ double red[4][4][64 * 4];  // [wave][tile][lane*4+reg]  32 KB
  const BlockRef br = map[blockIdx.x];
  const GramDesc d = descs[br.prob];
  const int ntp = d.nt * (d.nt + 1) / 2;
  const int ks = br.local / ntp;
  const int tp = br.local - ks * ntp;
  int ti, tj;
  tile_pair(tp, d.nt, ti, tj);
  const bool diag = (ti == tj);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;

  const int kbeg = ks * d.kchunk;
  const int kend = min(d.K, kbeg + d.kchunk);
  // each wave takes a contiguous quarter (multiple of 16) of [kbeg, kend)
  const int per = (((kend - kbeg + 3) / 4) + 15) & ~15;
  const int wk0 = min(kend, kbeg + wave * per);
  const int wk1 = min(kend, wk0 + per);

  double4_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const float* A = d.A;
  const int N = d.N;
  const int ra0 = ti * 32 + r, ra1 = ra0 + 16;   // G-rows of this lane for the A role
  const int rb0 = tj * 32 + r, rb1 = rb0 + 16;   // G-rows for the B role

  if (!d.trans) {
    // rows of A are G-indices, reduction runs along the contiguous dimension
    const int64_t ld = d.n;
    const bool vec = ((ld & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
    const float* pa0 = A + (int64_t)min(ra0, N - 1) * ld;
    const float* pa1 = A + (int64_t)min(ra1, N - 1) * ld;
    const float* pb0 = A + (int64_t)min(rb0, N - 1) * ld;
    const float* pb1 = A + (int64_t)min(rb1, N - 1) * ld;
    const float ma0 = ra0 < N ? 1.f : 0.f, ma1 = ra1 < N ? 1.f : 0.f;
    const float mb0 = rb0 < N ? 1.f : 0.f, mb1 = rb1 < N ? 1.f : 0.f;
    for (int kk = wk0; kk < wk1; kk += 16) {
      const int k = kk + 4 * q;  // this lane's 4 consecutive reduction indices (a permutation of the
                                 // MFMA k order that both operands share)
      float4 a0, a1, b0, b1;
      if (vec && k + 3 < wk1) {
        a0 = *reinterpret_cast<const float4*>(pa0 + k);
        a1 = *reinterpret_cast<const float4*>(pa1 + k);
        if (!diag) {
          b0 = *reinterpret_cast<const float4*>(pb0 + k);
          b1 = *reinterpret_cast<const float4*>(pb1 + k);
        }
      } else {
        float t[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (k + e) < wk1;
          t[0][e] = ok ? pa0[k + e] : 0.f;
          t[1][e] = ok ? pa1[k + e] : 0.f;
          t[2][e] = (ok && !diag) ? pb0[k + e] : 0.f;
          t[3][e] = (ok && !diag) ? pb1[k + e] : 0.f;
        }
        a0 = make_float4(t[0][0], t[0][1], t[0][2], t[0][3]);
        a1 = make_float4(t[1][0], t[1][1], t[1][2], t[1][3]);
        b0 = make_float4(t[2][0], t[2][1], t[2][2], t[2][3]);
        b1 = make_float4(t[3][0], t[3][1], t[3][2], t[3][3]);
      }
      a0.x *= ma0; a0.y *= ma0; a0.z *= ma0; a0.w *= ma0;
      a1.x *= ma1; a1.y *= ma1; a1.z *= ma1; a1.w *= ma1;
      if (diag) { b0 = a0; b1 = a1; } else {
        b0.x *= mb0; b0.y *= mb0; b0.z *= mb0; b0.w *= mb0;
        b1.x *= mb1; b1.y *= mb1; b1.z *= mb1; b1.w *= mb1;
      }
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double da0 = av0[e], da1 = av1[e], db0 = bv0[e], db1 = bv1[e];
        acc00 = mfma_f64(da0, db0, acc00);
        acc01 = mfma_f64(da0, db1, acc01);
        if (!diag) acc10 = mfma_f64(da1, db0, acc10);
        acc11 = mfma_f64(da1, db1, acc11);
      }
    }
  } else {
    // columns of A are G-indices, reduction runs over rows: lane (r,q) reads A[k+q][col]
    const int64_t ld = d.n;
    const int ca0 = min(ra0, N - 1), ca1 = min(ra1, N - 1), cb0 = min(rb0, N - 1), cb1 = min(rb1, N - 1);
    const float ma0 = ra0 < N ? 1.f : 0.f, ma1 = ra1 < N ? 1.f : 0.f;
    const float mb0 = rb0 < N ? 1.f : 0.f, mb1 = rb1 < N ? 1.f : 0.f;
    for (int kk = wk0; kk < wk1; kk += 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kk + 4 * e + q;
        const bool ok = k < wk1;
        const float* row = A + (int64_t)min(k, d.K - 1) * ld;
        const double da0 = ok ? row[ca0] * ma0 : 0.f;
        const double da1 = ok ? row[ca1] * ma1 : 0.f;
        double db0, db1;
        if (diag) { db0 = da0; db1 = da1; } else {
          db0 = ok ? row[cb0] * mb0 : 0.f;
          db1 = ok ? row[cb1] * mb1 : 0.f;
        }
        acc00 = mfma_f64(da0, db0, acc00);
        acc01 = mfma_f64(da0, db1, acc01);
        if (!diag) acc10 = mfma_f64(da1, db0, acc10);
        acc11 = mfma_f64(da1, db1, acc11);
      }
    }
  }

  // cross-wave reduction in a fixed order
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][0][lane * 4 + e] = acc00[e];
    red[wave][1][lane * 4 + e] = acc01[e];
    red[wave][2][lane * 4 + e] = acc10[e];
    red[wave][3][lane * 4 + e] = acc11[e];
  }
  __syncthreads();
  double* out = d.partial + ((int64_t)ks * ntp + tp) * 1024;
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int t = idx >> 8, le = idx & 255;        // tile, lane*4+reg
    const double v = (red[0][t][le] + red[1][t][le]) + (red[2][t][le] + red[3][t][le]);
    const int l = le >> 2, reg = le & 3;
    const int row = (t >> 1) * 16 + (l >> 4) + 4 * reg;
    const int col = (t & 1) * 16 + (l & 15);
    out[row * 32 + col] = v;
  }
}

// sum split-K partials, mirror, zero-pad.  local block = chunk of 1024 outputs of the [Npad][ld] image
__global__ __launch_bounds__(256) void gram_reduce_kernel(const GramDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const GramDesc d = descs[br.prob];
  const int ntp = d.nt * (d.nt + 1) / 2;
  const int64_t total = (int64_t)d.Npad * d.ld;
  const int64_t base = (int64_t)br.local * 1024;
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int64_t idx = base + t;
    if (idx >= total) break;
    const int j = (int)(idx / d.ld), i = (int)(idx - (int64_t)j * d.ld);
    double v = 0.0;
    if (j < d.N && i < d.N) {
      int a = j >> 5, b = i >> 5, jr = j & 31, ir = i & 31;
      if (a > b) { int s = a; a = b; b = s; s = jr; jr = ir; ir = s; }
      if (a == b && jr > ir) { const int s = jr; jr = ir; ir = s; }   // diagonal tiles: upper part only
      const int tp = a * d.nt - (a * (a - 1)) / 2 + (b - a);
      const double* p = d.partial + (int64_t)tp * 1024 + jr * 32 + ir;
      for (int ks = 0; ks < d.ksplit; ++ks) v += p[(int64_t)ks * ntp * 1024];
    }
    d.G[idx] = v;
  }
}

void launch_gram_partial(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

void launch_gram_reduce(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
