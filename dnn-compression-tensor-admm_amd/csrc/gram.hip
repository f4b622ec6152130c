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
                                                           const BlockRef* __restrict__ map,
                                                           const int32_t* __restrict__ skip) {
  __shared__ double red[4][4][64 * 4];  // [wave][tile][lane*4+reg]  32 KB
  __shared__ double tile[32][33];       // summed 32x32 tile of the direct (ksplit == 1) path
  const BlockRef br = map[blockIdx.x];
  if (skip && skip[br.prob]) return;
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
  const float ma0 = ra0 < N ? 1.f : 0.f, ma1 = ra1 < N ? 1.f : 0.f;
  const float mb0 = rb0 < N ? 1.f : 0.f, mb1 = rb1 < N ? 1.f : 0.f;
  const int64_t ld = d.n;

  struct Chunk { float a0[4], a1[4], b0[4], b1[4]; };   // this lane's 4 reduction indices of a 16-wide k chunk
  // 16 fp64 MFMAs per chunk; the next chunk's loads are issued before them (register double buffer), so the
  // matrix cores do not wait for HBM/L2 latency
  auto mma = [&](const Chunk& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double da0 = c.a0[e], da1 = c.a1[e];
      const double db0 = diag ? da0 : (double)c.b0[e], db1 = diag ? da1 : (double)c.b1[e];
      acc00 = mfma_f64(da0, db0, acc00);
      acc01 = mfma_f64(da0, db1, acc01);
      if (!diag) acc10 = mfma_f64(da1, db0, acc10);
      acc11 = mfma_f64(da1, db1, acc11);
    }
  };

  if (!d.trans) {
    // rows of A are G-indices, reduction runs along the contiguous dimension; lane (r,q) takes k = kk+4q .. +3
    // (a permutation of the MFMA k order that both operands share)
    const bool vec = ((ld & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
    const float* pa0 = A + (int64_t)min(ra0, N - 1) * ld;
    const float* pa1 = A + (int64_t)min(ra1, N - 1) * ld;
    const float* pb0 = A + (int64_t)min(rb0, N - 1) * ld;
    const float* pb1 = A + (int64_t)min(rb1, N - 1) * ld;
    auto fetch = [&](int kk, Chunk& c) {
      const int k = kk + 4 * q;
      if (vec && k + 3 < wk1) {
        const float4 t0 = *reinterpret_cast<const float4*>(pa0 + k);
        const float4 t1 = *reinterpret_cast<const float4*>(pa1 + k);
        c.a0[0] = t0.x * ma0; c.a0[1] = t0.y * ma0; c.a0[2] = t0.z * ma0; c.a0[3] = t0.w * ma0;
        c.a1[0] = t1.x * ma1; c.a1[1] = t1.y * ma1; c.a1[2] = t1.z * ma1; c.a1[3] = t1.w * ma1;
        if (!diag) {
          const float4 t2 = *reinterpret_cast<const float4*>(pb0 + k);
          const float4 t3 = *reinterpret_cast<const float4*>(pb1 + k);
          c.b0[0] = t2.x * mb0; c.b0[1] = t2.y * mb0; c.b0[2] = t2.z * mb0; c.b0[3] = t2.w * mb0;
          c.b1[0] = t3.x * mb1; c.b1[1] = t3.y * mb1; c.b1[2] = t3.z * mb1; c.b1[3] = t3.w * mb1;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (k + e) < wk1;
          c.a0[e] = ok ? pa0[k + e] * ma0 : 0.f;
          c.a1[e] = ok ? pa1[k + e] * ma1 : 0.f;
          c.b0[e] = (ok && !diag) ? pb0[k + e] * mb0 : 0.f;
          c.b1[e] = (ok && !diag) ? pb1[k + e] * mb1 : 0.f;
        }
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
  } else {
    // columns of A are G-indices, reduction runs over rows: lane (r,q) reads A[kk+4e+q][col]
    const int ca0 = min(ra0, N - 1), ca1 = min(ra1, N - 1), cb0 = min(rb0, N - 1), cb1 = min(rb1, N - 1);
    auto fetch = [&](int kk, Chunk& c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kk + 4 * e + q;
        const bool ok = k < wk1;
        const float* row = A + (int64_t)min(k, d.K - 1) * ld;
        c.a0[e] = ok ? row[ca0] * ma0 : 0.f;
        c.a1[e] = ok ? row[ca1] * ma1 : 0.f;
        c.b0[e] = (ok && !diag) ? row[cb0] * mb0 : 0.f;
        c.b1[e] = (ok && !diag) ? row[cb1] * mb1 : 0.f;
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
  if (d.ksplit > 1) {
    double* out = d.partial + ((int64_t)ks * ntp + tp) * 1024;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
      const int t = idx >> 8, le = idx & 255;        // tile, lane*4+reg
      const double v = (red[0][t][le] + red[1][t][le]) + (red[2][t][le] + red[3][t][le]);
      const int l = le >> 2, reg = le & 3;
      const int row = (t >> 1) * 16 + (l >> 4) + 4 * reg;
      const int col = (t & 1) * 16 + (l & 15);
      out[row * 32 + col] = v;
    }
    return;
  }
  // direct path (no split-K): this workgroup owns the whole reduction, so it writes the tile and its mirror
  // straight into the zero-padded [Npad][ld] image of the eigen-solver (Npad = 32*nt)
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int t = idx >> 8, le = idx & 255;
    if (diag && t == 2) continue;                    // lower-left block of a diagonal tile = mirror of block 1
    const double v = (red[0][t][le] + red[1][t][le]) + (red[2][t][le] + red[3][t][le]);
    const int l = le >> 2, reg = le & 3;
    const int row = (t >> 1) * 16 + (l >> 4) + 4 * reg;
    const int col = (t & 1) * 16 + (l & 15);
    tile[row][col] = v;
    if (diag && t == 1) tile[col][row] = v;
  }
  __syncthreads();
  double* __restrict__ G = d.G;
  const int64_t gld = d.ld;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int rr = e >> 5, cc = e & 31;
    G[(int64_t)(ti * 32 + rr) * gld + tj * 32 + cc] = tile[rr][cc];
    if (!diag) G[(int64_t)(tj * 32 + rr) * gld + ti * 32 + cc] = tile[cc][rr];
  }
  if (tj == d.nt - 1) {                              // zero the padding columns [Npad, ld) of row block ti
    const int padw = d.ld - d.Npad;
    for (int e = threadIdx.x; e < 32 * padw; e += 256) {
      const int rr = e / padw, cc = e - rr * padw;
      G[(int64_t)(ti * 32 + rr) * gld + d.Npad + cc] = 0.0;
    }
  }
}

// sum split-K partials, mirror, zero-pad.  local block = chunk of 1024 outputs of the [Npad][ld] image
__global__ __launch_bounds__(256) void gram_reduce_kernel(const GramDesc* __restrict__ descs,
                                                          const BlockRef* __restrict__ map,
                                                          const int32_t* __restrict__ skip) {
  const BlockRef br = map[blockIdx.x];
  if (skip && skip[br.prob]) return;
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
      const int64_t stride = (int64_t)ntp * 1024;
      int ks = 0;
      for (; ks + 8 <= d.ksplit; ks += 8) {          // independent loads in flight; the order of the adds is fixed
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(ks + u) * stride];
        v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
      }
      for (; ks < d.ksplit; ++ks) v += p[ks * stride];
    }
    d.G[idx] = v;
  }
}

void launch_gram_partial(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                         const int32_t* skip) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
}

void launch_gram_reduce(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                        const int32_t* skip) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
}

}  // namespace tadmm
