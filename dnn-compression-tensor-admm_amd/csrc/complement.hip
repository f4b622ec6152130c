// Complement route of the truncated eigen-solve (Z-only mode): when a TT step keeps r of N left singular vectors and
// DISCARDS only k = N - r <= 64 of them (DeiT-small: every `proj` / `fc2` unfolding, N = 288, keep 256), the cheap
// subspace is the discarded one.  The filtered solver (filter.hip) is run on the reflected matrix
//
//        G' = c I - G,      c >= lambda_max(G),
//
// whose LEADING k eigenvectors are the TRAILING k of G; the kept basis is then any orthonormal basis of their
// orthogonal complement -- valid because the projection Z depends only on the kept SUBSPACE (reference ttd.py:21-26:
// T_{s+1} = U_r^T A; a rotation of U_r inside its span changes no later step, DESIGN.md 2 "Z-only").  Reference
// step replaced: the `svd` of ttd.py:17 behind admm.py:103-111 (prune_linear_rank_tt).
//
//   comp_prepare : c from a few power steps on G (x1.08, never below the largest diagonal entry) and the image of G'.
//                  An under-estimated c leaves G' with a negative eigenvalue; the filter's own checks (degenerate
//                  bounds, residual test, guard) then reject the problem and the full Jacobi solve of G runs instead.
//   comp_form    : C^T[j][:] = e_j - sum_c B_c[j] B_c      (j < r: the first r coordinate vectors projected off B)
//   (CholQR twice on C, the kernels of chol.hip, its Gram by the tile GEMM: orthonormal to rounding)
//   comp_emit    : U_keep -> the fp32 factor the projection GEMM reads; singular values are not available on this route
//                  (NaN in the plan's sigma slot).
// A Cholesky breakdown (the chosen coordinate vectors nearly inside span(B)) marks the problem bad like any other
// filter failure.
#include "common.h"

namespace tadmm {

namespace {
__device__ __forceinline__ double chash(uint32_t a, uint32_t b) {
  uint64_t x = ((uint64_t)a << 32) ^ (uint64_t)b;
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(int64_t)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}
}  // namespace

__global__ __launch_bounds__(1024) void comp_prepare_kernel(const CompDesc* __restrict__ descs, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) double csm[];
  __shared__ double red[16];
  const CompDesc d = descs[blockIdx.x];
  const int N = d.N, Npad = d.Npad, ldg = d.ldg;
  const double* __restrict__ G = d.G;
  double* z = csm;
  double* w = z + Npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto block_sum = [&](double v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
  };
  double dmax = 0.0;
  for (int i = tid; i < N; i += 1024) dmax = fmax(dmax, G[(int64_t)i * ldg + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
  if (lane == 0) red[wave] = dmax;
  __syncthreads();
  dmax = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) dmax = fmax(dmax, red[k]);
  __syncthreads();
  for (int i = tid; i < Npad; i += 1024) z[i] = i < N ? chash((uint32_t)blockIdx.x * 40503u + 7u, (uint32_t)i) : 0.0;
  __syncthreads();
  double rho = 0.0;
  for (int s = 0; s < nsteps; ++s) {
    for (int i0 = wave * 4; i0 < N; i0 += 64) {           // w = G z: four rows per wave in flight
      const double* r0 = G + (int64_t)i0 * ldg;
      const bool h1 = i0 + 1 < N, h2 = i0 + 2 < N, h3 = i0 + 3 < N;
      const double* r1 = h1 ? r0 + ldg : r0;
      const double* r2 = h2 ? r0 + 2 * (int64_t)ldg : r0;
      const double* r3 = h3 ? r0 + 3 * (int64_t)ldg : r0;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int k = lane; k < N; k += 64) {
        const double zv = z[k];
        a0 += r0[k] * zv; a1 += r1[k] * zv; a2 += r2[k] * zv; a3 += r3[k] * zv;
      }
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
      if (lane == 0) {
        w[i0] = a0;
        if (h1) w[i0 + 1] = a1;
        if (h2) w[i0 + 2] = a2;
        if (h3) w[i0 + 3] = a3;
      }
    }
    __syncthreads();
    double zz = 0.0, zw = 0.0, ww = 0.0;
    for (int i = tid; i < N; i += 1024) { zz += z[i] * z[i]; zw += z[i] * w[i]; ww += w[i] * w[i]; }
    zz = block_sum(zz); zw = block_sum(zw); ww = block_sum(ww);
    if (zz > 0.0) rho = fmax(rho, zw / zz);               // Rayleigh quotient: a lower bound of lambda_max
    if (!(ww > 0.0)) break;                               // uniform
    const double inv = 1.0 / sqrt(ww);
    for (int i = tid; i < N; i += 1024) z[i] = w[i] * inv;
    __syncthreads();
  }
  const double c = fmax(1.08 * rho, 1.001 * dmax);
  if (tid == 0) *d.cshift = c;
  double* __restrict__ Gc = d.Gc;
  const int64_t total = (int64_t)Npad * ldg;
  for (int64_t idx = tid; idx < total; idx += 1024) {
    const int i = (int)(idx / ldg), j = (int)(idx - (int64_t)i * ldg);
    Gc[idx] = (i < N && j < N) ? ((i == j ? c : 0.0) - G[idx]) : 0.0;
  }
}

// C^T image [r][ldy]: row j = e_j - sum_{c < k} UT[c][j] * UT[c][:]   (UT rows = the trailing eigenvectors of G)
__global__ __launch_bounds__(256) void comp_form_kernel(const CompDesc* __restrict__ descs, const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const CompDesc d = descs[br.prob];
  if (d.st->bad) return;
  const int j0 = br.local * 4;                            // four rows per workgroup
  const int N = d.N, ldy = d.ldy, k = d.k;
  __shared__ double bj[4][64];
  for (int t = threadIdx.x; t < 4 * k; t += 256) {
    const int jj = t / k, c = t - jj * k;
    bj[jj][c] = (j0 + jj < d.r) ? d.UT[(int64_t)c * ldy + j0 + jj] : 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ldy; i += 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (i < N) {
      for (int c = 0; c < k; ++c) {
        const double u = d.UT[(int64_t)c * ldy + i];
        a0 += bj[0][c] * u; a1 += bj[1][c] * u; a2 += bj[2][c] * u; a3 += bj[3][c] * u;
      }
    }
    const double acc[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + jj;
      if (j < d.r) d.Cimg[(int64_t)j * ldy + i] = (i < N) ? ((i == j ? 1.0 : 0.0) - acc[jj]) : 0.0;
    }
  }
}

// U_keep (rows of the orthonormalised C^T image) -> out_a[i * R + c]  (EigDesc mode 0), sigma slot -> NaN
__global__ __launch_bounds__(256) void comp_emit_kernel(const CompDesc* __restrict__ descs, const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const CompDesc d = descs[br.prob];
  if (d.st->bad) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = br.local * 4 + wave;
  if (c >= d.r) return;
  const double* row = d.Cimg + (int64_t)c * d.ldy;
  const int R = d.ldo ? d.ldo : d.r;
  for (int i = lane; i < d.N; i += 64) d.out_a[(int64_t)i * R + c] = (float)row[i];
  if (lane == 0 && d.sigma_layer) d.sigma_layer[c] = __builtin_nan("");
}

void launch_comp_prepare(const CompDesc* descs_dev, int nprob, int npad_max, int steps, hipStream_t s) {
  if (nprob <= 0) return;
  hipLaunchKernelGGL(comp_prepare_kernel, dim3(nprob), dim3(1024), (size_t)2 * npad_max * sizeof(double), s, descs_dev, steps);
}
void launch_comp_form(const CompDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(comp_form_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}
void launch_comp_emit(const CompDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(comp_emit_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
