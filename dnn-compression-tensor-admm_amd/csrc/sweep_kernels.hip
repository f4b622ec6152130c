// HBM sweeps of the ADMM iteration (reference admm.py:45, :96, :99, :73-76, :80-85):
//   unfold      : T0 = unfold(W + U)            conv: (O,I,k2) -> (O,k2,I), LDS-staged transpose
//   fold_update : Z = fold(Zmat); diff = W-Z; U += diff; ||diff||^2 (deterministic two-stage reduce)
//   penalty     : 0.5*rho*||W-Z+U||^2 and rho*(W-Z+U)
// All are bandwidth-bound: 16 B per weight element per iteration for unfold+fold_update
// (read W,U ; write Z,U) plus the fp32 scratch round trip of T0/Zmat.
#include "common.h"

namespace tadmm {

constexpr int kSweepThreads = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ double block_sum_256(double v, double* red /*4 doubles*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = (red[0] + red[1]) + (red[2] + red[3]);
  return t;  // valid on thread 0
}

// ---- conv layout change (O, I, k2) <-> (O, k2, I) through an LDS tile of one output channel x `ni` input channels ----
// Generic path: element-wise, any k2 / alignment.  Vector path (K2 known at compile time, everything a multiple of 4
// floats): 16-byte global accesses on both sides and no run-time division -- the element-wise path moved 2 TB/s.
__device__ __forceinline__ bool conv_vec_ok(const SweepDesc& d, int ni) {
  return (d.I & 3) == 0 && (ni & 3) == 0 && (d.ichunk & 3) == 0 &&
         ((((uintptr_t)d.W | (uintptr_t)d.U | (uintptr_t)d.Z | (uintptr_t)d.T0 | (uintptr_t)d.Zmat) & 15) == 0);
}
__device__ __forceinline__ int conv_ldt(const SweepDesc& d) { return d.ichunk + 4; }   // rows start 16-byte aligned

__device__ __forceinline__ void unfold_conv_any(const SweepDesc& d, int o, int i0, int ni, int use_u, float* tile, int tid) {
  const int K2 = d.K2;
  const int ldt = conv_ldt(d);
  const int64_t base = ((int64_t)o * d.I + i0) * K2;
  const float* w = d.W + base;
  const float* u = d.U + base;
  const int cnt = ni * K2;
  for (int e = tid; e < cnt; e += kSweepThreads) {
    const int il = e / K2, p = e - il * K2;
    float v = w[e];
    if (use_u) v += u[e];
    tile[p * ldt + il] = v;
  }
  __syncthreads();
  float* t = d.T0 + (int64_t)o * K2 * d.I + i0;
  for (int e = tid; e < cnt; e += kSweepThreads) {
    const int p = e / ni, il = e - p * ni;
    t[(int64_t)p * d.I + il] = tile[p * ldt + il];
  }
}

template <int K2>
__device__ __forceinline__ void unfold_conv_vec(const SweepDesc& d, int o, int i0, int ni, int use_u, float* tile, int tid) {
  const int ldt = conv_ldt(d);
  const int64_t base = ((int64_t)o * d.I + i0) * K2;
  const float4* w4 = reinterpret_cast<const float4*>(d.W + base);
  const float4* u4 = reinterpret_cast<const float4*>(d.U + base);
  const int cnt4 = (ni * K2) >> 2;
  for (int v = tid; v < cnt4; v += kSweepThreads) {
    float4 a = w4[v];
    if (use_u) { const float4 b = u4[v]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    int il = (4 * v) / K2, p = 4 * v - il * K2;                  // K2 is a constant: multiply-shift
    const float vals[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[p * ldt + il] = vals[j];
      if (++p == K2) { p = 0; ++il; }
    }
  }
  __syncthreads();
  float* t = d.T0 + (int64_t)o * K2 * d.I + i0;
  const int n4 = ni >> 2;                                         // float4 per output row
  for (int v = tid; v < K2 * n4; v += kSweepThreads) {
    const int p = v / n4, c = v - p * n4;
    *reinterpret_cast<float4*>(t + (int64_t)p * d.I + 4 * c) = *reinterpret_cast<const float4*>(tile + p * ldt + 4 * c);
  }
}

// fold + update of one tile: Zmat (k2 rows of ni) -> Z (ni x k2 contiguous), diff = W - Z, U += diff, sum diff^2
__device__ __forceinline__ double fold_conv_any(const SweepDesc& d, int o, int i0, int ni, int update_u, float* tile, int tid) {
  const int K2 = d.K2;
  const int ldt = conv_ldt(d);
  const int cnt = ni * K2;
  const float* zm = d.Zmat + (int64_t)o * K2 * d.I + i0;
  for (int e = tid; e < cnt; e += kSweepThreads) {
    const int p = e / ni, il = e - p * ni;
    tile[p * ldt + il] = zm[(int64_t)p * d.I + il];
  }
  __syncthreads();
  const int64_t base = ((int64_t)o * d.I + i0) * K2;
  const float* w = d.W + base;
  float* u = d.U + base;
  float* z = d.Z + base;
  double acc = 0.0;
  for (int e = tid; e < cnt; e += kSweepThreads) {
    const int il = e / K2, p = e - il * K2;
    const float zz = tile[p * ldt + il];
    const float df = w[e] - zz;
    z[e] = zz;
    if (update_u) u[e] += df;
    acc += (double)df * df;
  }
  return acc;
}

template <int K2>
__device__ __forceinline__ double fold_conv_vec(const SweepDesc& d, int o, int i0, int ni, int update_u, float* tile, int tid) {
  const int ldt = conv_ldt(d);
  const float* zm = d.Zmat + (int64_t)o * K2 * d.I + i0;
  const int n4 = ni >> 2;
  for (int v = tid; v < K2 * n4; v += kSweepThreads) {
    const int p = v / n4, c = v - p * n4;
    *reinterpret_cast<float4*>(tile + p * ldt + 4 * c) = *reinterpret_cast<const float4*>(zm + (int64_t)p * d.I + 4 * c);
  }
  __syncthreads();
  const int64_t base = ((int64_t)o * d.I + i0) * K2;
  const float4* w4 = reinterpret_cast<const float4*>(d.W + base);
  float4* u4 = reinterpret_cast<float4*>(d.U + base);
  float4* z4 = reinterpret_cast<float4*>(d.Z + base);
  const int cnt4 = (ni * K2) >> 2;
  double acc = 0.0;
  for (int v = tid; v < cnt4; v += kSweepThreads) {
    int il = (4 * v) / K2, p = 4 * v - il * K2;
    float zz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      zz[j] = tile[p * ldt + il];
      if (++p == K2) { p = 0; ++il; }
    }
    const float4 a = w4[v];
    const float dx = a.x - zz[0], dy = a.y - zz[1], dz = a.z - zz[2], dw = a.w - zz[3];
    z4[v] = float4{zz[0], zz[1], zz[2], zz[3]};
    if (update_u) {
      float4 c = u4[v];
      c.x += dx; c.y += dy; c.z += dz; c.w += dw;
      u4[v] = c;
    }
    acc += ((double)dx * dx + (double)dy * dy) + ((double)dz * dz + (double)dw * dw);
  }
  return acc;
}

__global__ __launch_bounds__(kSweepThreads) void unfold_kernel(const SweepDesc* __restrict__ descs,
                                                               const BlockRef* __restrict__ map, int use_u) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const BlockRef br = map[blockIdx.x];
  const SweepDesc d = descs[br.prob];
  const int tid = threadIdx.x;
  if (d.K2 == 1) {
    const int64_t e0 = (int64_t)br.local * d.ichunk;
    const int64_t e1 = min(e0 + (int64_t)d.ichunk, d.numel);
    const int64_t n = e1 - e0;
    const float* w = d.W + e0;
    const float* u = d.U + e0;
    float* t = d.T0 + e0;
    const bool vec = ((((uintptr_t)w | (uintptr_t)u | (uintptr_t)t) & 15) == 0);
    if (vec) {
      const int64_t n4 = n >> 2;
      for (int64_t i = tid; i < n4; i += kSweepThreads) {
        float4 a = reinterpret_cast<const float4*>(w)[i];
        if (use_u) {
          const float4 b = reinterpret_cast<const float4*>(u)[i];
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(t)[i] = a;
      }
      for (int64_t i = (n4 << 2) + tid; i < n; i += kSweepThreads) t[i] = use_u ? w[i] + u[i] : w[i];
    } else {
      for (int64_t i = tid; i < n; i += kSweepThreads) t[i] = use_u ? w[i] + u[i] : w[i];
    }
    return;
  }
  // conv: block = (o, chunk of input channels)
  const int o = br.local / d.nchunk;
  const int ch = br.local - o * d.nchunk;
  const int i0 = ch * d.ichunk;
  const int ni = min(d.ichunk, d.I - i0);
  if (d.K2 == 9 && conv_vec_ok(d, ni)) unfold_conv_vec<9>(d, o, i0, ni, use_u, tile, tid);
  else unfold_conv_any(d, o, i0, ni, use_u, tile, tid);
}

__global__ __launch_bounds__(kSweepThreads) void fold_update_kernel(const SweepDesc* __restrict__ descs,
                                                                    const BlockRef* __restrict__ map, int update_u,
                                                                    double* __restrict__ resid_partial) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ double red[4];
  const BlockRef br = map[blockIdx.x];
  const SweepDesc d = descs[br.prob];
  const int tid = threadIdx.x;
  double acc = 0.0;
  if (d.K2 == 1) {
    const int64_t e0 = (int64_t)br.local * d.ichunk;
    const int64_t e1 = min(e0 + (int64_t)d.ichunk, d.numel);
    const int64_t n = e1 - e0;
    const float* w = d.W + e0;
    const float* zm = d.Zmat + e0;
    float* u = d.U + e0;
    float* z = d.Z + e0;
    const bool vec = ((((uintptr_t)w | (uintptr_t)u | (uintptr_t)z | (uintptr_t)zm) & 15) == 0);
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t i = tid; i < n4; i += kSweepThreads) {
      const float4 a = reinterpret_cast<const float4*>(w)[i];
      const float4 b = reinterpret_cast<const float4*>(zm)[i];
      reinterpret_cast<float4*>(z)[i] = b;
      const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
      if (update_u) {
        float4 c = reinterpret_cast<float4*>(u)[i];
        c.x += dx; c.y += dy; c.z += dz; c.w += dw;
        reinterpret_cast<float4*>(u)[i] = c;
      }
      acc += (double)dx * dx + (double)dy * dy + (double)dz * dz + (double)dw * dw;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += kSweepThreads) {
      const float zz = zm[i];
      const float df = w[i] - zz;
      z[i] = zz;
      if (update_u) u[i] += df;
      acc += (double)df * df;
    }
  } else {
    const int o = br.local / d.nchunk;
    const int ch = br.local - o * d.nchunk;
    const int i0 = ch * d.ichunk;
    const int ni = min(d.ichunk, d.I - i0);
    acc = (d.K2 == 9 && conv_vec_ok(d, ni)) ? fold_conv_vec<9>(d, o, i0, ni, update_u, tile, tid)
                                            : fold_conv_any(d, o, i0, ni, update_u, tile, tid);
  }
  const double t = block_sum_256(acc, red);
  if (tid == 0) resid_partial[d.blk_begin + br.local] = t;
}

// one wave per layer, fixed summation order -> run-to-run deterministic
__global__ __launch_bounds__(64) void resid_reduce_kernel(const SweepDesc* __restrict__ descs,
                                                          const double* __restrict__ resid_partial,
                                                          double* __restrict__ resid_sq,
                                                          const int32_t* __restrict__ out_index) {
  const SweepDesc d = descs[blockIdx.x];
  double acc = 0.0;
  for (int i = threadIdx.x; i < d.nblk; i += 64) acc += resid_partial[d.blk_begin + i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) resid_sq[out_index ? out_index[blockIdx.x] : blockIdx.x] = acc;
}

void launch_unfold(const SweepDesc* descs_dev, const BlockRef* map_dev, int nblocks, int use_u, hipStream_t s) {
  if (nblocks <= 0) return;
  // dynamic LDS sized for the largest conv tile: the host sizes ichunk so that K2*(ichunk+4) <= 12288 floats
  hipLaunchKernelGGL(unfold_kernel, dim3(nblocks), dim3(kSweepThreads), 49152, s, descs_dev, map_dev, use_u);
}

void launch_fold_update(const SweepDesc* descs_dev, const BlockRef* map_dev, int nblocks, int update_u,
                        double* resid_partial_dev, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(fold_update_kernel, dim3(nblocks), dim3(kSweepThreads), 49152, s, descs_dev, map_dev, update_u,
                     resid_partial_dev);
}

void launch_resid_reduce(const SweepDesc* descs_dev, int nlayers, const double* resid_partial_dev,
                         double* resid_sq_dev, hipStream_t s, const int32_t* out_index) {
  if (nlayers <= 0) return;
  hipLaunchKernelGGL(resid_reduce_kernel, dim3(nlayers), dim3(64), 0, s, descs_dev, resid_partial_dev, resid_sq_dev,
                     out_index);
}

// ---------------------------------------------------------------- penalty (admm.py:80-85)
// ptrs: [W_0..W_{n-1} | Z_0.. | U_0.. | G_0..] ; elements are walked as one concatenated index space so
// that one launch covers every layer.  Block partials are reduced in a fixed order by the last kernel.
__global__ __launch_bounds__(kSweepThreads) void penalty_kernel(int n, const void* const* __restrict__ ptrs,
                                                                const int64_t* __restrict__ numel, int64_t total,
                                                                float gscale, double* __restrict__ partial) {
  __shared__ double red[4];
  // balanced split of the concatenated element space over the grid (multiples of 4 elements)
  const int64_t per = (((total + gridDim.x - 1) / gridDim.x) + 3) & ~(int64_t)3;
  int64_t g0 = (int64_t)blockIdx.x * per;
  const int64_t g1 = min(g0 + per, total);
  double acc = 0.0;
  // locate the first layer
  int layer = 0;
  int64_t lbeg = 0;
  while (layer < n && lbeg + numel[layer] <= g0) { lbeg += numel[layer]; ++layer; }
  while (g0 < g1 && layer < n) {
    const int64_t lend = lbeg + numel[layer];
    const int64_t e0 = g0 - lbeg, e1 = min(g1, lend) - lbeg;
    const float* w = (const float*)ptrs[layer];
    const float* z = (const float*)ptrs[n + layer];
    const float* u = (const float*)ptrs[2 * n + layer];
    float* g = (float*)ptrs[3 * n + layer];
    // 16-byte accesses on the aligned middle of the segment (torch allocations are 256-byte aligned and `per` is
    // a multiple of 4, so whole layers take this path), scalar head / tail otherwise
    const bool al = ((((uintptr_t)w) | ((uintptr_t)z) | ((uintptr_t)u) | ((uintptr_t)g)) & 15) == 0;
    const int64_t v0 = al ? min(e1, (e0 + 3) & ~(int64_t)3) : e1;       // first 4-aligned element
    const int64_t v1 = al ? (v0 + ((e1 - v0) & ~(int64_t)3)) : e1;       // end of the vector part
    for (int64_t e = e0 + threadIdx.x; e < v0; e += kSweepThreads) {
      const float dlt = w[e] - z[e] + u[e];
      acc += (double)dlt * dlt;
      if (g) g[e] = gscale * dlt;
    }
    for (int64_t e = v0 + 4 * (int64_t)threadIdx.x; e < v1; e += 4 * kSweepThreads) {
      const float4 a = *reinterpret_cast<const float4*>(w + e);
      const float4 b = *reinterpret_cast<const float4*>(z + e);
      const float4 c = *reinterpret_cast<const float4*>(u + e);
      const float4 dl = make_float4(a.x - b.x + c.x, a.y - b.y + c.y, a.z - b.z + c.z, a.w - b.w + c.w);
      acc += ((double)dl.x * dl.x + (double)dl.y * dl.y) + ((double)dl.z * dl.z + (double)dl.w * dl.w);
      if (g) *reinterpret_cast<float4*>(g + e) = make_float4(gscale * dl.x, gscale * dl.y, gscale * dl.z, gscale * dl.w);
    }
    for (int64_t e = v1 + threadIdx.x; e < e1; e += kSweepThreads) {
      const float dlt = w[e] - z[e] + u[e];
      acc += (double)dlt * dlt;
      if (g) g[e] = gscale * dlt;
    }
    g0 = lbeg + e1;
    lbeg = lend;
    ++layer;
  }
  const double t = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void penalty_reduce_kernel(int nblk, float rho, const double* __restrict__ partial,
                                                            double* __restrict__ loss) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] += 0.5 * (double)rho * acc;
}

void launch_penalty(int n, const void* const* ptrs_dev, const int64_t* numel_dev, int64_t total, float rho,
                    float gscale, double* loss_dev, double* partial_dev, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(penalty_kernel, dim3(kPenaltyBlocks), dim3(kSweepThreads), 0, s, n, ptrs_dev, numel_dev, total,
                     gscale, partial_dev);
  hipLaunchKernelGGL(penalty_reduce_kernel, dim3(1), dim3(64), 0, s, kPenaltyBlocks, rho, partial_dev, loss_dev);
}

}  // namespace tadmm
