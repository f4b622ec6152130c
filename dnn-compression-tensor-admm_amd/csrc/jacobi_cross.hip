// Register/LDS-resident cross phase of the one-sided block Jacobi eigen-solver (see jacobi.hip for the
// algorithm).  One workgroup = one *super-pair* (A, B) of two 16-column super-blocks of X (stored
// transposed: XT[j][:] = column j):
//
//   * 16 waves; wave w keeps column A_w in REGISTERS (ld/64 doubles per lane, 16 B per lane per 1 KiB
//     chunk so every global/LDS access is a full-width coalesced / conflict-free 1 KiB row segment);
//   * the 16 columns of B live in LDS (16*ld*8 bytes: 64 KiB at N=512, so two workgroups share a CU);
//   * 16 barrier-separated steps; at step t wave w orthogonalises (A_w, B_{(w+t) mod 16}) by a direct
//     Hestenes rotation: g = <a,b> (wave all-reduce in DPP + readlane, no LDS), rotate both columns in
//     place (A in registers, B in LDS).  After 16 steps all 256 cross pairs of (A, B) have met once.
//
// No Gram matrix, no rotation accumulation, no separate apply pass: per launch the data makes one trip
// HBM -> registers/LDS -> HBM.  Squared column norms are carried along exactly
// (a' = c^2 a - 2cs g + s^2 b, b' = s^2 a + 2cs g + c^2 b) and recomputed from the data at every load.
// Pairs *inside* a super-block are rotated once per sweep by jacobi_tick_kernel in self mode.
#include "common.h"
#include <cstdio>

namespace tadmm {

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Sum over the 64 lanes, identical bits in every lane: xor-butterfly inside each row of 16 lanes on the
// DPP crossbar (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row
// sums are combined through scalar registers.
__device__ __forceinline__ double wave_allreduce_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ void rr_pair_x(int nb, int step, int q, int& a, int& b) {
  const int m = nb - 1;
  if (q == 0) { a = m; b = step % m; }
  else { a = (step + q) % m; b = (step - q + m) % m; }
}

constexpr int kSB = 2 * kJB;   // 16 columns per super-block

// Dynamic LDS (doubles): B[16][ld] | nB[16] | flags
template <int CHMAX>   // 16-byte chunks per lane: ld <= 128 * CHMAX
__global__ __launch_bounds__(1024) void jacobi_cross_kernel(const EigDesc* __restrict__ descs,
                                                            const BlockRef* __restrict__ map, int tick, double tol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const BlockRef br = map[blockIdx.x];
  const EigDesc d = descs[br.prob];
  if (*d.done) return;
  const int nbs = d.nb >> 1;                       // super-blocks of 16 columns
  const int steps = nbs - 1;
  const int sweep = tick / steps;
  const int step = tick - sweep * steps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (step == 0 && sweep > 0) {
    if (d.off[(sweep - 1) & 1] < tol) {           // previous sweep saw nothing left to rotate
      if (br.local == 0 && tid == 0) *d.done = 1;
      return;
    }
  }
  const int ld = d.ld;
  const int ch = ld >> 7;                          // ld is a multiple of 128
  double* Bs = smem;
  double* nB = Bs + kSB * ld;
  int sa, sb;
  rr_pair_x(nbs, step, br.local, sa, sb);
  double* __restrict__ rowA = d.XT + (int64_t)(sa * kSB + wave) * ld + 2 * lane;
  double* __restrict__ rowB = d.XT + (int64_t)(sb * kSB + wave) * ld + 2 * lane;

  // ---- load: A_w -> registers, B_w -> LDS, exact squared norms ----
  double2_t xa[CHMAX];
  double na = 0.0, nbw = 0.0;
#pragma unroll
  for (int m = 0; m < CHMAX; ++m) {
    if (m < ch) {
      xa[m] = *reinterpret_cast<const double2_t*>(rowA + 128 * m);
      const double2_t xb = *reinterpret_cast<const double2_t*>(rowB + 128 * m);
      *reinterpret_cast<double2_t*>(Bs + wave * ld + 128 * m + 2 * lane) = xb;
      na += xa[m].x * xa[m].x + xa[m].y * xa[m].y;
      nbw += xb.x * xb.x + xb.y * xb.y;
    } else {
      xa[m] = double2_t{0.0, 0.0};
    }
  }
  na = wave_allreduce_sum(na);
  nbw = wave_allreduce_sum(nbw);
  if (lane == 0) nB[wave] = nbw;
  __syncthreads();
  // every thread has taken its convergence decision by now: safe to clear the slot of the NEXT sweep
  if (step == steps - 1 && br.local == 0 && tid == 0) d.off[(sweep + 1) & 1] = 0.0;

  const double hmax = d.off[2];
  const double floor2 = hmax * 1e-28;              // lambda < 1e-14 lambda_max: padding / exact zeros
  const double wscale = 1e-14 / tol;
  const double inv_hmax = hmax > 0.0 ? 1.0 / hmax : 1.0;
  double mx = 0.0;
  int did = 0;
  for (int st = 0; st < kSB; ++st) {
    const int jb = (wave + st) & (kSB - 1);
    double* brow = Bs + jb * ld + 2 * lane;
    double2_t xb[CHMAX];
    double g = 0.0;
#pragma unroll
    for (int m = 0; m < CHMAX; ++m) {
      if (m < ch) {
        xb[m] = *reinterpret_cast<const double2_t*>(brow + 128 * m);
        g += xa[m].x * xb[m].x + xa[m].y * xb[m].y;
      }
    }
    g = wave_allreduce_sum(g);
    const double a = na, b = nB[jb];
    const double ab = a * b;
    const double hmin = fmin(a, b);
    if (hmin > floor2) {
      const double w = fmax(1.0, wscale * sqrt(hmax / hmin));     // lambda_max / lambda_min of the pair
      mx = fmax(mx, fabs(g) * rsqrt(ab) / w);
    }
    if (g * g > 1e-36 * ab && fabs(g) > 1e-300) {                 // wave-uniform branch
      // tan(theta) of the small-angle root estimated in fp32 (the pair is annihilated to ~1e-7 relative,
      // which does not slow the quadratic phase); c = (1+t^2)^-1/2 refined to fp64 so c^2+s^2 = 1.
      const float zf = (float)((b - a) * inv_hmax), wf = (float)(2.0 * g * inv_hmax);
      const float az = fabsf(zf), aw = fabsf(wf);
      float tf;
      if (az >= aw) {
        const float u = wf * __builtin_amdgcn_rcpf(az);
        tf = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(1.0f + u * u));
      } else {
        const float v = az * __builtin_amdgcn_rcpf(aw);
        tf = copysignf(__builtin_amdgcn_rcpf(v + __builtin_amdgcn_sqrtf(1.0f + v * v)), wf);
      }
      if (zf < 0.0f) tf = -tf;
      const double t = (double)tf;
      const double x = 1.0 + t * t;
      double c = __builtin_amdgcn_rsq(x);
      c = c * (1.5 - 0.5 * x * c * c);
      c = c * (1.5 - 0.5 * x * c * c);
      const double s = t * c;
      // new_a = c*a - s*b ; new_b = s*a + c*b
#pragma unroll
      for (int m = 0; m < CHMAX; ++m) {
        if (m < ch) {
          const double2_t va = xa[m], vb = xb[m];
          xa[m] = double2_t{c * va.x - s * vb.x, c * va.y - s * vb.y};
          *reinterpret_cast<double2_t*>(brow + 128 * m) = double2_t{s * va.x + c * vb.x, s * va.y + c * vb.y};
        }
      }
      const double c2 = c * c, s2 = s * s, csg = 2.0 * c * s * g;
      na = c2 * a - csg + s2 * b;
      if (lane == 0) nB[jb] = s2 * a + csg + c2 * b;
      did = 1;
    }
    __syncthreads();
  }
  if (lane == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(&d.off[sweep & 1]), (unsigned long long)__double_as_longlong(mx));
  // ---- store (skipped when nothing rotated in this workgroup) ----
  if (!__syncthreads_or(did)) return;
#pragma unroll
  for (int m = 0; m < CHMAX; ++m) {
    if (m < ch) {
      *reinterpret_cast<double2_t*>(rowA + 128 * m) = xa[m];
      *reinterpret_cast<double2_t*>(rowB + 128 * m) = *reinterpret_cast<const double2_t*>(Bs + wave * ld + 128 * m + 2 * lane);
    }
  }
}

size_t jacobi_cross_lds_bytes(int ld_max) { return ((size_t)kSB * ld_max + kSB + 2) * 8; }
int jacobi_cross_max_ld() { return 128 * 9; }

void launch_jacobi_cross(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                         int ld_max, hipStream_t s) {
  if (nblocks <= 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross_kernel<4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross_kernel<9>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  const size_t lds = jacobi_cross_lds_bytes(ld_max);
  if (ld_max <= 512)
    hipLaunchKernelGGL(jacobi_cross_kernel<4>, dim3(nblocks), dim3(1024), lds, s, descs_dev, map_dev, tick, tol);
  else
    hipLaunchKernelGGL(jacobi_cross_kernel<9>, dim3(nblocks), dim3(1024), lds, s, descs_dev, map_dev, tick, tol);
  const hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess)
    fprintf(stderr, "[tadmm] jacobi cross launch failed: %s (blocks=%d lds=%zu ld=%d)\n", hipGetErrorString(e), nblocks,
            lds, ld_max);
}

}  // namespace tadmm
