// Shared helpers of the direct eigen-solvers (tridiag.hip: problems of at most 64 columns; tridiag_mid.hip: the 128- and
// 192-column Rayleigh-Ritz problems of the filtered solver).
#pragma once
#include "common.h"
#include <cstdio>

namespace tadmm {

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kTN = 64;              // largest problem
constexpr int kTLd = kTN + 1;        // leading dimension of the LDS images (odd: conflict-free column walks)
constexpr int kTMaxCluster = 6;      // eigenvalues closer than kTClusterTol * ||T|| are orthogonalised against each other
constexpr double kTClusterTol = 1e-3;

template <int CTRL>
__device__ __forceinline__ double tdpp(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over each 16-lane DPP row, result in all 16 lanes
__device__ __forceinline__ double row16_sum(double v) {
  v += tdpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += tdpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += tdpp<0x141>(v);    // row_half_mirror
  v += tdpp<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ double quad_sum(double v) {
  v += tdpp<0xB1>(v);
  v += tdpp<0x4E>(v);
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

__device__ __forceinline__ double hash_pm1(uint32_t a, uint32_t b) {
  uint64_t x = ((uint64_t)a << 32) ^ (uint64_t)b;
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(int64_t)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}

#ifdef TADMM_TRI_STAMPS
__device__ long long g_tri_stamps[16];
#define TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tri_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define TSUB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long tnow = (long long)__builtin_readcyclecounter(); g_tri_stamps[8 + i] += tnow - tsub; tsub = tnow; } } while (0)
#else
#define TSTAMP(i) do { } while (0)
#define TSUB(i) do { } while (0)
#endif

}  // namespace

// ~2^-24 relative from v_rcp_f64 / v_rsq_f64, two Newton steps each: full fp64 without the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}
__device__ __forceinline__ double fast_sqrt(double x) {      // x > 0
  const double r = fast_rsqrt(x);
  const double s = x * r;
  return fma(fma(-s, s, x), 0.5 * r, s);
}


}  // namespace tadmm
