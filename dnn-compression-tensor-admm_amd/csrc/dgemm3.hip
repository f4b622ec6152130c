// Block products of the filtered eigen-solver at fp32 accuracy on the bf16 matrix cores.
//
//   C[j][i] = epilogue( sum_k Y[j][k] * G[i][k] )        Y: block image [rp][ldy] (fp64), G: Gram matrix (fp64)
//
// The same NT products dgemm.hip computes in fp64, for the filter stages that do not need fp64: the Chebyshev filter
// only has to ENRICH the block with the wanted eigen-directions; rounding errors of relative size 1e-7 in its products
// act as a perturbation that the later stages filter again.  The last stage of every problem, the polish, the
// Rayleigh-Ritz projection and the verification stay in fp64 (filter_host.h decides per stage, filter.hip makes sure on
// the device that no problem finishes without a full-precision stage), and the a-posteriori check with fallback is
// unchanged.  Numpy model of the scheme on the bench's spectra: all stages but the last at fp32 accuracy leave the
// verification figure at 1e-8..1e-6 (bar 1e-5) and Z at 1e-11..1e-9 of the fp64 result.
//
// Arithmetic: every operand value is rounded to fp32 and split exactly into three bf16 terms; the six partial products
// of weight >= 2^-16 go through v_mfma_f32_16x16x32_bf16 into one fp32 accumulator (chain.hip explains the scheme and
// its accuracy: that of an fp32 GEMM).  6 bf16 MFMAs cost 96 cycles per 16x16x32 block against 512 for the eight
// v_mfma_f64_16x16x4 of the fp64 kernel.
//
// G is packed once per level and iteration into three fragment-major bf16 planes (gplanes_kernel; one MFMA operand = one
// contiguous KiB, read straight from L2); the block rows are staged through LDS in chunks of 128 columns, converted and
// split on the way in.  A workgroup owns 32 block rows x 64 columns of C; epilogue modes 0 / 1 / 2 as in dgemm.hip
// (plain store, Chebyshev recurrence s0*acc + s1*P + s2*Q in fp64, store + Rayleigh-quotient partials).
#include "common.h"

namespace tadmm {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float4v_t __attribute__((ext_vector_type(4)));
typedef float float2v_t __attribute__((ext_vector_type(2)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4v_t __attribute__((ext_vector_type(4)));

constexpr int kTM = 32, kTN = 64, kNB = 1, kKC = 128, kSPC = kKC / 32, kPad = 8, kLd = kKC + kPad;

__device__ __forceinline__ const double* ring_sel(const DgemmDesc& d, int sel, const double* explicit_ptr, int base) {
  if (sel < 0) return explicit_ptr;
  int i = base + sel;
  i -= (i >= 3) ? 3 : 0;
  i -= (i >= 3) ? 3 : 0;
  // no runtime index, and prvalues: `c ? d.ring[0] : d.ring[1]` on lvalues selects between ADDRESSES inside the by-value
  // descriptor, which is a run-time index again and moves the whole copy to scratch
  const double* r0 = +d.ring[0]; const double* r1 = +d.ring[1]; const double* r2 = +d.ring[2];
  return i == 0 ? +r0 : (i == 1 ? +r1 : +r2);
}

// (x, y) -> three packed bf16 pairs with x = sum of the planes exactly (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ void split3(float x, float y, uint32_t (&o)[3]) {
  float2v_t v = {x, y};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    o[p] = __builtin_bit_cast(uint32_t, h);
    if (p < 2) v -= __builtin_convertvector(h, float2v_t);
  }
}

// One fragment block (16 rows x 32 columns of G) per wave: 8 consecutive doubles per lane -> three 16-byte pieces.
__global__ __launch_bounds__(256) void gplanes_kernel(const GPlaneDesc* __restrict__ descs,
                                                      const BlockRef* __restrict__ map) {
  const BlockRef br = map[blockIdx.x];
  const GPlaneDesc d = descs[br.prob];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fb = br.local * 4 + wave;
  if (fb >= d.nt * d.ks) return;
  const int ft = fb / d.ks, ks = fb - ft * d.ks;
  const int row = 16 * ft + (lane & 15), k = 32 * ks + 8 * (lane >> 4);
  const G<const double>* g = gp(d.Gm) + (int64_t)row * d.ldg + k;
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const double2_t t = *reinterpret_cast<const G<const double2_t>*>(g + i);
    v[i] = t.x; v[i + 1] = t.y;
  }
  uint32_t s[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3((float)v[2 * i], (float)v[2 * i + 1], s[i]);
  G<uint16_t>* out = gp(d.out) + ((int64_t)fb * 64 + lane) * 8;
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<G<uint4v_t>*>(out + p * d.plane) = uint4v_t{s[0][p], s[1][p], s[2][p], s[3][p]};
}

__global__ __launch_bounds__(256) void dgemm3_kernel(const DgemmDesc* __restrict__ descs,
                                                     const BlockRef* __restrict__ map) {
  __shared__ __attribute__((aligned(16))) uint16_t Xs[2 * 3 * kTM * kLd];     // [2 buffers][3 planes][kTM][kLd]: 51 KiB
  __shared__ double red[4][kTM];
  const BlockRef br = map[blockIdx.x];
  const DgemmDesc d = descs[br.prob];
  if (d.gate && *gp(d.gate) < d.gate_min) return;
  const int base = d.rot ? *gp(d.rot) : 0;
  const G<const double>* __restrict__ A = gp(ring_sel(d, d.selA, d.A, base));
  G<double>* __restrict__ C = gp(const_cast<double*>(ring_sel(d, d.selC, d.C, base)));
  const int tiles_n = (d.N + kTN - 1) / kTN;
  const int tm = br.local / tiles_n, tn = br.local - tm * tiles_n;
  const int m0 = tm * kTM, n0 = tn * kTN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int KS = d.K / 32;
  const int64_t lda = d.lda;
  const int nch = (KS + kSPC - 1) / kSPC;

  // this wave's columns of C = rows of G, fragment-major planes
  typedef const uint16_t __attribute__((address_space(1)))* gw_t;
  typedef const bf16x8_t __attribute__((address_space(1)))* gfrag_t;
  gw_t w[kNB];
#pragma unroll
  for (int j = 0; j < kNB; ++j) {
    int ft = n0 / 16 + wave * kNB + j;
    ft = min(ft, d.N / 16 - 1);                            // surplus tiles (N % kTN != 0) compute and are not stored
    w[j] = (gw_t)d.Gp + ((int64_t)ft * KS * 64 + lane) * 8;
  }
  const int64_t plane = d.g_plane;

  float4v_t acc[2][kNB];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < kNB; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
  // block rows -> LDS: 32 rows x 128 columns of doubles per chunk = 8 double2 per thread, converted and split.
  // Chunks past K are zero-filled (clamped address + select: a branch per load makes hipcc wait for vmcnt(0)).
  constexpr int kXL = kTM * kKC / 2 / 256;
  double2_t xa[kXL], xb[kXL];                              // two register sets: block-row chunks are fetched two ahead
  auto xload = [&](double2_t (&xr)[kXL], int k0) {
#pragma unroll
    for (int i = 0; i < kXL; ++i) {
      const int v = tid + 256 * i;
      const int row = v / (kKC / 2), k = k0 + 2 * (v % (kKC / 2));
      const double2_t t = *reinterpret_cast<const G<const double2_t>*>(A + (int64_t)(m0 + row) * lda + min(k, d.K - 2));
      xr[i] = k < d.K ? t : double2_t{0.0, 0.0};
    }
  };
  auto xstore = [&](const double2_t (&xr)[kXL], int buf) {
#pragma unroll
    for (int i = 0; i < kXL; ++i) {
      const int v = tid + 256 * i;
      const int row = v / (kKC / 2), kk = 2 * (v % (kKC / 2));
      uint32_t s[3];
      split3((float)xr[i].x, (float)xr[i].y, s);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint32_t*>(&Xs[((buf * 3 + p) * kTM + row) * kLd + kk]) = s[p];
    }
  };
  auto wload = [&](bf16x8_t (&b)[3][kNB], int s) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < kNB; ++j) b[p][j] = *(gfrag_t)(w[j] + p * plane + (int64_t)s * 512);
  };
  auto step = [&](int buf, int ks, const bf16x8_t (&b)[3][kNB]) {
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
    bf16x8_t a[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        a[p][mt] = *reinterpret_cast<const bf16x8_t*>(&Xs[((buf * 3 + p) * kTM + 16 * mt + r) * kLd + 32 * ks + 8 * q]);
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < kNB; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pb[pr]][j], a[pa[pr]][mt], acc[mt][j], 0, 0, 0);
  };
  // G fragments are requested THREE k-steps ahead (ring b0..b3, named so that nothing is indexed at run time), the block
  // rows TWO chunks ahead.  Nothing here branches on data: k-steps and chunks past K multiply zero-filled columns.
  bf16x8_t b0[3][kNB], b1[3][kNB], b2[3][kNB], b3[3][kNB];
  auto chunk = [&](int buf, int s) {                       // the four k-steps of the chunk in LDS buffer `buf`
    // sched_barrier: hipcc otherwise sinks the prefetch loads to just in front of their first use
    wload(b3, min(s + 3, KS - 1));
    __builtin_amdgcn_sched_barrier(0);
    step(buf, 0, b0);
    wload(b0, min(s + 4, KS - 1));
    __builtin_amdgcn_sched_barrier(0);
    step(buf, 1, b1);
    wload(b1, min(s + 5, KS - 1));
    __builtin_amdgcn_sched_barrier(0);
    step(buf, 2, b2);
    wload(b2, min(s + 6, KS - 1));
    __builtin_amdgcn_sched_barrier(0);
    step(buf, 3, b3);
    static_assert(kSPC == 4, "written out for four k-steps");
  };
  xload(xa, 0);
  xload(xb, kKC);
  wload(b0, 0);
  wload(b1, min(1, KS - 1));
  wload(b2, min(2, KS - 1));
  xstore(xa, 0);
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {
    xload(xa, (c + 2) * kKC);
    __builtin_amdgcn_sched_barrier(0);
    chunk(0, kSPC * c);
    xstore(xb, 1);
    __syncthreads();
    xload(xb, (c + 3) * kKC);
    __builtin_amdgcn_sched_barrier(0);
    chunk(1, kSPC * (c + 1));
    xstore(xa, 0);
    __syncthreads();
  }

  // ---- epilogue: lane holds columns f0..f0+3 of row t for each of its tiles
  const int mode = d.mode;
  double s0 = 1.0, s1 = 0.0, s2 = 0.0;
  if (mode == 1) { s0 = gp(d.coef)[0]; s1 = gp(d.coef)[1]; s2 = gp(d.coef)[2]; }
  const G<const double>* __restrict__ P = gp((mode >= 1) ? ring_sel(d, d.selP, d.P, base) : nullptr);
  const G<const double>* __restrict__ Q = gp((mode == 1 && s2 != 0.0) ? ring_sel(d, d.selQ, d.Q, base) : nullptr);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int t = m0 + 16 * mt + r;
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
      const int f0 = n0 + 16 * (wave * kNB + j) + 4 * q;   // N % 16 == 0: a lane's four columns are in or out together
      if (f0 < d.N && t < d.M) {
        const int64_t gi = (int64_t)t * d.ldc + f0;
        double o[4] = {(double)acc[mt][j][0], (double)acc[mt][j][1], (double)acc[mt][j][2], (double)acc[mt][j][3]};
        if (mode >= 1) {
          const double2_t p0 = *reinterpret_cast<const G<const double2_t>*>(P + gi);
          const double2_t p1 = *reinterpret_cast<const G<const double2_t>*>(P + gi + 2);
          if (mode == 1) {
            o[0] = s0 * o[0] + s1 * p0.x; o[1] = s0 * o[1] + s1 * p0.y;
            o[2] = s0 * o[2] + s1 * p1.x; o[3] = s0 * o[3] + s1 * p1.y;
            if (Q) {
              const double2_t q0 = *reinterpret_cast<const G<const double2_t>*>(Q + gi);
              const double2_t q1 = *reinterpret_cast<const G<const double2_t>*>(Q + gi + 2);
              o[0] += s2 * q0.x; o[1] += s2 * q0.y; o[2] += s2 * q1.x; o[3] += s2 * q1.y;
            }
          } else {
            part += (o[0] * p0.x + o[1] * p0.y) + (o[2] * p1.x + o[3] * p1.y);
          }
        }
        *reinterpret_cast<G<double2_t>*>(C + gi) = double2_t{o[0], o[1]};
        *reinterpret_cast<G<double2_t>*>(C + gi + 2) = double2_t{o[2], o[3]};
      }
    }
    if (mode == 2) {                                        // fixed-order sum over the workgroup's 128 columns
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      if (q == 0) red[wave][16 * mt + r] = part;
    }
  }
  if (mode == 2) {
    __syncthreads();
    if (tid < kTM && m0 + tid < d.M) {
      // dgemm.hip keeps one partial per 32-column tile ([tiles_n][M]): this workgroup covers four of them
      const int slot = (kTN / 32) * tn;
      d.rowpart[(int64_t)slot * d.M + m0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
#pragma unroll
      for (int u = 1; u < kTN / 32; ++u)
        if (slot + u < d.tiles_n) d.rowpart[(int64_t)(slot + u) * d.M + m0 + tid] = 0.0;
    }
  }
}

}  // namespace

void launch_gplanes(const GPlaneDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(gplanes_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

void launch_dgemm3(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(dgemm3_kernel, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev);
}

}  // namespace tadmm
