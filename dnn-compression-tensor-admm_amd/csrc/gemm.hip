// Grouped strided GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact-f32 fma chain).
//
//   C(i,j) = alpha * sum_k A(i,k) * B(k,j)  [+ beta * C(i,j)] [+ bias_n[j]] [+ bias_m[i]]
//
// Every operand is addressed with (row stride, col stride) so one kernel serves all contractions of
// the path without materialised transposes: U_r^T * T (TT projection), T * (V Sigma^-1) (core of a
// tall unfolding), the tt2ten chain (ttd.py:39-40), and the forward chains of the factorised layers
// (TTLinear.py:79-86, TTConv.py:133-147, TKConv.py:210-214, TKLinear.py:66-71).
//
// Tile: 64x64 per workgroup, BK=16, 4 waves in a 2x2 grid, each wave one 32x32 MFMA accumulator.
// LDS images are k-major (As[k][m], Bs[k][n]) so the MFMA operand reads (lane = row/col index) are
// conflict-free ds_read_b32; the global->LDS path picks its lane mapping from whichever stride is 1
// so global loads stay coalesced for both "N" and "T" operands.  Next tile's global loads are issued
// before the current tile's MFMAs (register double buffering).
#include "common.h"

namespace tadmm {

typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int BM = kGemmBM, BN = kGemmBN, BK = 16;
constexpr int LDA = BM + 4, LDB = BN + 4;

struct Frag4 { float v[4]; };

// load 4 elements of a (rows x BK) tile.  kmajor==true: the operand is contiguous along k.
__device__ __forceinline__ Frag4 load_tile4(const float* __restrict__ P, int64_t rs, int64_t cs, bool kcontig,
                                            int row0, int nrows, int k0, int K, int tid) {
  // returns values for (row, k) pairs given by the mapping below
  Frag4 f;
  if (kcontig) {
    const int rr = tid >> 2, kk = (tid & 3) * 4;           // 64 rows x 4 groups of 4 k
    const int row = row0 + rr;
    const bool rok = row < nrows;
    const float* p = P + (int64_t)row * rs + (int64_t)(k0 + kk) * cs;   // cs == 1 here
    if (rok && (k0 + kk + 3) < K && ((((uintptr_t)p) & 15) == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) f.v[e] = (rok && (k0 + kk + e) < K) ? p[e] : 0.f;
    }
  } else {
    const int kk = tid >> 4, rr = (tid & 15) * 4;          // 16 k x 16 groups of 4 rows
    const int k = k0 + kk;
    const bool kok = k < K;
    const float* p = P + (int64_t)(row0 + rr) * rs + (int64_t)k * cs;   // rs == 1 here
    if (kok && (row0 + rr + 3) < nrows && ((((uintptr_t)p) & 15) == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) f.v[e] = (kok && (row0 + rr + e) < nrows) ? p[e] : 0.f;
    }
  }
  return f;
}

__device__ __forceinline__ void store_tile4(float* __restrict__ S, int ldS, bool kcontig, const Frag4& f, int tid) {
  if (kcontig) {
    const int rr = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) S[(kk + e) * ldS + rr] = f.v[e];
  } else {
    const int kk = tid >> 4, rr = (tid & 15) * 4;
    *reinterpret_cast<float4*>(&S[kk * ldS + rr]) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  }
}

// one 64x64 output tile (`local` = tile index inside the problem).  kComp: chunked accumulation with an fp64 total
// (the product default); false = one fp32 accumulator over all of K (TADMM_GEMM_PLAIN=1: A/B measurements only)
template <bool kComp>
__device__ __forceinline__ void gemm_tile(const GemmDesc& d, int local, float* __restrict__ As, float* __restrict__ Bs) {
  const int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const bool a_k = (d.a_cs == 1), b_k = (d.b_rs == 1) && (d.b_cs != 1);
  // For B the "rows" of the tile loader are the N index: element (n,k) at B + k*b_rs + n*b_cs
  const int K = d.K;
  // Accumulation.  One fp32 MFMA accumulator over all of K rounds K times; when every product is equal (constant or
  // rank-1 weights: all partial sums round the same way) the error grows like K eps instead of sqrt(K) eps and reached
  // 1.3e-5 of ||W|| at K = 4608 -- above the 1e-5 parity bar.  The K range is therefore accumulated in chunks of 4 BK = 64
  // (32 MFMAs); a finished chunk is added to a second set of 16 registers on the vector ALU and the accumulator restarts
  // from zero: the rounding bias of a chunk is at most 64 eps / 2 of ITS sum, that of the K / 64 additions of chunks the
  // same again -- two orders of magnitude below the single-accumulator figure at K = 4608 (measured: <= 4e-7).  A first
  // version with two alternating accumulators and an fp64 total cost 25 % of the kernel in-plan (32 more live
  // registers); this one waits once per chunk for the last MFMA (~3 %).
  const float16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float16_t acc0 = zero16, tot = zero16;

  Frag4 fa = load_tile4(d.A, d.a_rs, d.a_cs, a_k, m0, d.M, 0, K, tid);
  Frag4 fb = load_tile4(d.B, d.b_cs, d.b_rs, b_k, n0, d.N, 0, K, tid);
  const int ai = wm * 32 + (lane & 31), bj = wn * 32 + (lane & 31), kq = lane >> 5;
  int chunk = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    store_tile4(As, LDA, a_k, fa, tid);
    store_tile4(Bs, LDB, b_k, fb, tid);
    __syncthreads();
    if (k0 + BK < K) {
      fa = load_tile4(d.A, d.a_rs, d.a_cs, a_k, m0, d.M, k0 + BK, K, tid);
      fb = load_tile4(d.B, d.b_cs, d.b_rs, b_k, n0, d.N, k0 + BK, K, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[(kk + kq) * LDA + ai];
      const float b = Bs[(kk + kq) * LDB + bj];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    }
    if (kComp && ++chunk == 4) {
      chunk = 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) tot[e] += acc0[e];
      acc0 = zero16;
    }
    __syncthreads();
  }
  float16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = kComp ? tot[e] + acc0[e] : acc0[e];

  // epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
  const int col = n0 + wn * 32 + (lane & 31);
  if (col < d.N) {
    const float bn = d.bias_n ? d.bias_n[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (row < d.M) {
        float* c = d.C + (int64_t)row * d.c_rs + (int64_t)col * d.c_cs;
        float v = d.alpha * acc[reg] + bn;
        if (d.bias_m) v += d.bias_m[row];
        if (d.beta != 0.f) v += d.beta * (*c);
        *c = v;
      }
    }
  }
}

template <bool kComp>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmDesc* __restrict__ descs,
                                                   const BlockRef* __restrict__ map,
                                                   const int32_t* __restrict__ skip) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  const BlockRef br = map[blockIdx.x];
  if (skip && skip[br.prob]) return;
  const GemmDesc d = descs[br.prob];
  gemm_tile<kComp>(d, br.local, As, Bs);
}

// A single GEMM whose descriptor travels as a kernel argument: no descriptor upload, no block map
// (the per-call path of the factorised layers' forward / backward products).
template <bool kComp>
__global__ __launch_bounds__(256) void gemm_one_kernel(const GemmDesc d) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  gemm_tile<kComp>(d, blockIdx.x, As, Bs);
}

static bool gemm_plain() {
  const char* e = getenv("TADMM_GEMM_PLAIN");
  return e && atoi(e);
}

void launch_gemm_one(const GemmDesc& d, hipStream_t s) {
  const int nblocks = d.tiles_m * d.tiles_n;
  if (nblocks <= 0) return;
  if (gemm_plain()) hipLaunchKernelGGL(gemm_one_kernel<false>, dim3(nblocks), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(gemm_one_kernel<true>, dim3(nblocks), dim3(256), 0, s, d);
}

void launch_gemm(const GemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s, const int32_t* skip) {
  if (nblocks <= 0) return;
  if (gemm_plain()) hipLaunchKernelGGL(gemm_kernel<false>, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
  else hipLaunchKernelGGL(gemm_kernel<true>, dim3(nblocks), dim3(256), 0, s, descs_dev, map_dev, skip);
}

}  // namespace tadmm
