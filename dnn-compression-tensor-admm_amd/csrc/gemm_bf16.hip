// bf16 "NT" GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//
//   C[M][N] (bf16) = A[M][K] (bf16) * Bt[N][K]^T (bf16)  [+ bias_n[j] (f32)]
//
// Both operands are contiguous along K.  That is the one shape every product of the TT-linear forward chain
// has (TTLinear.py:79-86: `mm(out.reshape(-1, k), core.reshape(-1, k).t())` for the input modes,
// `mm(core.reshape(-1, r), out.reshape(-1, r).t())` for the output modes), so the bf16 inference path of
// TTLinearM needs nothing else ("bf16 MFMA path").
//
// Tile: 64x64 per workgroup, BK = 32, 4 waves in a 2x2 grid, each wave one 32x32 accumulator (2 MFMAs per BK).
// Operand fragments (cdna_hip_programming.md section 3): lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j],
// j = 0..7 -- eight consecutive K elements = one 16-byte LDS read; B likewise with its column index on the lane.
// LDS rows are padded to 40 bf16 (80 bytes) so the 16-byte reads of 32 consecutive rows spread over the banks.
#include "common.h"

namespace tadmm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float16v_t __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 64, TN = 64, TK = 32, LDS_LD = TK + 8;   // bf16 elements per LDS row

struct Bf16GemmArgs {
  const uint16_t* A; const uint16_t* Bt; uint16_t* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  const float* bias_n;
  int tiles_n;
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// 8 consecutive K elements of one row (zero beyond the matrix); vectorised when the row segment is 16-byte aligned
__device__ __forceinline__ uint4 load8(const uint16_t* __restrict__ P, int64_t ld, int row, int nrows, int k, int K,
                                      bool vec) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row >= nrows || k >= K) return v;
  const uint16_t* p = P + (int64_t)row * ld + k;
  if (vec && k + 8 <= K) return *reinterpret_cast<const uint4*>(p);
  uint16_t e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = (k + j < K) ? p[j] : (uint16_t)0;
  v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
  v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
  return v;
}

__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(const Bf16GemmArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t As[TM * LDS_LD];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[TN * LDS_LD];
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // loader mapping: thread t moves 8 K-elements of tile row t/4, K segment (t%4)*8
  const int lrow = tid >> 2, lk = (tid & 3) * 8;
  const bool veca = ((g.lda & 7) == 0) && ((((uintptr_t)g.A) & 15) == 0);
  const bool vecb = ((g.ldb & 7) == 0) && ((((uintptr_t)g.Bt) & 15) == 0);
  float16v_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint4 fa = load8(g.A, g.lda, m0 + lrow, g.M, lk, g.K, veca);
  uint4 fb = load8(g.Bt, g.ldb, n0 + lrow, g.N, lk, g.K, vecb);
  for (int k0 = 0; k0 < g.K; k0 += TK) {
    *reinterpret_cast<uint4*>(&As[lrow * LDS_LD + lk]) = fa;
    *reinterpret_cast<uint4*>(&Bs[lrow * LDS_LD + lk]) = fb;
    __syncthreads();
    if (k0 + TK < g.K) {      // next tile's loads are in flight during this tile's MFMAs
      fa = load8(g.A, g.lda, m0 + lrow, g.M, k0 + TK + lk, g.K, veca);
      fb = load8(g.Bt, g.ldb, n0 + lrow, g.N, k0 + TK + lk, g.K, vecb);
    }
#pragma unroll
    for (int s = 0; s < TK / 16; ++s) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&As[(wm * 32 + r) * LDS_LD + 16 * s + 8 * h]);
      const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&Bs[(wn * 32 + r) * LDS_LD + 16 * s + 8 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = n0 + wn * 32 + r;
  if (col < g.N) {
    const float bn = g.bias_n ? g.bias_n[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row < g.M) g.C[(int64_t)row * g.ldc + col] = f32_to_bf16_rne(acc[reg] + bn);
    }
  }
}

}  // namespace

void launch_gemm_bf16_nt(const void* A, const void* Bt, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                         int64_t ldc, const float* bias_n, hipStream_t s) {
  Bf16GemmArgs g;
  g.A = (const uint16_t*)A; g.Bt = (const uint16_t*)Bt; g.C = (uint16_t*)C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.bias_n = bias_n;
  const int tiles_m = (M + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(tiles_m * g.tiles_n), dim3(256), 0, s, g);
}

}  // namespace tadmm
