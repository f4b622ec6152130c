// The factorised convolutions of small images in ONE launch:
//
//   y = W3 * conv_kxk( W1 * x ; Wc ) + bias          TTConv2dM (TTConv.py:130-153: input-core chain, k x k core
//                                                     convolution, output-core chain) and TKConv2dC / TKConv2dM
//                                                     (TKConv.py:93-98, :210-214: 1x1, k x k, 1x1)
//
// One workgroup per TILE OF OUTPUT ROWS of an image (at most 64 output pixels: the whole plane for 8x8 / 7x7, four rows of
// a 14x14 plane, two of a 28x28, one of a 56x56); it first builds H1 for the input rows the tile's taps reach (its halo:
// at most 192 pixels, recomputed by the neighbouring tiles), so the two intermediates (r1 and r2 channels per pixel, both
// <= 256) never leave LDS.  Three products on the bf16 matrix cores, fp32 through the exact three-plane split (chain.hip):
//   1. H1[pixel][r1]  = X[pixel][C] * W1^T             tokens = input pixels, X read in place from the NCHW tensor
//   2. H2[opixel][r2] = sum_tap H1[src(opixel, tap)][:] * Wc[tap]^T     K = taps * r1; the token fragment of a tap is the
//      H1 row of the shifted input pixel (a per-lane LDS gather), zero outside the image (padding)
//   3. Y[opixel][O]   = H2[opixel][r2] * W3^T + bias   written to the NCHW tensor
// Weights are fragment-major bf16 planes (chain.hip); Wc is packed as an (r2 x taps*r1) matrix, tap-major.
// Wider planes (more than 64 output columns) and intermediates beyond the LDS take the three-launch path (tadmm_ttconv_chain_in, the device library's conv2d, tadmm_ttconv_chain_out).
#include "chain_common.h"

namespace tadmm {
namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// gathered token fragments of one k-step: row src[mt] of the LDS image (or zeros), three or one plane
template <int P>
__device__ __forceinline__ void gather_x(bf16x8_t (&a)[P], const uint16_t* img, int prow, int ld, int row, int kloc, int q) {
  const int rr = row < 0 ? 0 : row;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(&img[(p * prow + rr) * ld + 32 * kloc + 8 * q]);
    if (row < 0) v = __builtin_bit_cast(bf16x8_t, u32x4_t{0u, 0u, 0u, 0u});
    a[p] = v;
  }
}

template <int P, int NB>
__device__ __forceinline__ void mma_tile(const bf16x8_t (&a)[P], const bf16x8_t (&b)[P][NB], float4v_t (&acc)[NB]) {
  if constexpr (P == 1) {
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0], acc[j], 0, 0, 0);
  } else {
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pb[pr]][j], a[pa[pr]], acc[j], 0, 0, 0);
  }
}

// acc (features 4q..4q+3 of token row0 + r, tile j) -> P planes of an LDS image [P][prow][ld]
template <int P, int TM, int NB>
__device__ __forceinline__ void acc_to_lds(const float4v_t (&acc)[TM / 16][NB], uint16_t* img, int prow, int row0, int ld,
                                           int f_base, int nfeat, int r, int q) {
#pragma unroll
  for (int mt = 0; mt < TM / 16; ++mt)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f0 = f_base + 16 * j + 4 * q;
      if (f0 >= nfeat) continue;
      uint32_t s0[P], s1[P];
      split2<P>(acc[mt][j][0], acc[mt][j][1], s0);
      split2<P>(acc[mt][j][2], acc[mt][j][3], s1);
#pragma unroll
      for (int p = 0; p < P; ++p)
        *reinterpret_cast<uint2*>(&img[(p * prow + row0 + 16 * mt + r) * ld + f0]) = make_uint2(s0[p], s1[p]);
    }
}

// Chunk of the image's input plane into the registers of a ChunkLoader (image layout): the workgroup's 64 token rows
// are the pixels of ONE image, so element (pixel, channel) sits at X + (img*C + c)*hw + pixel -- no division per element
// as in the general image loader, which matters for planes whose size is not a multiple of the vector width (7x7).
template <int P, int TM, int KC, typename T>
__device__ __forceinline__ void conv_load(ChunkLoader<P, TM, KC, T, true>& ld, const T* X, int img, int C, int hw,
                                          int p_off, int k0, bool vec, int tid) {
  constexpr int EPL = 16 / sizeof(T);
#pragma unroll
  for (int i = 0; i < ChunkLoader<P, TM, KC, T, true>::NV; ++i) {
    const int v = tid + 256 * i;
    const int c = k0 + v / (TM / EPL), p0 = p_off + (v % (TM / EPL)) * EPL;     // p0: pixel of the plane
    const T* base = X + ((int64_t)img * C + min(c, C - 1)) * hw;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (vec) {
      if (c < C && p0 < hw) r = *reinterpret_cast<const uint4*>(base + p0);        // hw % EPL == 0: inside or outside
    } else {
      alignas(16) T e[EPL];
#pragma unroll
      for (int j = 0; j < EPL; ++j) e[j] = (c < C && p0 + j < hw) ? base[p0 + j] : T(0);
      r = *reinterpret_cast<const uint4*>(e);
    }
    ld.regs[i] = r;
  }
}

// TM: pixels per workgroup (64, or 32 when three planes of a TT-rank intermediate would not fit the LDS otherwise).
// NBW: feature tiles (16 wide) per wave in products 1 and 2 (4 * NBW * 16 >= max(R1, R2)) and per pass in product 3
template <int P, int TM, int KC, int NBW, typename T>
__global__ __launch_bounds__(256) void tt_conv_kernel(const ConvChainDesc d) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = TM / 16, LDX = KC + kPad, SPC = KC / 32;
  constexpr int kStageBytes = 2 * P * TM * LDX * 2 / 4;    // per wave: a quarter of the chunk buffers (free after product 1)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int img = blockIdx.x / d.tiles, tile = blockIdx.x - img * d.tiles;
  const int hw_in = d.H * d.W, hw_out = d.Ho * d.Wo;
  // this workgroup's output rows and the input rows their taps reach (clipped to the image)
  const int oy0 = tile * d.TR, orows = min(d.TR, d.Ho - oy0), n_out = orows * d.Wo;
  const int iy_lo = max(0, oy0 * d.sh - d.ph), iy_hi = min(d.H - 1, (oy0 + orows - 1) * d.sh - d.ph + (d.kh - 1) * d.dh);
  const int n_in = max(0, iy_hi - iy_lo + 1) * d.W, p_in0 = iy_lo * d.W;
  const int ntt = (n_in + TM - 1) / TM;                    // token tiles of product 1 (<= d.NT)
  const int prow1 = TM * d.NT;
  uint16_t* Xs = lds;                                          // [2][P][64][LDX]
  const int ld1 = d.R1 + kPad, ld2 = d.R2 + kPad;
  uint16_t* H1s = lds + 2 * P * TM * LDX;                    // [P][64 * NT][ld1]
  uint16_t* H2s = H1s + P * prow1 * ld1;                       // [P][64][ld2]

  // ---------------- product 1: H1 = X W1^T over the halo's input pixels, 64 at a time (chain.hip, product 1)
  {
    const int KS1 = (d.C + 31) / 32, nt1 = d.R1 / 16;
    gw_t w1[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      int ft = wave * NBW + j;
      ft = ft < nt1 ? ft : nt1 - 1;
      w1[j] = (gw_t)d.W1 + ((int64_t)ft * KS1 * 64 + lane) * 8;
    }
    const int nchunks = (d.C + KC - 1) / KC;
    const T* X = static_cast<const T*>(d.X);
    for (int tt = 0; tt < ntt; ++tt) {
      const int p_off = p_in0 + TM * tt;                     // first plane pixel of this token tile
      const bool xvec = d.x_vec != 0 && (p_off % (16 / (int)sizeof(T))) == 0;
      float4v_t acc[MT][NBW];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
      ChunkLoader<P, TM, KC, T, true> ld;
      // weight fragments RD1 k-steps ahead (ring indexed statically: RD1 divides the k-steps of a chunk)
      constexpr int RD1 = (SPC % 4 == 0) ? 4 : 2;
      bf16x8_t b[RD1][P][NBW];
      conv_load<P, TM, KC, T>(ld, X, img, d.C, hw_in, p_off, 0, xvec, tid);
#pragma unroll
      for (int u = 0; u < RD1 - 1; ++u) load_w<P, NBW>(b[u], w1, d.w1_plane, min(u, KS1 - 1));
      ld.store(Xs, tid);
      __syncthreads();
      for (int c = 0; c < nchunks; ++c) {
        const uint16_t* Xc = Xs + (c & 1) * (P * TM * LDX);
        conv_load<P, TM, KC, T>(ld, X, img, d.C, hw_in, p_off, min(c + 1, nchunks - 1) * KC, xvec, tid);
#pragma unroll
        for (int ks = 0; ks < SPC; ++ks) {
          load_w<P, NBW>(b[(ks + RD1 - 1) % RD1], w1, d.w1_plane, min(c * SPC + ks + RD1 - 1, KS1 - 1));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            bf16x8_t a[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
              a[p] = *reinterpret_cast<const bf16x8_t*>(&Xc[(p * TM + 16 * mt + r) * LDX + 32 * ks + 8 * q]);
            mma_tile<P, NBW>(a, b[ks % RD1], acc[mt]);
          }
        }
        static_assert(SPC % RD1 == 0, "fragment ring returns to slot 0 at every chunk boundary");
        ld.store(Xs + ((c + 1) & 1) * (P * TM * LDX), tid);
        __syncthreads();
      }
      acc_to_lds<P, TM, NBW>(acc, H1s, prow1, TM * tt, ld1, wave * NBW * 16, d.R1, r, q);
    }
  }
  __syncthreads();

  // ---------------- product 2: the k x k core convolution as taps * R1/32 k-steps with gathered token rows
  {
    float4v_t acc[MT][NBW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NBW; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
    const int KSR = d.R1 / 32, taps = d.kh * d.kw, S = taps * KSR, nt2 = d.R2 / 16;
    gw_t w2[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      int ft = wave * NBW + j;
      ft = ft < nt2 ? ft : nt2 - 1;
      w2[j] = (gw_t)d.W2 + ((int64_t)ft * S * 64 + lane) * 8;
    }
    // output pixel of (mt, r) -> its coordinates; the source row of a tap is computed on the fly
    int oy[MT], ox[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int px = 16 * mt + r;
      oy[mt] = px < n_out ? oy0 + px / d.Wo : -(1 << 20);     // invalid output pixels gather nothing
      ox[mt] = px - (px / d.Wo) * d.Wo;
    }
    // weight fragments RD - 1 steps ahead, ring indexed statically (RD steps per trip); surplus steps of the last trip
    // multiply zero fragments
    constexpr int RD = (P == 1) ? 4 : 3;
    bf16x8_t b[RD][P][NBW];
#pragma unroll
    for (int u = 0; u < RD - 1; ++u) load_w<P, NBW>(b[u], w2, d.w2_plane, min(u, S - 1));
    int tap = 0, ks = 0;                                       // position of the current step
    for (int s = 0; s < S; s += RD) {
#pragma unroll
      for (int u = 0; u < RD; ++u) {
        load_w<P, NBW>(b[(u + RD - 1) % RD], w2, d.w2_plane, min(s + u + RD - 1, S - 1));
        const bool live = s + u < S;
        const int dy = tap / d.kw, dx = tap - dy * d.kw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int iy = oy[mt] * d.sh - d.ph + dy * d.dh, ix = ox[mt] * d.sw - d.pw + dx * d.dw;
          const bool in = live && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
          bf16x8_t a[P];
          gather_x<P>(a, H1s, prow1, ld1, in ? (iy - iy_lo) * d.W + ix : -1, ks, q);
          mma_tile<P, NBW>(a, b[u], acc[mt]);
        }
        ks += 1;
        if (ks == KSR) { ks = 0; tap += 1; }
        if (tap >= taps) { tap = taps - 1; }                  // keeps the surplus steps' addresses valid
      }
    }
    acc_to_lds<P, TM, NBW>(acc, H2s, TM, 0, ld2, wave * NBW * 16, d.R2, r, q);
  }
  __syncthreads();

  // ---------------- product 3: Y = H2 W3^T + bias over the output pixels, straight into the NCHW tensor
  {
    constexpr int NB3 = NBW;
    const int KS3 = d.R2 / 32, nt3 = (d.Nout + 15) / 16, ngroups = (nt3 + NB3 - 1) / NB3;
    T* Y = static_cast<T*>(d.Y);
    for (int g = wave; g < ngroups; g += 4) {
      float4v_t acc[MT][NB3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NB3; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
      gw_t w3[NB3];
#pragma unroll
      for (int j = 0; j < NB3; ++j) {
        int ft = g * NB3 + j;
        ft = ft < nt3 ? ft : nt3 - 1;
        w3[j] = (gw_t)d.W3 + ((int64_t)ft * KS3 * 64 + lane) * 8;
      }
      constexpr int RD = (P == 1) ? 4 : 3;
      bf16x8_t b[RD][P][NB3];
#pragma unroll
      for (int u = 0; u < RD - 1; ++u) load_w<P, NB3>(b[u], w3, d.w3_plane, min(u, KS3 - 1));
      for (int s = 0; s < KS3; s += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
          load_w<P, NB3>(b[(u + RD - 1) % RD], w3, d.w3_plane, min(s + u + RD - 1, KS3 - 1));
          const int t = s + u;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            bf16x8_t a[P];
            gather_x<P>(a, H2s, TM, ld2, t < KS3 ? 16 * mt + r : -1, min(t, KS3 - 1), q);
            mma_tile<P, NB3>(a, b[u], acc[mt]);
          }
        }
      }
      // Stores.  A group of feature rows is staged in a wave-private LDS area as [feature][n_out pixels] and copied out in
      // units of 16 bytes (whole plane per workgroup: the [Nout][hw] block of the image is contiguous in memory) or
      // 8 bytes (row tiles: every feature row of the tile is a separate run); the direct path (a lane's values one by
      // one, 2-4 bytes each) is the fallback for unaligned runs.
      constexpr int SZ = sizeof(T);
      const int f_base = g * NB3 * 16;
      const int nf = min(NB3 * 16, d.Nout - f_base);                         // features of this group that exist
      T* yblk = Y + ((int64_t)img * d.Nout + f_base) * hw_out + oy0 * d.Wo;   // first pixel of the tile, feature f_base
      const bool fits = nf * n_out * SZ <= kStageBytes;
      const bool vec16 = fits && d.tiles == 1 && (((uintptr_t)yblk) & 15) == 0 && ((nf * hw_out * SZ) & 15) == 0;
      const bool vec8 = fits && !vec16 && (((uintptr_t)yblk) & 7) == 0 && ((hw_out * SZ) & 7) == 0 && ((n_out * SZ) & 7) == 0;
      if (vec16 || vec8) {
        T* st = reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(lds) + wave * kStageBytes);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int px = 16 * mt + r;
#pragma unroll
          for (int j = 0; j < NB3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int fl = j * 16 + 4 * q + e;
              if (px < n_out && fl < nf) {
                const float v = acc[mt][j][e] + (d.bias ? d.bias[f_base + fl] : 0.f);
                if constexpr (SZ == 4) st[fl * n_out + px] = v;
                else st[fl * n_out + px] = bf16_rne(v);
              }
            }
        }
        if (vec16) {                                                           // n_out == hw_out: one contiguous block
          const int nvec = nf * hw_out * SZ / 16;
          for (int i = lane; i < nvec; i += 64)
            reinterpret_cast<uint4*>(yblk)[i] = reinterpret_cast<const uint4*>(st)[i];
        } else {
          const int upr = n_out * SZ / 8, nunits = nf * upr;                   // 8-byte units per feature row
          for (int i = lane; i < nunits; i += 64) {
            const int fl = i / upr, u = i - fl * upr;
            reinterpret_cast<uint2*>(yblk + (int64_t)fl * hw_out)[u] = reinterpret_cast<const uint2*>(st + fl * n_out)[u];
          }
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int px = 16 * mt + r;
          if (px >= n_out) continue;
#pragma unroll
          for (int j = 0; j < NB3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int f = f_base + j * 16 + 4 * q + e;
              if (f >= d.Nout) continue;
              const float v = acc[mt][j][e] + (d.bias ? d.bias[f] : 0.f);
              T* p = Y + ((int64_t)img * d.Nout + f) * hw_out + oy0 * d.Wo + px;
              if constexpr (SZ == 4) *p = v;
              else *p = bf16_rne(v);
            }
        }
      }
    }
  }
}

template <int P, int TM, int KC, int NBW, typename T>
int launch_conv_nbw(const ConvChainDesc& d, hipStream_t s) {
  auto kern = tt_conv_kernel<P, TM, KC, NBW, T>;
  const size_t lds = ((size_t)2 * P * TM * (KC + kPad) + (size_t)P * TM * d.NT * (d.R1 + kPad) + (size_t)P * TM * (d.R2 + kPad)) * 2;
  if (lds > 160 * 1024) return -1;
  static bool attr_done[64] = {false};
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (!attr_done[devi & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done[devi & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(d.B * d.tiles), dim3(256), lds, s, d);
  return 0;
}

template <int P, int TM, int KC, typename T>
int launch_conv_variant(const ConvChainDesc& d, hipStream_t s) {
  const int tiles = (d.R1 > d.R2 ? d.R1 : d.R2) / 16;                  // spread the feature tiles over the four waves
  if (tiles <= 4) return launch_conv_nbw<P, TM, KC, 1, T>(d, s);
  if (tiles <= 8) return launch_conv_nbw<P, TM, KC, 2, T>(d, s);
  return launch_conv_nbw<P, TM, KC, 4, T>(d, s);
}

}  // namespace

// dtype 0: fp32 through three bf16 planes; 1: bf16.  -1: the intermediates do not fit the LDS (the caller takes the
// three-launch path).
int launch_tt_conv(const ConvChainDesc& d, int dtype, hipStream_t s) {
  if (d.B <= 0) return 0;
  if (d.TM == 32) {
    if (dtype == 1) return launch_conv_variant<1, 32, 128, uint16_t>(d, s);
    return launch_conv_variant<3, 32, 64, float>(d, s);
  }
  if (dtype == 1) return launch_conv_variant<1, 64, 128, uint16_t>(d, s);
  return launch_conv_variant<3, 64, 64, float>(d, s);
}

}  // namespace tadmm
