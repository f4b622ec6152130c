// Shared pieces of the forward-chain kernels (chain.hip, convchain.hip): three-plane bf16 split, MFMA step, LDS chunk
// loader, fragment loads, staged stores.  See chain.hip for the scheme.  Everything lives in an unnamed namespace: each
// translation unit gets its own copy.
#pragma once
#include "common.h"

namespace tadmm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float4v_t __attribute__((ext_vector_type(4)));
// weight pointers carry their address space: through arrays and selects hipcc otherwise degrades them to generic
// pointers, and flat loads count on vmcnt AND lgkmcnt -- every wait behind them becomes a full drain
typedef const uint16_t __attribute__((address_space(1)))* gw_t;
typedef const bf16x8_t __attribute__((address_space(1)))* gfrag_t;

namespace {

constexpr int kPad = 8;          // bf16 elements of row padding in LDS: row stride = 4 words mod 64 banks
constexpr int kNB1 = 4;          // feature tiles (16 wide) per wave in product 1: 256 features per workgroup pass

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2v_t){x, y}, bf16x2_t));
}
__device__ __forceinline__ uint16_t bf16_rne(float f) { return (uint16_t)pack_bf16(f, 0.f); }

// (x, y) -> P packed pairs with x = sum_p plane_p exactly (P == 3), or its rounding (P == 1)
template <int P> __device__ __forceinline__ void split2(float x, float y, uint32_t (&o)[P]) {
  float2v_t v = {x, y};
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    o[p] = __builtin_bit_cast(uint32_t, h);
    if (p + 1 < P) v -= __builtin_convertvector(h, float2v_t);
  }
}

// acc[mt][j] += sum over the kept plane pairs of  W-fragment(plane pb, tile j) x token-fragment(plane pa, tile mt)
template <int P, int MT, int NB>
__device__ __forceinline__ void mma_step(const bf16x8_t (&a)[P][MT], const bf16x8_t (&b)[P][NB],
                                         float4v_t (&acc)[MT][NB]) {
  if constexpr (P == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0][mt], acc[mt][j], 0, 0, 0);
  } else {
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pb[pr]][j], a[pa[pr]][mt], acc[mt][j], 0, 0, 0);
  }
}

// element address of (token t, channel c): row-major rows (hw == 0) or channels-first images of hw pixels
__device__ __forceinline__ int64_t elem_off(int64_t t, int c, int64_t ld, int hw, int nchan) {
  if (hw == 0) return t * ld + c;
  const int64_t b = t / hw;
  return (b * nchan + c) * hw + (t - b * hw);
}

// 4 consecutive features [f0, f0+4) of token t; partial groups and image layouts go element by element
template <typename TOut, bool YIMG>
__device__ __forceinline__ void store4(TOut* Y, int64_t t, int f0, int N, int64_t ldy, int hw, const float4v_t v,
                                       const float4 bv) {
  const float o[4] = {v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w};
  if (!YIMG && f0 + 4 <= N && (ldy & 3) == 0) {
    TOut* p = Y + t * ldy + f0;
    if constexpr (sizeof(TOut) == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]));
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (f0 + e < N) {
      TOut* p = Y + elem_off(t, f0 + e, ldy, YIMG ? hw : 0, N);
      if constexpr (sizeof(TOut) == 4) *p = o[e];
      else *p = bf16_rne(o[e]);
    }
}

// P planes of TM x KC token-tile columns [k0, k0+KC): global -> registers -> (split) -> LDS
template <int P, int TM, int KC, typename TIn, bool XIMG> struct ChunkLoader {
  static constexpr int EPL = 16 / sizeof(TIn);                 // elements per 16-byte load
  static constexpr int NV = TM * KC / EPL / 256;               // loads per thread
  static_assert(TM * KC / EPL % 256 == 0, "chunk must split evenly over 256 threads");
  uint4 regs[NV];

  __device__ __forceinline__ void load(const ChainDesc& d, int64_t m0, int k0, int tid) {
    const TIn* X = static_cast<const TIn*>(d.X);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      uint4 r = make_uint4(0, 0, 0, 0);
      if constexpr (!XIMG) {                                      // EPL consecutive channels of one token
        const int row = v / (KC / EPL), c = k0 + (v % (KC / EPL)) * EPL;
        const int64_t t = m0 + row;
        if (t < d.T && c < d.Kin) r = *reinterpret_cast<const uint4*>(X + t * d.ldx + c);
      } else {                                                    // EPL consecutive pixels of one channel
        const int c = k0 + v / (TM / EPL);
        const int64_t t = m0 + (v % (TM / EPL)) * EPL;
        if (c < d.Kin && t < d.T) {
          if (d.x_vec && t + EPL <= d.T) {
            r = *reinterpret_cast<const uint4*>(X + elem_off(t, c, 0, d.x_hw, d.Kin));
          } else {
            alignas(16) TIn e[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) e[j] = (t + j < d.T) ? X[elem_off(t + j, c, 0, d.x_hw, d.Kin)] : TIn(0);
            r = *reinterpret_cast<const uint4*>(e);
          }
        }
      }
      regs[i] = r;
    }
  }

  __device__ __forceinline__ void store(uint16_t* Xs, int tid) const {
    constexpr int LDX = KC + kPad;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      alignas(16) TIn e[EPL];
      *reinterpret_cast<uint4*>(e) = regs[i];
      if constexpr (!XIMG) {
        const int row = v / (KC / EPL), c = (v % (KC / EPL)) * EPL;
        if constexpr (P == 1) {
          *reinterpret_cast<uint4*>(&Xs[row * LDX + c]) = regs[i];
        } else {
          uint32_t s0[P], s1[P];
          split2<P>(e[0], e[1], s0);
          split2<P>(e[2], e[3], s1);
#pragma unroll
          for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(&Xs[(p * TM + row) * LDX + c]) = make_uint2(s0[p], s1[p]);
        }
      } else {
        const int c = v / (TM / EPL), row = (v % (TM / EPL)) * EPL;
#pragma unroll
        for (int j = 0; j < EPL; j += 2) {
          if constexpr (P == 1) {
            Xs[(row + j) * LDX + c] = reinterpret_cast<const uint16_t*>(e)[j];
            Xs[(row + j + 1) * LDX + c] = reinterpret_cast<const uint16_t*>(e)[j + 1];
          } else {
            uint32_t s[P];
            split2<P>(e[j], e[j + 1], s);
#pragma unroll
            for (int p = 0; p < P; ++p) {
              Xs[(p * TM + row + j) * LDX + c] = (uint16_t)s[p];
              Xs[(p * TM + row + j + 1) * LDX + c] = (uint16_t)(s[p] >> 16);
            }
          }
        }
      }
    }
  }
};

// fragments of NB weight tiles at k-step ks: one contiguous KiB per tile and plane
template <int P, int NB>
__device__ __forceinline__ void load_w(bf16x8_t (&b)[P][NB], gw_t (&base)[NB], int64_t plane, int ks) {
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < NB; ++j) b[p][j] = *(gfrag_t)(base[j] + p * plane + (int64_t)ks * 512);
}
// token fragments of k-step `kloc` of an LDS image [P][TM][ld]
template <int P, int TM>
__device__ __forceinline__ void load_x(bf16x8_t (&a)[P][TM / 16], const uint16_t* img, int ld, int kloc, int r, int q) {
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int mt = 0; mt < TM / 16; ++mt)
      a[p][mt] = *reinterpret_cast<const bf16x8_t*>(&img[(p * TM + 16 * mt + r) * ld + 32 * kloc + 8 * q]);
}

#ifdef TADMM_CHAIN_STAMPS
#define STAMP()                                                                                                  \
  do {                                                                                                           \
    if (d.stamps && tid == 0 && nstamp < 32)                                                                     \
      d.stamps[(int64_t)blockIdx.x * 32 + nstamp++] = (long long)__builtin_readcyclecounter();                   \
  } while (0)
#else
#define STAMP() do { } while (0)
#endif

// Epilogue of one feature group of one wave: NB tiles x TM tokens.  With `vec` the tile goes through a wave-private
// LDS staging area so that every store instruction writes whole 16-byte units of contiguous rows (token rows of
// NB*16 features, or -- image layout -- feature rows of TM pixels); the direct path writes the 4 features a lane
// holds (8/16-byte pieces, 16 tokens apart: the memory pipe takes those an order of magnitude slower).
template <int TM, int NB, typename TOut, bool YIMG, int STAGE_BYTES>
__device__ __forceinline__ void store_group(const ChainDesc& d, float4v_t (&acc)[TM / 16][NB], int64_t m0, int f_base,
                                            int N, uint8_t* stage, bool vec, int lane) {
  constexpr int MT = TM / 16, SZ = sizeof(TOut);
  const int r = lane & 15, q = lane >> 4;
  TOut* Y = static_cast<TOut*>(d.Y);
  float4 bq[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int f0 = f_base + 16 * j + 4 * q;
    bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias) {
      if (f0 + 4 <= N) bq[j] = *reinterpret_cast<const float4*>(d.bias + f0);
      else {
        if (f0 < N) bq[j].x = d.bias[f0];
        if (f0 + 1 < N) bq[j].y = d.bias[f0 + 1];
        if (f0 + 2 < N) bq[j].z = d.bias[f0 + 2];
      }
    }
  }
  if (!vec) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t t = m0 + 16 * mt + r;
      if (t >= d.T) continue;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int f0 = f_base + 16 * j + 4 * q;
        if (f0 < N) store4<TOut, YIMG>(Y, t, f0, N, d.ldy, d.y_hw, acc[mt][j], bq[j]);
      }
    }
    return;
  }
  if constexpr (!YIMG) {
    constexpr int ROWB = NB * 16 * SZ, SLD = ROWB + 16;            // staged token row, padded
    constexpr int TS0 = (STAGE_BYTES / SLD) / 16 * 16;
    constexpr int TS = TS0 > TM ? TM : TS0;                         // tokens per staging pass
    static_assert(TS >= 16, "staging area too small");
    constexpr int UPR = ROWB / 16, TPI = 64 / UPR;                  // 16-byte units per row, token rows per instruction
    const int tl = lane / UPR, u = lane - tl * UPR;
#pragma unroll
    for (int t0 = 0; t0 < TM; t0 += TS) {
#pragma unroll
      for (int mt = t0 / 16; mt < (t0 + TS) / 16 && mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          uint8_t* p = stage + (16 * mt - t0 + r) * SLD + (16 * j + 4 * q) * SZ;
          const float o0 = acc[mt][j][0] + bq[j].x, o1 = acc[mt][j][1] + bq[j].y, o2 = acc[mt][j][2] + bq[j].z,
                      o3 = acc[mt][j][3] + bq[j].w;
          if constexpr (SZ == 4) *reinterpret_cast<float4*>(p) = make_float4(o0, o1, o2, o3);
          else {
            *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
          }
        }
#pragma unroll
      for (int tt = 0; tt < TS && t0 + tt < TM; tt += TPI) {
        const int tok = tt + tl;
        const int64_t t = m0 + t0 + tok;
        const int f = f_base + u * (16 / SZ);
        if (tl < TPI && tok < TS && t0 + tok < TM && t < d.T && f < N)
          *reinterpret_cast<uint4*>(Y + t * d.ldy + f) = *reinterpret_cast<const uint4*>(stage + tok * SLD + u * 16);
      }
    }
  } else {
    // image layout: staged as [feature][TM pixels]; one instruction writes 64 / (TM*SZ/16) feature rows of TM pixels
    constexpr int ROWB = TM * SZ, SLD = ROWB + 16;
    constexpr int FS0 = (STAGE_BYTES / SLD) / 16 * 16;
    constexpr int FS = FS0 > NB * 16 ? NB * 16 : FS0;               // features per staging pass (whole tiles)
    static_assert(FS >= 16, "staging area too small");
    constexpr int UPR = ROWB / 16, FPI = 64 / UPR;
    const int fl = lane / UPR, u = lane - fl * UPR;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += FS / 16) {
#pragma unroll
      for (int j = j0; j < j0 + FS / 16 && j < NB; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float o[4] = {acc[mt][j][0] + bq[j].x, acc[mt][j][1] + bq[j].y, acc[mt][j][2] + bq[j].z,
                              acc[mt][j][3] + bq[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint8_t* p = stage + (16 * (j - j0) + 4 * q + e) * SLD + (16 * mt + r) * SZ;
            if constexpr (SZ == 4) *reinterpret_cast<float*>(p) = o[e];
            else *reinterpret_cast<uint16_t*>(p) = bf16_rne(o[e]);
          }
        }
#pragma unroll
      for (int ff = 0; ff < FS && 16 * j0 + ff < NB * 16; ff += FPI) {
        const int fi = ff + fl;
        const int f = f_base + 16 * j0 + fi;
        const int64_t t = m0 + u * (16 / SZ);
        if (fl < FPI && fi < FS && 16 * j0 + fi < NB * 16 && f < N && t < d.T)
          *reinterpret_cast<uint4*>(Y + elem_off(t, f, 0, d.y_hw, N)) = *reinterpret_cast<const uint4*>(stage + fi * SLD + u * 16);
      }
    }
  }
}

// ---- three-plane mode: token fragments stream through a two-slot ring, one 16-token tile (3 planes) at a time, so
// that only 24 registers of token fragments are live instead of 12 per tile of the workgroup's token block
template <int TM>
__device__ __forceinline__ void load_x3(bf16x8_t (&a)[3], const uint16_t* img, int ld, int kloc, int mt, int r, int q) {
#pragma unroll
  for (int p = 0; p < 3; ++p)
    a[p] = *reinterpret_cast<const bf16x8_t*>(&img[(p * TM + 16 * mt + r) * ld + 32 * kloc + 8 * q]);
}
template <int TM, int NB>
__device__ __forceinline__ void mma_stream3(const uint16_t* img, int ld, int kloc, int r, int q, const bf16x8_t (&a0)[3],
                                            const bf16x8_t (&b)[3][NB], float4v_t (&acc)[TM / 16][NB]) {
  constexpr int MT = TM / 16;
  constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
  bf16x8_t ring[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) ring[0][p] = a0[p];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt + 1 < MT) load_x3<TM>(ring[(mt + 1) & 1], img, ld, kloc, mt + 1, r, q);
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pb[pr]][j], ring[mt & 1][pa[pr]], acc[mt][j], 0, 0, 0);
  }
}

}  // namespace
}  // namespace tadmm
