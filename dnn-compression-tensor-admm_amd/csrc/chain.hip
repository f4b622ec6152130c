// Forward chains of the factorised layers as ONE launch on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).
//
//   fused  : Y[t][:] = Wout * (Win * X[t][:]) + bias        TTLinearM (TTLinear.py:75-93): Win = the contracted input
//            cores (R x Kin), Wout = the contracted output cores (Nout x R), R = the middle TT rank.  The R-vector of
//            a token never leaves the CU: product 1 leaves it in LDS, product 2 reads it from there.
//   single : Y[t][:] = Win * X[t][:] + bias                  in-/out-core chains of TTConv2dM around the k x k core
//            (TTConv.py:130-153) and the 1x1 stages of TKConv2dC (TKConv.py:93-98), channels-first images read and
//            written in place (no NHWC copy).
//
// Two arithmetic modes, one code path (template parameter P = number of bf16 planes per operand):
//   P = 1  bf16 in / bf16 out, fp32 accumulate.
//   P = 3  fp32 in / fp32 out.  Every fp32 operand value is split EXACTLY into three bf16 terms x = x1 + x2 + x3
//          (8 significant bits each); the six partial products whose weight is >= 2^-16 of the leading one
//          (x1y1, x1y2, x2y1, x1y3, x3y1, x2y2) go through the bf16 matrix cores into one fp32 accumulator, smallest
//          first.  Each bf16 x bf16 product is exact in fp32, so the only error beside the fp32 accumulation the
//          fp32 matrix cores would also make is the three dropped terms (<= 2^-23 |x||y| together): measured against
//          fp64 it is as accurate as the fp32 GEMM (tests/test_gpu_chain.py).  Six bf16 MFMAs cost 6/16 of one fp32
//          MFMA of the same shape on gfx950 (2.5 PFLOP/s bf16 vs 157 TFLOP/s fp32).
//
// Work split: one workgroup (4 waves) per TM tokens.  MFMA operand roles are swapped (A = weight rows, B = tokens)
// so that a lane ends up with 4 CONSECUTIVE features of one token: 8/16-byte LDS and global stores, no transposes.
// Weights are read straight from global memory (they are L2-resident: 0.2-2 MB per layer) in FRAGMENT-MAJOR order:
// the 64 x 8 bf16 a wave needs for one MFMA operand are 1 KiB contiguous, so one global_load_dwordx4 per fragment is
// perfectly coalesced and every 128-byte line is used whole:
//     element (row n, col k) of plane p  ->  W[p * plane + (((n / 16) * KS + k / 32) * 64 + (k % 32 / 8) * 16 + n % 16) * 8 + k % 8]
// with KS = ceil(K / 32), rows padded to 16, columns to 32 (zeros).  The token tile is staged through LDS in chunks
// of KC columns, split into planes on the way in.  Weight fragments and LDS fragments of step s+1 are requested before
// the MFMAs of step s (one wave per SIMD: nothing else hides the L2 latency).  Workgroups walk the weights in rotated
// order (by blockIdx) so that the CUs of an XCD do not all ask the same L2 channel for the same line at once.
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

template <int P, int TM, int KC, int NB2, int KS2T, typename TIn, typename TOut, bool FUSED, bool XIMG, bool YIMG>
__global__ __launch_bounds__(256) void tt_chain_kernel(const ChainDesc d) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = TM / 16, LDX = KC + kPad, SPC = KC / 32;  // k-steps per chunk
  constexpr int XS_BYTES = 2 * P * TM * LDX * 2;               // both chunk buffers; reused as store staging
  // token fragments are double-buffered only in bf16 mode: with three planes a k-step is 6x as many MFMAs, the LDS
  // latency is a few percent of it, and the second fragment set would push the kernel into AGPR copies
  constexpr bool ADB = (P == 1);
  constexpr int AS = ADB ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * TM;
  const int n1_0 = blockIdx.y * (4 * kNB1 * 16);             // first product-1 feature of this workgroup
  uint16_t* Xs = lds;                                          // [2][P][TM][LDX]
  const int ldh = d.R + kPad;
  uint16_t* Hs = lds + 2 * P * TM * LDX;                       // [P][TM][ldh]   (fused mode)
  uint8_t* stage = reinterpret_cast<uint8_t*>(lds) + wave * (XS_BYTES / 4);
#ifdef TADMM_CHAIN_STAMPS
  int nstamp = 0;
#endif
  STAMP();

  // ---------------- product 1:  H (or Y) tile = Win[n1_0 + ...][:] x X-tile^T
  // The loop nest below is free of data-dependent branches on purpose: with a branch between a prefetch and the MFMAs
  // that do not need it, hipcc falls back to s_waitcnt vmcnt(0) and the prefetch is waited for at once.  Ragged edges
  // are handled by clamping (k-steps past Kin multiply zero-filled tokens; the last chunk prefetches chunk 0 again).
  float4v_t acc[MT][kNB1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < kNB1; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
  const int KS1 = (d.Kin + 31) / 32;
  const int ntiles1 = (d.R + 15) / 16;
  gw_t w1[kNB1];
#pragma unroll
  for (int j = 0; j < kNB1; ++j) {
    int ft = n1_0 / 16 + wave * kNB1 + j;
    ft = ft < ntiles1 ? ft : ntiles1 - 1;                      // surplus tiles compute and are not stored
    w1[j] = (gw_t)d.Win + ((int64_t)ft * KS1 * 64 + lane) * 8;
  }
  {
    ChunkLoader<P, TM, KC, TIn, XIMG> ld;
    const int nchunks = (d.Kin + KC - 1) / KC;
    const int rot = blockIdx.x % nchunks;                      // chunk order rotated per workgroup
    bf16x8_t a[AS][ADB ? P : 1][ADB ? MT : 1], a0[2][3], b[2][P][kNB1];
    ld.load(d, m0, rot * KC, tid);
    load_w<P, kNB1>(b[0], w1, d.win_plane, min(rot * SPC, KS1 - 1));
    ld.store(Xs, tid);
    __syncthreads();
    STAMP();
    if constexpr (ADB) load_x<P, TM>(a[0], Xs, LDX, 0, r, q);
    else load_x3<TM>(a0[0], Xs, LDX, 0, 0, r, q);
    int ca = rot;
    for (int c = 0; c < nchunks; ++c) {
      const int cn = (ca + 1 == nchunks) ? 0 : ca + 1;
      const uint16_t* Xc = Xs + (c & 1) * (P * TM * LDX);
      uint16_t* Xn = Xs + ((c + 1) & 1) * (P * TM * LDX);
      ld.load(d, m0, cn * KC, tid);                            // (the last trip re-reads a chunk nobody uses)
#pragma unroll
      for (int ks = 0; ks < SPC; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        const int snext = (ks + 1 < SPC) ? ca * SPC + ks + 1 : cn * SPC;
        load_w<P, kNB1>(b[nxt], w1, d.win_plane, min(snext, KS1 - 1));
        if constexpr (ADB) {
          if (ks + 1 < SPC) load_x<P, TM>(a[nxt], Xc, LDX, ks + 1, r, q);
          mma_step<P, MT, kNB1>(a[cur], b[cur], acc);
        } else {
          if (ks + 1 < SPC) load_x3<TM>(a0[nxt], Xc, LDX, ks + 1, 0, r, q);
          mma_stream3<TM, kNB1>(Xc, LDX, ks, r, q, a0[cur], b[cur], acc);
        }
      }
      static_assert(SPC % 2 == 0, "the fragment ring returns to slot 0 at every chunk boundary");
      STAMP();
      ld.store(Xn, tid);
      __syncthreads();
      STAMP();
      if constexpr (ADB) load_x<P, TM>(a[0], Xn, LDX, 0, r, q);
      else load_x3<TM>(a0[0], Xn, LDX, 0, 0, r, q);
      ca = cn;
    }
  }
  const bool yvec = d.y_vec != 0;

  if constexpr (!FUSED) {
    store_group<TM, kNB1, TOut, YIMG, XS_BYTES / 4>(d, acc, m0, n1_0 + wave * kNB1 * 16, d.R, stage, yvec, lane);
    return;
  } else {
    // H -> LDS, split into planes again (the fp32 chain of the reference rounds H to fp32 here as well)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < kNB1; ++j) {
        const int f0 = (wave * kNB1 + j) * 16 + 4 * q;
        if (f0 >= d.R) continue;
        uint32_t s0[P], s1[P];
        split2<P>(acc[mt][j][0], acc[mt][j][1], s0);
        split2<P>(acc[mt][j][2], acc[mt][j][3], s1);
#pragma unroll
        for (int p = 0; p < P; ++p)
          *reinterpret_cast<uint2*>(&Hs[(p * TM + 16 * mt + r) * ldh + f0]) = make_uint2(s0[p], s1[p]);
      }
    __syncthreads();
    STAMP();

    // ---------------- product 2:  Y tile = Wout x H-tile^T + bias, NB2 feature tiles per pass, R % 64 == 0
    const int ntiles2 = (d.Nout + 15) / 16;
    const int ngroups = (ntiles2 + NB2 - 1) / NB2;
    constexpr int KS2 = KS2T;                                    // = R / 32, even: the k-loop unrolls completely
    const int rot = blockIdx.x % ngroups;                        // group order rotated per workgroup
    auto bases = [&](int g, gw_t (&w2)[NB2]) {
#pragma unroll
      for (int j = 0; j < NB2; ++j) {
        int ft = g * NB2 + j;
        ft = ft < ntiles2 ? ft : ntiles2 - 1;
        w2[j] = (gw_t)d.Wout + ((int64_t)ft * KS2 * 64 + lane) * 8;
      }
    };
    auto group_of = [&](int pos) { const int g = pos + rot; return g >= ngroups ? g - ngroups : g; };
    bf16x8_t a[AS][ADB ? P : 1][ADB ? MT : 1], a0[2][3], b[2][P][NB2];
    gw_t w2[NB2], w2n[NB2];
    int pos = wave;
    if (pos < ngroups) {
      bases(group_of(pos), w2);
      load_w<P, NB2>(b[0], w2, d.wout_plane, 0);
      if constexpr (ADB) load_x<P, TM>(a[0], Hs, ldh, 0, r, q);
      else load_x3<TM>(a0[0], Hs, ldh, 0, 0, r, q);
    }
    while (pos < ngroups) {
      const int g = group_of(pos);
      const int npos = pos + 4;
      bases(group_of(npos < ngroups ? npos : pos), w2n);         // no next group: re-read this one's first step
      float4v_t acc2[MT][NB2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NB2; ++j) acc2[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
      // two k-steps per trip: the fragment ring (slot 0 / slot 1) is addressed statically, and every trip issues the
      // same loads whatever its position (the last one fetches the next group's first step)
#pragma unroll
      for (int ks = 0; ks < KS2; ks += 2) {
        const bool last = ks + 2 >= KS2;
        load_w<P, NB2>(b[1], w2, d.wout_plane, ks + 1);
        if constexpr (ADB) {
          load_x<P, TM>(a[1], Hs, ldh, ks + 1, r, q);
          mma_step<P, MT, NB2>(a[0], b[0], acc2);
        } else {
          load_x3<TM>(a0[1], Hs, ldh, ks + 1, 0, r, q);
          mma_stream3<TM, NB2>(Hs, ldh, ks, r, q, a0[0], b[0], acc2);
        }
        if (last) load_w<P, NB2>(b[0], w2n, d.wout_plane, 0);
        else load_w<P, NB2>(b[0], w2, d.wout_plane, ks + 2);
        if constexpr (ADB) {
          load_x<P, TM>(a[0], Hs, ldh, last ? 0 : ks + 2, r, q);
          mma_step<P, MT, NB2>(a[1], b[1], acc2);
        } else {
          load_x3<TM>(a0[0], Hs, ldh, last ? 0 : ks + 2, 0, r, q);
          mma_stream3<TM, NB2>(Hs, ldh, ks + 1, r, q, a0[1], b[1], acc2);
        }
      }
      STAMP();
      store_group<TM, NB2, TOut, YIMG, XS_BYTES / 4>(d, acc2, m0, g * NB2 * 16, d.Nout, stage, yvec, lane);
      STAMP();
#pragma unroll
      for (int j = 0; j < NB2; ++j) w2[j] = w2n[j];
      pos = npos;
    }
  }
}

template <int P, int TM, int KC, int NB2, int KS2T, typename TIn, typename TOut, bool FUSED, bool XIMG, bool YIMG>
int launch_variant(const ChainDesc& d, hipStream_t s) {
  auto kern = tt_chain_kernel<P, TM, KC, NB2, KS2T, TIn, TOut, FUSED, XIMG, YIMG>;
  size_t lds = (size_t)2 * P * TM * (KC + kPad) * 2;
  if (FUSED) lds += (size_t)P * TM * (d.R + kPad) * 2;
  if (lds > 160 * 1024) return -1;
  static bool attr_done[64] = {false};
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (!attr_done[devi & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_done[devi & 63] = true;
  }
  const int gx = (int)((d.T + TM - 1) / TM);
  const int gy = FUSED ? 1 : (d.R + 4 * kNB1 * 16 - 1) / (4 * kNB1 * 16);
  ChainDesc dd = d;
#ifdef TADMM_CHAIN_STAMPS
  static long long* stamps = nullptr;
  if (!stamps) { (void)hipMalloc(&stamps, 32 * 8 * 4096); (void)hipMemset(stamps, 0, 32 * 8 * 4096); }
  dd.stamps = (gx <= 4096) ? stamps : nullptr;
  if (getenv("TADMM_CHAIN_STAMPS_DUMP")) {
    static long long host[32 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, stamps, sizeof host, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; w += 7) {
      fprintf(stderr, "[stamps wg %d]", w);
      for (int i = 1; i < 32 && host[w * 32 + i]; ++i) fprintf(stderr, " %lld", host[w * 32 + i] - host[w * 32 + i - 1]);
      fprintf(stderr, "\n");
    }
  }
#endif
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, s, dd);
  return 0;
}

template <int P, int TM, int KC, typename T>
int launch_single(const ChainDesc& d, hipStream_t s) {
  if (d.x_hw > 0)
    return d.y_hw > 0 ? launch_variant<P, TM, KC, 1, 2, T, T, false, true, true>(d, s)
                      : launch_variant<P, TM, KC, 1, 2, T, T, false, true, false>(d, s);
  return d.y_hw > 0 ? launch_variant<P, TM, KC, 1, 2, T, T, false, false, true>(d, s)
                    : launch_variant<P, TM, KC, 1, 2, T, T, false, false, false>(d, s);
}

template <int KS2T>
int launch_fused(const ChainDesc& d, int dtype, int tile_tokens, hipStream_t s) {
  if (dtype == 1) {
    if (tile_tokens == 64) return launch_variant<1, 64, 128, 6, KS2T, uint16_t, uint16_t, true, false, false>(d, s);
    return launch_variant<1, 32, 128, 6, KS2T, uint16_t, uint16_t, true, false, false>(d, s);
  }
  if (tile_tokens == 32) return launch_variant<3, 32, 128, 6, KS2T, float, float, true, false, false>(d, s);
  return launch_variant<3, 64, 64, 3, KS2T, float, float, true, false, false>(d, s);
}
int launch_fused_ks(const ChainDesc& d, int dtype, int tile_tokens, hipStream_t s) {
  switch (d.R / 32) {
    case 2: return launch_fused<2>(d, dtype, tile_tokens, s);
    case 4: return launch_fused<4>(d, dtype, tile_tokens, s);
    case 6: return launch_fused<6>(d, dtype, tile_tokens, s);
    default: return launch_fused<8>(d, dtype, tile_tokens, s);
  }
}

}  // namespace

// dtype 0: fp32 in/out through three bf16 planes per operand; dtype 1: bf16 in/out.  Returns 0, or -1 when the shape
// does not fit the kernel (the caller reports it; there is no other path inside the library).
int launch_tt_chain(const ChainDesc& d, int dtype, int tile_tokens, hipStream_t s) {
  if (d.T <= 0) return 0;
  if (!d.fused) {
    if (dtype == 1) return launch_single<1, 64, 128, uint16_t>(d, s);
    return launch_single<3, 64, 64, float>(d, s);
  }
  if (d.x_hw > 0 || d.y_hw > 0 || d.R % 64 || d.R > 256) return -1;
  return launch_fused_ks(d, dtype, tile_tokens, s);
}

}  // namespace tadmm
