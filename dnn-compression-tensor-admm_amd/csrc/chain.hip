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

namespace {

constexpr int kPad = 8;          // bf16 elements of row padding in LDS: row stride = 4 words mod 64 banks
constexpr int kNB1 = 4;          // feature tiles (16 wide) per wave in product 1: 256 features per workgroup pass

__device__ __forceinline__ uint16_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

template <int P> __device__ __forceinline__ void split(float x, uint16_t (&o)[P]) {
  o[0] = bf16_rne(x);
  if constexpr (P == 3) {
    float r = x - bf16_f32(o[0]);
    o[1] = bf16_rne(r);
    r -= bf16_f32(o[1]);
    o[2] = bf16_rne(r);
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
      uint2 w;
      w.x = bf16_rne(o[0]) | ((uint32_t)bf16_rne(o[1]) << 16);
      w.y = bf16_rne(o[2]) | ((uint32_t)bf16_rne(o[3]) << 16);
      *reinterpret_cast<uint2*>(p) = w;
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
          uint16_t s[EPL][P];
#pragma unroll
          for (int j = 0; j < EPL; ++j) split<P>(e[j], s[j]);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            uint2 w;
            w.x = s[0][p] | ((uint32_t)s[1][p] << 16);
            w.y = s[2][p] | ((uint32_t)s[3][p] << 16);
            *reinterpret_cast<uint2*>(&Xs[(p * TM + row) * LDX + c]) = w;
          }
        }
      } else {
        const int c = v / (TM / EPL), row = (v % (TM / EPL)) * EPL;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          if constexpr (P == 1) {
            Xs[(row + j) * LDX + c] = reinterpret_cast<const uint16_t*>(e)[j];
          } else {
            uint16_t s[P];
            split<P>(e[j], s);
#pragma unroll
            for (int p = 0; p < P; ++p) Xs[(p * TM + row + j) * LDX + c] = s[p];
          }
        }
      }
    }
  }
};

// fragments of NB weight tiles at k-step ks: one contiguous KiB per tile and plane
template <int P, int NB>
__device__ __forceinline__ void load_w(bf16x8_t (&b)[P][NB], const uint16_t* (&base)[NB], int64_t plane, int ks) {
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < NB; ++j) b[p][j] = *reinterpret_cast<const bf16x8_t*>(base[j] + p * plane + (int64_t)ks * 512);
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

template <int P, int TM, int KC, int NB2, typename TIn, typename TOut, bool FUSED, bool XIMG, bool YIMG>
__global__ __launch_bounds__(256) void tt_chain_kernel(const ChainDesc d) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = TM / 16, LDX = KC + kPad, SPC = KC / 32;  // k-steps per chunk
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

  // ---------------- product 1:  H (or Y) tile = Win[n1_0 + ...][:] x X-tile^T
  float4v_t acc[MT][kNB1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < kNB1; ++j) acc[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
  const int KS1 = (d.Kin + 31) / 32;
  const int ntiles1 = (d.R + 15) / 16;
  const uint16_t* w1[kNB1];
#pragma unroll
  for (int j = 0; j < kNB1; ++j) {
    int ft = n1_0 / 16 + wave * kNB1 + j;
    ft = ft < ntiles1 ? ft : ntiles1 - 1;                      // surplus tiles compute and are not stored
    if (d.dbg & 4) ft = 0;
    w1[j] = d.Win + ((int64_t)ft * KS1 * 64 + lane) * 8;
  }
  // Touch every 128-byte line of the token tile once, up front: the chunk loads below then find their lines in L2
  // instead of paying the HBM latency once per chunk (one workgroup per CU: nothing else would hide it).
  constexpr int kTouch = 8;
  uint32_t touch[kTouch];
  {
    constexpr int EPLINE = 128 / sizeof(TIn);
    const TIn* X = static_cast<const TIn*>(d.X);
    const int lpr = XIMG ? TM / EPLINE : (d.Kin + EPLINE - 1) / EPLINE;       // lines per row (channel for images)
    const int nlines = XIMG ? d.Kin * (lpr > 0 ? lpr : 1) : TM * lpr;
#pragma unroll
    for (int i = 0; i < kTouch; ++i) {
      const int li = tid + 256 * i;
      touch[i] = 0;
      if (li < nlines) {
        const int row = li / (lpr > 0 ? lpr : 1), l = li - row * (lpr > 0 ? lpr : 1);
        if constexpr (!XIMG) {
          const int64_t t = m0 + row;
          const int c = l * EPLINE;
          if (t < d.T && c < d.Kin) touch[i] = *reinterpret_cast<const uint32_t*>(X + t * d.ldx + (c / (4 / (int)sizeof(TIn))) * (4 / (int)sizeof(TIn)));
        } else {
          const int64_t t = m0 + (int64_t)l * EPLINE;
          if (t < d.T && d.x_vec) touch[i] = *reinterpret_cast<const uint32_t*>(X + elem_off(t, row, 0, d.x_hw, d.Kin));
        }
      }
    }
  }
  {
    ChunkLoader<P, TM, KC, TIn, XIMG> ld;
    const int nchunks = (d.Kin + KC - 1) / KC;
    const int rot = blockIdx.x % nchunks;                      // chunk order rotated per workgroup
    bf16x8_t a[AS][P][MT], b[2][P][kNB1];
    ld.load(d, m0, rot * KC, tid);
    load_w<P, kNB1>(b[0], w1, d.win_plane, rot * SPC);
    ld.store(Xs, tid);
    __syncthreads();
    if constexpr (ADB) load_x<P, TM>(a[0], Xs, LDX, 0, r, q);
    for (int c = 0; c < nchunks; ++c) {
      const int ca = (c + rot) % nchunks, cn = (c + 1 + rot) % nchunks;
      const uint16_t* Xc = Xs + (c & 1) * (P * TM * LDX);
      uint16_t* Xn = Xs + ((c + 1) & 1) * (P * TM * LDX);
      const bool more = c + 1 < nchunks;
      if (more) ld.load(d, m0, cn * KC, tid);
      const int ksn = min(SPC, KS1 - ca * SPC);                // k-steps of this chunk (the last one may be short)
#pragma unroll
      for (int ks = 0; ks < SPC; ++ks) {
        if (ks < ksn) {
          const int cur = ks & 1, nxt = cur ^ 1;
          if (ks + 1 < ksn) {
            load_w<P, kNB1>(b[nxt], w1, d.win_plane, ca * SPC + ks + 1);
            if constexpr (ADB) load_x<P, TM>(a[nxt], Xc, LDX, ks + 1, r, q);
          } else if (more) {
            load_w<P, kNB1>(b[nxt], w1, d.win_plane, cn * SPC);
          }
          if constexpr (!ADB) load_x<P, TM>(a[0], Xc, LDX, ks, r, q);
          mma_step<P, MT, kNB1>(a[ADB ? cur : 0], b[cur], acc);
          if (ks + 1 == ksn && (nxt != 0)) {                   // odd-length chunk: keep the ring aligned to slot 0
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
              for (int j = 0; j < kNB1; ++j) b[0][p][j] = b[1][p][j];
          }
        }
      }
      if (more) {
        ld.store(Xn, tid);
        __syncthreads();
        if constexpr (ADB) load_x<P, TM>(a[0], Xn, LDX, 0, r, q);
      }
    }
  }

  {   // the touched words are dead data; this impossible store only keeps their loads from being dropped
    uint32_t tx = 0;
#pragma unroll
    for (int i = 0; i < kTouch; ++i) tx ^= touch[i];
    if (d.T < 0 && tx == 0x9e3779b9u) *static_cast<uint32_t*>(d.Y) = tx;
  }
  if constexpr (!FUSED) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t t = m0 + 16 * mt + r;
      if (t >= d.T) continue;
#pragma unroll
      for (int j = 0; j < kNB1; ++j) {
        const int f0 = n1_0 + (wave * kNB1 + j) * 16 + 4 * q;
        if (f0 >= d.R) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
          if (f0 + 4 <= d.R) bv = *reinterpret_cast<const float4*>(d.bias + f0);
          else {
            bv.x = d.bias[f0];
            if (f0 + 1 < d.R) bv.y = d.bias[f0 + 1];
            if (f0 + 2 < d.R) bv.z = d.bias[f0 + 2];
          }
        }
        store4<TOut, YIMG>(static_cast<TOut*>(d.Y), t, f0, d.R, d.ldy, d.y_hw, acc[mt][j], bv);
      }
    }
    return;
  } else {
    // H -> LDS, split into planes again (the fp32 chain of the reference rounds H to fp32 here as well)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < kNB1; ++j) {
        const int f0 = (wave * kNB1 + j) * 16 + 4 * q;
        if (f0 >= d.R) continue;
        uint16_t s[4][P];
#pragma unroll
        for (int e = 0; e < 4; ++e) split<P>(acc[mt][j][e], s[e]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          uint2 w;
          w.x = s[0][p] | ((uint32_t)s[1][p] << 16);
          w.y = s[2][p] | ((uint32_t)s[3][p] << 16);
          *reinterpret_cast<uint2*>(&Hs[(p * TM + 16 * mt + r) * ldh + f0]) = w;
        }
      }
    __syncthreads();

    // ---------------- product 2:  Y tile = Wout x H-tile^T + bias, NB2 feature tiles per pass
    const int ntiles2 = (d.Nout + 15) / 16;
    const int ngroups = (ntiles2 + NB2 - 1) / NB2;
    const int KS2 = d.R / 32;
    const int rot = blockIdx.x % ngroups;                        // group order rotated per workgroup
    auto bases = [&](int pos, const uint16_t* (&w2)[NB2]) {
      const int g = (pos + rot) % ngroups;
#pragma unroll
      for (int j = 0; j < NB2; ++j) {
        int ft = g * NB2 + j;
        ft = ft < ntiles2 ? ft : ntiles2 - 1;
        if (d.dbg & 4) ft = 0;
        w2[j] = d.Wout + ((int64_t)ft * KS2 * 64 + lane) * 8;
      }
      return g;
    };
    bf16x8_t a[AS][P][MT], b[2][P][NB2];
    const uint16_t* w2[NB2];
    int pos = (d.dbg & 2) ? ngroups : wave;
    if (pos < ngroups) {
      bases(pos, w2);
      load_w<P, NB2>(b[0], w2, d.wout_plane, 0);
      if constexpr (ADB) load_x<P, TM>(a[0], Hs, ldh, 0, r, q);
    }
    while (pos < ngroups) {
      const int g = (pos + rot) % ngroups;
      float4v_t acc2[MT][NB2];
      float4 bq[NB2];
#pragma unroll
      for (int j = 0; j < NB2; ++j) {                             // bias of this lane's 4 features per tile, fetched early
        const int f0 = (g * NB2 + j) * 16 + 4 * q;
        bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
          if (f0 + 4 <= d.Nout) bq[j] = *reinterpret_cast<const float4*>(d.bias + f0);
          else {
            if (f0 < d.Nout) bq[j].x = d.bias[f0];
            if (f0 + 1 < d.Nout) bq[j].y = d.bias[f0 + 1];
            if (f0 + 2 < d.Nout) bq[j].z = d.bias[f0 + 2];
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NB2; ++j) acc2[mt][j] = float4v_t{0.f, 0.f, 0.f, 0.f};
      const int npos = pos + 4;
      // two k-steps per trip so that the fragment ring (slot 0 / slot 1) is addressed statically
      for (int ks = 0; ks < KS2; ks += 2) {
        if (ks + 1 < KS2) {
          load_w<P, NB2>(b[1], w2, d.wout_plane, ks + 1);
          if constexpr (ADB) load_x<P, TM>(a[1], Hs, ldh, ks + 1, r, q);
        }
        if constexpr (!ADB) load_x<P, TM>(a[0], Hs, ldh, ks, r, q);
        mma_step<P, MT, NB2>(a[0], b[0], acc2);
        if (ks + 1 < KS2) {
          if (ks + 2 < KS2) {
            load_w<P, NB2>(b[0], w2, d.wout_plane, ks + 2);
            if constexpr (ADB) load_x<P, TM>(a[0], Hs, ldh, ks + 2, r, q);
          } else if (npos < ngroups) {
            bases(npos, w2);
            load_w<P, NB2>(b[0], w2, d.wout_plane, 0);
            if constexpr (ADB) load_x<P, TM>(a[0], Hs, ldh, 0, r, q);
          }
          if constexpr (!ADB) load_x<P, TM>(a[0], Hs, ldh, ks + 1, r, q);
          mma_step<P, MT, NB2>(a[ADB ? 1 : 0], b[1], acc2);
        } else if (npos < ngroups) {                             // odd KS2: the next group's first step goes to slot 0
          bases(npos, w2);
          load_w<P, NB2>(b[0], w2, d.wout_plane, 0);
          if constexpr (ADB) load_x<P, TM>(a[0], Hs, ldh, 0, r, q);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int64_t t = m0 + 16 * mt + r;
        if (t >= d.T) continue;
#pragma unroll
        for (int j = 0; j < NB2; ++j) {
          const int f0 = (g * NB2 + j) * 16 + 4 * q;
          if (f0 < d.Nout && !(d.dbg & 1)) store4<TOut, YIMG>(static_cast<TOut*>(d.Y), t, f0, d.Nout, d.ldy, d.y_hw, acc2[mt][j], bq[j]);
        }
      }
      pos = npos;
    }
  }
}

template <int P, int TM, int KC, int NB2, typename TIn, typename TOut, bool FUSED, bool XIMG, bool YIMG>
int launch_variant(const ChainDesc& d, hipStream_t s) {
  auto kern = tt_chain_kernel<P, TM, KC, NB2, TIn, TOut, FUSED, XIMG, YIMG>;
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
  static const int dbg = getenv("TADMM_CHAIN_DBG") ? atoi(getenv("TADMM_CHAIN_DBG")) : 0;
  dd.dbg = dbg;
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), lds, s, dd);
  return 0;
}

template <int P, int TM, int KC, typename T>
int launch_single(const ChainDesc& d, hipStream_t s) {
  if (d.x_hw > 0)
    return d.y_hw > 0 ? launch_variant<P, TM, KC, 1, T, T, false, true, true>(d, s)
                      : launch_variant<P, TM, KC, 1, T, T, false, true, false>(d, s);
  return d.y_hw > 0 ? launch_variant<P, TM, KC, 1, T, T, false, false, true>(d, s)
                    : launch_variant<P, TM, KC, 1, T, T, false, false, false>(d, s);
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
  if (d.x_hw > 0 || d.y_hw > 0) return -1;
  if (dtype == 1) {
    if (tile_tokens == 32) return launch_variant<1, 32, 128, 6, uint16_t, uint16_t, true, false, false>(d, s);
    return launch_variant<1, 64, 128, 6, uint16_t, uint16_t, true, false, false>(d, s);
  }
  if (tile_tokens == 32) return launch_variant<3, 32, 128, 6, float, float, true, false, false>(d, s);
  return launch_variant<3, 64, 64, 3, float, float, true, false, false>(d, s);
}

}  // namespace tadmm
