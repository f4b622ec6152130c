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
#include "chain_common.h"

namespace tadmm {
namespace {

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
