// Internal declarations shared by the HIP translation units of libtadmm_hip.so.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/tadmm.h"

namespace tadmm {

constexpr int kWave = 64;

// block -> (problem, local block) map used by every grouped launch
struct BlockRef { int32_t prob; int32_t local; };

// ---------------------------------------------------------------- sweep kernels
// One compressed parameter as seen by the unfold / fold+update sweeps.
//   conv (K2>1): W is (O, I, K2) row-major, the unfolded tensor T is (O, K2, I)
//   K2==1      : no permutation, flat copy
struct SweepDesc {
  const float* W; float* U; float* Z;
  float* T0;          // unfolded W+U (input of TT step 0)
  const float* Zmat;  // reconstructed Z in unfolded layout
  int32_t O, I, K2;
  int32_t ichunk;     // input channels per block (conv) or elements per block (flat)
  int32_t nchunk;     // chunks per o-slab (conv) ; total chunks (flat)
  int64_t numel;
  int32_t blk_begin;  // first global block of this layer (for the residual partials)
  int32_t nblk;
};

void launch_unfold(const SweepDesc* descs_dev, const BlockRef* map_dev, int nblocks, int use_u, hipStream_t s);
void launch_fold_update(const SweepDesc* descs_dev, const BlockRef* map_dev, int nblocks, int update_u,
                        double* resid_partial_dev, hipStream_t s);
// out_index (nullable): layer l's sum goes to resid_sq_dev[out_index[l]] (sub-plans of a two-lane plan)
void launch_resid_reduce(const SweepDesc* descs_dev, int nlayers, const double* resid_partial_dev,
                         double* resid_sq_dev, hipStream_t s, const int32_t* out_index = nullptr);

// ---------------------------------------------------------------- Gram (fp64 MFMA)
struct GramDesc {
  const float* A;     // row-major m x n
  int32_t m, n;
  int32_t trans;      // 0: G = A A^T (N=m, reduce over n) ; 1: G = A^T A (N=n, reduce over m)
  int32_t N, K;
  int32_t nt;         // number of 32-wide tiles along N
  int32_t ksplit, kchunk;
  double* partial;    // [ksplit][ntp][32*32]
  double* G;          // [Npad][ld]  (row j = column j of the symmetric matrix), zero padded
  int32_t Npad, ld;
};
// `skip` (nullable, here and below): device int per problem; a non-zero entry turns every block of that problem
// into a no-op.  Lets a data-dependent outer loop (Tucker HOOI) drop finished layers without rebuilding maps.
void launch_gram_partial(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                         const int32_t* skip = nullptr);
void launch_gram_reduce(const GramDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                        const int32_t* skip = nullptr);

// ---------------------------------------------------------------- Jacobi eigen-solver
constexpr int kJB = 8;          // columns per block; a pair is 2*kJB = 16 columns = one MFMA tile
struct EigDesc {
  double* XT;         // [Npad][ld]: row j holds column j of X (starts as G, converges to G*V)
  int32_t N, Npad, ld, nb;       // nb = Npad / kJB (multiple of 4)
  double* off;        // [3]: [0],[1] max relative off-diagonal per sweep parity; [2] max squared column norm
  int32_t* done;      // [1] sticky convergence flag
  // finalize outputs
  double* lam;        // [Npad] eigenvalues (column norms), unsorted
  int32_t* order;     // [Npad] order[c] = column index of c-th largest
  double* sigma;      // [r] sqrt(lam) of kept columns, descending
  int32_t r;          // kept rank
  int32_t mode;       // 0: left vectors  (m<=n): Uf[m][r] = V           (core), needs project GEMM
                      // 1: right vectors (m>n) : Vs[n][r] = V/sigma, Tn[r][n] = sigma*V^T ; 2: evec_out only
                      // 3: out_a = V/sigma only
  float* out_a;       // mode0: Uf (N x r) ; mode1: Vs (N x r)
  float* out_b;       // mode1: Tnext (r x N) ; mode0: unused
  double* evec_out;   // optional: [r][N] eigenvectors as rows in fp64 (tadmm_eigh_f64), nullable
  double* sblk;       // [Npad/16][16*16] carried self-Gram of every 16-column super-block (tick3), nullable
  int32_t ldo;        // leading dimension of out_a (0: r)
  double* warm;       // jacobi_small only, nullable: [Npad][Npad] eigenvectors of this problem's PREVIOUS solve (row j =
                      // vector of column j); the solve then starts from X = G * V_prev instead of X = G
  int32_t* warm_ok;   // [1] the image is valid (written by a converged solve whose spectrum allowed normalising every column)
  double* scratch;    // optional: 4 * N * N doubles of global scratch -- the problem may take the direct route of tridiag_mid.hip
  int32_t period;     // ticks per sweep of the GROUP's schedule (tick3 groups; 0: this problem's own players - 1): with one
                      // period for every problem of a group all sweeps start at the same tick, so the self pass is launched
                      // once per global sweep; a problem with fewer players idles in the ticks beyond its own
};
void launch_jacobi_init(const EigDesc* descs_dev, int nprob, hipStream_t s, const int32_t* skip = nullptr,
                        double* prev_dev = nullptr);   // prev_dev: [nprob] history of jacobi_conv_kernel, reset here
// device-side convergence decision after a global sweep (prev_dev: [nprob] doubles, zeroed by the caller per run;
// verdict_pinned: [1 + nprob] ints of device-visible pinned host memory)
void launch_jacobi_conv(const EigDesc* descs_dev, int nprob, int tick, double tol, bool super, double* prev_dev,
                        int* verdict_pinned, hipStream_t s);
// Largest eigen-problem the Jacobi kernels take: rows longer than the LDS-resident pair (N = 1152) go through the streamed
// pair kernel (jacobi_tick_stream_kernel); the bound itself is the 64 KiB eigenvalue table of eig_sort_kernel.
constexpr int kJacobiMaxN = 8160;
bool jacobi_size_supported(int n);
size_t jacobi_tick_stream_lds_bytes();
size_t jacobi_tick_lds_bytes(int ld_max);    // dynamic LDS of a tick1 launch whose largest problem has row length ld_max
size_t jacobi_tick2_lds_bytes(int ld_max);   // same for the LDS-resident super-pair kernel
bool jacobi_tick2_fits(int ld_max);
// super=false: one workgroup per pair of 8-column blocks (nb/2 per problem, nb-1 ticks per sweep)
// super=true : one workgroup per pair of 16-column super-blocks (nb/4 per problem, nb/2-1 ticks per sweep)
void launch_jacobi_tick(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                        int inner_sweeps, size_t lds_bytes, bool super, hipStream_t s);
void dump_stamps();
size_t jacobi_tick3_lds_bytes(int ld_max);
bool jacobi_tick3_fits(int ld_max);
// tick3: super-pair kernel with carried self-Grams (one cross-Gram per launch, round-2 solve overlapped with the
// round-1 update).  Needs launch_jacobi_self on the first tick of every sweep (it refreshes EigDesc::sblk).
void launch_jacobi_tick3(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                         int ld_max, hipStream_t s);
// rows of an eigen-solver image are whole 1 KiB chunks (128 doubles: LDS-DMA loads).  The LDS-resident pair kernel
// holds 16 such rows up to ld = 1152; longer rows go through the streamed pair kernel.
constexpr int kLdResidentMax = 1152;
static inline int eig_ld(int N) { return (N + 127) / 128 * 128; }
// once-per-sweep companion of tick3 (tick1 in self mode): within-block pairs + refresh of the carried self-Grams
void launch_jacobi_self(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, int tick, double tol,
                        int inner_sweeps, int ld_max, hipStream_t s);
// whole eigen-solve of small problems (Npad <= 64) in one launch, one workgroup per problem; converged flags go to
// verdict_pinned[1 + p]
bool jacobi_small_fits(int npad_max);
// warm: reserve the second LDS image a warm-started problem needs (EigDesc::warm)
// `fast_done` (optional): per-problem words set by eig_small_direct_kernel -- those problems are finished already
void launch_jacobi_small(const EigDesc* descs_dev, int nprob, int npad_max, double tol, int max_sweeps,
                         const int32_t* skip, int* verdict_pinned, hipStream_t s, bool warm,
                         const int32_t* fast_done = nullptr);
// tridiag.hip: direct solver (tridiagonalisation + bisection + inverse iteration) for problems of at most 64 columns;
// verified results only, everything else is left to jacobi_small_kernel
bool eig_small_direct_on();
// tridiag_mid.hip: the same route for the 128- / 192-column Rayleigh-Ritz problems (EigDesc::scratch set), one launch per size
bool eig_mid_direct_on();
bool eig_mid_direct_size(int n);
size_t eig_mid_scratch_bytes(int n);
void launch_eig_mid_direct(const EigDesc* descs_dev, int nprob, int ns, const int32_t* skip, int* verdict_pinned, hipStream_t s);
void launch_eig_small_direct(const EigDesc* descs_dev, int nprob, const int32_t* skip, int32_t* fast_done_dev,
                             int* verdict_pinned, hipStream_t s);
void launch_eig_norms(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                      const int32_t* skip = nullptr);
// npad_max: largest padded problem size of the launch (sizes the eigenvalue table in LDS; 0 = the largest supported)
void launch_eig_sort(const EigDesc* descs_dev, int nprob, hipStream_t s, const int32_t* skip = nullptr, int npad_max = 0);
void launch_eig_extract(const EigDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                        const int32_t* skip = nullptr);

// ---------------------------------------------------------------- grouped GEMM (fp32 MFMA)
struct GemmDesc {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
  float alpha, beta;
  const float* bias_n; const float* bias_m;
  int32_t tiles_m, tiles_n;
};
constexpr int kGemmBM = 64, kGemmBN = 64;
void launch_gemm(const GemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s,
                 const int32_t* skip = nullptr);
void launch_gemm_one(const GemmDesc& d, hipStream_t s);   // descriptor passed as a kernel argument
// bf16 C[M][N] = A[M][K] * Bt[N][K]^T (+ f32 bias over N), fp32 accumulate (gemm_bf16.hip)
void launch_gemm_bf16_nt(const void* A, const void* Bt, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                         int64_t ldc, const float* bias_n, hipStream_t s);

// ---------------------------------------------------------------- address spaces
// A pointer fetched from a descriptor array has lost its address space as far as hipcc can tell, and the loads and
// stores through it become flat_load / flat_store.  Those count on vmcnt AND lgkmcnt: every wait for an LDS read then
// also drains the global loads in flight (and the other way round), which undoes any global -> LDS software pipeline.
// `gp(p)` pins the global address space; a G<T>* dereferences to global_load / global_store.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
template <typename T> using G = T __attribute__((address_space(1)));
template <typename T> __device__ __forceinline__ G<T>* gp(T* p) { return (G<T>*)p; }
#endif

// ---------------------------------------------------------------- forward chains of the factorised layers (chain.hip)
struct ChainDesc {
  const void* X; void* Y;
  const uint16_t* Win; const uint16_t* Wout;    // bf16 planes in fragment-major order (chain.hip)
  const float* bias;
  int64_t T;
  int32_t Kin, R, Nout;
  int64_t ldx, ldy, win_plane, wout_plane;
  int32_t x_hw, y_hw, x_vec, y_vec, fused;
  long long* stamps;                            // debugging: per-workgroup cycle stamps (nullable)
};
int launch_tt_chain(const ChainDesc& d, int dtype, int tile_tokens, hipStream_t s);

// factorised convolution of a small image in one launch (convchain.hip): 1x1 / chain-in, k x k core, 1x1 / chain-out
struct ConvChainDesc {
  const void* X; void* Y;
  const uint16_t* W1; const uint16_t* W2; const uint16_t* W3;   // fragment-major bf16 planes
  const float* bias;
  int64_t w1_plane, w2_plane, w3_plane;
  int32_t B, C, R1, R2, Nout;                                   // R1, R2 multiples of 32 (zero padded), <= 256
  int32_t H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
  int32_t TM, TR, tiles, NT;                                    // pixels per workgroup (64 | 32), its output rows, tiles per image, TM-pixel tiles of its halo
  int32_t x_vec;
};
int launch_tt_conv(const ConvChainDesc& d, int dtype, hipStream_t s);

// ---------------------------------------------------------------- grouped GEMM (fp64 MFMA) -- filtered eigen-solver
// See dgemm.hip.  M, N multiples of 32, K multiple of 16; every leading dimension even (16-byte rows).
struct DgemmDesc {
  const double* A; const double* B; const double* C;     // explicit operands (used when the selector is < 0)
  const double* P; const double* Q;                       // epilogue operands, shaped like C
  double* ring[3];                                        // ring of block images, addressed as (*rot + sel) % 3
  const int32_t* rot;                                     // device word: ring index of the current basis (nullable)
  int32_t selA, selB, selC, selP, selQ;                   // ring offsets, or -1 = explicit pointer
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t tiles_m, tiles_n;
  int32_t mode;                                           // epilogue, see dgemm.hip
  const double* coef;                                     // mode 1: s0, s1, s2 (device)
  const double* theta;                                    // mode 3: per-row shift
  double* rowpart;                                        // modes 2, 3: [tiles_n][M] fixed-order row partials
  const int32_t* gate; int32_t gate_min;                  // no-op unless *gate >= gate_min (gate nullable)
  const uint16_t* Gp; int64_t g_plane;                    // dgemm3.hip: B as three fragment-major bf16 planes (nullable)
};
void launch_dgemm(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, bool b_transposed, hipStream_t s);
// NT product with LDS-staged 64 x tile_n tiles, tile_n = 64 | 32 (block map: local = tile index over
// ceil(M/64) x ceil(N/tile_n); K % 32 == 0; row partials of modes 2/3 per tile_n-column tile)
void launch_dgemm_nt64(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s, int tile_n = 64,
                       int tile_m = 64);   // tile_m = 32: 32 x 32 tiles (tile_n is then 32 as well)
// C <- coef[0]*C + coef[1]*P over M x ldc elements; block map: local = chunk of 1024 elements
void launch_daxpby(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);
// dgemm3.hip: the same NT products with B = G at fp32 accuracy on the bf16 matrix cores (three-plane split).
// Block map: local = tile index over (M/32) x ceil(N/128); modes 0, 1, 2; d.Gp = planes written by launch_gplanes.
struct GPlaneDesc {
  const double* Gm; int32_t ldg;          // Gram image [16*nt][ldg], zero padded
  int32_t nt, ks;                         // 16-row tiles, 32-column k-steps
  uint16_t* out; int64_t plane;           // [3][nt][ks][64][8] bf16, plane stride in elements
};
void launch_gplanes(const GPlaneDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);   // local = 4 fragment blocks
void launch_dgemm3(const DgemmDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);

// ---------------------------------------------------------------- Cholesky QR of a block (chol.hip)
// C = R^T R of an n x n Gram matrix (n multiple of 16, <= 256), one workgroup per problem, upper tiles resident in
// registers; writes the off-diagonal tiles of R and the inverses Wd[k] = R_kk^{-T} of the diagonal tiles, which
// is what the block forward substitution Q^T = R^{-T} Y^T (chol_solve) consumes.
struct CholDesc {
  const double* C; int32_t ldc;         // Gram matrix (symmetric; only the upper tiles are read)
  int32_t n;
  double* R; int32_t ldr;               // upper factor, off-diagonal tiles only are meaningful
  double* Wd;                           // [n/16][16*16] row-major inverse-transposed diagonal tiles
  double* ring[3]; const int32_t* rot; int32_t sel;   // block image Y^T [n][ldy] to be solved in place
  int32_t ldy, ncols;                   // ncols multiple of 16 (a wave of chol_solve owns 16 columns)
  int32_t* bad;                         // set to 1 on a non-positive pivot (sticky, device word)
  const int32_t* gate; int32_t gate_min;
  int32_t* rot_out;                     // optional: *rot_out <- (*rot + sel) % 3 after the solve (new basis index)
};
void launch_chol_factor(const CholDesc* descs_dev, int nprob, hipStream_t s);
void launch_chol_solve(const CholDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);

// ---------------------------------------------------------------- filtered eigen-solver (filter.hip)
// Device-resident state of one filtered problem: the stage logic runs on the device (filt_plan_kernel), launches
// read their gates / ring indices / Chebyshev scalars from here, the host only learns "anyone still filtering?".
struct FiltState {
  int32_t base, res;          // ring index of the current basis Q / of the block the coming CholQR works on
  int32_t nsteps;             // recurrence steps of the current stage (0: none)
  int32_t active;             // 1 while another stage is wanted
  int32_t alive;              // 0 once the problem has been handed to the fallback solver
  int32_t bad;                // sticky failure word (pivot breakdown, degenerate bounds, failed verification)
  int32_t stage;
  int32_t products;           // block products with G this run (stage-0 product, recurrence steps, final T)
  int32_t products_fast;      // ... of which at fp32 accuracy on the bf16 matrix cores (dgemm3.hip)
  int32_t precise_stages;     // filter stages run in fp64 so far: a problem may not finish before it had one
  double logamp;              // accumulated log-amplification of the boundary Ritz vector relative to the damped part
  double logamp_precise;      // ... the share of it contributed by fp64 stages
  double coef1[2];            // first step of a stage:  Y1 = coef1[0]*T + coef1[1]*Q           (T = G Q)
  double coefk[3];            // later steps: Y_{k+1} = coefk[0]*G*Y_k + coefk[1]*Y_k + coefk[2]*Y_{k-1}
  double b, lr, l1;           // bounds used by the last stage (diagnostics)
  double crit;                // verification figure: bound on ||sin Theta||_F of the accepted subspace
  double guard;               // power-iteration estimate of the largest eigenvalue OUTSIDE the block (filt_guard_kernel)
  double mom[64 * 5];         // filt_moments_kernel: per-workgroup partials of ||G||_F^2, tr G, ||GQ||_F^2, ||H||_F^2, tr H
  int32_t mom_on;             // 1 when the partials above belong to this run
};
struct FiltProb {
  FiltState* st;
  double* ring[3];            // block images [rp][ldy]
  int32_t N, Npad, rp, r, r32;
  int32_t ldy;
  const double* rqpart; int32_t rq_tiles;   // [rq_tiles][rp] Rayleigh-quotient partials of the current basis
  const double* vpart; int32_t v_tiles;     // [v_tiles][r32] residual partials of the Ritz pairs
  const double* lam; const int32_t* order;  // Ritz values of the Rayleigh-Ritz solve (unsorted) and their order
  double* sigma;                            // [r] sqrt of the kept Ritz values, descending (shared with the full path)
  double* theta;                            // [r32] kept Ritz values (0 beyond r)
  const double* UT;                         // [r32][ldy] Ritz vectors as rows
  int32_t mode, ldo;                        // output convention of EigDesc
  float* out_a; float* out_b;
  int32_t* skip_slot;                       // word of the eig group's skip array that belongs to this problem
  int32_t* fb_skip;                         // word of the fallback group's skip array: 1 = filtered result accepted
  const double* G; int32_t ldg;             // the problem's matrix [Npad][ldg] (guard: products with single vectors)
  const double* H; int32_t ldh;             // Rayleigh-Ritz image H = Q^T G Q [rp][ldh] (moments guard; overwritten by the solve)
};
struct FiltParams {
  int32_t max_degree;         // recurrence steps per stage (D)
  double log_target;          // ln(2/eps): wanted total log-amplification
  double cond_max;            // largest tolerated growth of the block's condition number per stage
  double cond_first;          // the same for the first stage, whose bounds come from the Rayleigh quotients of one power step
  double sin_tol;             // acceptance threshold of the verification
  double log_precise;         // log-amplification the fp64 stages must contribute once fp32-accuracy stages were used
};
// complement route (complement.hip): a step that discards few vectors is solved for the discarded subspace of G' = cI - G
struct CompDesc {
  FiltState* st;                             // the filtered problem's state: `bad` gates everything
  const double* G; double* Gc;               // [Npad][ldg] Gram image and its reflection (the filter's operand)
  int32_t N, Npad, ldg;
  double* cshift;                            // [1] the shift c
  const double* UT; int32_t ldy, k;          // Ritz vectors of G' as rows [align32(k)][ldy]: the k trailing eigenvectors of G
  double* Cimg; int32_t r;                   // [r][ldy] the kept basis, rows = vectors
  float* out_a; int32_t ldo;                 // EigDesc mode-0 output: out_a[i * R + c]
  double* sigma_layer;                       // [r] singular-value slot of the plan (filled with NaN: not available)
};
void launch_comp_prepare(const CompDesc* descs_dev, int nprob, int npad_max, int steps, hipStream_t s);
void launch_comp_form(const CompDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);   // local = 4 rows of C^T
void launch_comp_emit(const CompDesc* descs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);   // local = 4 vectors
void launch_filt_init(const FiltProb* probs_dev, const BlockRef* map_dev, int nblocks, int nprob, hipStream_t s);
void launch_filt_guard(const FiltProb* probs_dev, int nprob, int npad_max, int rp_max, int steps, hipStream_t s);
void launch_filt_moments(const FiltProb* probs_dev, int nprob, hipStream_t s);
// stage_fast: bit 0 = this stage's products run in dgemm3 (fp32 accuracy), bit 1 = the stage-0 product did
void launch_filt_plan(const FiltProb* probs_dev, int nprob, FiltParams prm, int last_stage, int stage_fast, int* verdict_pinned,
                      hipStream_t s);
void launch_filt_flags(const FiltProb* probs_dev, int nprob, hipStream_t s);
void launch_filt_theta(const FiltProb* probs_dev, int nprob, hipStream_t s);
void launch_filt_verdict(const FiltProb* probs_dev, int nprob, FiltParams prm, int* verdict_pinned, hipStream_t s);
void launch_filt_emit(const FiltProb* probs_dev, const BlockRef* map_dev, int nblocks, hipStream_t s);

// ---------------------------------------------------------------- penalty
constexpr int kPenaltyBlocks = 1024;
void launch_penalty(int n, const void* const* ptrs_dev, const int64_t* numel_dev, int64_t total, float rho,
                    float gscale, double* loss_dev, double* partial_dev, hipStream_t s);

}  // namespace tadmm
