/*
 * tadmm.h -- C ABI of the MI355X-native ADMM low-rank projection path.
 *
 * The reference (miaoyin390/DNN-Compression-Tensor-ADMM) is pure Python and has
 * no FFI layer; its boundary for this path is the Python API of admm.py /
 * ttd.py / TT*.py / TK*.py.  This header is the C ABI
 * that sits *behind* that Python surface: plain pointers and sizes, no torch
 * types, extern "C".  Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - all tensors are row-major contiguous DEVICE pointers (float32 unless said
 *     otherwise), owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     call is stream-ordered and asynchronous unless documented otherwise;
 *   - return value 0 = success, negative = tadmm_status; a human-readable
 *     message for the last failure of a handle is returned by
 *     tadmm_last_error();
 *   - no exceptions cross the boundary; a handle is not thread-safe.
 */
#ifndef TADMM_H_
#define TADMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TADMM_MAX_MODES 8

typedef enum {
  TADMM_OK = 0,
  TADMM_ERR_INVALID = -1,     /* bad argument / inconsistent descriptor          */
  TADMM_ERR_WORKSPACE = -2,   /* caller workspace too small                      */
  TADMM_ERR_HIP = -3,         /* a HIP runtime call failed                       */
  TADMM_ERR_NOCONVERGE = -4,  /* Jacobi eigen-solver hit its sweep cap           */
  TADMM_ERR_UNSUPPORTED = -5
} tadmm_status;

typedef enum {
  TADMM_KIND_TT_CONV = 0,   /* admm.py:91-101  prune_conv_rank_tt  ((O,I,k2)->(O,k2,I) unfold) */
  TADMM_KIND_TT_LINEAR = 1, /* admm.py:103-111 prune_linear_rank_tt (no permutation)           */
  TADMM_KIND_SVD = 2,       /* admm.py:129-149 prune_{conv,linear}_rank_svd (2-mode TT)        */
  TADMM_KIND_TUCKER2 = 3    /* admm.py:113-127 prune_{conv,linear}_rank_tk (HOSVD + HOOI)      */
} tadmm_kind;

/* flags */
#define TADMM_FLAG_SKIP_ROTATIONS 1u /* Z-only mode: a TT step whose kept rank equals the row
                                        count of its unfolding is an orthogonal change of basis
                                        that cannot change Z; skip its eigen-solve.  Must be 0
                                        when cores are requested.                              */

/* Plain-old-data description of one compressed parameter. */
typedef struct {
  int32_t kind;                      /* tadmm_kind                                        */
  int32_t ndim;                      /* 2 or 4                                            */
  int64_t dims[4];                   /* weight shape: (O,I,kh,kw) or (out,in)             */
  int32_t d;                         /* number of TT modes (TT kinds); 2 for SVD          */
  int32_t tt_shapes[TADMM_MAX_MODES];/* n_0..n_{d-1}  (hp_dict.tt_shapes[name])           */
  int32_t ranks[TADMM_MAX_MODES + 1];/* r_0..r_d      (hp_dict.ranks[name]); Tucker: [r_out,r_in] */
  uint32_t flags;
  int32_t hooi_max_iter;             /* Tucker only (tensorly default 100)                */
  float hooi_tol;                    /* Tucker only (tensorly default 1e-4)               */
} tadmm_layer_desc;

typedef struct tadmm_ctx_s* tadmm_handle;
typedef struct tadmm_plan_s* tadmm_plan;

/* ---- library / handle ------------------------------------------------- */
int tadmm_version(void);
/* sizeof(tadmm_layer_desc) / sizeof(tadmm_gemm_desc) as compiled: lets a foreign binding verify its struct layout */
int tadmm_abi_sizes(int* layer_desc_bytes, int* gemm_desc_bytes);
int tadmm_create(int device, tadmm_handle* out);
int tadmm_destroy(tadmm_handle h);
const char* tadmm_last_error(tadmm_handle h);

/* ---- rank clamp (ttd.py:18-19) ---------------------------------------- */
/* The reference clamps r_{i+1} to the number of singular values of the i-th
 * unfolding, min(r_i*n_i, prod(n_{i+1..})) -- a pure function of the shapes.
 * Applies it to desc->ranks in place (host only, no device work); returns the
 * number of entries changed.  Rank selection is therefore bit-exact by
 * construction. */
int tadmm_tt_clamp_ranks(tadmm_layer_desc* desc);

/* ---- projection plan: Z <- proj(W+U), U += W-Z, ||W-Z||^2 -------------- */
/* Replaces ADMM.update (admm.py:42-78) for a set of layers that are processed
 * together, phase by phase, in grouped launches.  Pointers are captured at
 * creation; descriptors (with clamped ranks) are copied.
 *   W[i], U[i], Z[i] : device float32, prod(dims) elements each
 *   cores[i]         : NULL, or device float32 buffer receiving the TT cores
 *                      back to back (core_0 .. core_{d-1}, each (r_j,n_j,r_{j+1}));
 *                      for Tucker: core (r_out,r_in,k2...) then U_out (O,r_out)
 *                      then U_in (I,r_in)
 *   workspace        : device scratch of at least tadmm_plan_workspace_bytes */
int tadmm_plan_workspace_bytes(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, size_t* bytes);
int tadmm_plan_create(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs,
                      const float* const* W, float* const* U, float* const* Z, float* const* cores,
                      void* workspace, size_t workspace_bytes, tadmm_plan* out);
/* Runs one ADMM projection over all layers of the plan.
 *   update_u     : 0 -> only Z is written (ADMM.__init__ + update(update_u=False), engines.py:245)
 *   use_u        : 0 -> project W alone (the --decompose path: TTConv.py:96-109, TTLinear.py:61-66)
 *   resid_sq_dev : NULL or device double[n_layers]  <- ||W-Z||_2^2 per layer (admm.py:73-76)
 * Convergence of the Jacobi sweeps is decided on the device; the host reads the verdict through pinned
 * memory one sweep late, so the call blocks only at the end of each eigen-solve level. */
int tadmm_plan_run(tadmm_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream);
/* Singular values kept at TT step `step` of layer `layer` in the last run (device->host copy,
 * synchronous).  out must hold ranks[step+1] doubles. */
int tadmm_plan_singular_values(tadmm_plan p, int layer, int step, double* out_host, void* stream);
/* Per-phase device time of the last run in milliseconds:
 * [0]=unfold [1]=gram [2]=eig [3]=project [4]=reconstruct [5]=fold/update, [6]=jacobi sweeps (count) */
int tadmm_plan_last_timing(tadmm_plan p, double out_ms[8]);
int tadmm_plan_enable_timing(tadmm_plan p, int on);
/* Jacobi tunables: tol = largest relative off-diagonal a sweep may observe and still be the last one
 * (quadratic convergence leaves ~tol^2 afterwards; default 1e-9), inner_sweeps = reserved (the 16x16
 * sub-problems always get one cyclic sweep per visit), max_sweeps = cap before TADMM_ERR_NOCONVERGE.
 * <=0 keeps a value. */
int tadmm_plan_set_jacobi(tadmm_plan p, double tol, int inner_sweeps, int max_sweeps);
/* Filtered eigen-solver statistics (csrc/filter.hip): out[0] = eigen-problems of the plan served by the Chebyshev-
 * filtered subspace path (those whose kept rank is a fraction of their size), and for the LAST run out[1] = filtered
 * solves performed, out[2] = how many of them failed their a-posteriori check and were redone by the full Jacobi
 * solve, out[3] = largest number of filter stages a level needed.  TADMM_FILTER=0 in the environment disables the
 * path (every problem takes the full solve). */
int tadmm_plan_filter_stats(tadmm_plan p, int32_t out[4]);
/* With timing enabled (tadmm_plan_enable_timing) the fp64 GEMM launches of the filtered eigen-solver
 * (dgemm_nt_tile_kernel: block products, Gram matrices, Rayleigh-Ritz projection, residuals) are timed one by one
 * with HIP events on the launch stream: out[0] = their summed duration in ms, out[1] = number of launches,
 * out[2] = floating-point operations they executed (2*M*N*K of every product that was not gated off). */
int tadmm_plan_filter_timing(tadmm_plan p, double out[4]);
/* The same for the block products that ran at fp32 accuracy on the bf16 matrix cores (dgemm3_kernel: every filter
 * stage but the last one a level needed in the previous run; TADMM_FILTER_FAST=0 keeps all products in fp64):
 * out[0] = ms, out[1] = launches, out[2] = ALGORITHMIC flops 2*M*N*K (six bf16 products are executed per flop pair). */
int tadmm_plan_filter_timing_fast(tadmm_plan p, double out[4]);
/* The same for the launches of jacobi_tick3_kernel (the block-Jacobi tournament of the Rayleigh-Ritz and full solves):
 * out[0] = ms, out[1] = launches, out[2] = matrix-core flops the launches executed (2560 * row length per workgroup of
 * a problem the host did not yet know to be finished: cross Gram + two rounds of column updates), out[3] = those
 * workgroups. */
int tadmm_plan_jacobi_timing(tadmm_plan p, double out[4]);
/* clamped ranks of a layer (r_0..r_d); returns d+1 */
int tadmm_plan_ranks(tadmm_plan p, int layer, int32_t* ranks_out);
/* Lanes.  A plan whose table mixes long chains of eigen-solves (e.g. the 3x3 kernels of ResNet layer3/layer4) with
 * short ones runs as two sub-plans on two device streams -- the long chains on a high-priority stream, the rest
 * filling the CUs they leave idle; the caller's stream is joined in front and behind, the second lane is driven by a
 * worker thread owned by the plan.  Returns the number of lanes (1 or 2) and, when lane_of_out is not NULL, each
 * layer's lane.  A big table of like layers (>= 16, every chain within the threshold of the longest) is split into two halves.
 * TADMM_LANES=1 in the environment keeps every plan in one lane; TADMM_LANE_THRESHOLD (default 0.6)
 * is the fraction of the longest modelled chain from which a layer counts as long. */
int tadmm_plan_lanes(tadmm_plan p, int32_t* lane_of_out);
/* The split rule alone (pure host function of the descriptors, no device needed): what tadmm_plan_create would do. */
int tadmm_lane_split(int n_layers, const tadmm_layer_desc* descs, int32_t* lane_of_out);
int tadmm_plan_destroy(tadmm_plan p);

/* ---- Tucker-2 projection plan (admm.py:113-127: tensorly partial_tucker + tucker_to_tensor) ---- */
/* All layers of kind TADMM_KIND_TUCKER2 (ranks[0] = r_out, ranks[1] = r_in; 4-D (O,I,kh,kw) or 2-D (out,in))
 * are processed together: HOSVD initialisation, then HOOI sweeps until each layer meets tensorly's stopping
 * rule (|err_it - err_{it-1}| < hooi_tol from the third sweep, at most hooi_max_iter sweeps; 0 selects the
 * tensorly defaults 1e-4 / 100), then Z = core x_0 U_out x_1 U_in, U += W-Z and ||W-Z||^2 as in tadmm_plan_run.
 * The stopping rule is evaluated on the device; layers that are finished drop out of the later grouped launches.
 * PARITY UNPINNED: tensorly is not vendored by the reference. */
typedef struct tadmm_tucker_plan_s* tadmm_tucker_plan;
int tadmm_tucker_workspace_bytes(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, size_t* bytes);
int tadmm_tucker_create(tadmm_handle h, int n_layers, const tadmm_layer_desc* descs, const float* const* W,
                        float* const* U, float* const* Z, void* workspace, size_t workspace_bytes,
                        tadmm_tucker_plan* out);
int tadmm_tucker_run(tadmm_tucker_plan p, int update_u, int use_u, double* resid_sq_dev, void* stream);
/* Device pointers (inside the workspace, valid after a run) of a layer's factors:
 * core (r_out, k2, r_in) row-major -- note the kernel-position axis in the middle --, U_out (O, r_out),
 * U_in (I, r_in).  Columns beyond the number of singular values of an unfolding are zero. */
int tadmm_tucker_factors(tadmm_tucker_plan p, int layer, const float** core, const float** u_out, const float** u_in);
/* HOOI sweeps each layer ran in the last call and its final relative reconstruction error (host arrays of
 * n_layers entries, either may be NULL); synchronises the stream. */
int tadmm_tucker_iterations(tadmm_tucker_plan p, int32_t* iters_out_host, double* errors_out_host, void* stream);
/* Jacobi sweeps summed over every eigen-solve group of the last run (HOSVD start + two per HOOI sweep): the figure the
 * warm start of the HOOI solves lowers; host-side counter, no synchronisation.  Negative on a null plan. */
int tadmm_tucker_jacobi_sweeps(tadmm_tucker_plan p);
/* Instrumented runs (bench.py): every launch of the eigen-solver is bracketed by HIP events on the launch stream (adds a
 * stream sync per launch).  last_timing: out[0] = summed ms of those launches in the last run, [1] = their number,
 * [2] = the 8 N^3 model FLOPs they stand for, [3] = ms of the whole run, [4] = HOOI sweeps (max over layers). */
int tadmm_tucker_enable_timing(tadmm_tucker_plan p, int on);
int tadmm_tucker_last_timing(tadmm_tucker_plan p, double out[8]);
int tadmm_tucker_destroy(tadmm_tucker_plan p);

/* ---- augmented-Lagrangian penalty (admm.py:80-85) ---------------------- */
/* loss_dev[0] += 0.5*rho*sum_i ||W_i - Z_i + U_i||^2 ; gradW[i] (nullable) = grad_scale*(W_i-Z_i+U_i)
 * (grad_scale = rho * upstream gradient of the scalar loss).
 * ptrs_dev: device array of 4*n pointers laid out [W_0..W_{n-1} | Z.. | U.. | gradW..] ;
 * numel_dev: device int64[n]; partial_dev: device double scratch >= tadmm_penalty_scratch_doubles(). */
int tadmm_penalty_scratch_doubles(void);
int tadmm_penalty(tadmm_handle h, int n, const void* const* ptrs_dev, const int64_t* numel_dev,
                  int64_t total_numel, float rho, float grad_scale, double* loss_dev, double* partial_dev,
                  void* stream);

/* ---- building blocks (also used by the factorised layers) -------------- */
/* C[i,j] = alpha * sum_k A(i,k) B(k,j) (+ beta*C) with arbitrary element strides; one of the two
 * strides of each operand must be 1.  Replaces the torch.mm / F.linear chains of
 * TTLinear.py:79-86, TTConv.py:133-147, TKConv.py:210-214, TKLinear.py:66-71 and np.dot of
 * ttd.py:39-40.  fp32 in, fp32 MFMA accumulate. */
typedef struct {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
  float alpha, beta;
  const float* bias_n;   /* nullable: added along N (C[i,j] += bias_n[j]) */
  const float* bias_m;   /* nullable: added along M */
} tadmm_gemm_desc;
/* pack: writes GemmDesc[n] + block map into a host blob the caller uploads (and may cache) itself;
 * run: launches one grouped kernel over the uploaded blob. */
size_t tadmm_gemm_pack_bytes(int n, const tadmm_gemm_desc* descs);
int tadmm_gemm_pack(int n, const tadmm_gemm_desc* descs, void* blob_host, size_t blob_bytes, int* nblocks_out);
int tadmm_gemm_run(tadmm_handle h, const void* blob_dev, int n, int nblocks, void* stream);
/* One GEMM, descriptor passed by value to the kernel (no upload): the per-call path of the layers' forward and
 * backward products (TTLinear.py:79-86, TTConv.py:133-147, TKConv.py:210-214, TKLinear.py:66-71). */
int tadmm_gemm(tadmm_handle h, const tadmm_gemm_desc* desc, void* stream);
/* bf16 inference path of the TT-linear chain (TTLinear.py:79-86; every product there is A * Bt^T with both
 * operands contiguous along K):  C[M][N] = A[M][K] * Bt[N][K]^T (+ bias_n[j]), bf16 in / bf16 out, fp32
 * accumulate on the matrix cores.  lda / ldb / ldc in elements. */
int tadmm_gemm_bf16_nt(tadmm_handle h, const void* A, const void* Bt, void* C, int M, int N, int K, int64_t lda,
                       int64_t ldb, int64_t ldc, const float* bias_n, void* stream);

/* ---- forward chains of the factorised layers (csrc/chain.hip) ---------------------------------------------------
 * One launch per chain on the bf16 matrix cores; the per-token intermediate stays in LDS.
 *   fused  : Y[t][:] = Wout (Win X[t][:]) + bias     TTLinearM (TTLinear.py:75-93); Win (R x Kin) / Wout (Nout x R)
 *            are the contracted input / output cores, R the middle TT rank (multiple of 64, <= 256; pad with zeros).
 *   single : Y[t][:] = Win X[t][:] + bias            N = R output features, any size.
 * dtype TADMM_CHAIN_F32: X, Y float32; weights as THREE bf16 planes (w = w1 + w2 + w3 exactly, plane p at
 *   W + p * plane_stride); six bf16 products per fp32 product, fp32 accumulate: fp32-GEMM accuracy.
 * dtype TADMM_CHAIN_BF16: X, Y bfloat16; one weight plane.
 * Layouts: x_hw / y_hw == 0: token rows of ldx / ldy elements.  x_hw / y_hw > 0: channels-first images
 *   (batch, channel, pixel) of that many pixels -- the NCHW tensors of TTConv.py:131 / TKConv.py:94 in place.
 * Weights are FRAGMENT-MAJOR (one MFMA operand = one contiguous KiB): element (row n, col k) of plane p of an
 *   N x K weight lives at W[p*plane + (((n/16)*KS + k/32)*64 + (k%32/8)*16 + n%16)*8 + k%8], KS = ceil(K/32), rows
 *   padded to 16 and columns to 32 with zeros (tadmm.ops.weight_planes builds it).  Token rows: Kin % 8 == 0
 *   (% 4 for float32) and 16-byte aligned rows.  bias: float32[N], 16-byte aligned, or NULL.  tile_tokens: 0 (default), 32 or 64. */
enum { TADMM_CHAIN_F32 = 0, TADMM_CHAIN_BF16 = 1 };
typedef struct {
  const void* X; void* Y;
  const void* Win; const void* Wout;          /* bf16 planes */
  const float* bias;
  int64_t T;                                   /* tokens (rows, or batch * pixels) */
  int32_t Kin, R, Nout;
  int64_t ldx, ldy, win_plane, wout_plane;   /* elements */
  int32_t x_hw, y_hw;
  int32_t dtype, tile_tokens;
} tadmm_chain_desc;
/* sizeof(tadmm_chain_desc) as the library was built (bindings check their struct layout against it) */
int tadmm_chain_desc_bytes(void);
/* TTLinearM forward (TTLinear.py:75-93), fused. */
int tadmm_ttlinear_fwd(tadmm_handle h, const tadmm_chain_desc* d, void* stream);
/* TTLinearM data gradient dX = (dY Wout) Win: the same fused kernel with X = dY, Win = Wout^T planes (R x Nout),
 * Wout = Win^T planes (Kin x R); weight gradients are plain products (tadmm_gemm). */
int tadmm_ttlinear_bwd(tadmm_handle h, const tadmm_chain_desc* d, void* stream);
/* TTConv2dM input-core chain (TTConv.py:131-137): image (B, C, H, W) -> (B, r, H, W), single product. */
int tadmm_ttconv_chain_in(tadmm_handle h, const tadmm_chain_desc* d, void* stream);
/* TTConv2dM output-core chain + bias (TTConv.py:141-151): (B, r, H', W') -> (B, O, H', W'), single product. */
int tadmm_ttconv_chain_out(tadmm_handle h, const tadmm_chain_desc* d, void* stream);
/* TKConv2dC first / last 1x1 stage (TKConv.py:93-98): per-pixel channel mixing, single product. */
int tadmm_tucker_1x1(tadmm_handle h, const tadmm_chain_desc* d, void* stream);

/* The whole factorised convolution of a SMALL image in one launch (csrc/convchain.hip): y = W3 conv_kxk(W1 x; Wc) + bias
 * for NCHW tensors with output rows of at most 64 pixels: one workgroup per tile of output rows (<= 64 output pixels, a halo of
 * <= 192 input pixels); the two intermediates stay in LDS.  TTConv2dM
 * (TTConv.py:130-153), TKConv2dC / TKConv2dM (TKConv.py:93-98, :210-214).  W1 (R1 x C), W2 (R2 x kh*kw*R1, tap-major:
 * column (dy*kw + dx)*R1 + c) and W3 (Nout x R2) are fragment-major bf16 planes as for tadmm_chain_desc, R1 and R2
 * multiples of 32 (zero padded), both <= 256; groups = 1.  Returns TADMM_ERR_UNSUPPORTED when the image or the
 * intermediates do not fit (the caller then uses tadmm_ttconv_chain_in / conv2d / tadmm_ttconv_chain_out). */
typedef struct {
  const void* X; void* Y;
  const void* W1; const void* W2; const void* W3;
  const float* bias;
  int64_t w1_plane, w2_plane, w3_plane;
  int32_t B, C, R1, R2, Nout;
  int32_t H, W, Ho, Wo, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
  int32_t dtype;                              /* TADMM_CHAIN_F32 | TADMM_CHAIN_BF16 */
} tadmm_conv_chain_desc;
int tadmm_conv_chain_desc_bytes(void);
int tadmm_ttconv_fused(tadmm_handle h, const tadmm_conv_chain_desc* d, void* stream);

/* G = A A^T (m<=n) or A^T A (m>n) of a row-major float32 m x n matrix, exact fp32 products
 * accumulated in fp64 on v_mfma_f64_16x16x4_f64.  G is written as double[Npad][ldg] (zero padded; see tadmm_gram_ld), N=min(m,n).
 * partial_dev: scratch of tadmm_gram_scratch_bytes(m,n). */
size_t tadmm_gram_scratch_bytes(int m, int n);
/* returns N=min(m,n); *Npad rows and *ld doubles per row of the G image */
int tadmm_gram_ld(int m, int n, int* Npad, int* ld);
int tadmm_gram_f64(tadmm_handle h, const float* A, int m, int n, double* G, int ldg,
                   void* partial_dev, size_t partial_bytes, void* stream);

/* Symmetric eigen-decomposition of double[N][N] G (row-major, symmetric PSD) by one-sided block
 * Jacobi.  evals_out: double[N] descending; evecs_out: double[N][N], row j = eigenvector j.
 * Synchronous (polls convergence).  scratch: tadmm_eigh_scratch_bytes(N). */
size_t tadmm_eigh_scratch_bytes(int N);
int tadmm_eigh_f64(tadmm_handle h, const double* G, int N, double* evals_out, double* evecs_out,
                   void* scratch_dev, size_t scratch_bytes, int* sweeps_out, void* stream);

/* ---- building blocks of the filtered eigen-solver (csrc/dgemm.hip, csrc/chol.hip), exposed for tests ---- */
/* C[M][N] = A[M][K] * B^T (b_transposed=1: B is [N][K]) or A * B (b_transposed=0: B is [K][N]); row-major fp64 on
 * v_mfma_f64_16x16x4_f64.  M, N multiples of 32, K multiple of 16, even leading dimensions. */
size_t tadmm_dgemm_scratch_bytes(int M, int N);
int tadmm_dgemm_f64(tadmm_handle h, const double* A, const double* B, double* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int b_transposed, void* scratch, size_t scratch_bytes, void* stream);
/* C[M][N] = A[M][N] * G[N][N]^T at fp32 accuracy on the bf16 matrix cores (csrc/dgemm3.hip: the block products of the
 * filter's early stages; every value rounded to fp32 and split exactly into three bf16 terms, six MFMA products per
 * fp32 product).  M, N multiples of 32; `repeats` launches of the product (timing). */
size_t tadmm_dgemm3_scratch_bytes(int M, int N);
int tadmm_dgemm3_f64(tadmm_handle h, const double* A, const double* G, double* C, int M, int N, int lda, int ldg, int ldc,
                     int repeats, void* scratch, size_t scratch_bytes, void* stream);
/* Cholesky QR of a block stored as its transposed image YT[n][ldy] (row j = column j, `ncols` entries): on return the
 * rows are orthonormal (one pass: to ~cond^2 * 1e-16).  *bad_out_host = 1 when a pivot broke down (numerically
 * rank-deficient block; YT is then unspecified).  n multiple of 32, <= 256; ncols multiple of 64.  Synchronous. */
size_t tadmm_cholqr_scratch_bytes(int n, int ncols);
int tadmm_cholqr_f64(tadmm_handle h, double* YT, int n, int ncols, int ldy, void* scratch, size_t scratch_bytes,
                     int* bad_out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TADMM_H_ */
