/*
 * teal_hip.h — C ABI of libteal_hip.so, the MI355X (gfx950) implementation of TEAL's
 * activation-sparsity decode hot path.
 *
 * Every entry point replaces one piece of the reference's Triton path; citations are
 * into the reference tree (FasterDecoding/TEAL @ 2024-10-22):
 *
 *   teal_sparse_gemv       <- kernels/sparse_gemv.py:87-142  splitk_sparse_gemv()  + :50-83 kernel
 *                             (+ the init_to_zero("Y") pre-hook launch, :8-12, which disappears)
 *   teal_sparse_qkv_gemv   <- kernels/sparse_gemv.py:196-237 qkv_gemv()            + :152-194 kernel
 *   teal_compact           <- kernels/sparse_gemv.py:75      idx = tl.abs(x0) > threshold
 *                             (exposed standalone so index sets can be tested bit-exactly)
 *   teal_dense_gemv        <- kernels/sparse_gemv.py:301-307 DenseGEMV / torch.matmul(x, W.T) at S == 1
 *   teal_sparse_gateup_silu <- gpt-fast/model.py:258-259       silu(gemv1(x, w1)) * gemv1(x, w3)
 *   teal_fused_gemv        <- gpt-fast/model.py:158-161,289-291 residual adds + RMSNorm, :258-259 silu * up,
 *                             folded into the GEMV launch as producers
 *   teal_decode_attention* <- gpt-fast/model.py:170-186       RoPE, kv_cache.update, SDPA at S == 1
 *                             (TEAL_OUT_QKV_ROPE + teal_decode_attention_split_roped: RoPE and the cache append in the wqkv
 *                             projection's epilogue, :170-178, SDPA in the attention launch)
 *   teal_sample_topk       <- gpt-fast/generate.py:49-66      logits_to_probs + multinomial_sample_one
 *   teal_sparse_qkv_gemv_i8 <- gpt-fast/quantize.py:339-357   WeightOnlyInt8Linear.forward on the masked x
 *   teal_sparse_qkv_gemv_i4 <- gpt-fast/quantize.py:58-162,483-526 group-quantised int4 linear on the masked x
 *   teal_cmp_flag_gemv     <- scripts/benchmark_gemv.py:32-107,170-172 the Deja Vu comparator of the kernel benchmark
 *   teal_prefill_*         <- kernels/sparse_gemv.py:271,298 (dense matmul when S > 1) + gpt-fast/model.py:107-121,158-186,258-259:
 *                             the prompt pass of a short prompt, counted by the reference's tokens/sec (generate.py:458,487-496)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions across the boundary.
 *   - All pointers are DEVICE pointers owned by the caller (PyTorch); the library borrows them for
 *     the duration of the stream-ordered launch and never allocates.  Library state: the properties of each device,
 *     cached the first time it is used (teal_init(); immutable), and a host-side registry of the workspaces prepared by
 *     teal_workspace_init() (which device memory holds a valid header; mutex-protected).  Nothing else: no entry point
 *     of libteal_hip.so changes how a later call behaves, and what a call did is reported through its own arguments
 *     (teal_gemv_out_t.desc).  The tuning / phase-stamp switches of the last section exist only in the DIAGNOSTICS
 *     build of the same sources, libteal_hip_diag.so (-DTEAL_DIAGNOSTICS), which the product path never loads.
 *     No device memory belongs to the library: the arrival counters of the single-launch split-K GEMVs and the scratch of the
 *     multi-workgroup sampler live in the header of the CALLER's workspace, so two streams, two captured graphs or two
 *     devices can only collide if the caller hands them the same workspace.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Every call is
 *     asynchronous, allocation-free and hipGraph-capture safe.
 *   - dtype: 0 = fp16, 1 = bf16 (x, weights and y share it; int8 weights carry scales in that dtype).
 *   - Weight layout: the reference's "column major" weight[N, Z] with strides (1, N), i.e. the
 *     memory image is W^T row-major [Z][N]: element (m, n) at wT[m * N + n]
 *     (kernels/sparse_gemv.py:68,106).  N % 8 == 0 (16-byte rows), 1 <= Z <= 65536.
 *   - Keep rule: row m is kept iff float32(|x[m]|) > float32(tau)  (strict; kernels/sparse_gemv.py:75).
 *     In the GEMV entry points a NaN x[m] additionally propagates NaN into every output column of its
 *     threshold group, as the reference's `0 * NaN` on masked rows does.
 *   - Arithmetic: fp32 multiply-accumulate over the kept rows, ONE rounding to dtype at the end
 *     (the reference rounds through fp16 atomics, kernels/sparse_gemv.py:83); no atomics, results are
 *     bit-reproducible run to run.
 *   - Return value: 0 on success, negative TEAL_ERR_* otherwise (teal_strerror()).
 */
#ifndef TEAL_HIP_H
#define TEAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEAL_OK 0
#define TEAL_ERR_ARG (-1)       /* null pointer / non-positive size */
#define TEAL_ERR_DTYPE (-2)     /* dtype not 0 (fp16) / 1 (bf16) */
#define TEAL_ERR_SHAPE (-3)     /* N % 8 != 0, Z > 65536, bad N_q / N_kv */
#define TEAL_ERR_ALIGN (-4)     /* pointer not 16-byte aligned */
#define TEAL_ERR_WORKSPACE (-5) /* workspace too small (teal_workspace_bytes) */
#define TEAL_ERR_LAUNCH (-6)    /* hipGetLastError() after the launch */
#define TEAL_ERR_NO_DEVICE (-7) /* no HIP device / teal_init failed */
#define TEAL_ERR_CONFIG (-8)    /* invalid tuning override */

#define TEAL_F16 0
#define TEAL_BF16 1

int teal_version(void);
const char* teal_strerror(int code);

/* Cache the immutable properties of the CURRENT device (CU count, per-kernel attributes).  Call once per device
 * before any launch that may happen during stream capture (every entry point does it lazily otherwise).  Returns the
 * CU count (> 0) or a negative error. */
int teal_init(void);

/* Bytes of workspace a GEMV with N output columns may need: the library's header (arrival counters, sampler scratch)
 * plus an upper bound of the fp32 split-K slabs over all launch geometries (the deepest split x two N-column
 * segments).  Does not depend on Z — accepted for symmetry with the GEMV entry points.  The caller allocates once and
 * reuses; distinct streams need distinct workspaces. */
size_t teal_workspace_bytes(int Z, int N);

/* Prepare a workspace: zero its header (asynchronously on `stream`) and remember the pointer.  Once per allocation,
 * before the first launch that uses it and not under stream capture.  A prepared workspace lets a split-K GEMV with a
 * rounded output run as ONE launch (per-tile arrival tickets; the last slice to arrive sums the partials in slice
 * order) and a large-vocabulary sampler as several workgroups; every launch re-arms what it used, so a graph that was
 * captured with the workspace can be replayed indefinitely.  An unprepared workspace (plain memory of
 * teal_workspace_bytes) is still valid everywhere: 16-bit weights give the SAME bits from GEMV + ordered reduce launch (the
 * same slices summed in the same order) and the sampler the same tokens from a single workgroup; int8 / int4 weights pick
 * their split-K factor by whether tickets are available, so there the two ways agree to the tolerance of the fp32
 * summation order (one ulp of the 16-bit output), not bit for bit.  If a launch is aborted (device reset), prepare again.  teal_workspace_release() forgets the
 * pointer; call it before freeing the memory. */
int teal_workspace_init(void* ws, size_t ws_bytes, void* stream);
int teal_workspace_release(void* ws);

/* idx_out[0..count) = ascending m with float32(|x[m]|) > float32(tau); *count_out = count.
 * idx_out must hold Z int32.  (kernels/sparse_gemv.py:75) */
int teal_compact(const void* x, float tau, int Z, int dtype, int32_t* idx_out, int32_t* count_out,
                 void* stream);

/* y[n] = sum_{m kept} wT[m*N + n] * x[m]            (kernels/sparse_gemv.py:87-142) */
int teal_sparse_gemv(const void* x, const void* wT, void* y, float tau, int Z, int N, int dtype,
                     void* ws, size_t ws_bytes, void* stream);

/* Same on the fused wqkv weight with a threshold per column range:
 * columns [0,N_q) tau_q, [N_q,N_q+N_kv) tau_k, [N_q+N_kv,N) tau_v   (kernels/sparse_gemv.py:196-237;
 * N_q = N - 2*kv_size, N_kv = kv_size there).  N_q and N_kv must be multiples of 8. */
int teal_sparse_qkv_gemv(const void* x, const void* wT, void* y, float tau_q, float tau_k,
                         float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                         size_t ws_bytes, void* stream);

/* Same with an explicit row stride `ld >= N` (elements, multiple of 8) of the W^T image, i.e. weight[N, Z]
 * with strides (1, ld).  A stride that is NOT a multiple of 1024 bytes (ld = N + 64) makes consecutive
 * rows start in different 128-byte DRAM-channel residues, so a workgroup's column tile rotates over all
 * channels instead of hammering one. */
int teal_sparse_qkv_gemv_ld(const void* x, const void* wT, int ld, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                            size_t ws_bytes, void* stream);

/* int8 weight-only variant of teal_sparse_qkv_gemv_ld (the reference ships int8 only for its
 * dense path, gpt-fast/quantize.py:339-357, and lists quantised TEAL as missing, README.md:110).  wqT = int8 image of
 * W^T, row-major [Z][ld] bytes (ld % 8 == 0, ld >= N); scale[N] in the activation dtype.  N_kv = 0, N_q = N: one
 * threshold (tau_q). */
int teal_sparse_qkv_gemv_i8(const void* x, const void* wqT, const void* scale, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int ld, int dtype, void* ws, size_t ws_bytes,
                            void* stream);

/* y = x @ W^T with every row kept (prefill-free decode of un-sparsified layers, e.g. lm_head).
 * (kernels/sparse_gemv.py:301-307) */
int teal_dense_gemv(const void* x, const void* wT, void* y, int Z, int N, int dtype, void* ws,
                    size_t ws_bytes, void* stream);

/* int4 group-quantised weight-only variant (the reference ships int4-g32/64/128/256 for its dense
 * path only: gpt-fast/quantize.py:58-162 group q-params / quantise / dequantise, :483-526 WeightOnlyInt4Linear over a
 * CUDA-only packed layout).  w[n][m] = (q - 8) * scale[m / G][n] + zero[m / G][n], q in 0..15.
 *   wq               nibble image of W^T by ROW PAIRS: [Z / 2][ldb] BYTES (ldb >= N, ldb % 8 == 0, 8-byte aligned; for speed
 *                    ldb % 128 == 0 and a 128-byte aligned base, so that a tile's 128-byte segment is ONE memory line); the
 *                    32-bit word g of pair-row p holds columns 4g .. 4g+3 of row 2p in its low half (nibble j = column
 *                    4g + j) and of row 2p + 1 in its high half — two rows of a column unpack into one half2 for a packed
 *                    dot product; a pair is fetched when either of its rows is kept
 *   scales_and_zeros bf16 [Z / G][N][2] = {scale, zero}, exactly the reference's tensor (quantize.py:79-93)
 * N, N_q, N_kv multiples of 128; Z a multiple of G; x / y fp16 or bf16 (dtype).  N_kv = 0, N_q = N: one threshold.
 * One launch (split-K over 32-row units folded in by arrival tickets); fp32 accumulation, scale / zero applied once per
 * (unit, column), one rounding. */
int teal_sparse_qkv_gemv_i4(const void* x, const void* wq, const void* scales_and_zeros, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int ldb, int groupsize, int dtype, void* ws,
                            size_t ws_bytes, void* stream);

/* ---- fusions around the path ------------------------------------------ */

/* h[n] = silu(gate[n]) * up[n] with gate = sparse_gemv(x, w1T, tau_gate), up = sparse_gemv(x, w3T,
 * tau_up), both [Z][N]; gate/up are rounded to dtype before the activation and the product, as the
 * unfused reference sequence does (gpt-fast/model.py:258-259).  One launch, one shared read of x. */
int teal_sparse_gateup_silu(const void* x, const void* w1T, const void* w3T, void* h, float tau_gate,
                            float tau_up, int Z, int N, int dtype, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- fused decode step: producers/consumers folded into the GEMV launch - */

#define TEAL_IN_PLAIN 0      /* x given as is */
#define TEAL_IN_RESID_NORM 1 /* x = RMSNorm(resid + round(sum slabs)) * w   (model.py:158-161, 289-291) */
#define TEAL_IN_SILU_MUL 2   /* x = silu(gate) * up, gate|up contiguous [2Z] (model.py:258-259) */
#define TEAL_IN_MASKED 3     /* x given together with its keep masks (one uint64 per 64 elements, bit i =
                              * element 64*c+i kept) as emitted by the producing launch: the consumer skips
                              * the compare/ballot phase.  The masks must be those of tau[0]. */
#define TEAL_IN_ATTN_MERGE 4 /* x = attention output merged from the split-KV partials per head written by
                              * teal_decode_attention_split(y = NULL, nsplit = 4 or 8): x points at the fp32
                              * partials [Z/head_dim][nsplit][head_dim + 2]; att_head_dim = head_dim,
                              * att_nsplit = nsplit (0 means 4); Z <= 16384 (nsplit 4) / 8192 (nsplit 8) */
#define TEAL_OUT_ROUNDED 0   /* y rounded to dtype (runs the ordered slab reduce when split-K is used) */
#define TEAL_OUT_SLABS 1     /* leave the fp32 split-K slabs for the next launch's RESID_NORM producer */
#define TEAL_OUT_QKV_ROPE 3  /* nseg == 3 = q | k | v of ONE fused wqkv image, RESID_NORM input: when the launch needs no
                              * split-K (and the lean kernel takes it: *nslabs_out = 0), its epilogue applies RoPE to q and
                              * to the new k row and appends k, v to the caches (gpt-fast/model.py:170-178): y[0] = rotated
                              * q [ncols[0]], rounded exactly as teal_decode_attention* would have; follow it with
                              * teal_decode_attention_split_roped.  Otherwise (*nslabs_out >= 1) the launch behaved as
                              * TEAL_OUT_SLABS (slabs required) and teal_decode_attention_split_slabs finishes the job */
#define TEAL_OUT_SLAB_SUM 4  /* nseg == 1: `slabs` (plain memory, fp32 [ncols]) receives the UNROUNDED sum over the row slices,
                              * added in slice order by the last slice of each tile to arrive (arrival tickets: `ws` must be a
                              * prepared workspace when the launch uses split-K) — bit-identical to summing the TEAL_OUT_SLABS
                              * slabs in slice order.  *nslabs_out = 1; the consumer is a RESID_NORM producer with nslabs = 1,
                              * slabs_interleaved = 0.  Tensor parallelism: the [ncols] fp32 vector is what the ranks all-reduce
                              * (gpt-fast/tp.py:120-121,139-140) instead of [ncols][4..8] slabs.  16-bit and int8 weights; shapes
                              * outside the lean kernel return TEAL_ERR_CONFIG without launching */
#define TEAL_OUT_PAIR_SILU 2 /* nseg == 2 (gate, up of equal shape): every workgroup streams the same column tile
                              * of both matrices and stores h = silu(gate) * up to y[0] (model.py:258-259); optional
                              * mask_out[ncols/64] = keep masks of h against mask_tau for a TEAL_IN_MASKED consumer */

typedef struct teal_gemv_in {
    int mode;                 /* TEAL_IN_* */
    const void* x;            /* PLAIN: [Z];  SILU_MUL: gate[Z] followed by up[Z] */
    const void* resid_in;     /* RESID_NORM: residual stream [Z] (or a [rows][Z] table with row_index) */
    const int32_t* row_index; /* RESID_NORM, optional: device int32; row = row_index[0] (embedding lookup) */
    const float* slabs;       /* RESID_NORM, optional: fp32 [nslabs][Z] partial sums folded into the residual */
    int nslabs;
    const void* norm_weight;  /* RESID_NORM: RMSNorm weight [Z] */
    float eps;
    void* resid_out;          /* RESID_NORM, optional: updated residual [Z]; must not alias resid_in */
    const void* masks;        /* MASKED: uint64 [ceil(Z/64)] keep masks of x */
    int att_head_dim;         /* ATTN_MERGE: head_dim */
    int att_nsplit;           /* ATTN_MERGE: partials per head, 4 or 8 (0 = 4) */
    int slabs_interleaved;    /* RESID_NORM: slabs are [Z][(nslabs+3)&~3] (as written by a producer with
                               * slabs_interleaved = 1) instead of planar [nslabs][Z]; nslabs <= 8 */
    int gate_activated;       /* SILU_MUL: the gate half already holds round(silu(gate)) — written by a launch with
                               * act_seg0 = 1 — so x = round(gate * up): the same roundings as the unfused sequence
                               * (model.py:258-259), the activation computed once per column instead of in every consumer
                               * workgroup's prologue.  16-bit and int8 weights (not the int4 kernel) */
} teal_gemv_in_t;

typedef struct teal_gemv_out {
    int nseg;            /* 1..3 column segments, each with its own threshold (q|k|v, gate|up, ...) */
    const void* w[3];    /* weight image of the segment: row-major [Z][ld] */
    int ld[3];           /* row stride (elements) */
    int col0[3];         /* first column of the segment inside a row */
    int ncols[3];        /* columns in the segment */
    float tau[3];        /* keep threshold of the segment */
    void* y[3];          /* ROUNDED: output of the segment [ncols] */
    int mode;            /* TEAL_OUT_* */
    float* slabs;        /* SLABS: destination, fp32 [nslabs][sum ncols]; plain memory, NOT a prepared workspace */
    size_t slabs_bytes;
    void* mask_out;      /* PAIR_SILU, optional: uint64 [ceil(ncols/64)] keep masks of h */
    float mask_tau;      /* PAIR_SILU: threshold of the consumer (the down projection) */
    int slabs_interleaved; /* SLABS: write slabs[col][slice] with row stride (nslabs+3)&~3 so that the consumer
                            * fetches every partial of an element with one 16-byte load */
    int weight_bits;       /* 0 or 16: fp16/bf16 weights (the activation dtype); 8: int8 weight-only quantisation
                            * (gpt-fast/quantize.py:339-357): w[i] = int8 image of W^T, row-major [Z][ld] BYTES, and
                            * scale[i] = per-output-column scales in the activation dtype (element 0 = column col0[i]);
                            * y = round(fp32(sum q*x) * fp32(scale)) — one rounding, where the reference's
                            * F.linear(x, w.to(dtype)) * scales rounds twice */
    const void* scale[3];  /* int8: see above.  int4 (weight_bits = 4, gpt-fast/quantize.py:58-162, 483-526): w[i] = packed image
                            * of W^T by row pairs, [Z / 2][ld BYTES] (see teal_sparse_qkv_gemv_i4); scale[i] = the
                            * reference's scales_and_zeros tensor of that image, bf16 [Z / groupsize][scale_ld[i]][2]; col0
                            * addresses both.  ncols multiples of 128; in modes PLAIN, RESID_NORM (interleaved slabs, <= 8),
                            * SILU_MUL, ATTN_MERGE; out modes ROUNDED and SLABS */
    int scale_ld[3];       /* int4 only: columns per group row of scale[i] (the image's N) */
    int groupsize;         /* int4 only: 32, 64, 128 or 256 rows per quantisation group; Z a multiple */
    /* TEAL_OUT_QKV_ROPE only (zero otherwise): */
    const void* rope;          /* (cos, sin) table [rope_max_seq][rope_head_dim / 2][2], activation dtype */
    const int32_t* rope_pos;   /* device int32: position of the token being decoded (clamped to the cache) */
    void* k_cache;             /* [ncols[1] / rope_head_dim][rope_max_seq][rope_head_dim] */
    void* v_cache;
    int rope_head_dim;         /* 64 or 128 */
    int rope_max_seq;
    int act_seg0;              /* TEAL_OUT_ROUNDED: y[0] = round(silu(round(sum))) for segment 0 (the gate projection of an
                                * unpaired gate | up launch, model.py:258); the other segments are stored as usual.  16-bit
                                * and int8 weights (not the int4 kernel) */
    char* desc;                /* optional HOST buffer: receives the kernel template instantiation and grid of the launch THIS
                                * call made, NUL-terminated (as rocprofv3 prints it; e.g. to name the kernel in a benchmark
                                * record) — per call, no library state */
    int desc_bytes;            /* capacity of desc (160 is enough) */
} teal_gemv_out_t;

/* One launch: [fused producer] -> mask + compaction -> gathered GEMV over every segment.
 * *nslabs_out receives the split-K factor used (the number of slabs written in SLABS mode). */
int teal_fused_gemv(const teal_gemv_in_t* in, const teal_gemv_out_t* out, int Z, int dtype, void* ws,
                    size_t ws_bytes, int* nslabs_out, void* stream);

/* Single-token attention between gemv1 and gemv2 (gpt-fast/model.py:170-186): RoPE(q, k_new) with the
 * (cos, sin) table rope[max_pos][head_dim/2][2], KV-cache append at *pos, softmax(q K^T / sqrt(d)) V.
 * qkv = [q | k | v] as produced by the fused wqkv GEMV; caches are [n_kv_head][max_seq][head_dim];
 * y = [n_head * head_dim].  head_dim 64 or 128.  A position >= max_seq is clamped to the last cache slot (no
 * out-of-bounds write; the output of such a step is meaningless). */
int teal_decode_attention(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                          void* y, int n_head, int n_kv_head, int head_dim, int max_seq, int dtype, void* stream);

/* Same, also emitting the keep masks of y against mask_tau (uint64 [n_head*head_dim/64]) for a
 * TEAL_IN_MASKED wo projection. */
int teal_decode_attention_masked(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                 void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                 int max_seq, int dtype, void* stream);

/* Long-context form (flash-decoding): the cached positions of every head are split over `nsplit` workgroups
 * that write un-normalised partials {max, sum, o[head_dim]} (fp32, n_head*nsplit*(head_dim+2) floats), then
 * a merge launch rescales, sums, rounds once and emits the masks.  Use when max_seq is in the thousands:
 * one workgroup per head would leave most CUs idle while n_head of them stream the whole KV cache.
 * nsplit <= 64.  Grouped-query shapes (n_head / n_kv_head = 8 with max_seq >= 2048, = 4 with max_seq >= 4096) run as ONE workgroup per
 * (KV head, split) that serves all query heads of the group, so every K/V row is read once instead of once per query
 * head; choose nsplit so that n_kv_head * nsplit is about the CU count (32 for 8 KV heads).  Same partial format. */
int teal_decode_attention_split(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                int max_seq, int nsplit, void* partials, size_t partials_bytes, int dtype, void* stream);
/* Same, with the qkv projection taken from the fp32 split-K slabs of a TEAL_OUT_SLABS launch (slabs_interleaved = 1,
 * layout [col][(nslabs + 3) & ~3], nslabs <= 8): the kernel sums each element's partials in slice order and rounds
 * once — exactly what the ordered reduce launch would have written — so a narrow (GQA) wqkv can be row-sliced over
 * all CUs without a reduce launch between projection and attention. */
int teal_decode_attention_split_slabs(const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                      void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                      int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials,
                                      size_t partials_bytes, int dtype, void* stream);
/* General form: exactly one of qkv (rounded projection) / qkv_slabs (+ qkv_nslabs) is given.  Same launches and results as
 * teal_decode_attention_split / _split_slabs.  `ws` / `ws_bytes` are accepted and unused (round 3 folded the merge launch into
 * the split launch through arrival counters in a prepared workspace: bit-identical y, measured no faster — equal at 4-8
 * splits, slower at 16-32 — and removed in round 5). */
int teal_decode_attention_split_ws(const void* qkv, const float* qkv_slabs, int qkv_nslabs, const void* rope,
                                   const int32_t* pos, void* k_cache, void* v_cache, void* y, void* mask_out,
                                   float mask_tau, int n_head, int n_kv_head, int head_dim, int max_seq, int nsplit,
                                   void* partials, size_t partials_bytes, int dtype, void* ws, size_t ws_bytes, void* stream);
/* After a TEAL_OUT_QKV_ROPE projection that reported *nslabs_out = 0: q = the rotated, rounded query [n_head * head_dim];
 * the token's k / v rows are already in the caches.  Always the per-query-head split kernel (no grouped-query form); same
 * partials, merge and results as teal_decode_attention_split_ws. */
int teal_decode_attention_split_roped(const void* q, const int32_t* pos, const void* k_cache, const void* v_cache, void* y,
                                      void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim, int max_seq,
                                      int nsplit, void* partials, size_t partials_bytes, int dtype, void* ws, size_t ws_bytes,
                                      void* stream);
/* y == NULL: only the partials are written (no merge launch); with nsplit 4 or 8 a TEAL_IN_ATTN_MERGE wo
 * projection merges them in its own prologue (nsplit 4 or 8) — one launch less per layer, and 4 CUs per head pull the KV
 * cache instead of one (a single CU sustains ~50 GB/s, which bounds the one-workgroup-per-head kernel). */

/* Sampling step of the decode loop (gpt-fast/generate.py:49-66): logits / temperature, top-k filter
 * (ties at the pivot kept), softmax, exponential-race multinomial.  rng_state = device uint64[2]
 * {seed, draw counter}; the kernel bumps the counter so hipGraph replays draw fresh numbers.
 * top_k <= 0 or >= vocab disables the filter.  token_out = device int32[1] (may be the buffer the next
 * decode step reads its token from).  Optional in-graph loop-carried state, so that one graph replay
 * IS one decode step with no host-side glue: pos_inout[0] += 1; history[draw counter] = token.
 * teal_sample_topk_ws with a prepared workspace (teal_workspace_init): vocabularies of 8193..131072 entries (multiple
 * of 8) with an active filter run as one workgroup per 8192 logits: local top-k candidates -> workspace header -> the
 * last workgroup to arrive picks the token; same tokens as the single-workgroup kernels, which serve every other case
 * (no / unprepared workspace, other shapes). */
int teal_sample_topk(const void* logits, int vocab, int dtype, int top_k, float temperature, void* rng_state,
                     int32_t* token_out, int32_t* pos_inout, int32_t* history, int history_len, void* stream);
int teal_sample_topk_ws(const void* logits, int vocab, int dtype, int top_k, float temperature, void* rng_state,
                        int32_t* token_out, int32_t* pos_inout, int32_t* history, int history_len, void* ws,
                        size_t ws_bytes, void* stream);


/* ---- dense prompt pass for short prompts (T <= 16 tokens), teal_amd/csrc/teal_prefill.hip ----- */

/* The reference's prefill is dense (its ops run torch.matmul when the sequence is longer than one token,
 * kernels/sparse_gemv.py:271,298; the rest is the stock model, gpt-fast/model.py:107-121,158-186,258-259,289-291) and its
 * tokens/sec counts it (gpt-fast/generate.py:458,487-496).  These entry points make one layer of that pass seven launches over
 * the decode step's own weight images.  Every hand-over is TRANSPOSED, [feature][R] with R = 8 for T <= 8 and R = 16 for
 * 9 <= T <= 16 (every call of one pass takes the same T, hence the same R; "[..][8]" below reads "[..][R]"): the tokens of a
 * feature in one or two 16-byte words (16-bit activations: xt, ht, yt) or R / 4 of them (fp32 slabs [slice][feature][R]); token
 * slots >= T of the 16-bit vectors are written as zero, of the slabs left untouched (consumers ignore them).  1 <= T <= 16.
 * Floating-point results: fp32 sums, the rounding points of the module path's 16-bit tensors. */

/* What a GEMM launch builds its activations from (every workgroup builds the rows of its own slice, once, while staging them) */
#define TEAL_PREFILL_IN_XT 0        /* xt [Z][8] as given */
#define TEAL_PREFILL_IN_NORM 1      /* x = RMSNorm(h) * norm_w from the residual rows xt = ht [Z][8] and the per-workgroup sums of squares
                                     * sumsq [nwg][8] a teal_prefill_resid_norm call left in its scratch (nwg = Z / 256 rounded up) */
#define TEAL_PREFILL_IN_SILU_MUL 2  /* x = round(round(silu(round(gate))) * round(up)) from the slabs [gu_split][2 Z][8] of a gate | up launch */
typedef struct teal_prefill_in {
    int mode;
    const void* xt;
    const float* sumsq;
    int nwg;
    const void* norm_w;
    float eps;
    const float* gu_slabs;
    int gu_split;
} teal_prefill_in_t;

/* slabs[slice][n][s] = sum over the slice's rows m of W^T[m][n] * x[m][s]: w0T [Z][ld0] (n0 columns) and, optionally, w1T [Z][ld1]
 * (n1 columns, output columns n0 ..: gate | up in one launch).  Z, n0, n1 multiples of 256.  *split_out = slices written (<= 16;
 * the consumer sums them in slice order and rounds once); slabs must hold 16 * (n0 + n1) * R floats and must not be the buffer a
 * TEAL_PREFILL_IN_SILU_MUL launch reads. */
int teal_prefill_gemm(const teal_prefill_in_t* in, const void* w0T, int ld0, int n0, const void* w1T, int ld1, int n1, float* slabs,
                      size_t slabs_bytes, int Z, int T, int dtype, int* split_out, void* stream);
/* h = (embedding rows of tokens[0..T) | ht_in) + round(sum of `split` slabs) (split 0: nothing to add); x = RMSNorm(h) * norm_w.
 * Exactly one of tokens (+ emb [vocab][dim]) / ht_in is given; with neither xt_out nor x_last only the first of the two launches runs
 * (h and the sums of squares: a TEAL_PREFILL_IN_NORM GEMM normalises while it stages).  Outputs: ht_out [dim][8] (required; must not alias ht_in's
 * words of other columns — the same buffer is fine), xt_out [dim][8] and x_last [dim] (optional) = the normalised vector of
 * token T - 1 as a plain vector (input of the lm_head GEMV).  sumsq_scratch: (dim / 256 rounded up) * R floats of caller memory
 * (the per-workgroup sums of squares between the two launches this call makes).  dim <= 16384. */
int teal_prefill_resid_norm(const void* emb, const int32_t* tokens, int T, const void* ht_in, const float* slabs, int split,
                            const void* norm_w, float eps, int dim, void* ht_out, void* xt_out, void* x_last, float* sumsq_scratch,
                            int dtype, void* stream);
/* q | k | v from the slabs [split][(n_head + 2 n_kv_head) * head_dim][8] of the wqkv launch: RoPE(q, k) at positions 0 .. T-1,
 * cache rows 0 .. T-1 written, causal softmax(q K^T / sqrt(d)) V -> yt [n_head * head_dim][8].  head_dim 64 or 128. */
int teal_prefill_attention(const float* qkv_slabs, int split, const void* rope, void* k_cache, void* v_cache, void* yt, int T, int n_head,
                           int n_kv_head, int head_dim, int max_seq, int dtype, void* stream);

/* ---- benchmark comparator (scripts/benchmark_gemv.py only; not on the decode path) ----------- */

/* The Deja Vu gather GEMV the reference's kernel benchmark plots next to TEAL's (scripts/benchmark_gemv.py:32-107,170-172),
 * restated for CDNA4: flags[m] = |x[m]| > tau by a launch of its own, y32 zeroed, then a (row block, column tile) grid adds
 * fp32 partial sums of the flagged rows into y32 with atomics.  y32: fp32 [N]; flags: Z bytes of scratch.  Three launches. */
int teal_cmp_flag_gemv(const void* x, const void* wT, int ld, float* y32, unsigned char* flags, float tau, int Z, int N,
                       int dtype, void* stream);

/* ---- diagnostics build only (libteal_hip_diag.so = these sources with -DTEAL_DIAGNOSTICS) ------------------------
 * Process-global switches, NOT thread-safe, for benchmarks, phase-stamp probes and the parity tests that force the general
 * kernel or a launch geometry.  libteal_hip.so exports none of them. */
#ifdef TEAL_DIAGNOSTICS

/* Override the launch geometry picked from (Z, N, CU count): lanes per row segment (8/16/32/64) and split-K factor
 * (>= 1); waves per workgroup and unroll depth are fixed at 16 and 4 (the other values of round 1 were sweep-only
 * and are no longer built: pass 0 or the fixed value).  0 = automatic.  Process-global; meant for benchmark sweeps only. */
int teal_set_tuning(int lanes_per_row, int waves, int split, int unroll);

/* Wave-local compaction (default on): each wave streams the rows it ballots itself instead of an even share
 * of a workgroup-wide list; removes every barrier between the activation and the first weight load. */
int teal_set_wave_local(int on);

/* Diagnostics: kernel template instantiation and grid of the most recent GEMV launch of this process (host-side
 * string; e.g. for naming the kernel in a benchmark record). */
const char* teal_last_launch_desc(void);

/* Lean kernel for qualifying shapes (default on; 0 forces the general kernel everywhere: A/B and parity tests). */
int teal_set_fast(int on);

/* Diagnostics: when set (device pointer to >= 32 * workgroups uint64 — 32 stamps per workgroup of the launch), every
 * GEMV workgroup stores 100 MHz wall-clock stamps of its phases, row w of the buffer = workgroup w:
 *   [0] kernel entry, [1] kernel arguments in registers, [2] activation ready, [3] row list ready, [4] first weight
 *   batch consumed (wave 0), [5] wave 0 done streaming, [6] past the reduce barrier, [7] done,
 *   [12] hardware id << 32 | XCC id, [13] waves << 32 | workgroups, [16 + w] end of stream of wave w (w < 16).
 * The attention, sampler and int4 launches (the latter with fp16 activations only) stamp rows of the same width with their own
 * phase meanings (scripts/attn_phase.py, scripts/sampler_phase.py, scripts/int4_phase.py).  NULL (default) disables.
 * Process-global. */
int teal_set_phase_buffer(void* dev_u64);
/* > 0: consecutive GEMV / attention launches stamp consecutive regions of `u64_per_launch` uint64 of the phase buffer
 * (so that a chain of launches can be timed against each other): u64_per_launch >= 32 * the largest launch's
 * workgroups, and the buffer must hold u64_per_launch * (launches between two teal_set_phase_stride calls) uint64;
 * 0 (default): every launch stamps the start of the buffer. */
int teal_set_phase_stride(size_t u64_per_launch);
#endif /* TEAL_DIAGNOSTICS */

/* ---- introspection (pure function of its arguments and the device's CU count) ------------------ */

/* The geometry a GEMV of this shape would use: out[0..5) = {lanes_per_row, waves, split, unroll,
 * workgroups}. */
int teal_get_config(int Z, int N, int nseg, int* out);

#ifdef __cplusplus
}
#endif
#endif /* TEAL_HIP_H */
