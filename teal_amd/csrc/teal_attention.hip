// teal_attention.hip — batch-1 decode attention (single-workgroup and split-KV / flash-decoding forms) and the
// fused top-k sampler, with their C-ABI entry points (include/teal_hip.h).
//   gpt-fast/model.py:170-186   RoPE + kv_cache.update + SDPA at S = 1   -> decode_attention*_kernel
//   gpt-fast/generate.py:49-66  logits_to_probs + multinomial_sample_one -> sample_topk_kernel
#include "teal_common.h"

namespace teal {

constexpr int kPhaseRowA = 32;  // phase-stamp row of the split attention kernel (same layout as the GEMV's)

// ------------------------------------------------------------------------------------------------
// Single-token attention over a static KV cache (the step between gemv1 and gemv2 of
// gpt-fast/model.py:163-190): RoPE on q and the new k, KV-cache append, softmax(q K^T / sqrt(d)) V.
// One workgroup (256 threads) per query head; GQA by head group.  Rounding points follow the
// reference's fp16/bf16 tensors: rotated q/k, scores, probabilities and the output are rounded to
// dtype; accumulation is fp32.
// ------------------------------------------------------------------------------------------------
template <bool BF16, int NT, int HD>
__global__ __launch_bounds__(NT) void decode_attention_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ rope, const int* __restrict__ pos_ptr,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache, uint16_t* __restrict__ y,
    unsigned long long* __restrict__ mask_out, const float mask_tau,
    const int n_head, const int n_kv, const int max_seq, const float scale, unsigned long long* __restrict__ phase) {
    constexpr int NW = NT / 64;
    constexpr int hd = HD;
    auto stamp_a = [&](const int i) { if (phase && threadIdx.x == 0) phase[(size_t)blockIdx.x * 8 + i] = wall_clock64(); };
    stamp_a(0);
    constexpr int SL = HD / 8;   // 16-byte slices per row (16 for hd=128, 8 for hd=64)
    constexpr int RW = 64 / SL;  // V rows per wave step (4 or 8)
    constexpr int VPF = 256 / (NW * RW) > 0 ? 256 / (NW * RW) : 1;  // V steps prefetched: the first 256 rows (hd=128)
    extern __shared__ __align__(16) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);  // [hd] rotated q
    float* kn = qs + hd;                         // [hd] rotated new k
    float* vn = kn + hd;                         // [hd] new v
    float* red = vn + hd;                        // [2 * NW] block reductions
    float* part = red + 2 * NW;                  // [NW][hd] per-wave partial outputs
    float* sc = part + NW * hd;                  // [max_seq] scores / probabilities
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x;
    const int rep = n_head / n_kv;
    const int kvh = h / rep;
    // a position past the cache (a caller decoding more tokens than it allocated) must not write out of bounds:
    // it is clamped to the last slot — the output is then meaningless, the memory stays intact
    const int pos = min(max(pos_ptr[0], 0), max_seq - 1);
    const int dim = n_head * hd, kvs = n_kv * hd;
    const uint16_t* qh = qkv + (size_t)h * hd;
    const uint16_t* kh = qkv + dim + (size_t)kvh * hd;
    const uint16_t* vh = qkv + dim + kvs + (size_t)kvh * hd;
    uint16_t* kc = k_cache + (size_t)kvh * max_seq * hd;
    uint16_t* vc = v_cache + (size_t)kvh * max_seq * hd;

    // ---- everything that only depends on `pos` is requested first: this thread's cached K row and
    //      its V slices are in flight while q/k are rotated (one memory round trip instead of three)
    // (lane = (row-in-wave rw, 16-byte slice ds): a wave reads 64/SL whole rows = 1 KiB contiguous per load)
    const int ds = lane % SL, rw = lane / SL;
    u32x4 kreg[VPF], vreg[VPF];
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        kreg[i] = *reinterpret_cast<const u32x4*>(kc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
    }
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        vreg[i] = *reinterpret_cast<const u32x4*>(vc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
    }

    // RoPE on interleaved pairs (model.py apply_rotary_emb), table rows are (cos, sin) in dtype
    if (tid < hd / 2) {
        const float c = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2], BF16);
        const float sn = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2 + 1], BF16);
        const float q0 = bits_to_float(qh[2 * tid], BF16), q1 = bits_to_float(qh[2 * tid + 1], BF16);
        const float k0 = bits_to_float(kh[2 * tid], BF16), k1 = bits_to_float(kh[2 * tid + 1], BF16);
        const uint16_t qa = float_to_bits<BF16>(rope_even(q0, q1, c, sn)), qb = float_to_bits<BF16>(rope_odd(q0, q1, c, sn));
        const uint16_t ka = float_to_bits<BF16>(rope_even(k0, k1, c, sn)), kb = float_to_bits<BF16>(rope_odd(k0, k1, c, sn));
        qs[2 * tid] = bits_to_float(qa, BF16);
        qs[2 * tid + 1] = bits_to_float(qb, BF16);
        kn[2 * tid] = bits_to_float(ka, BF16);
        kn[2 * tid + 1] = bits_to_float(kb, BF16);
        if (h % rep == 0) {  // one writer per KV head
            kc[(size_t)pos * hd + 2 * tid] = ka;
            kc[(size_t)pos * hd + 2 * tid + 1] = kb;
        }
    } else if (tid >= 128 && tid < 128 + hd) {
        const int d = tid - 128;
        const uint16_t vb = vh[d];
        vn[d] = bits_to_float(vb, BF16);
        if (h % rep == 0) vc[(size_t)pos * hd + d] = vb;
    }
    __syncthreads();
    stamp_a(1);

    // scores: each lane multiplies its 8-dim slice, the SL lanes of a row are summed with DPP; the new
    // token's own key comes from LDS, not from the cache line being written
    float qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = qs[ds * 8 + j];
    float lmax = -INFINITY;
    auto score_row = [&](const int t, const u32x4 w) {
        float a = 0.0f;
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a += qv[j] * kn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += qv[2 * j] * bits_to_float(w[j] & 0xFFFFu, BF16);
                a += qv[2 * j + 1] * bits_to_float(w[j] >> 16, BF16);
            }
        }
        a = row_slices_sum<SL>(a);
        const float sv = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        if (t <= pos) {
            if (ds == 0) sc[t] = sv;
            lmax = fmaxf(lmax, sv);
        }
    };
#pragma unroll
    for (int i = 0; i < VPF; ++i) score_row(wave * RW + rw + i * NW * RW, kreg[i]);
    for (int tb = VPF * NW * RW; tb <= pos; tb += NW * RW) {  // beyond the prefetched rows (wave-uniform trip count)
        const int t = tb + wave * RW + rw;
        const u32x4 w = *reinterpret_cast<const u32x4*>(kc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
        score_row(t, w);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, d));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    stamp_a(2);
    float mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float lsum = 0.0f;
    for (int t = tid; t <= pos; t += NT) {
        const float e = expf(sc[t] - mx);
        sc[t] = e;
        lsum += e;
    }
    lsum = wave_sum_f(lsum);
    if (lane == 0) red[NW + wave] = lsum;
    __syncthreads();
    stamp_a(3);
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    const float inv = 1.0f / tot;

    // output: 16-byte slices of V rows; lane = (row-in-wave rw, 8-dim slice ds)
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    auto accum = [&](const int t, const u32x4 w) {
        const float pr = bits_to_float(float_to_bits<BF16>(sc[t] * inv), BF16);
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += pr * vn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] += pr * bits_to_float(w[j] & 0xFFFFu, BF16);
                o[2 * j + 1] += pr * bits_to_float(w[j] >> 16, BF16);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        if (t <= pos) accum(t, vreg[i]);
    }
#pragma unroll 4
    for (int t = wave * RW + rw + VPF * NW * RW; t <= pos; t += NW * RW) {
        const u32x4 w = (t < pos) ? *reinterpret_cast<const u32x4*>(vc + (size_t)t * hd + ds * 8) : (u32x4){0u, 0u, 0u, 0u};
        accum(t, w);
    }
    for (int off = SL; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], off);
    }
    if (lane < SL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave * hd + lane * 8 + j] = o[j];
    }
    __syncthreads();
    stamp_a(4);
    if (tid < hd) {  // whole waves (hd = 64 or 128)
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[w * hd + tid];
        const uint16_t yb = float_to_bits<BF16>(acc);
        y[(size_t)h * hd + tid] = yb;
        if (mask_out) {  // keep masks of y for the wo projection (TEAL_IN_MASKED consumer)
            const float yv = bits_to_float(yb, BF16);
            const unsigned long long mk = __ballot(keep_rule(yv, mask_tau) || (yv != yv));
            if (lane == 0) mask_out[((size_t)h * hd + tid) >> 6] = mk;
        }
    }
    stamp_a(5);
}

// Merge of one head's split-KV partials {m_s, l_s, o_s[hd]} (s < nsplit <= 64), one thread per output column t < hd
// (whole waves): lane s of every wave holds {m_s, l_s}, so the scale factors cost one load round trip; the o columns are
// then summed in split order, 32 independent loads at a time.  Same arithmetic, in the same order, as the merge producer
// of the GEMV launch.
template <bool BF16>
__device__ __forceinline__ void merge_head(const float* __restrict__ p, uint16_t* __restrict__ y_head,
                                           unsigned long long* __restrict__ mask_head, const float mask_tau, const int hd,
                                           const int nsplit, const int t, const int lane) {
    auto ld = [&](const float* q) -> float { return *q; };
    float m = -INFINITY, l = 0.0f;
    if (lane < nsplit) {
        m = ld(p + (size_t)lane * (hd + 2));
        l = ld(p + (size_t)lane * (hd + 2) + 1);
    }
    const float M = wave_max_f(m);
    const float f = l > 0.0f ? expf(m - M) : 0.0f;  // empty splits (l == 0, m == -inf) contribute nothing
    const float lf = l * f;
    float L = 0.0f, O = 0.0f;
    constexpr int MB = 32;  // loads per round trip
    for (int s0 = 0; s0 < nsplit; s0 += MB) {
        float v[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) v[u] = ld(p + (size_t)min(s0 + u, nsplit - 1) * (hd + 2) + 2 + t);
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            if (s0 + u < nsplit) {  // uniform
                const float fs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), s0 + u));
                const float ls = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lf), s0 + u));
                if (ls > 0.0f) {
                    L += ls;
                    O = fmaf(v[u], fs, O);  // explicit (not left to -ffp-contract): the merge launch must round like this everywhere
                }
            }
        }
    }
    const uint16_t yb = float_to_bits<BF16>(O / L);
    y_head[t] = yb;
    if (mask_head) {
        const float yv = bits_to_float(yb, BF16);
        const unsigned long long mk = __ballot(keep_rule(yv, mask_tau) || (yv != yv));
        if (lane == 0) mask_head[t >> 6] = mk;
    }
}

// ------------------------------------------------------------------------------------------------
// Split-KV decode attention (flash-decoding): the cached positions of a head are dealt to `nsplit` workgroups in
// groups of STEP = (waves x rows per wave) rows, round-robin (group g belongs to workgroup g mod nsplit), so that
//   * the work is balanced at every position without knowing it, and
//   * the K/V rows a workgroup will read do NOT depend on *pos: the first PF row groups of K and V, the q/k/v
//     projection (rounded, or the fp32 split-K slabs of the projection launch) and *pos itself are all requested at
//     the first instruction — one memory round trip before RoPE instead of three dependent ones (pos -> addresses ->
//     rows).  The pointer arguments are scalar kernel parameters preloaded into SGPRs (teal_amd/_lib.py: PRELOAD).
// Each workgroup writes an un-normalised partial {running max m, sum l, o[hd]}; the wo projection's merge producer
// (or the merge kernel) rescales and sums them and rounds once.
// ------------------------------------------------------------------------------------------------
template <bool BF16, int HD, int NT, bool ROPED = false>
__global__ __launch_bounds__(NT) void decode_attention_split_kernel(
    // the first 12 dwords are preloaded into SGPRs (teal_amd/_lib.py: PRELOAD): everything the FIRST loads need — the K / V
    // prefetch, q and *pos — with no scalar load and no integer division in front of them (round 4: the kernel used to open
    // with s_load n_head / n_kv / max_seq / nsplit, a wait, and three divisions: blockIdx / nsplit, n_head / n_kv, h / rep;
    // now the grid is (split, query head inside its group, KV head): no division for any ratio)
    const int* __restrict__ pos_ptr, uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
    const uint16_t* __restrict__ qkv, const int max_seq, const int nsplit, const int rep, const int qkv_nslabs,
    const float* __restrict__ qkv_slabs, float* __restrict__ partials, const uint16_t* __restrict__ rope, const int n_head,
    const int n_kv, const float scale, unsigned long long* __restrict__ phase) {
    constexpr int NW = NT / 64, hd = HD, SL = HD / 8, RW = 64 / SL;
    // PF row groups of K and of V leave before *pos is known; a workgroup with more than PF groups in range (cache positions
    // beyond PF x STEP x nsplit) keeps RD groups of each in flight from the moment it knows (rolling refill, see below)
    constexpr int PF = 4, RD = 8, STEP = NW * RW;
    const unsigned long long t_entry = wall_clock64();
    const int sp = blockIdx.x, kvh = blockIdx.z, h = kvh * rep + blockIdx.y;  // grid (nsplit, rep, n_kv): no division
    const int wg = h * nsplit + sp;            // linear index: partials, phase stamps
    auto stamp_p = [&](const int i) { if (phase && threadIdx.x == 0) phase[(size_t)wg * kPhaseRowA + i] = wall_clock64(); };
    extern __shared__ __align__(16) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    float* kn = qs + hd;
    float* vn = kn + hd;
    float* red = vn + hd;            // [2 * NW]
    float* part = red + 2 * NW;      // [NW][hd]
    float* sc = part + NW * hd;      // [local steps][STEP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto first_of_group = [&]() { return blockIdx.y == 0; };  // the query head that appends the group's k / v row
    const int dim = n_head * hd, kvs = n_kv * hd;
    uint16_t* kc = k_cache + (size_t)kvh * max_seq * hd;
    uint16_t* vc = v_cache + (size_t)kvh * max_seq * hd;
    // lanes = (row rw, 16-byte slice ds): a wave load covers RW whole cache rows (coalesced)
    const int ds = lane % SL, rw = lane / SL;
    const int rbase = wave * RW + rw;                  // row inside a step
    auto row_of = [&](const int i) { return (sp + i * nsplit) * STEP + rbase; };  // local step i -> cache row
    u32x4 kreg[RD], vreg[RD];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const size_t off = (size_t)min(row_of(i), max_seq - 1) * hd + ds * 8;
        kreg[i] = *reinterpret_cast<const u32x4*>(kc + off);
        vreg[i] = *reinterpret_cast<const u32x4*>(vc + off);
        // group by group: the first two loads leave after the first group's address arithmetic, not after all four groups'
        __builtin_amdgcn_sched_barrier(0);
    }
    // q, k, v of this head: the rounded projection, or its fp32 split-K slabs, interleaved [col][(nslabs + 3) & ~3],
    // summed in slice order and rounded once here (what the ordered reduce launch would have written)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int sstride = (qkv_nslabs + 3) & ~3;
    auto raw_at = [&](const int col, f32x4& a, f32x4& b, uint16_t& r) {
        if (qkv_slabs) {
            const float* p = qkv_slabs + (size_t)col * sstride;
            a = *reinterpret_cast<const f32x4*>(p);
            b = qkv_nslabs > 4 ? *reinterpret_cast<const f32x4*>(p + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
            r = qkv[col];
        }
    };
    auto fin_at = [&](const f32x4& a, const f32x4& b, const uint16_t r) -> float {
        if (!qkv_slabs) return bits_to_float(r, BF16);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (j < qkv_nslabs) ? a[j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (4 + j < qkv_nslabs) ? b[j] : 0.0f;
        return bits_to_float(float_to_bits<BF16>(s), BF16);
    };
    f32x4 la[4], lb[4];
    uint16_t lr[4] = {0, 0, 0, 0};
    // ROPED (after a TEAL_OUT_QKV_ROPE projection): qkv = the rotated, rounded query; the token's k / v rows are in the caches
    u32x4 qraw = {0u, 0u, 0u, 0u};
    if constexpr (ROPED) {
        qraw = *reinterpret_cast<const u32x4*>(qkv + (size_t)h * hd + ds * 8);
        // nothing below may be scheduled above these loads: the compiler otherwise hoists address arithmetic on arguments that
        // are NOT preloaded (partials, phase, n_head ...) — and with it an s_waitcnt for their scalar loads — in front of them
        __builtin_amdgcn_sched_barrier(0);
    }
    const bool rot = !ROPED && tid < hd / 2, vld = !ROPED && tid >= 128 && tid < 128 + hd;
    if (rot) {
        raw_at(h * hd + 2 * tid, la[0], lb[0], lr[0]);
        raw_at(h * hd + 2 * tid + 1, la[1], lb[1], lr[1]);
        raw_at(dim + kvh * hd + 2 * tid, la[2], lb[2], lr[2]);
        raw_at(dim + kvh * hd + 2 * tid + 1, la[3], lb[3], lr[3]);
    } else if (vld) {
        raw_at(dim + kvs + kvh * hd + (tid - 128), la[0], lb[0], lr[0]);
    }
    const int pos = min(max(pos_ptr[0], 0), max_seq - 1), n = pos + 1;  // clamped: never writes past the cache
    float* out = partials + (size_t)wg * (hd + 2);
    if (phase && threadIdx.x == 0) { phase[(size_t)wg * kPhaseRowA] = t_entry; phase[(size_t)wg * kPhaseRowA + 13] = ((unsigned long long)NW << 32) | (gridDim.x * gridDim.y * gridDim.z); }
    stamp_p(1);
    if (sp * STEP >= n) {  // no row group of this workgroup is in range yet (short sequence, many splits)
        if (tid < hd) out[2 + tid] = 0.0f;
        if (tid == 0) { out[0] = -INFINITY; out[1] = 0.0f; }
        return;
    }
    const int nsteps = ((n + STEP - 1) / STEP - sp + nsplit - 1) / nsplit;  // local steps with a row in range
    // More row groups than the blind prefetch covers (workgroup-uniform): groups PF .. RD-1 of K are requested ahead of the
    // score loop, V's right behind it, and both loops refill a register the moment they have consumed it — RD loads of the
    // stream in flight per lane until the rows run out.  (Until round 4 the groups past PF were loaded one at a time,
    // `global_load; s_waitcnt vmcnt(0)` per group in the ISA: a memory round trip per row group; same box, 7B heads, split +
    // merge launch: 4096 positions x 8 splits 21.1 -> 19.3 us, x 4 splits 30.2 -> 21.6 us, 512 positions x 4 11.5 -> 9.9 us,
    // profiles/r04_attention_context_sweep.txt.)  Rows past the end are clamped to row n - 1: every lane then reads the same
    // line, no HBM traffic, and the loads stay unconditional so that the compiler's vmcnt bookkeeping stays exact
    // (`s_waitcnt vmcnt(7)` in the steady state).  Measured and not kept: K's and V's second halves requested together
    // (K's registers are still live: 20.1 us at 4096 x 8); both requested unconditionally where *pos arrives in the 16-wave
    // instantiations (21.0 us, and 546 against 554 tok/s at ~1000 positions where the 8 extra loads buy nothing).
    const bool rolling = nsteps > PF;
    auto k_roll = [&](const int i) { return *reinterpret_cast<const u32x4*>(kc + (size_t)min(row_of(i), n - 1) * hd + ds * 8); };
    auto v_roll = [&](const int i) { return *reinterpret_cast<const u32x4*>(vc + (size_t)min(row_of(i), n - 1) * hd + ds * 8); };
    const bool has_new = !ROPED && ((pos / STEP) % nsplit) == sp;  // this workgroup's rows include the token being decoded
    if (rot) {
        const float c = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2], BF16);
        const float sn = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2 + 1], BF16);
        const float q0 = fin_at(la[0], lb[0], lr[0]), q1 = fin_at(la[1], lb[1], lr[1]);
        qs[2 * tid] = bits_to_float(float_to_bits<BF16>(rope_even(q0, q1, c, sn)), BF16);
        qs[2 * tid + 1] = bits_to_float(float_to_bits<BF16>(rope_odd(q0, q1, c, sn)), BF16);
        if (has_new) {
            const float k0 = fin_at(la[2], lb[2], lr[2]), k1 = fin_at(la[3], lb[3], lr[3]);
            const uint16_t ka = float_to_bits<BF16>(rope_even(k0, k1, c, sn)), kb = float_to_bits<BF16>(rope_odd(k0, k1, c, sn));
            kn[2 * tid] = bits_to_float(ka, BF16);
            kn[2 * tid + 1] = bits_to_float(kb, BF16);
            if (first_of_group()) {
                kc[(size_t)pos * hd + 2 * tid] = ka;
                kc[(size_t)pos * hd + 2 * tid + 1] = kb;
            }
        }
    } else if (has_new && vld) {
        const int d = tid - 128;
        const float vf = fin_at(la[0], lb[0], lr[0]);
        vn[d] = vf;
        if (first_of_group()) vc[(size_t)pos * hd + d] = float_to_bits<BF16>(vf);
    }
    if constexpr (!ROPED) __syncthreads();
    stamp_p(2);
    float qv[8];
    if constexpr (ROPED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qv[2 * j] = bits_to_float(qraw[j] & 0xFFFFu, BF16);
            qv[2 * j + 1] = bits_to_float(qraw[j] >> 16, BF16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = qs[ds * 8 + j];
    }
    float lmax = -INFINITY;
    auto score_row = [&](const int i, const u32x4 w) {
        const int t = row_of(i);
        float a = 0.0f;
        if (!ROPED && t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a += qv[j] * kn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += qv[2 * j] * bits_to_float(w[j] & 0xFFFFu, BF16);
                a += qv[2 * j + 1] * bits_to_float(w[j] >> 16, BF16);
            }
        }
        a = row_slices_sum<SL>(a);
        const float sv = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        if (t < n) {
            if (ds == 0) sc[i * STEP + rbase] = sv;
            lmax = fmaxf(lmax, sv);
        }
    };
    if (!rolling) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < nsteps) score_row(i, kreg[i]);  // workgroup-uniform guard
    } else {
        // (issued inside this branch, not where *pos arrives: after a conditional issue the compiler's merged wait counts
        // make every later wait of the shared code drain these loads too)
#pragma unroll
        for (int i = PF; i < RD; ++i) kreg[i] = k_roll(i);
        for (int i0 = 0; i0 < nsteps; i0 += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                if (i0 + u < nsteps) score_row(i0 + u, kreg[u]);
                kreg[u] = k_roll(i0 + u + RD);  // RD groups ahead, into the register just consumed
            }
        }
        // V groups PF .. RD-1: in flight across the softmax (K's registers are free now; asking for them together with K's
        // costs 32 more VGPRs through the score loop: spills at 16 waves per workgroup)
#pragma unroll
        for (int i = PF; i < RD; ++i) vreg[i] = v_roll(i);
    }
    lmax = wave_max_f(lmax);  // row_shr DPP: lanes without a predecessor keep their own value (old = src)
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    stamp_p(3);
    float mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float lsum = 0.0f;
    for (int e = tid; e < nsteps * STEP; e += NT) {
        const int t = (sp + (e / STEP) * nsplit) * STEP + (e % STEP);
        if (t < n) {
            const float ex = expf(sc[e] - mx);
            sc[e] = ex;
            lsum += ex;
        }
    }
    lsum = wave_sum_f(lsum);
    if (lane == 0) red[NW + wave] = lsum;
    __syncthreads();
    stamp_p(4);
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    auto pv_row = [&](const int i, const u32x4 w) {
        const int t = row_of(i);
        if (t >= n) return;
        const float pr = sc[i * STEP + rbase];
        if (!ROPED && t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += pr * vn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] += pr * bits_to_float(w[j] & 0xFFFFu, BF16);
                o[2 * j + 1] += pr * bits_to_float(w[j] >> 16, BF16);
            }
        }
    };
    if (!rolling) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < nsteps) pv_row(i, vreg[i]);
    } else {
        for (int i0 = 0; i0 < nsteps; i0 += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                if (i0 + u < nsteps) pv_row(i0 + u, vreg[u]);
                vreg[u] = v_roll(i0 + u + RD);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (SL <= 8) o[j] = xor_add<8>(o[j]);
        o[j] = xor_add<32>(xor_add<16>(o[j]));
    }
    if (lane < SL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave * hd + lane * 8 + j] = o[j];
    }
    __syncthreads();
    stamp_p(5);
    if (tid < hd) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[w * hd + tid];
        out[2 + tid] = acc;
    }
    if (tid == 0) { out[0] = mx; out[1] = tot; }
    stamp_p(7);
}

// ------------------------------------------------------------------------------------------------
// Grouped-query split-KV decode attention: one workgroup = one KV head x one share of its cached positions, for ALL
// REP = n_head / n_kv query heads that read that KV head.  The per-query-head kernel above reads every K/V row REP
// times (from L2 / the Infinity Cache when the workgroups of a group land on different XCDs): at 16 k positions the
// Llama-2-70B cache (67 MB per layer) took 83 us = 0.8 TB/s of algorithmic bytes (scripts/attention_context_sweep.py).
// Here each cache byte is loaded once and feeds REP heads.  That makes the launch instruction-bound on the vector ALU
// (REP x head_dim multiply-adds per row for Q.K and again for P.V, 4 clocks per wave instruction), so the inner loops
// are written for instruction count:
//   * Q.K: the 16 bytes a lane holds of a K row stay packed; q (already rounded to the cache dtype) is held as packed
//     pairs, one v_dot2c_f32_{f16,bf16} per pair and head (fp32 accumulate) — 4 per head instead of 8 FMAs + unpacking;
//   * the row sum over the SL lanes of a cache row is a REDUCE-SCATTER over the REP heads (halve the value set at
//     every butterfly level: 22 instead of 32 DPP adds for REP = 8) that leaves head (lane's bits) in each lane, so
//     that scaling, rounding, the running max and the LDS store run once per row for all heads, not once per head;
//   * P.V: probabilities of a row are read as one or two 16-byte LDS words ([row][REP] layout), V is unpacked once per
//     row and the REP x 8 accumulators advance as pairs (v_pk_fma_f32 until round 6; the library is built without packed fp32
//     since — profiles/r06_concurrent_packed_fp32.txt — at 2-3 % on this launch: profiles/r06_attention_context_sweep.txt).
// Same lane layout, row dealing (round-robin row groups, independent of *pos), rounding points and partial format
// {m, l, o[hd]} per (query head, split) as decode_attention_split_kernel; the loads run PF row groups ahead.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_lane(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// One butterfly level over `a[0..NV)`: the partner lane is the DPP pattern CTRL (an involution that flips lane bit BIT).
// NV > 1: the lane keeps the half of the values its bit selects and adds the partner's copy of that half; NV == 1: plain add.
template <int NV, int CTRL, int BIT>
__device__ __forceinline__ void fold_level(float* a, const int lane) {
    if constexpr (NV > 1) {
        const bool hi = (lane & BIT) != 0;
#pragma unroll
        for (int k = 0; k < NV / 2; ++k) {
            const float keep = hi ? a[NV / 2 + k] : a[k];
            const float send = hi ? a[k] : a[NV / 2 + k];
            a[k] = keep + dpp_lane<CTRL>(send);
        }
    } else {
        a[0] += dpp_lane<CTRL>(a[0]);
    }
}

// Sum a[0..REP) over the SL lanes of a cache row; returns the total of head gqa_lane_head<REP, SL>(lane).
template <int REP, int SL>
__device__ __forceinline__ float gqa_reduce_scatter(float* a, const int lane) {
    constexpr int N8 = REP, N4 = (SL == 16 && N8 > 1) ? N8 / 2 : N8, N2 = N4 > 1 ? N4 / 2 : N4, N1 = N2 > 1 ? N2 / 2 : N2;
    static_assert((N1 > 1 ? N1 / 2 : N1) == 1, "more heads than lanes per row");
    if constexpr (SL == 16) fold_level<N8, 0x128, 8>(a, lane);  // row_ror:8        lane ^ 8
    fold_level<N4, 0x141, 4>(a, lane);                         // row_half_mirror  lane ^ 7 (flips bit 2)
    fold_level<N2, 0x4E, 2>(a, lane);                          // quad_perm[2,3,0,1]  lane ^ 2
    fold_level<N1, 0xB1, 1>(a, lane);                          // quad_perm[1,0,3,2]  lane ^ 1
    return a[0];
}

template <int REP, int SL>
__device__ __forceinline__ int gqa_lane_head(const int lane) {
    constexpr int N8 = REP, N4 = (SL == 16 && N8 > 1) ? N8 / 2 : N8, N2 = N4 > 1 ? N4 / 2 : N4, N1 = N2 > 1 ? N2 / 2 : N2;
    int h = 0;
    if constexpr (SL == 16 && N8 > 1) h = (h << 1) | ((lane >> 3) & 1);
    if constexpr (N4 > 1) h = (h << 1) | ((lane >> 2) & 1);
    if constexpr (N2 > 1) h = (h << 1) | ((lane >> 1) & 1);
    if constexpr (N1 > 1) h = (h << 1) | (lane & 1);
    return h;
}

template <bool BF16>
__device__ __forceinline__ float dot2_acc(const uint32_t a, const uint32_t b, const float c) {
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
    if constexpr (BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b16x2, a), __builtin_bit_cast(b16x2, b), c, false);
    else return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, a), __builtin_bit_cast(h16x2, b), c, false);
}

template <bool BF16, int HD, int NT, int REP>
__global__ __launch_bounds__(NT) void decode_attention_gqa_kernel(
    const int* __restrict__ pos_ptr, const float* __restrict__ qkv_slabs, uint16_t* __restrict__ k_cache,
    uint16_t* __restrict__ v_cache, const uint16_t* __restrict__ qkv, float* __restrict__ partials,
    const uint16_t* __restrict__ rope, const int n_head, const int n_kv, const int max_seq, const int nsplit,
    const float scale, const int qkv_nslabs, const int lrows) {
    constexpr int NW = NT / 64, hd = HD, SL = HD / 8, RW = 64 / SL;
    constexpr int PF = 4, RD = 8, STEP = NW * RW;  // blind prefetch depth; depth of the stream once *pos is known
    constexpr int NPAIR = (REP + 2) * (HD / 2), NITEM = (NPAIR + NT - 1) / NT;  // q pairs of REP heads, k pairs, v pairs
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    static_assert(NT % REP == 0 && (REP == 4 || REP == 8), "softmax pass assigns head tid % REP to a thread");
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* qsb = reinterpret_cast<uint32_t*>(smem);  // [REP][hd / 2] rotated, rounded q (packed pairs)
    uint32_t* knb = qsb + REP * (hd / 2);               // [hd / 2] the token's own rotated k
    uint32_t* vnb = knb + hd / 2;                       // [hd / 2] ... and v
    float* red = reinterpret_cast<float*>(vnb + hd / 2);  // [2][REP][NW]: running max, sum
    float* sc = red + 2 * REP * NW;                     // [lrows][REP] scores -> probabilities; later [NW][REP * hd] partial o
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kvh = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int dim = n_head * hd, kvs = n_kv * hd;
    uint16_t* kc = k_cache + (size_t)kvh * max_seq * hd;
    uint16_t* vc = v_cache + (size_t)kvh * max_seq * hd;
    const int ds = lane % SL, rw = lane / SL;
    const int rbase = wave * RW + rw;
    auto row_of = [&](const int i) { return (sp + i * nsplit) * STEP + rbase; };
    auto k_at = [&](const int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + (size_t)min(row_of(i), max_seq - 1) * hd + ds * 8)); };
    auto v_at = [&](const int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + (size_t)min(row_of(i), max_seq - 1) * hd + ds * 8)); };
    u32x4 kb[RD], vb[RD];  // the first PF row groups of K and of V leave at the first instruction
#pragma unroll
    for (int i = 0; i < PF; ++i) kb[i] = k_at(i);
#pragma unroll
    for (int i = 0; i < PF; ++i) vb[i] = v_at(i);
    // the projection's columns of this group: pairs (2c, 2c + 1); q of heads kvh * REP + r, then k, then v
    const int sstride = (qkv_nslabs + 3) & ~3;
    auto pair_col = [&](const int it) -> int {
        if (it < REP * (hd / 2)) return (kvh * REP) * hd + 2 * it;
        if (it < (REP + 1) * (hd / 2)) return dim + kvh * hd + 2 * (it - REP * (hd / 2));
        return dim + kvs + kvh * hd + 2 * (it - (REP + 1) * (hd / 2));
    };
    f32x4 la[NITEM][2], lb[NITEM][2];
    uint32_t lr[NITEM];
#pragma unroll
    for (int u = 0; u < NITEM; ++u) {
        const int it = min(u * NT + tid, NPAIR - 1), col = pair_col(it);
        if (qkv_slabs) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float* p = qkv_slabs + (size_t)(col + e) * sstride;
                la[u][e] = *reinterpret_cast<const f32x4*>(p);
                lb[u][e] = qkv_nslabs > 4 ? *reinterpret_cast<const f32x4*>(p + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
            lr[u] = *reinterpret_cast<const uint32_t*>(qkv + col);
        }
    }
    auto fin = [&](const int u, const int e) -> float {
        if (!qkv_slabs) return bits_to_float((lr[u] >> (16 * e)) & 0xFFFFu, BF16);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (j < qkv_nslabs) ? la[u][e][j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (4 + j < qkv_nslabs) ? lb[u][e][j] : 0.0f;
        return bits_to_float(float_to_bits<BF16>(s), BF16);
    };
    const int pos = min(max(pos_ptr[0], 0), max_seq - 1), n = pos + 1;
    auto out_of = [&](const int r) { return partials + ((size_t)(kvh * REP + r) * nsplit + sp) * (hd + 2); };
    if (sp * STEP >= n) {  // no row group of this workgroup is in range yet
        for (int c = tid; c < REP * hd; c += NT) out_of(c / hd)[2 + (c % hd)] = 0.0f;
        if (tid < REP) { out_of(tid)[0] = -INFINITY; out_of(tid)[1] = 0.0f; }
        return;
    }
    const int nsteps = ((n + STEP - 1) / STEP - sp + nsplit - 1) / nsplit;
    const bool has_new = ((pos / STEP) % nsplit) == sp;
    auto k_roll = [&](const int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + (size_t)min(row_of(i), n - 1) * hd + ds * 8)); };
    auto v_roll = [&](const int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + (size_t)min(row_of(i), n - 1) * hd + ds * 8)); };
#pragma unroll
    for (int u = 0; u < NITEM; ++u) {
        const int it = u * NT + tid;
        if (it >= NPAIR) continue;
        const float a0 = fin(u, 0), a1 = fin(u, 1);
        if (it < (REP + 1) * (hd / 2)) {  // q or k pair: rotate by the position's angle, round
            const int pr = it % (hd / 2);
            const float c = bits_to_float(rope[((size_t)pos * (hd / 2) + pr) * 2], BF16);
            const float sn = bits_to_float(rope[((size_t)pos * (hd / 2) + pr) * 2 + 1], BF16);
            const uint32_t rr = (uint32_t)float_to_bits<BF16>(rope_even(a0, a1, c, sn)) | ((uint32_t)float_to_bits<BF16>(rope_odd(a0, a1, c, sn)) << 16);
            if (it < REP * (hd / 2)) {
                qsb[it] = rr;
            } else if (has_new) {
                knb[pr] = rr;
                *reinterpret_cast<uint32_t*>(kc + (size_t)pos * hd + 2 * pr) = rr;
            }
        } else if (has_new) {
            const int pr = it - (REP + 1) * (hd / 2);
            const uint32_t rr = (uint32_t)float_to_bits<BF16>(a0) | ((uint32_t)float_to_bits<BF16>(a1) << 16);
            vnb[pr] = rr;
            *reinterpret_cast<uint32_t*>(vc + (size_t)pos * hd + 2 * pr) = rr;
        }
    }
    __syncthreads();
    const int myhead = gqa_lane_head<REP, SL>(lane);
    float lmax = -INFINITY;  // of head `myhead`, over this lane's rows
    {
        u32x4 qp[REP];
#pragma unroll
        for (int r = 0; r < REP; ++r) qp[r] = *reinterpret_cast<const u32x4*>(qsb + r * (hd / 2) + ds * 4);
        const u32x4 knw = *reinterpret_cast<const u32x4*>(knb + ds * 4);
        auto k_step = [&](const int i, u32x4 w) {
            const int t = row_of(i);
            if (t == pos) w = knw;
            float a[REP];
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = dot2_acc<BF16>(qp[r][j], w[j], acc);
                a[r] = acc;
            }
            const float tot = gqa_reduce_scatter<REP, SL>(a, lane);
            const float sv = bits_to_float(float_to_bits<BF16>(tot * scale), BF16);
            if (t < n) {
                sc[(i * STEP + rbase) * REP + myhead] = sv;  // the lanes that share a head write the same value
                lmax = fmaxf(lmax, sv);
            }
        };
        // Past the blind prefetch (more than PF row groups in range, workgroup-uniform): RD groups of the stream in flight per
        // lane, a register refilled when consumed, rows past the position clamped to row n - 1 (one line for every lane: no HBM
        // traffic — the blind form's clamp, max_seq - 1, made every refill past the position fetch real rows: up to
        // PF x STEP x nsplit rows per phase whenever the cache is longer than the sequence).  Same structure as
        // decode_attention_split_kernel (round 4); inside the prefetch no refill is requested at all.
        if (nsteps <= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < nsteps) k_step(u, kb[u]);  // workgroup-uniform guard
        } else {
#pragma unroll
            for (int u = PF; u < RD; ++u) kb[u] = k_roll(u);
            for (int i0 = 0; i0 < nsteps; i0 += RD) {
#pragma unroll
                for (int u = 0; u < RD; ++u) {
                    if (i0 + u < nsteps) k_step(i0 + u, kb[u]);
                    kb[u] = k_roll(i0 + u + RD);
                }
            }
#pragma unroll
            for (int u = PF; u < RD; ++u) vb[u] = v_roll(u);  // in flight across the softmax
        }
    }
    // max per head: lanes with the same `myhead` differ in the row group bits (lane / SL) and in the duplicated low bits
#pragma unroll
    for (int off = 32; off >= SL; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane < SL) red[myhead * NW + wave] = lmax;  // duplicates write the same value
    __syncthreads();
    {
        const int r = tid % REP;  // NT % REP == 0: a thread only ever sees entries of head r
        float m = red[r * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[r * NW + w]);
        float lsum = 0.0f;
        for (int e = tid; e < nsteps * STEP * REP; e += NT) {
            const int le = e / REP;
            const int t = (sp + (le / STEP) * nsplit) * STEP + (le % STEP);
            if (t < n) {
                const float ex = expf(sc[e] - m);
                sc[e] = ex;
                lsum += ex;
            }
        }
#pragma unroll
        for (int off = REP; off < 64; off <<= 1) lsum += __shfl_xor(lsum, off);
        if (lane < REP) red[(REP + lane) * NW + wave] = lsum;
    }
    __syncthreads();
    f32x2 o[REP][4];
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[r][j] = (f32x2){0.0f, 0.0f};
    const u32x4 vnw = *reinterpret_cast<const u32x4*>(vnb + ds * 4);
    auto v_step = [&](const int i, u32x4 w) {
        const int t = row_of(i);
        if (t < n) {
            if (t == pos) w = vnw;
            f32x2 vf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vf[j] = (f32x2){bits_to_float(w[j] & 0xFFFFu, BF16), bits_to_float(w[j] >> 16, BF16)};
            float pr[REP];
            const f32x4* pp = reinterpret_cast<const f32x4*>(sc + (size_t)(i * STEP + rbase) * REP);
#pragma unroll
            for (int q4 = 0; q4 < REP / 4; ++q4) {
                const f32x4 pv = pp[q4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pr[q4 * 4 + e] = pv[e];
            }
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                const f32x2 p2 = (f32x2){pr[r], pr[r]};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[r][j] += p2 * vf[j];
            }
        }
    };
    if (nsteps <= PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < nsteps) v_step(u, vb[u]);
    } else {
        for (int i0 = 0; i0 < nsteps; i0 += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                if (i0 + u < nsteps) v_step(i0 + u, vb[u]);
                vb[u] = v_roll(i0 + u + RD);
            }
        }
    }
    // sum over the row groups of the wave (lanes with the same 16-byte slice)
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = o[r][j][e];
                if constexpr (SL <= 8) v = xor_add<8>(v);
                v = xor_add<32>(xor_add<16>(v));
                o[r][j][e] = v;
            }
    __syncthreads();  // every wave is done reading probabilities: the region becomes the per-wave partial o
    float* part = sc;
    if (lane < SL) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x2*>(part + (wave * REP + r) * hd + lane * 8 + 2 * j) = o[r][j];
    }
    __syncthreads();
    for (int c = tid; c < REP * hd; c += NT) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[w * REP * hd + c];
        out_of(c / hd)[2 + (c % hd)] = acc;
    }
    if (tid < REP) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[(REP + tid) * NW + w];
        float m = red[tid * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[tid * NW + w]);
        out_of(tid)[0] = m;
        out_of(tid)[1] = tot;
    }
}

// Merge launch (split counts the wo launch's merge producer does not take): one workgroup per head.  Lane s of every
// wave holds {m_s, l_s} of split s (nsplit <= 64), so the scale factors cost one load round trip; the o columns are then
// summed in split order, 32 independent loads at a time (the dependent-load loop this replaces took 9-16 us at
// 16-32 splits).  Same arithmetic, in the same order, as the merge producer of the GEMV launch.
template <bool BF16>
__global__ __launch_bounds__(128) void decode_attention_merge_kernel(const float* __restrict__ partials,
                                                                     uint16_t* __restrict__ y,
                                                                     unsigned long long* __restrict__ mask_out,
                                                                     const float mask_tau, const int hd, const int nsplit) {
    const int h = blockIdx.x, tid = threadIdx.x;
    if (tid >= hd) return;  // hd = 64 or 128: whole waves
    merge_head<BF16>(partials + (size_t)h * nsplit * (hd + 2), y + (size_t)h * hd,
                            mask_out ? mask_out + ((size_t)h * hd >> 6) : nullptr, mask_tau, hd, nsplit, tid, tid & 63);
}

// ------------------------------------------------------------------------------------------------
// Fused sampler (gpt-fast/generate.py:49-66): logits / T -> keep the top-k -> softmax -> exponential-
// race multinomial (argmax p_i / q_i, q_i ~ Exp(1)), no host sync.  One workgroup; the k-th largest
// logit is found EXACTLY by a two-pass radix select on the 16-bit keys (ties at the pivot are all
// kept, as `logits < pivot -> -inf` does).  Randomness: counter-based hash of (seed, draw counter,
// index); the draw counter lives on the device and is bumped by the kernel, so hipGraph replays
// draw fresh numbers.  Token streams are not pinned by the reference (they depend on torch's RNG).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key16(uint32_t b, bool bf16) {
    (void)bf16;  // fp16 and bf16 share sign-magnitude ordering
    return (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u);
}

__device__ __forceinline__ uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// sel[0] = the bin b in which the `need`-th largest key falls, sel[1] = its rank inside that bin.  One wave does
// it without workgroup barriers (lane l owns bins 4l..4l+3, suffix sums over lanes by shuffles): the Hillis-Steele
// scan over 1024 threads it replaces cost 18 barriers per call, and barriers were most of this kernel's time.
// All threads must call this; hist[] must be complete (barrier before), sel[] is valid after the trailing barrier.
__device__ __forceinline__ void select_bin(const unsigned int* hist, unsigned int* suf, unsigned int* sel,
                                           const unsigned int need, const int tid) {
    (void)suf;
    if (tid < 64) {
        const unsigned int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const unsigned int own = h0 + h1 + h2 + h3;
        unsigned int incl = own;  // keys in this lane's bins and all higher lanes'
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int t = (unsigned int)__shfl_down((int)incl, d);
            if (tid + d < 64) incl += t;
        }
        // keys in strictly higher bins, for each of the four bins (highest first)
        const unsigned int a3 = incl - own, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1;
        if (a3 + h3 >= need && a3 < need) { sel[0] = 4u * tid + 3u; sel[1] = need - a3; }
        if (a2 + h2 >= need && a2 < need) { sel[0] = 4u * tid + 2u; sel[1] = need - a2; }
        if (a1 + h1 >= need && a1 < need) { sel[0] = 4u * tid + 1u; sel[1] = need - a1; }
        if (a0 + h0 >= need && a0 < need) { sel[0] = 4u * tid; sel[1] = need - a0; }
        if (tid == 0 && incl < need) { sel[0] = 0u; sel[1] = need; }  // fewer keys than requested: keep all
    }
    __syncthreads();
}

template <bool BF16>
__device__ __forceinline__ void sample_full(const uint16_t* __restrict__ logits, const int V, const int top_k,
                                            const float inv_temp, unsigned long long* __restrict__ rng_state,
                                            int* __restrict__ token_out, int* __restrict__ pos_inout,
                                            int* __restrict__ history, const int history_len) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int whist[16][256];  // per-wave sub-histograms: logits cluster in a few bins, a single
                                             // shared histogram serialises on LDS atomics
    __shared__ float fred[16];
    __shared__ int ired[16];
    __shared__ unsigned int sel[2];
    __shared__ unsigned int suf[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool filter = top_k > 0 && top_k < V;
    const int V8 = V >> 3;  // 16-byte vectors (vocab sizes are multiples of 8; the tail is handled scalar)
    const u32x4* lv = reinterpret_cast<const u32x4*>(logits);
    uint32_t pivot_key = 0;  // keep keys >= pivot_key
    float mx = -INFINITY;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
    __syncthreads();
    // pass 1: high-byte histogram of the order-preserving 16-bit keys + global max
    for (int i = tid; i < V8; i += 1024) {
        const u32x4 w = lv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = w[j] & 0xFFFFu, hi = w[j] >> 16;
            mx = fmaxf(mx, fmaxf(bits_to_float(lo, BF16), bits_to_float(hi, BF16)));
            if (filter) {
                atomicAdd(&whist[wave][order_key16(lo, BF16) >> 8], 1u);
                atomicAdd(&whist[wave][order_key16(hi, BF16) >> 8], 1u);
            }
        }
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) {
        mx = fmaxf(mx, bits_to_float(logits[i], BF16));
        if (filter) atomicAdd(&whist[wave][order_key16(logits[i], BF16) >> 8], 1u);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0) fred[wave] = mx;
    __syncthreads();
    if (tid < 256) {
        unsigned int a = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) a += whist[w][tid];
        hist[tid] = a;
    }
    __syncthreads();
    mx = fred[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, fred[w]);
    if (filter) {
        // suffix counts over the 256 bins (parallel scan), then the bin holding the top_k-th key
        select_bin(hist, suf, sel, (unsigned int)top_k, tid);
        __syncthreads();
        const unsigned int hb = sel[0], need2 = sel[1];
        __syncthreads();
        for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
        __syncthreads();
        // pass 2: low-byte histogram inside the selected high-byte bin
        for (int i = tid; i < V8; i += 1024) {
            const u32x4 w = lv[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k0 = order_key16(w[j] & 0xFFFFu, BF16), k1 = order_key16(w[j] >> 16, BF16);
                if ((k0 >> 8) == hb) atomicAdd(&whist[wave][k0 & 0xFFu], 1u);
                if ((k1 >> 8) == hb) atomicAdd(&whist[wave][k1 & 0xFFu], 1u);
            }
        }
        for (int i = (V8 << 3) + tid; i < V; i += 1024) {
            const uint32_t k = order_key16(logits[i], BF16);
            if ((k >> 8) == hb) atomicAdd(&whist[wave][k & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            unsigned int a = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += whist[w][tid];
            hist[tid] = a;
        }
        __syncthreads();
        select_bin(hist, suf, sel, need2, tid);
        if (tid == 0) sel[0] = (hb << 8) | sel[0];
        __syncthreads();
        pivot_key = sel[0];
    }
    // exponential race: argmax_i exp((x_i - max)/T) / q_i  over the kept set (the softmax
    // normaliser is common to all i and cannot change the argmax)
    const uint32_t seed = (uint32_t)rng_state[0], ctr = (uint32_t)rng_state[1];
    float best = -1.0f;
    int besti = 0x7FFFFFFF;
    auto consider = [&](const uint32_t b, const int i) {
        if (filter && order_key16(b, BF16) < pivot_key) return;
        const float pnum = expf((bits_to_float(b, BF16) - mx) * inv_temp);
        const float u = ((float)(hash3(seed, ctr, (uint32_t)i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float scv = pnum / (-logf(u));
        if (scv > best || (scv == best && i < besti)) { best = scv; besti = i; }
    };
    for (int i = tid; i < V8; i += 1024) {
        const u32x4 w = lv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            consider(w[j] & 0xFFFFu, i * 8 + 2 * j);
            consider(w[j] >> 16, i * 8 + 2 * j + 1);
        }
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) consider(logits[i], i);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(besti, d);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if (lane == 0) { fred[wave] = best; ired[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < besti)) { best = fred[w]; besti = ired[w]; }
        token_out[0] = besti;
        const unsigned long long c = rng_state[1];
        if (history && (long long)c < (long long)history_len) history[c] = besti;
        rng_state[1] = c + 1ull;
        if (pos_inout) pos_inout[0] = pos_inout[0] + 1;
    }
}

template <bool BF16>
__global__ __launch_bounds__(1024) void sample_topk_kernel(const uint16_t* __restrict__ logits, const int V,
                                                            const int top_k, const float inv_temp,
                                                            unsigned long long* __restrict__ rng_state,
                                                            int* __restrict__ token_out, int* __restrict__ pos_inout,
                                                            int* __restrict__ history, const int history_len) {
    sample_full<BF16>(logits, V, top_k, inv_temp, rng_state, token_out, pos_inout, history, history_len);
}


// ------------------------------------------------------------------------------------------------
// Register-resident sampler for vocab % 8 == 0, vocab <= NV * 8192 (Llama-2: NV = 4, Llama-3: NV = 16).
// Phase stamps of the generic kernel above (scripts/sampler_phase.py): its time is the high-byte histogram
// pass — every key does an LDS atomic, and bf16 logits fall into 4-5 of the 256 high-byte bins (sign + 7
// exponent bits), so the atomics serialise: 45 of 72 us at vocab 128256 — plus dependent global loads in
// every pass and in the last thread's epilogue.  Here:
//   * every thread loads its NV vectors ONCE and keeps the order-preserving keys in registers;
//   * the k-th largest key is found in a WINDOW below the maximum: bin = (kmax - key) >> SH for the keys
//     within 256 << SH of kmax, everything further away does no atomic at all.  The top-k of a peaked
//     distribution sits within ~2 octaves of the maximum (SH = 0 for bf16, 3 for fp16), i.e. a few per cent of
//     the vocabulary, spread over 256 bins.  If the window holds fewer than k keys it is widened (SH += 3, up
//     to 8 where it covers every key) and the pass repeated; a bin wider than one key is resolved by a second
//     histogram of the low SH bits of its (few) members.  Exact: same pivot, ties kept, same tokens as the
//     generic kernel (tests/test_engine.py).
// ------------------------------------------------------------------------------------------------
template <bool BF16, int NV>
__global__ __launch_bounds__(1024) void sample_topk_window_kernel(const uint16_t* __restrict__ logits, const int V,
                                                                   const int top_k, const float inv_temp,
                                                                   unsigned long long* __restrict__ rng_state,
                                                                   int* __restrict__ token_out, int* __restrict__ pos_inout,
                                                                   int* __restrict__ history, const int history_len,
                                                                   unsigned long long* __restrict__ phase) {
    auto stamp_s = [&](const int i) { if (phase && threadIdx.x == 0) phase[i] = wall_clock64(); };
    stamp_s(0);
    __shared__ unsigned int hist[256];
    __shared__ unsigned int whist[16][256];
    __shared__ float fred[16];
    __shared__ int ired[16];
    __shared__ unsigned int sel[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool filter = top_k > 0 && top_k < V;
    const int V8 = V >> 3;
    const u32x4* lv = reinterpret_cast<const u32x4*>(logits);
    u32x4 kv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) kv[v] = lv[min(v * 1024 + tid, V8 - 1)];
    // everything the epilogue needs from memory is requested now, not by the last thread at the very end
    const unsigned long long seed64 = rng_state[0], ctr64 = rng_state[1];
    const int pos0 = pos_inout ? pos_inout[0] : 0;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
    uint32_t kmax = 0u;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool ok = v * 1024 + tid < V8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k0 = order_key16(kv[v][j] & 0xFFFFu, BF16), k1 = order_key16(kv[v][j] >> 16, BF16);
            kv[v][j] = ok ? (k0 | (k1 << 16)) : 0u;  // vectors past the vocabulary: key 0, never considered (index check)
            kmax = ok ? max(kmax, max(k0, k1)) : kmax;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d));
    if (lane == 0) ired[wave] = (int)kmax;
    __syncthreads();
    kmax = (uint32_t)ired[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) kmax = max(kmax, (uint32_t)ired[w]);
    auto key_bits = [](const uint32_t k) -> uint32_t { return (k & 0x8000u) ? (k ^ 0x8000u) : (~k & 0xFFFFu); };  // order_key16^-1
    const float mx = bits_to_float(key_bits(kmax), BF16);
    stamp_s(1);
    auto merge_hist = [&]() {  // whist[16][256] -> hist[256]; barriers on both sides
        __syncthreads();
        if (tid < 256) {
            unsigned int a = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += whist[w][tid];
            hist[tid] = a;
        }
        __syncthreads();
    };
    uint32_t pivot_key = 0u;  // keep keys >= pivot_key
    if (filter) {
        for (int sh = BF16 ? 0 : 3;; sh += 3) {
            if (sh > 8) sh = 8;  // 256 << 8 covers every key
            // window pass: bin 255 = kmax, bin 255 - d = keys (d << sh) .. below it
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v * 1024 + tid < V8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t d0 = (kmax - (kv[v][j] & 0xFFFFu)) >> sh, d1 = (kmax - (kv[v][j] >> 16)) >> sh;
                        if (d0 < 256u) atomicAdd(&whist[wave][255u - d0], 1u);
                        if (d1 < 256u) atomicAdd(&whist[wave][255u - d1], 1u);
                    }
                }
            }
            merge_hist();
            unsigned int inwin = 0;  // keys inside the window (workgroup-uniform)
            {
                unsigned int a = (tid < 256) ? hist[tid] : 0u;
                a = (unsigned int)wave_sum_f((float)a);  // <= 131072: exact in fp32
                if (lane == 0) fred[wave] = (float)a;
                __syncthreads();
                inwin = (unsigned int)(fred[0] + fred[1] + fred[2] + fred[3]);
            }
            if (inwin >= (unsigned int)top_k || sh == 8) {
                select_bin(hist, nullptr, sel, (unsigned int)top_k, tid);
                const unsigned int d = 255u - sel[0], need2 = sel[1];
                __syncthreads();
                if (sh == 0) {
                    pivot_key = kmax - d;
                } else {
                    // resolve the bin: histogram of the low `sh` bits of its members (bin 255 = largest key)
                    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
                    __syncthreads();
                    const uint32_t lowmask = (1u << sh) - 1u;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        if (v * 1024 + tid < V8) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t e0 = kmax - (kv[v][j] & 0xFFFFu), e1 = kmax - (kv[v][j] >> 16);
                                if ((e0 >> sh) == d) atomicAdd(&whist[wave][255u - (e0 & lowmask)], 1u);
                                if ((e1 >> sh) == d) atomicAdd(&whist[wave][255u - (e1 & lowmask)], 1u);
                            }
                        }
                    }
                    merge_hist();
                    select_bin(hist, nullptr, sel, need2, tid);
                    pivot_key = kmax - ((d << sh) | (255u - sel[0]));
                    __syncthreads();
                }
                break;
            }
            for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;  // widen the window and count again
            __syncthreads();
        }
    }
    stamp_s(2);
    // exponential race over the kept set (see sample_topk_kernel)
    const uint32_t seed = (uint32_t)seed64, ctr = (uint32_t)ctr64;
    float best = -1.0f;
    int besti = 0x7FFFFFFF;
    auto consider = [&](const uint32_t key, const int i) {
        if (key < pivot_key) return;
        const float pnum = expf((bits_to_float(key_bits(key), BF16) - mx) * inv_temp);
        const float u = ((float)(hash3(seed, ctr, (uint32_t)i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float scv = pnum / (-logf(u));
        if (scv > best || (scv == best && i < besti)) { best = scv; besti = i; }
    };
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int i = v * 1024 + tid;
        if (i < V8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                consider(kv[v][j] & 0xFFFFu, i * 8 + 2 * j);
                consider(kv[v][j] >> 16, i * 8 + 2 * j + 1);
            }
        }
    }
    stamp_s(3);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(besti, d);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if (lane == 0) { fred[wave] = best; ired[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < besti)) { best = fred[w]; besti = ired[w]; }
        token_out[0] = besti;
        if (history && (long long)ctr64 < (long long)history_len) history[ctr64] = besti;
        rng_state[1] = ctr64 + 1ull;
        if (pos_inout) pos_inout[0] = pos0 + 1;
    }
    stamp_s(4);
}

// ------------------------------------------------------------------------------------------------
// Multi-workgroup sampler (vocab % 8 == 0, vocab <= 16 x 8192, 0 < top_k < vocab): the single workgroup above spends
// its time in NV sequential passes over its registers (14 us at 32 k, 37 us at 128 k logits).  Here workgroup g owns
// 8192 logits (one 16-byte vector per thread) and
//   stage A  bounds ITS k-th largest key from below with one pass of the same window select (the lower edge of the
//            histogram bin that holds it) and appends every key >= that bound, with its index, to a candidate list
//            in global memory (write-through stores), then takes an arrival ticket;
//   stage B  (the last workgroup to arrive) reads the <= G x kSampCap candidates, finds the global pivot among them,
//            and runs the exponential race over the candidates that survive it.
// Exact: a key >= the global pivot P is >= its chunk's pivot (the k-th largest of a subset is <= the k-th largest of
// the whole) and so >= the chunk's bound: the union of the candidate sets contains every key >= P, hence its k-th
// largest is P; the race
// uses the same counter-based random numbers by vocabulary index and the same tie-break as the single-workgroup
// kernels, so the tokens are identical (tests/test_engine.py).  A chunk with more than kSampCap candidates (top_k
// beyond the cap, or massive ties) raises an overflow flag and the last arriver runs the generic two-pass select over
// the whole vocabulary instead.
// ------------------------------------------------------------------------------------------------
// k-th largest of the keys this workgroup's threads hold (NK per thread, absent entries flagged in `valid` bits): the
// window select of sample_topk_window_kernel.  All 1024 threads call; returns the pivot key (keep keys >= pivot).
// COARSE: return the lower edge of the histogram bin that holds the k-th largest key instead of resolving the bin — a
// bound BELOW the exact pivot (a superset of the top-k, by up to one bin of 1 << sh keys), one pass cheaper.
template <bool BF16, int NK, bool COARSE = false>
__device__ __forceinline__ uint32_t window_pivot(const uint32_t (&keys)[NK], const uint32_t valid, const uint32_t kmax,
                                                 const unsigned int top_k, unsigned int* hist, unsigned int (*whist)[256],
                                                 float* fred, unsigned int* sel, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    auto merge_hist = [&]() {
        __syncthreads();
        if (tid < 256) {
            unsigned int a = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += whist[w][tid];
            hist[tid] = a;
        }
        __syncthreads();
    };
    uint32_t pivot_key = 0u;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
    __syncthreads();
    for (int sh = BF16 ? 0 : 3;; sh += 3) {
        if (sh > 8) sh = 8;  // 256 << 8 covers every key
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const uint32_t d = (kmax - keys[j]) >> sh;
            if (((valid >> j) & 1u) && d < 256u) atomicAdd(&whist[wave][255u - d], 1u);
        }
        merge_hist();
        unsigned int inwin = 0;
        {
            unsigned int a = (tid < 256) ? hist[tid] : 0u;
            a = (unsigned int)wave_sum_f((float)a);  // <= 131072: exact in fp32
            if (lane == 0) fred[wave] = (float)a;
            __syncthreads();
            inwin = (unsigned int)(fred[0] + fred[1] + fred[2] + fred[3]);
        }
        if (inwin >= top_k || sh == 8) {
            select_bin(hist, nullptr, sel, top_k, tid);
            const unsigned int d = 255u - sel[0], need2 = sel[1];
            __syncthreads();
            if (sh == 0) {
                pivot_key = kmax - d;
            } else if (COARSE) {
                const uint32_t span = ((d + 1u) << sh) - 1u;  // kmax - key <= span for every key of bins 0..d
                pivot_key = span >= kmax ? 0u : kmax - span;
            } else {
                for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
                __syncthreads();
                const uint32_t lowmask = (1u << sh) - 1u;
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    const uint32_t e = kmax - keys[j];
                    if (((valid >> j) & 1u) && (e >> sh) == d) atomicAdd(&whist[wave][255u - (e & lowmask)], 1u);
                }
                merge_hist();
                select_bin(hist, nullptr, sel, need2, tid);
                pivot_key = kmax - ((d << sh) | (255u - sel[0]));
                __syncthreads();
            }
            break;
        }
        for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;  // widen the window and count again
        __syncthreads();
    }
    return pivot_key;
}

__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, int* ired, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    __syncthreads();  // ired may still be read from a previous use
    if (lane == 0) ired[wave] = (int)v;
    __syncthreads();
    v = (uint32_t)ired[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) v = max(v, (uint32_t)ired[w]);
    return v;
}

template <bool BF16>
__global__ __launch_bounds__(1024) void sample_topk_multi_kernel(const uint16_t* __restrict__ logits, const int V,
                                                                  const int top_k, const float inv_temp,
                                                                  unsigned long long* __restrict__ rng_state,
                                                                  int* __restrict__ token_out, int* __restrict__ pos_inout,
                                                                  int* __restrict__ history, const int history_len,
                                                                  unsigned char* __restrict__ slot) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int whist[16][256];
    __shared__ float fred[16];
    __shared__ int ired[16];
    __shared__ unsigned int sel[2];
    __shared__ unsigned int lcnt;
    __shared__ unsigned int lflag;
    __shared__ unsigned int gcount[kSampMaxGroups];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, G = gridDim.x;
    const int V8 = V >> 3;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(slot);
    unsigned int* counts = reinterpret_cast<unsigned int*>(slot + (size_t)kSampMaxGroups * kSampCap * 8);
    unsigned int* ticket = counts + kSampMaxGroups;
    // ---- stage A: this workgroup's 8192 logits -------------------------------------------------------------
    const int vi = g * 1024 + tid;
    const bool ok = vi < V8;
    const u32x4 raw = reinterpret_cast<const u32x4*>(logits)[min(vi, V8 - 1)];
    // what the last arriver's epilogue needs from memory is requested now
    const unsigned long long seed64 = rng_state[0], ctr64 = rng_state[1];
    const int pos0 = pos_inout ? pos_inout[0] : 0;
    if (tid == 0) lcnt = 0u;
    uint32_t keys[8];
    uint32_t kmax = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        keys[2 * j] = order_key16(raw[j] & 0xFFFFu, BF16);
        keys[2 * j + 1] = order_key16(raw[j] >> 16, BF16);
        if (ok) kmax = max(kmax, max(keys[2 * j], keys[2 * j + 1]));
    }
    kmax = block_max_u32(kmax, ired, tid);
    const unsigned int chunk_keys = (unsigned int)(min(V8 - g * 1024, 1024) * 8);
    // fewer keys in the chunk than requested: every key is a candidate (and overflows the cap: generic path)
    const uint32_t lpivot = chunk_keys <= (unsigned int)top_k ? 0u
                            : window_pivot<BF16, 8, true>(keys, ok ? 0xFFu : 0u, kmax, (unsigned int)top_k, hist, whist, fred, sel, tid);
    __syncthreads();
    if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (keys[j] >= lpivot) {
                const unsigned int s = atomicAdd(&lcnt, 1u);
                if (s < (unsigned int)kSampCap)
                    __hip_atomic_store(&cand[(size_t)g * kSampCap + s], ((unsigned long long)keys[j] << 32) | (unsigned int)(vi * 8 + j),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&counts[g], lcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // arrival ticket (see gemv_fast_kernel): the stores above are write-through and complete before the increment
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lflag = (t == (unsigned)G - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (lflag == 0u) return;
    // ---- stage B: the last arriver ---------------------------------------------------------------------------
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch / replay
    constexpr int NC = kSampMaxGroups * kSampCap / 1024;  // candidate slots per thread
    unsigned long long craw[NC];  // requested together with the counts (one round trip); slots past a count are stale
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int sidx = j * 1024 + tid;
        craw[j] = sidx / kSampCap < G ? __hip_atomic_load(&cand[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    if (tid < kSampMaxGroups) gcount[tid] = tid < G ? __hip_atomic_load(&counts[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    bool overflow = false;
#pragma unroll
    for (int q = 0; q < kSampMaxGroups; ++q) overflow |= gcount[q] > (unsigned int)kSampCap;
    if (overflow) {  // workgroup-uniform
        sample_full<BF16>(logits, V, top_k, inv_temp, rng_state, token_out, pos_inout, history, history_len);
        return;
    }
    uint32_t ck[NC], ci[NC], cvalid = 0u;
    uint32_t gmax = 0u;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int sidx = j * 1024 + tid, gq = sidx / kSampCap, jj = sidx % kSampCap;
        ck[j] = 0u; ci[j] = 0u;
        if (gq < G && (unsigned int)jj < gcount[gq]) {
            ck[j] = (uint32_t)(craw[j] >> 32); ci[j] = (uint32_t)craw[j];
            cvalid |= 1u << j;
            gmax = max(gmax, ck[j]);
        }
    }
    gmax = block_max_u32(gmax, ired, tid);
    const uint32_t pivot = window_pivot<BF16, NC>(ck, cvalid, gmax, (unsigned int)top_k, hist, whist, fred, sel, tid);
    auto key_bits = [](const uint32_t k) -> uint32_t { return (k & 0x8000u) ? (k ^ 0x8000u) : (~k & 0xFFFFu); };  // order_key16^-1
    const float mx = bits_to_float(key_bits(gmax), BF16);
    const uint32_t seed = (uint32_t)seed64, ctr = (uint32_t)ctr64;
    float best = -1.0f;
    int besti = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if (((cvalid >> j) & 1u) && ck[j] >= pivot) {
            const int i = (int)ci[j];
            const float pnum = expf((bits_to_float(key_bits(ck[j]), BF16) - mx) * inv_temp);
            const float u = ((float)(hash3(seed, ctr, (uint32_t)i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float scv = pnum / (-logf(u));
            if (scv > best || (scv == best && i < besti)) { best = scv; besti = i; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(besti, d);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if (lane == 0) { fred[wave] = best; ired[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < besti)) { best = fred[w]; besti = ired[w]; }
        token_out[0] = besti;
        if (history && (long long)ctr64 < (long long)history_len) history[ctr64] = besti;
        rng_state[1] = ctr64 + 1ull;
        if (pos_inout) pos_inout[0] = pos0 + 1;
    }
}

constexpr int kGqaMinSeq8 = 2048;           // cache length from which 8-heads-per-KV-head models take the grouped kernel
constexpr int kGqaMinSeq = 4096;            // ... and 4-heads-per-KV-head models
constexpr size_t kGqaMaxLds = 128 * 1024;   // ... if its scores fit this much LDS (mirrored by engine.py's split choice)

// First use of a device (device_ctx(), teal_kernels.hip): one workgroup per CU, so the grouped-query kernel may take more
// than the default 64 KB of the CU's 160 KB LDS (scores of a long share); opted in once per device, outside any stream
// capture.  Returns whether the opt-in succeeded.
bool attention_device_init() {
    bool ok = true;
#define TEAL_OPT(BF, HDV, REPV) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attention_gqa_kernel<BF, HDV, 512, REPV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGqaMaxLds) == hipSuccess
    TEAL_OPT(false, 128, 8); TEAL_OPT(false, 128, 4); TEAL_OPT(false, 64, 8); TEAL_OPT(false, 64, 4);
    TEAL_OPT(true, 128, 8); TEAL_OPT(true, 128, 4); TEAL_OPT(true, 64, 8); TEAL_OPT(true, 64, 4);
#undef TEAL_OPT
    if (!ok) (void)hipGetLastError();
    return ok;
}

}  // namespace teal

using namespace teal;

extern "C" {

int teal_decode_attention_masked(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                 void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                 int max_seq, int dtype, void* stream) {
    if (!qkv || !rope || !pos || !k_cache || !v_cache || !y) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || max_seq <= 0)
        return TEAL_ERR_SHAPE;
    // 16 waves per head: a wave load covers whole cache rows, one pass covers 1024 positions
    const int nt = 1024;
    const size_t lds = (size_t)(3 * head_dim + 2 * (nt / 64) + (nt / 64) * head_dim + max_seq) * sizeof(float);
    if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const dim3 grid(n_head), block(nt);
    auto* q = reinterpret_cast<const uint16_t*>(qkv);
    auto* r = reinterpret_cast<const uint16_t*>(rope);
    auto* kc = reinterpret_cast<uint16_t*>(k_cache);
    auto* vc = reinterpret_cast<uint16_t*>(v_cache);
    auto* yo = reinterpret_cast<uint16_t*>(y);
    auto* mo = reinterpret_cast<unsigned long long*>(mask_out);
#define TEAL_ATT(BF, NTV, HDV) hipLaunchKernelGGL((decode_attention_kernel<BF, NTV, HDV>), grid, block, lds, st, q, r, pos, kc, vc, yo, mo, mask_tau, n_head, n_kv_head, max_seq, scale, phase_start_only())
#define TEAL_ATT_HD(BF, NTV) do { if (head_dim == 128) TEAL_ATT(BF, NTV, 128); else TEAL_ATT(BF, NTV, 64); } while (0)
    if (dtype == TEAL_BF16) TEAL_ATT_HD(true, 1024);
    else TEAL_ATT_HD(false, 1024);
#undef TEAL_ATT_HD
#undef TEAL_ATT
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

static int attention_split_impl(const void* qkv, const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials, size_t partials_bytes,
                                int dtype, void* ws, size_t ws_bytes, void* stream, bool roped = false) {
    if ((!qkv && !qkv_slabs) || (!rope && !roped) || !pos || !k_cache || !v_cache || !partials) return TEAL_ERR_ARG;
    if (roped && (!qkv || !aligned16(qkv))) return TEAL_ERR_ARG;
    if (qkv_slabs && (qkv_nslabs < 1 || qkv_nslabs > 8 || !aligned16(qkv_slabs))) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || max_seq <= 0 ||
        nsplit < 1 || nsplit > 64)
        return TEAL_ERR_SHAPE;
    if (partials_bytes < (size_t)n_head * nsplit * (head_dim + 2) * sizeof(float)) return TEAL_ERR_WORKSPACE;
    DeviceCtx* dctx = device_ctx();  // per-device kernel attributes (not under capture the first time: teal_init())
    if (!dctx) return TEAL_ERR_NO_DEVICE;
    const int chunk_max = (max_seq + nsplit - 1) / nsplit;
    // bandwidth of one workgroup = bytes in flight / latency: long shares get 16 waves, short ones 4 waves (cheaper
    // barriers).  Rows are dealt to the workgroups of a head in groups of STEP = waves x rows-per-wave, round-robin.
    const int nt = chunk_max > 128 ? 1024 : 256;
    const int step = (nt / 64) * (64 / (head_dim / 8));
    const int steps_total = (max_seq + step - 1) / step;
    const int local_steps = (steps_total + nsplit - 1) / nsplit;
    const size_t lds = (size_t)(3 * head_dim + 2 * (nt / 64) + (nt / 64) * head_dim + local_steps * step) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    auto* q = reinterpret_cast<const uint16_t*>(qkv);
    auto* r = reinterpret_cast<const uint16_t*>(rope);
    auto* kc = reinterpret_cast<uint16_t*>(k_cache);
    auto* vc = reinterpret_cast<uint16_t*>(v_cache);
    auto* pw = reinterpret_cast<float*>(partials);
    const dim3 grid(n_head * nsplit), block(nt);
    // stride mode (teal_set_phase_stride): the attention launch takes the next region like a GEMV launch does
    unsigned long long* ph = phase_strided_only();
    // grouped-query models at long contexts: one workgroup per (KV head, split) serves all the query heads of the group
    // (the per-query-head kernel is 2-3 us faster below ~2 k positions: scripts/attention_context_sweep.py)
    const int rep = n_head / n_kv_head;
    const dim3 grid2(nsplit, rep, n_kv_head);  // the per-query-head kernel: (split, query head of the group, KV head), no division
    bool gqa = ((rep == 8 && max_seq >= kGqaMinSeq8) || (rep == 4 && max_seq >= kGqaMinSeq)) && !roped;
    // (y requested — the consumer does not merge: a merge launch follows.  Folding the merge into the split launch by arrival
    // tickets was built in round 3 and measured no faster — equal at 4-8 splits, 1-7 us slower at 16-32,
    // profiles/r03_attention_context_sweep.txt — and is gone since round 5; `ws` is accepted for ABI stability and unused.)
    (void)ws; (void)ws_bytes;
    auto* yo = reinterpret_cast<uint16_t*>(y);
    auto* mo = reinterpret_cast<unsigned long long*>(mask_out);
    if (gqa) {
        constexpr int GNT = 512, GNW = GNT / 64;
        const int gstep = GNW * (64 / (head_dim / 8));
        const int glocal = (((max_seq + gstep - 1) / gstep + nsplit - 1) / nsplit) * gstep;  // rows a workgroup may own
        const size_t region = (size_t)rep * (glocal > GNW * head_dim ? glocal : GNW * head_dim);
        const size_t glds = ((size_t)(rep + 2) * (head_dim / 2) + 2 * rep * GNW + region) * sizeof(float);
        if (glds > (dctx->gqa_lds_ok ? kGqaMaxLds : 64 * 1024)) gqa = false;
        else {
            const dim3 ggrid(n_kv_head * nsplit), gblock(GNT);
#define TEAL_ATTG(BF, HDV, REPV) hipLaunchKernelGGL((decode_attention_gqa_kernel<BF, HDV, GNT, REPV>), ggrid, gblock, glds, st, pos, qkv_slabs, kc, vc, q, pw, r, n_head, n_kv_head, max_seq, nsplit, scale, qkv_nslabs, glocal)
#define TEAL_ATTG_R(BF, HDV) do { if (rep == 8) TEAL_ATTG(BF, HDV, 8); else TEAL_ATTG(BF, HDV, 4); } while (0)
            if (dtype == TEAL_BF16) { if (head_dim == 128) TEAL_ATTG_R(true, 128); else TEAL_ATTG_R(true, 64); }
            else { if (head_dim == 128) TEAL_ATTG_R(false, 128); else TEAL_ATTG_R(false, 64); }
#undef TEAL_ATTG_R
#undef TEAL_ATTG
        }
    }
    if (!gqa) {
        if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
#define TEAL_ATTS_R(BF, HDV, NTV, RP) hipLaunchKernelGGL((decode_attention_split_kernel<BF, HDV, NTV, RP>), grid2, block, lds, st, pos, kc, vc, q, max_seq, nsplit, rep, qkv_nslabs, qkv_slabs, pw, r, n_head, n_kv_head, scale, ph)
#define TEAL_ATTS(BF, HDV, NTV) do { if (roped) TEAL_ATTS_R(BF, HDV, NTV, true); else TEAL_ATTS_R(BF, HDV, NTV, false); } while (0)
#define TEAL_ATTS_NT(BF, HDV) do { if (nt == 1024) TEAL_ATTS(BF, HDV, 1024); else TEAL_ATTS(BF, HDV, 256); } while (0)
    if (dtype == TEAL_BF16) { if (head_dim == 128) TEAL_ATTS_NT(true, 128); else TEAL_ATTS_NT(true, 64); }
    else { if (head_dim == 128) TEAL_ATTS_NT(false, 128); else TEAL_ATTS_NT(false, 64); }
#undef TEAL_ATTS_NT
#undef TEAL_ATTS
#undef TEAL_ATTS_R
    }
    if (hipGetLastError() != hipSuccess) return TEAL_ERR_LAUNCH;
    if (!y) return TEAL_OK;  // partials only: the consumer merges (TEAL_IN_ATTN_MERGE)
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((decode_attention_merge_kernel<true>), dim3(n_head), dim3(128), 0, st, pw, yo, mo, mask_tau, head_dim, nsplit);
    else
        hipLaunchKernelGGL((decode_attention_merge_kernel<false>), dim3(n_head), dim3(128), 0, st, pw, yo, mo, mask_tau, head_dim, nsplit);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_decode_attention_split(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                int max_seq, int nsplit, void* partials, size_t partials_bytes, int dtype, void* stream) {
    if (!qkv) return TEAL_ERR_ARG;
    return attention_split_impl(qkv, nullptr, 0, rope, pos, k_cache, v_cache, y, mask_out, mask_tau, n_head, n_kv_head,
                                head_dim, max_seq, nsplit, partials, partials_bytes, dtype, nullptr, 0, stream);
}

int teal_decode_attention_split_slabs(const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                      void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                      int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials,
                                      size_t partials_bytes, int dtype, void* stream) {
    if (!qkv_slabs) return TEAL_ERR_ARG;
    return attention_split_impl(nullptr, qkv_slabs, qkv_nslabs, rope, pos, k_cache, v_cache, y, mask_out, mask_tau, n_head,
                                n_kv_head, head_dim, max_seq, nsplit, partials, partials_bytes, dtype, nullptr, 0, stream);
}

int teal_decode_attention_split_ws(const void* qkv, const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                   void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                   int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials, size_t partials_bytes,
                                   int dtype, void* ws, size_t ws_bytes, void* stream) {
    if ((qkv != nullptr) == (qkv_slabs != nullptr)) return TEAL_ERR_ARG;  // exactly one form of the projection
    return attention_split_impl(qkv, qkv_slabs, qkv_nslabs, rope, pos, k_cache, v_cache, y, mask_out, mask_tau, n_head,
                                n_kv_head, head_dim, max_seq, nsplit, partials, partials_bytes, dtype, ws, ws_bytes, stream);
}

int teal_decode_attention_split_roped(const void* q, const int32_t* pos, const void* k_cache, const void* v_cache, void* y,
                                      void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim, int max_seq,
                                      int nsplit, void* partials, size_t partials_bytes, int dtype, void* ws, size_t ws_bytes,
                                      void* stream) {
    return attention_split_impl(q, nullptr, 0, nullptr, pos, const_cast<void*>(k_cache), const_cast<void*>(v_cache), y, mask_out,
                                mask_tau, n_head, n_kv_head, head_dim, max_seq, nsplit, partials, partials_bytes, dtype, ws,
                                ws_bytes, stream, true);
}

int teal_decode_attention(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                          void* y, int n_head, int n_kv_head, int head_dim, int max_seq, int dtype, void* stream) {
    return teal_decode_attention_masked(qkv, rope, pos, k_cache, v_cache, y, nullptr, 0.0f, n_head, n_kv_head, head_dim,
                                        max_seq, dtype, stream);
}

int teal_sample_topk(const void* logits, int vocab, int dtype, int top_k, float temperature, void* rng_state,
                     int32_t* token_out, int32_t* pos_inout, int32_t* history, int history_len, void* stream) {
    return teal_sample_topk_ws(logits, vocab, dtype, top_k, temperature, rng_state, token_out, pos_inout, history, history_len,
                               nullptr, 0, stream);
}

int teal_sample_topk_ws(const void* logits, int vocab, int dtype, int top_k, float temperature, void* rng_state,
                        int32_t* token_out, int32_t* pos_inout, int32_t* history, int history_len, void* ws, size_t ws_bytes,
                        void* stream) {
    if (!logits || !rng_state || !token_out || vocab <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (!aligned16(logits)) return TEAL_ERR_ALIGN;
    const float inv_temp = 1.0f / fmaxf(temperature, 1e-5f);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto* lg = reinterpret_cast<const uint16_t*>(logits);
    auto* rs = reinterpret_cast<unsigned long long*>(rng_state);
#define TEAL_SAMPLE(KERNEL) hipLaunchKernelGGL((KERNEL), dim3(1), dim3(1024), 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len)
#define TEAL_SAMPLE_W(KERNEL) hipLaunchKernelGGL((KERNEL), dim3(1), dim3(1024), 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len, phase_start_only())
    const bool bf = dtype == TEAL_BF16;
    if ((vocab & 7) == 0 && vocab > 8192 && vocab <= kSampMaxGroups * 8192 && top_k > 0 && top_k < vocab && ws_prepared(ws, ws_bytes)) {
        // one workgroup per 8192 logits + the last arriver (sample_topk_multi_kernel)
        unsigned char* slot = ws_sampler(ws);  // scratch of the caller's prepared workspace (one per stream)
        const dim3 grid((vocab / 8 + 1023) / 1024), block(1024);
        if (bf) hipLaunchKernelGGL((sample_topk_multi_kernel<true>), grid, block, 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len, slot);
        else hipLaunchKernelGGL((sample_topk_multi_kernel<false>), grid, block, 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len, slot);
    } else if ((vocab & 7) == 0 && vocab <= 4 * 8192) {  // register-resident keys, window select: 4 vectors per thread
        if (bf) TEAL_SAMPLE_W((sample_topk_window_kernel<true, 4>)); else TEAL_SAMPLE_W((sample_topk_window_kernel<false, 4>));
    } else if ((vocab & 7) == 0 && vocab <= 16 * 8192) {  // 16 vectors per thread (Llama-3's 128256)
        if (bf) TEAL_SAMPLE_W((sample_topk_window_kernel<true, 16>)); else TEAL_SAMPLE_W((sample_topk_window_kernel<false, 16>));
    } else {  // any size: two full radix passes over memory
        if (bf) TEAL_SAMPLE((sample_topk_kernel<true>)); else TEAL_SAMPLE((sample_topk_kernel<false>));
    }
#undef TEAL_SAMPLE
#undef TEAL_SAMPLE_W
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}


}  // extern "C"
