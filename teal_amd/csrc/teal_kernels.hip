// teal_kernels.hip — host side of libteal_hip.so: launch-geometry choice (pick_config / run_gemv), the C-ABI
// entry points of the sparse GEMV (include/teal_hip.h), and the small kernels around it (ordered split-K
// reduce, gate|up epilogue, standalone compaction).  gfx950 / CDNA4, wave64.
//
//   teal_gemv_kernel.h          the sparse GEMV kernel template (the hot kernel; design notes there)
//   teal_gemv_w*_*.hip          its instantiations, one translation unit per (weight width, dtype)
//   teal_attention.hip          decode attention + fused sampler and their entry points
//
// Replaces (reference tree FasterDecoding/TEAL @ 2024-10-22):
//   kernels/sparse_gemv.py:87-142   splitk_sparse_gemv (host wrapper, autotune, grid)  -> run_gemv
//   kernels/sparse_gemv.py:196-237  qkv_gemv                                           -> teal_sparse_qkv_gemv*
//   kernels/sparse_gemv.py:8-12     init_to_zero("Y") pre-hook memset                  -> gone
#include "teal_common.h"
#include "teal_gemv_fast_decl.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

namespace teal {

// ------------------------------------------------------------------------------------------------
// Workgroup-wide compaction of x against one threshold.
//   phase A: one ballot per 64-element chunk -> masks[] in LDS
//   phase B: wave 0 turns popcounts into an exclusive prefix (prefix[nch] = total kept)
// `nan_keeps`: GEMV mode — a NaN activation is kept so that it poisons the output like the
// reference's masked `0 * NaN` does; teal_compact uses the pure rule.
// ------------------------------------------------------------------------------------------------
template <int WAVES, bool BF16>
__device__ __forceinline__ void wg_ballot_prefix(const uint16_t* __restrict__ x, const int Z,
                                                 const float tau, const bool nan_keeps,
                                                 unsigned long long* masks, int* prefix) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = (Z + 63) >> 6;
    for (int c = wave; c < nch; c += WAVES) {
        const int m = (c << 6) + lane;
        bool k = false;
        if (m < Z) {
            const float v = bits_to_float(x[m], BF16);
            k = keep_rule(v, tau) || (nan_keeps && (v != v));
        }
        const unsigned long long mask = __ballot(k);
        if (lane == 0) masks[c] = mask;
    }
    __syncthreads();
    if (wave == 0) {
        int base = 0;
        for (int g0 = 0; g0 < nch; g0 += 64) {
            const int c = g0 + lane;
            const int v = (c < nch) ? __popcll(masks[c]) : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (c < nch) prefix[c] = base + incl - v;
            base += __shfl(incl, 63);
        }
        if (lane == 0) prefix[nch] = base;
    }
    __syncthreads();
}


// y[n] = round(sum_s ws[s][n]) in slice order; one thread per column.
template <bool BF16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Params p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.ws_ld) return;
    int s = 0;
    if (p.nseg > 1 && n >= p.seg[1].ws_off) s = 1;
    if (p.nseg > 2 && n >= p.seg[2].ws_off) s = 2;
    float sum = 0.0f;
    for (int k = 0; k < p.split; ++k) sum += p.ws[(size_t)k * p.ws_ld + n];
    reinterpret_cast<uint16_t*>(p.seg[s].y)[n - p.seg[s].ws_off] = (p.act0 && s == 0) ? silu_bits<BF16>(sum) : float_to_bits<BF16>(sum);
}

// h[n] = silu(gate[n]) * up[n] from the fp32 slabs of a 2-segment (gate | up) GEMV.
// gate and up are rounded to dtype first, silu is rounded, then the product is rounded — the same
// roundings the unfused fp16 sequence F.silu(g) * u performs (gpt-fast/model.py:258-259).
template <bool BF16>
__global__ __launch_bounds__(256) void gateup_silu_epilogue_kernel(const Params p, void* h) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int N = p.seg[0].ncols;
    if (n >= N) return;
    float g = 0.0f, u = 0.0f;
    for (int k = 0; k < p.split; ++k) {
        g += p.ws[(size_t)k * p.ws_ld + n];
        u += p.ws[(size_t)k * p.ws_ld + N + n];
    }
    const float g16 = bits_to_float(float_to_bits<BF16>(g), BF16);
    const float u16 = bits_to_float(float_to_bits<BF16>(u), BF16);
    const float sl = g16 / (1.0f + expf(-g16));
    const float sl16 = bits_to_float(float_to_bits<BF16>(sl), BF16);
    reinterpret_cast<uint16_t*>(h)[n] = float_to_bits<BF16>(sl16 * u16);
}

// Standalone compaction (one workgroup): ascending kept indices + count to global memory.
template <bool BF16>
__global__ __launch_bounds__(1024) void compact_kernel(const uint16_t* __restrict__ x, const int Z,
                                                       const float tau, int32_t* __restrict__ idx_out,
                                                       int32_t* __restrict__ count_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nch = (Z + 63) >> 6;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem);
    int* prefix = reinterpret_cast<int*>(masks + nch);
    wg_ballot_prefix<16, BF16>(x, Z, tau, false, masks, prefix);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int c = wave; c < nch; c += 16) {
        const unsigned long long mask = masks[c];
        if ((mask >> lane) & 1ull) idx_out[prefix[c] + lane_rank(mask)] = (c << 6) + lane;
    }
    if (threadIdx.x == 0) *count_out = prefix[nch];
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------


#ifdef TEAL_DIAGNOSTICS  // libteal_hip_diag.so only (teal_common.h): the product library has no mutable switches
Config g_override = {0, 0, 0, 0};
unsigned long long* g_phase = nullptr;
size_t g_phase_stride = 0;  // > 0: consecutive GEMV launches stamp consecutive regions of this many uint64
int g_phase_seq = 0;
int g_wave_local = 1;
int g_fast = 1;  // lean kernel (teal_gemv_fast.h) where the shape qualifies
char g_last_desc[160] = "";  // template instantiation + grid of the most recent GEMV launch (teal_last_launch_desc)
#endif

// ---- per-device properties (immutable once cached) ---------------------------------------------------------------
constexpr int kMaxDevices = 64;
static DeviceCtx g_dev[kMaxDevices] = {};
static std::mutex g_dev_mu;

DeviceCtx* device_ctx() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    DeviceCtx* c = &g_dev[dev];
    if (c->num_cu > 0) return c;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (c->num_cu > 0) return c;
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) {
        (void)hipGetLastError();
        return nullptr;
    }
    c->gqa_lds_ok = attention_device_init();
    c->num_cu = cu;
    return c;
}

// ---- prepared workspaces (teal_workspace_init): host-side registry, keyed by the device pointer -------------------
struct WsEntry { const void* ws; size_t bytes; };
static std::vector<WsEntry> g_ws_reg;
static std::mutex g_ws_mu;

bool ws_prepared(const void* ws, size_t ws_bytes) {
    if (!ws || ws_bytes < kWsHeaderBytes) return false;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (const WsEntry& e : g_ws_reg)
        if (e.ws == ws) return true;
    return false;
}


size_t lds_bytes(int Z, int cap, int waves, int lpr, bool pair = false) {
    const int nch = (Z + 63) >> 6;
    return (size_t)nch * 8 + 128 + (size_t)cap * 4 + (size_t)waves * lpr * 8 * 4 * (pair ? 2 : 1) + 128;  // + int8 bias sums
}

int count_tiles(const Params& p, int bn) {
    int t = 0;
    for (int i = 0; i < p.nseg; ++i) t += (p.seg[i].ncols + bn - 1) / bn;
    return t;
}

// Launch geometry from (Z, columns, CU count).  Deterministic: no autotune at first call.
// Measured on MI355X (profiles/, scripts/tune_gemv.py): the kernel wants ONE 16-wave workgroup per
// CU (every workgroup repeats the compaction prologue, so more workgroups only add latency), and
// a single launch (split == 1) whenever the column tiles alone can occupy >= ~60 % of the CUs.
Config pick_config(int Z, int ncols_total, int nseg_tiles_hint) {
    (void)nseg_tiles_hint;
    const int ncu = num_cu_or(256);
    Config c;
    c.waves = g_override.waves ? g_override.waves : 16;
    c.unroll = 4;
    c.lpr = 0;
    c.split = 1;
    // widest tile whose tile count still covers most CUs -> no split-K, no second launch
    for (int lpr = 64; lpr >= 8; lpr >>= 1) {
        const int tiles = (ncols_total + lpr * 8 - 1) / (lpr * 8);
        if (tiles <= ncu + ncu / 8 && tiles * 5 >= ncu * 3) { c.lpr = lpr; break; }
    }
    if (c.lpr == 0) {
        const int t8 = (ncols_total + 63) / 64;
        if (t8 > ncu) {  // more 64-column tiles than CUs: widest tile that keeps >= ncu workgroups
            c.lpr = 8;
            for (int lpr = 64; lpr > 8; lpr >>= 1)
                if ((ncols_total + lpr * 8 - 1) / (lpr * 8) >= ncu) { c.lpr = lpr; break; }
        } else {  // few columns: 64-column tiles, split the kept rows to reach ~1 workgroup per CU.  With the
                  // wave-local compaction a slice is a set of 16-chunk rounds, so split <= rounds (and <= 8
                  // keeps the slab count small for consumers that re-read them); without it, 512-byte row
                  // segments and a deeper split measured best.
            const int rounds = (((Z + 63) >> 6) + c.waves - 1) / c.waves;
            if (g_wave_local && rounds >= 2) {
                c.lpr = 8;
                const int tiles = (ncols_total + 63) / 64;
                int split = ncu / tiles;
                if (split > rounds) split = rounds;
                if (split > 8) split = 8;
                if (split < 1) split = 1;
                c.split = split;
            } else {
                c.lpr = ncols_total >= 2048 ? 32 : 8;
                const int tiles = (ncols_total + c.lpr * 8 - 1) / (c.lpr * 8);
                int split = ncu / tiles;
                const int max_by_rows = Z / (4 * c.waves * (64 / c.lpr));  // >= ~2 steps per workgroup at 50 %
                if (split > max_by_rows) split = max_by_rows;
                if (split < 1) split = 1;
                if (split > kMaxSplit) split = kMaxSplit;
                c.split = split;
            }
        }
    }
    if (g_override.lpr) c.lpr = g_override.lpr;
    if (g_override.waves) c.waves = g_override.waves;
    if (g_override.split) c.split = g_override.split;
    if (g_override.unroll) c.unroll = g_override.unroll;
    // LDS list capacity: stay inside the 64 KB a workgroup gets without opting in to more
    while ((size_t)((Z + c.split - 1) / c.split) * 4 > 40 * 1024 && c.split < kMaxSplit) ++c.split;
    return c;
}


// Does this launch qualify for the lean kernel (teal_gemv_fast.h)?  Whole chunks and tiles, one weight image (or two:
// the gate|up pair, or gate | up as two unpaired segments), interleaved slabs or a single rounded output, wave-local
// lists that fit; 16-bit weights, or int8 images in 128-column tiles (never paired).
bool fast_eligible(const Params& p, const Config& c, bool to_ws, size_t ws_bytes, FastLaunch& f) {
    if (!g_fast || c.waves != 16 || c.unroll != 4 || (c.lpr != 8 && c.lpr != 16)) return false;
    if (p.w8 && (c.lpr != 16 || p.pair)) return false;
    if ((p.Z & 63) || p.Z > 65536 || !p.wl) return false;
    const int mode = p.in.mode;
    if (mode != 0 && mode != 1 && mode != 2 && mode != 3 && mode != 4) return false;
    const int bn = c.lpr * 8, nch = p.Z >> 6, owned_all = (nch + 15) / 16;
    // outputs: interleaved slabs for the next launch; one rounded vector (split == 1); or split > 1 folded into ONE
    // launch by arrival tickets (the last slice of a tile sums the partials in slice order)
    const bool ticketed = !to_ws && c.split > 1;
    if (to_ws ? (!p.ws_il || c.split > 8) : (c.split > 8)) return false;
    if (ticketed && (!p.tickets || p.ntiles > kTicketTiles || !p.ws ||
                     ws_bytes < (size_t)((c.split + 3) & ~3) * (size_t)p.ws_ld * sizeof(float)))
        return false;
    const int nseg = p.pair ? 2 : p.nseg;
    int off = 0;
    for (int i = 0; i < nseg; ++i) {
        const Seg& sg = p.seg[i];
        if (sg.ncols % bn) return false;
        if (p.pair) continue;
        // one weight image cut into threshold segments (q|k|v) — or exactly two images, one per segment (gate | up unpaired)
        const bool same_image = sg.w == p.seg[0].w && sg.ld == p.seg[0].ld && sg.col0 == p.seg[0].col0 + off;
        if (!same_image && !(nseg == 2 && i == 1)) return false;
        if (p.w8 && same_image && reinterpret_cast<const uint16_t*>(sg.scale) != reinterpret_cast<const uint16_t*>(p.seg[0].scale) + off)
            return false;  // one image: its scale vector is one vector
        if (!to_ws && reinterpret_cast<const uint16_t*>(sg.y) != reinterpret_cast<const uint16_t*>(p.seg[0].y) + off) return false;
        off += sg.ncols;
    }
    if (p.pair && (mode != 1 || p.seg[0].ncols != p.seg[1].ncols)) return false;
    int kr, rounds_owned;
    if (mode == 1) {
        if (owned_all > 16 || c.split > owned_all) return false;
        kr = owned_all <= 4 ? 4 : (owned_all <= 8 ? 8 : 16);
        rounds_owned = (kr + c.split - 1) / c.split;
    } else {
        const int need = (owned_all + c.split - 1) / c.split;
        if (need > 8 || c.split > owned_all) return false;
        kr = need <= 1 ? 1 : (need <= 4 ? 4 : 8);
        rounds_owned = need;
        if (mode == 4 && kr * p.in.att_ns > 64) return false;
    }
    f = FastLaunch{};
    f.a.cap = rounds_owned * 64;
    f.lds = 64 + (size_t)16 * f.a.cap * 4 + (size_t)16 * bn * 4 * (p.pair ? 2 : 1) + (p.w8 ? 128 : 0);
    if (f.lds > 64 * 1024) return false;
    f.w8 = p.w8;
    f.mode = mode; f.pair = p.pair; f.lpr = c.lpr; f.kr = kr; f.ntiles = p.ntiles; f.split = c.split;
    f.Z = p.Z; f.nslabs = p.in.nslabs; f.eps = p.in.eps;
    switch (mode) {
        case 1: f.in0 = p.in.resid_in; f.in1 = p.in.slabs; f.in2 = p.in.norm_w; f.row_index = p.in.row_index;
                // interleaved slabs, or ONE planar fp32 vector (the hand-over of a TEAL_OUT_SLAB_SUM launch): bit 8 of the count
                if (p.in.nslabs > 0 && !p.in.slabs_il && p.in.nslabs != 1) return false;
                if (p.in.nslabs > 8) return false;
                if (p.in.nslabs == 1 && !p.in.slabs_il) f.nslabs = 1 | 0x100;
                break;
        case 2: f.in0 = p.x; break;  // gate | up contiguous [2Z]
        case 3: f.in0 = p.x; f.in1 = p.in.masks; break;
        case 4: f.in0 = p.in.att; f.a.att_hd = p.in.att_hd; f.a.att_ns = p.in.att_ns;
                f.nslabs = p.in.att_ns | ((p.in.att_hd == 128 ? 7 : 6) << 8);  // the preloaded slot of the merge producer
                break;
        default: f.in0 = p.x; break;
    }
    const size_t wb = p.w8 ? 1 : 2;  // bytes per weight
    f.a.w0 = reinterpret_cast<const char*>(p.seg[0].w) + (size_t)p.seg[0].col0 * wb;
    f.a.ld0 = p.seg[0].ld;
    const bool two_images = !p.pair && p.nseg == 2 && !(p.seg[1].w == p.seg[0].w && p.seg[1].ld == p.seg[0].ld &&
                                                        p.seg[1].col0 == p.seg[0].col0 + p.seg[0].ncols);
    f.a.w1 = (p.pair || two_images) ? reinterpret_cast<const char*>(p.seg[1].w) + (size_t)p.seg[1].col0 * wb : nullptr;
    f.a.scale0 = reinterpret_cast<const uint16_t*>(p.seg[0].scale);
    f.a.scale1 = two_images ? reinterpret_cast<const uint16_t*>(p.seg[1].scale) : nullptr;
    f.a.ld1 = (p.pair || two_images) ? p.seg[1].ld : 0;
    f.a.w1_tile = two_images ? p.seg[1].tile0 : INT_MAX;
    f.a.y = p.seg[0].y;
    f.a.ws = p.ws;
    f.a.mask_out = p.mask_out;
    f.a.mask_tau = p.mask_tau;
    f.a.resid_out = p.in.resid_out;
    f.a.phase = p.phase;
    f.a.tau0 = p.seg[0].tau; f.a.tau1 = p.seg[1].tau; f.a.tau2 = p.seg[2].tau;
    f.a.seg_tile1 = (!p.pair && p.nseg > 1) ? p.seg[1].tile0 : INT_MAX;
    f.a.seg_tile2 = (!p.pair && p.nseg > 2) ? p.seg[2].tile0 : INT_MAX;
    f.a.act0 = p.act0;
    f.a.sum32 = p.sum32;
    if (p.sum32 && (to_ws || p.pair || p.nseg != 1 || p.act0 || p.rope)) return false;
    f.a.gate_act = p.in.gate_act;
    f.a.ws_stride = (to_ws || ticketed) ? ((c.split + 3) & ~3) : 0;
    f.a.ticket = ticketed ? p.tickets : nullptr;
    if (p.rope) {  // RoPE + KV append epilogue: one rounded q|k|v vector, no split-K, tiles inside one head
        if (mode != 1 || p.pair || p.w8 || to_ws || c.split != 1 || p.nseg != 3 || (p.rope_hd != 64 && p.rope_hd != 128) ||
            p.rope_hd % bn || p.seg[1].ncols != p.seg[2].ncols || p.seg[0].ncols % p.rope_hd || p.seg[1].ncols % p.rope_hd)
            return false;
        f.a.rope = p.rope; f.a.rope_pos = p.rope_pos; f.a.kc = p.kc; f.a.vc = p.vc;
        f.a.rope_hd = p.rope_hd; f.a.rope_dim = p.seg[0].ncols; f.a.rope_kv = p.seg[1].ncols; f.a.rope_max_seq = p.rope_max_seq;
    }
    return true;
}

// one translation unit per (weight width, dtype): see teal_common.h
inline hipError_t launch_gemv(const Params& p, int dtype, size_t lds, const Config& c, hipStream_t st) {
    if (p.w8) return dtype == TEAL_BF16 ? launch_gemv_w8_bf16(p, lds, c, st) : launch_gemv_w8_f16(p, lds, c, st);
    return dtype == TEAL_BF16 ? launch_gemv_w16_bf16(p, lds, c, st) : launch_gemv_w16_f16(p, lds, c, st);
}

// Common driver: fills geometry fields of `p` (segments' w/y/tau/ld/col0/ncols are set by the
// caller), launches the GEMV and, if needed, the ordered slab reduce.
int run_gemv(Params& p, int dtype, void* ws, size_t ws_bytes, bool to_ws, hipStream_t st,
             Config* used, bool few_slabs = false, bool interleave = false, bool caller_ws = true, bool* rope_taken = nullptr,
             char* desc = nullptr, int desc_bytes = 0) {
    // desc / desc_bytes: the caller's host buffer for the description of the launch this call makes (teal_gemv_out_t.desc)
    // *rope_taken: the launch ran a ROPE instantiation of the lean kernel (TEAL_OUT_QKV_ROPE bookkeeping; per call, no global)
    if (rope_taken) *rope_taken = false;
    // caller_ws: `ws` is the caller's workspace (as opposed to an explicit slab destination, TEAL_OUT_SLABS).  A workspace
    // prepared by teal_workspace_init() starts with the header that holds the arrival counters: slabs go behind it.
    p.tickets = nullptr;
    if (caller_ws && ws_prepared(ws, ws_bytes)) {
        p.tickets = ws_tickets(ws);
        ws = ws_slabs(ws);
        ws_bytes -= kWsHeaderBytes;
    }
    int total_cols = 0;
    for (int i = 0; i < p.nseg; ++i) total_cols += p.seg[i].ncols;
    Config c = pick_config(p.Z, total_cols, p.nseg);
    bool wide_sliced = false;
    // (a TEAL_OUT_SLAB_SUM launch — p.sum32 — is a slab launch whose last slice folds the slabs: same geometry)
    if (few_slabs && (to_ws || p.sum32) && (size_t)p.Z * total_cols >= (size_t)8192 * 8192 && !p.w8 && !p.pair && !g_override.split &&
        !g_override.lpr) {
        // slab output of a 70B-class matrix (>= 8192 x 8192): 128-column tiles (256-byte row segments) with the kept
        // rows sliced so that tiles x slices ~ the CU count — fewer, longer row requests per CU.  Measured +2-8 % on
        // those launches; on 7B / 8B matrices the extra slices cost more in the consumers' prologues than they save
        // (7B: 529 -> 518 tok/s), hence the size gate.
        const int ncu = num_cu_or(256);
        const int tiles = (total_cols + 127) / 128;
        const int rounds = (((p.Z + 63) >> 6) + c.waves - 1) / c.waves;
        int split = ncu / tiles;
        if (split > rounds) split = rounds;
        if (split > 8) split = 8;
        if (split < 1) split = 1;
        if (tiles * split * 10 >= ncu * 7 && (size_t)((p.Z + split - 1) / split) * 4 <= 40 * 1024) {
            c.lpr = 16;
            c.split = split;
            wide_sliced = true;
        }
    }
    if (!wide_sliced && few_slabs && c.split > 1 && !g_override.split && !g_override.lpr) {
        // the consumer re-reads every slab in each of its workgroups: prefer narrow tiles and a
        // shallow split (64-column tiles, <= 8 slabs) over 512-byte row segments
        const int ncu = num_cu_or(256);
        c.lpr = 8;
        const int tiles = (total_cols + 63) / 64;
        int split = ncu / tiles;
        if (split > 8) split = 8;
        // a slice of the wave-local compaction is a set of 16-chunk rounds: no more slices than rounds, or the launch falls
        // back to the workgroup-wide list of the general kernel (round 5: the rank-local shapes of tensor parallelism — wo
        // 2048 -> 4096 has two rounds, Llama-3-8B's wqkv / 2 four — ran 7.9-9.2 us that way, profiles/r05_tp_rank_local_launches.txt)
        const int rounds = (((p.Z + 63) >> 6) + c.waves - 1) / c.waves;
        if (g_wave_local && c.waves == 16 && split > rounds) split = rounds;
        if (split < 1) split = 1;
        c.split = split;
        while ((size_t)((p.Z + c.split - 1) / c.split) * 4 > 40 * 1024 && c.split < kMaxSplit) ++c.split;
        // a long vector can force more slices than CUs / tiles (LDS list capacity): then fill whole rounds of
        // workgroups (Llama-2-70B down, Z = 28672: 128 tiles x 3 slices = 1.5 rounds -> x 4 = 2 full rounds)
        while (tiles * c.split > ncu && (tiles * c.split) % ncu != 0 && c.split < 8) ++c.split;
    }
    if (p.w8) {
        // int8: 8 bytes per lane.  The stream is bound by cache-line REQUESTS per CU (measured: a 64-byte and a
        // 128-byte row segment cost the same, ~3.2 ns per row and CU), so a row segment should be a whole 128-byte
        // line = 16 lanes = 128 columns, as long as tiles x slices still cover most of the CUs
        if (c.lpr > 32) c.lpr = 32;
        if (!g_override.lpr && !g_override.split && !p.pair && c.lpr < 16) {
            const int ncu = num_cu_or(256);
            const int tiles = (total_cols + 127) / 128;
            const int rounds = (((p.Z + 63) >> 6) + 15) / 16;
            int split = 1;
            if (to_ws || p.tickets) {  // slab output, or a rounded output folded in by arrival tickets: the kept rows may be
                                       // sliced (wave-local: <= rounds, interleaved slabs: <= 8)
                split = ncu / tiles;
                if (split > rounds) split = rounds;
                if (split > 8) split = 8;
                if (split < 1) split = 1;
            }
            // (half the CUs with 128-byte segments stream as much as all of them with 64-byte segments — the same number of
            // requests per microsecond — and 128-column tiles are what the lean kernel is built for: wo, 32 tiles x 4 slices)
            if (tiles * split * 2 >= ncu) {
                c.lpr = 16;
                c.split = split;
            }
        }
        for (int i = 0; i < (p.pair ? 2 : p.nseg); ++i)
            if (!p.seg[i].scale || (p.seg[i].ld & 7) || (p.seg[i].col0 & 7)) return TEAL_ERR_ARG;
    }
    if (p.w8 || p.in.mode != 0 || p.pair) {  // fused / int8 variants exist for 16-wave workgroups
        c.waves = 16;
        c.unroll = 4;
    }
    if ((p.in.mode == 1 || p.in.mode == 4) && p.Z > c.waves * 64 * 16) return TEAL_ERR_SHAPE;  // register-resident producer
    if (p.pair) {  // both matrices in one workgroup; the activation needs complete sums: no split-K
        c.split = 1;
        if ((size_t)(p.Z + 1) * 4 > 44 * 1024) return TEAL_ERR_SHAPE;
        if (!g_override.lpr) {
            const int ncu = num_cu_or(256);
            const int n1 = p.seg[0].ncols;
            c.lpr = 8;
            for (int lpr = 64; lpr > 8; lpr >>= 1)  // widest tile that still gives >= 2/3 of the CUs a tile
                if (((n1 + lpr * 8 - 1) / (lpr * 8)) * 3 >= ncu * 2) { c.lpr = lpr; break; }
            // a ragged second round (a 16-wave workgroup owns its CU): between one and 1.5 rounds of 64-column tiles — Llama-30B,
            // inter 17920 = 280 tiles on 256 CUs — run as ONE round of 128-column tiles when those cover at least half the CUs
            // (round 6, profiles/r06_ratio_vs_width.txt: 30B gate | up 56.2 us at 280 workgroups)
            const int t8 = (n1 + 63) / 64, t16 = (n1 + 127) / 128;
            if (c.lpr == 8 && t8 > ncu && t8 * 2 < ncu * 3 && t16 * 2 >= ncu) c.lpr = 16;
        }
    }
    const int bn = c.lpr * 8;
    int t = 0, off = 0;
    for (int i = 0; i < (p.pair ? 1 : p.nseg); ++i) {
        p.seg[i].tile0 = t;
        p.seg[i].ws_off = off;
        t += (p.seg[i].ncols + bn - 1) / bn;
        off += p.seg[i].ncols;
    }
    if (p.pair) p.nseg = 1;  // tile space = the gate columns; seg[1] rides along
    p.ntiles = t;
    p.split = c.split;
    p.ws_ld = off;
    p.cap = (p.Z + c.split - 1) / c.split + 1;
    p.wl = 0;
    p.sl = 0;
    p.krt = 0;
    if (g_wave_local && c.waves == 16) {
        const int nch = (p.Z + 63) >> 6;
        const int owned = (nch + c.waves - 1) / c.waves;  // rounds of `waves` chunks
        // element-wise producers (everything but the RMSNorm, which needs the whole vector in every workgroup)
        // cache only the rounds of the workgroup's slice
        const bool slice_local = p.in.mode != 1 && c.split > 1;
        const int need = slice_local ? (owned + c.split - 1) / c.split : owned;
        const int krt = need <= 4 ? 4 : (need <= 8 ? 8 : 16);
        const int capw = (slice_local ? need : (krt + c.split - 1) / c.split) * 64;  // entries one wave can own
        const bool merge_ok = p.in.mode != 4 || krt * p.in.att_ns <= 64;  // one lane per (chunk, split) in the merge
        if (need <= krt && c.split <= owned && (size_t)c.waves * capw * 4 <= 40 * 1024 && merge_ok) {
            p.wl = 1;
            p.sl = slice_local ? 1 : 0;
            p.krt = krt;
            p.cap = capw;
        }
    }
    if (p.in.mode == 4) {  // the merge producer holds one lane per (cached chunk, split): cache depth x splits <= 64
        const int owned_all = (((p.Z + 63) >> 6) + c.waves - 1) / c.waves;
        const int kr = p.krt ? p.krt : (owned_all <= 4 ? 4 : (owned_all <= 8 ? 8 : 16));
        if (kr * p.in.att_ns > 64 || (!p.wl && owned_all > kr)) return TEAL_ERR_SHAPE;
    }
    p.to_ws = to_ws ? 1 : 0;
    p.ws = reinterpret_cast<float*>(ws);
    p.phase = phase_next_region();  // (nullptr in the product library)
    const size_t lds = lds_bytes(p.Z, p.wl ? p.cap * c.waves : p.cap, c.waves, c.lpr, p.pair != 0);
    if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
    p.ws_il = (interleave && to_ws && c.split <= 8) ? 1 : 0;
    if (c.split > 1 || to_ws) {
        const size_t slabs = p.ws_il ? (size_t)((c.split + 3) & ~3) : (size_t)c.split;
        if (!ws || ws_bytes < slabs * off * sizeof(float)) return TEAL_ERR_WORKSPACE;
        if (!aligned16(ws)) return TEAL_ERR_ALIGN;
    }
    if (used) *used = c;
    {
        FastLaunch f;
        const bool lean = fast_eligible(p, c, to_ws, ws_bytes, f);
        if (p.rope && !lean) return TEAL_ERR_CONFIG;  // the RoPE epilogue exists in the lean kernel only: never fall through unrotated
        if (p.sum32 && !lean) return TEAL_ERR_CONFIG; // so does the fp32 slice-sum output (needs a prepared workspace when split-K is used)
        if (lean) {
            if (rope_taken) *rope_taken = f.a.rope != nullptr;
            // (the instantiation as rocprofv3 prints it: BF16, MODE, PAIR, LPR, KR, EXACT, PHASE, U, W8, ROPE)
            if (desc || kDiagnostics) {
                char d[160];
                snprintf(d, sizeof d, "gemv_fast_kernel<%s,%d,%s,%d,%d,%s,%s%s,%s> grid (%d,%d) x 1024",
                         dtype == TEAL_BF16 ? "true" : "false", f.mode, f.pair ? "true" : "false", f.lpr, f.kr,
                         (f.mode == 1 && f.Z == 1024 * f.kr) ? "true" : "false", (f.a.phase && !f.w8) ? "true" : "false",
                         f.w8 ? ",4,true" : ",4,false", f.a.rope ? "true" : "false", f.ntiles, f.split);
                publish_desc(d, desc, desc_bytes);
            }
            const hipError_t e = f.w8 ? (dtype == TEAL_BF16 ? launch_fast_w8_bf16(f, st) : launch_fast_w8_f16(f, st))
                                      : (dtype == TEAL_BF16 ? launch_fast_bf16(f, st) : launch_fast_f16(f, st));
            return e == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
        }
    }
    if (desc || kDiagnostics) {
        const int owned = p.krt ? p.krt : (((p.Z + 63) >> 6) + c.waves - 1) / c.waves;
        const int krt = c.waves == 16 ? (owned <= 4 ? 4 : (owned <= 8 ? 8 : 16)) : 16;
        char d[160];
        snprintf(d, sizeof d, "sparse_gemv_kernel<%d,%d,%d,%s,%d,%d,%s,%s> grid %d x %d", c.lpr, c.waves, c.unroll,
                 dtype == TEAL_BF16 ? "true" : "false", p.in.mode, krt, p.pair ? "true" : "false", p.w8 ? "true" : "false",
                 p.ntiles * p.split, c.waves * 64);
        publish_desc(d, desc, desc_bytes);
    }
    if (launch_gemv(p, dtype, lds, c, st) != hipSuccess) return TEAL_ERR_LAUNCH;
    if (c.split > 1 && !to_ws) {
        const dim3 grid((off + 255) / 256), block(256);
        if (dtype == TEAL_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<false>), grid, block, 0, st, p);
        if (hipGetLastError() != hipSuccess) return TEAL_ERR_LAUNCH;
    }
    return TEAL_OK;
}

int check_common(const void* x, const void* w, const void* y, int Z, int N, int dtype) {
    if (!x || !w || !y || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((N & 7) != 0 || Z > 65536) return TEAL_ERR_SHAPE;
    if (!aligned16(w) || (reinterpret_cast<uintptr_t>(x) & 1u) || (reinterpret_cast<uintptr_t>(y) & 1u))
        return TEAL_ERR_ALIGN;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    return TEAL_OK;
}


}  // namespace teal

using namespace teal;

extern "C" {


int teal_version(void) { return 100; }

const char* teal_strerror(int code) {
    switch (code) {
        case TEAL_OK: return "ok";
        case TEAL_ERR_ARG: return "bad argument (null pointer or non-positive size)";
        case TEAL_ERR_DTYPE: return "unsupported dtype (0 = fp16, 1 = bf16)";
        case TEAL_ERR_SHAPE: return "unsupported shape (need N % 8 == 0, Z <= 65536, segment sizes % 8 == 0)";
        case TEAL_ERR_ALIGN: return "pointer not sufficiently aligned (weights/workspace 16 B)";
        case TEAL_ERR_WORKSPACE: return "split-K workspace missing or too small (teal_workspace_bytes)";
        case TEAL_ERR_LAUNCH: return "HIP kernel launch failed";
        case TEAL_ERR_NO_DEVICE: return "no HIP device available";
        case TEAL_ERR_CONFIG: return "invalid tuning override";
        default: return "unknown error";
    }
}

int teal_init(void) {
    DeviceCtx* c = device_ctx();
    return c ? c->num_cu : TEAL_ERR_NO_DEVICE;
}

size_t teal_workspace_bytes(int Z, int N) {
    (void)Z;
    if (N <= 0) return 0;
    // the library's header (arrival counters, sampler scratch) + kMaxSplit slabs of N columns; the fused gate|up GEMV
    // uses two segments of N columns
    return kWsHeaderBytes + (size_t)kMaxSplit * (size_t)N * 2 * sizeof(float);
}

int teal_workspace_init(void* ws, size_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < kWsHeaderBytes) return TEAL_ERR_WORKSPACE;
    if (!aligned16(ws)) return TEAL_ERR_ALIGN;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    if (hipMemsetAsync(ws, 0, kWsHeaderBytes, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
        (void)hipGetLastError();
        return TEAL_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_ws_mu);
    // entries that overlap the new range are stale by construction (their memory was freed and handed out again
    // without teal_workspace_release): drop them
    const char* lo = reinterpret_cast<const char*>(ws);
    const char* hi = lo + ws_bytes;
    for (size_t i = 0; i < g_ws_reg.size();) {
        const char* elo = reinterpret_cast<const char*>(g_ws_reg[i].ws);
        if (elo < hi && lo < elo + g_ws_reg[i].bytes) g_ws_reg.erase(g_ws_reg.begin() + i);
        else ++i;
    }
    g_ws_reg.push_back({ws, ws_bytes});
    return TEAL_OK;
}

int teal_workspace_release(void* ws) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (size_t i = 0; i < g_ws_reg.size(); ++i)
        if (g_ws_reg[i].ws == ws) { g_ws_reg.erase(g_ws_reg.begin() + i); return TEAL_OK; }
    return TEAL_ERR_ARG;
}

#ifdef TEAL_DIAGNOSTICS  // exported by libteal_hip_diag.so only
int teal_set_tuning(int lanes_per_row, int waves, int split, int unroll) {
    auto in = [](int v, std::initializer_list<int> ok) {
        for (int o : ok) if (v == o) return true;
        return false;
    };
    if (!in(lanes_per_row, {0, 8, 16, 32, 64}) || !in(waves, {0, 16}) ||
        !in(unroll, {0, 4}) || split < 0 || split > kMaxSplit)
        return TEAL_ERR_CONFIG;
    g_override = {lanes_per_row, waves, split, unroll};
    return TEAL_OK;
}

const char* teal_last_launch_desc(void) { return g_last_desc; }

int teal_set_fast(int on) {
    g_fast = on ? 1 : 0;
    return TEAL_OK;
}

int teal_set_wave_local(int on) {
    g_wave_local = on ? 1 : 0;
    return TEAL_OK;
}

int teal_set_phase_buffer(void* dev_u64) {
    g_phase = reinterpret_cast<unsigned long long*>(dev_u64);
    g_phase_seq = 0;
    return TEAL_OK;
}

int teal_set_phase_stride(size_t u64_per_launch) {
    g_phase_stride = u64_per_launch;
    g_phase_seq = 0;
    return TEAL_OK;
}
#endif  // TEAL_DIAGNOSTICS

int teal_get_config(int Z, int N, int nseg, int* out) {
    if (!out || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    const Config c = pick_config(Z, N, nseg);
    out[0] = c.lpr;
    out[1] = c.waves;
    out[2] = c.split;
    out[3] = c.unroll;
    out[4] = ((N + c.lpr * 8 - 1) / (c.lpr * 8)) * c.split;
    return TEAL_OK;
}

int teal_compact(const void* x, float tau, int Z, int dtype, int32_t* idx_out, int32_t* count_out,
                 void* stream) {
    if (!x || !idx_out || !count_out || Z <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (Z > 65536) return TEAL_ERR_SHAPE;
    const int nch = (Z + 63) >> 6;
    const size_t lds = (size_t)nch * 8 + (size_t)(nch + 2) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const uint16_t* xp = reinterpret_cast<const uint16_t*>(x);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((compact_kernel<true>), dim3(1), dim3(1024), lds, st, xp, Z, tau, idx_out, count_out);
    else
        hipLaunchKernelGGL((compact_kernel<false>), dim3(1), dim3(1024), lds, st, xp, Z, tau, idx_out, count_out);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_sparse_qkv_gemv(const void* x, const void* wT, void* y, float tau_q, float tau_k,
                         float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                         size_t ws_bytes, void* stream) {
    return teal_sparse_qkv_gemv_ld(x, wT, N, y, tau_q, tau_k, tau_v, Z, N, N_q, N_kv, dtype, ws, ws_bytes, stream);
}

int teal_sparse_qkv_gemv_ld(const void* x, const void* wT, int ld, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                            size_t ws_bytes, void* stream) {
    int rc = check_common(x, wT, y, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    if (ld < N || (ld & 7)) return TEAL_ERR_SHAPE;
    if (N_q < 0 || N_kv < 0 || N_q + N_kv > N || (N_q & 7) || (N_kv & 7)) return TEAL_ERR_SHAPE;
    Params p = {};
    p.x = x;
    p.Z = Z;
    const int widths[3] = {N_q, N_kv, N - N_q - N_kv};
    const float taus[3] = {tau_q, tau_k, tau_v};
    int col = 0, ns = 0;
    for (int i = 0; i < 3; ++i) {
        if (widths[i] > 0) {
            Seg& sgm = p.seg[ns++];
            sgm.w = wT;
            sgm.y = reinterpret_cast<uint16_t*>(y) + col;
            sgm.tau = taus[i];
            sgm.ld = ld;
            sgm.col0 = col;
            sgm.ncols = widths[i];
        }
        col += widths[i];
    }
    p.nseg = ns;
    return run_gemv(p, dtype, ws, ws_bytes, false, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int teal_sparse_gemv(const void* x, const void* wT, void* y, float tau, int Z, int N, int dtype,
                     void* ws, size_t ws_bytes, void* stream) {
    return teal_sparse_qkv_gemv(x, wT, y, tau, tau, tau, Z, N, N, 0, dtype, ws, ws_bytes, stream);
}

int teal_sparse_qkv_gemv_i8(const void* x, const void* wqT, const void* scale, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int ld, int dtype, void* ws, size_t ws_bytes,
                            void* stream) {
    if (!scale || N_q <= 0 || N_kv < 0 || N_q + 2 * N_kv != N || ld < N) return TEAL_ERR_ARG;
    if ((N_q & 7) || (N_kv & 7) || (ld & 7)) return TEAL_ERR_SHAPE;
    int rc = check_common(x, wqT, y, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    teal_gemv_in_t in = {};
    in.mode = TEAL_IN_PLAIN;
    in.x = x;
    teal_gemv_out_t out = {};
    out.mode = TEAL_OUT_ROUNDED;
    out.weight_bits = 8;
    const float taus[3] = {tau_q, tau_k, tau_v};
    const int col0[3] = {0, N_q, N_q + N_kv}, ncols[3] = {N_q, N_kv, N_kv};
    out.nseg = N_kv > 0 ? 3 : 1;
    for (int i = 0; i < out.nseg; ++i) {
        out.w[i] = wqT;
        out.ld[i] = ld;
        out.col0[i] = col0[i];
        out.ncols[i] = ncols[i];
        out.tau[i] = taus[i];
        out.y[i] = reinterpret_cast<uint16_t*>(y) + col0[i];
        out.scale[i] = reinterpret_cast<const uint16_t*>(scale) + col0[i];
    }
    return teal_fused_gemv(&in, &out, Z, dtype, ws, ws_bytes, nullptr, stream);
}

int teal_dense_gemv(const void* x, const void* wT, void* y, int Z, int N, int dtype, void* ws,
                    size_t ws_bytes, void* stream) {
    // |x| > -inf keeps every finite and infinite activation; NaN propagates via nan_keeps.
    return teal_sparse_gemv(x, wT, y, -INFINITY, Z, N, dtype, ws, ws_bytes, stream);
}

int teal_sparse_gateup_silu(const void* x, const void* w1T, const void* w3T, void* h, float tau_gate,
                            float tau_up, int Z, int N, int dtype, void* ws, size_t ws_bytes,
                            void* stream) {
    int rc = check_common(x, w1T, h, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    if (!w3T) return TEAL_ERR_ARG;
    if (!aligned16(w3T)) return TEAL_ERR_ALIGN;
    Params p = {};
    p.x = x;
    p.Z = Z;
    p.nseg = 2;
    p.seg[0].w = w1T; p.seg[0].y = nullptr; p.seg[0].tau = tau_gate; p.seg[0].ld = N; p.seg[0].col0 = 0; p.seg[0].ncols = N;
    p.seg[1].w = w3T; p.seg[1].y = nullptr; p.seg[1].tau = tau_up;   p.seg[1].ld = N; p.seg[1].col0 = 0; p.seg[1].ncols = N;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    rc = run_gemv(p, dtype, ws, ws_bytes, true, st, nullptr);
    if (rc != TEAL_OK) return rc;
    const dim3 grid((N + 255) / 256), block(256);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((gateup_silu_epilogue_kernel<true>), grid, block, 0, st, p, h);
    else
        hipLaunchKernelGGL((gateup_silu_epilogue_kernel<false>), grid, block, 0, st, p, h);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_fused_gemv(const teal_gemv_in_t* in, const teal_gemv_out_t* out, int Z, int dtype, void* ws,
                    size_t ws_bytes, int* nslabs_out, void* stream) {
    if (!in || !out || Z <= 0 || out->nseg < 1 || out->nseg > kMaxSeg) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (Z > 65536) return TEAL_ERR_SHAPE;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    if (out->act_seg0 && out->mode != TEAL_OUT_ROUNDED) return TEAL_ERR_ARG;
    if (out->weight_bits == 4) {
        if (out->act_seg0 || in->gate_activated || out->mode == TEAL_OUT_QKV_ROPE || out->mode == TEAL_OUT_SLAB_SUM) return TEAL_ERR_ARG;  // 16-bit / int8 launches only
        return fused_gemv_i4(in, out, Z, dtype, ws, ws_bytes, nslabs_out, reinterpret_cast<hipStream_t>(stream));
    }
    Params p = {};
    p.Z = Z;
    p.in.mode = in->mode;
    switch (in->mode) {
        case TEAL_IN_PLAIN:
        case TEAL_IN_SILU_MUL:
            if (!in->x) return TEAL_ERR_ARG;
            p.x = in->x;
            p.in.gate_act = (in->mode == TEAL_IN_SILU_MUL && in->gate_activated) ? 1 : 0;
            break;
        case TEAL_IN_ATTN_MERGE:
            if (!in->x || (in->att_head_dim != 64 && in->att_head_dim != 128) || Z % in->att_head_dim ||
                (in->att_nsplit != 0 && in->att_nsplit != 4 && in->att_nsplit != 8))
                return TEAL_ERR_ARG;
            p.in.att_ns = in->att_nsplit ? in->att_nsplit : 4;
            if (Z > (64 / p.in.att_ns) * 1024) return TEAL_ERR_SHAPE;  // one lane per (chunk, split)
            p.x = in->x;
            p.in.att = reinterpret_cast<const float*>(in->x);
            p.in.att_hd = in->att_head_dim;
            break;
        case TEAL_IN_MASKED:
            if (!in->x || !in->masks) return TEAL_ERR_ARG;
            p.x = in->x;
            p.in.masks = reinterpret_cast<const unsigned long long*>(in->masks);
            break;
        case TEAL_IN_RESID_NORM:
            if (!in->resid_in || !in->norm_weight || in->nslabs < 0 || (in->nslabs > 0 && !in->slabs)) return TEAL_ERR_ARG;
            if (in->resid_out == in->resid_in && !in->row_index) return TEAL_ERR_ARG;  // must ping-pong
            p.x = in->resid_in;
            p.in.resid_in = in->resid_in;
            p.in.row_index = in->row_index;
            p.in.slabs = in->slabs;
            p.in.nslabs = in->nslabs;
            p.in.slabs_il = in->slabs_interleaved ? 1 : 0;
            if (p.in.slabs_il && in->nslabs > 8) return TEAL_ERR_ARG;
            p.in.norm_w = in->norm_weight;
            p.in.resid_out = in->resid_out;
            p.in.eps = in->eps;
            break;
        default: return TEAL_ERR_ARG;
    }
    p.nseg = out->nseg;
    for (int i = 0; i < out->nseg; ++i) {
        if (!out->w[i] || out->ncols[i] <= 0 || (out->ncols[i] & 7) || (out->col0[i] & 7) || (out->ld[i] & 7)) return TEAL_ERR_SHAPE;
        if (!aligned16(out->w[i])) return TEAL_ERR_ALIGN;
        if (out->mode == TEAL_OUT_ROUNDED && !out->y[i]) return TEAL_ERR_ARG;
        p.seg[i].w = out->w[i];
        p.seg[i].y = out->y[i];
        p.seg[i].tau = out->tau[i];
        p.seg[i].ld = out->ld[i];
        p.seg[i].col0 = out->col0[i];
        p.seg[i].ncols = out->ncols[i];
        p.seg[i].scale = out->scale[i];
    }
    if (out->weight_bits != 0 && out->weight_bits != 16 && out->weight_bits != 8) return TEAL_ERR_ARG;
    p.w8 = out->weight_bits == 8 ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Config used = {};
    int rc;
    if (out->mode == TEAL_OUT_PAIR_SILU) {
        // seg 0 = gate (w1), seg 1 = up (w3), same shape; y[0] receives h = silu(gate) * up
        if (out->nseg != 2 || in->mode != TEAL_IN_RESID_NORM || out->ncols[0] != out->ncols[1] || !out->y[0]) return TEAL_ERR_ARG;
        p.pair = 1;
        p.mask_out = reinterpret_cast<unsigned long long*>(out->mask_out);
        p.mask_tau = out->mask_tau;
        rc = run_gemv(p, dtype, ws, ws_bytes, false, st, &used, false, false, true, nullptr, out->desc, out->desc_bytes);
    } else if (out->mode == TEAL_OUT_SLABS) {
        if (!out->slabs) return TEAL_ERR_ARG;
        if (ws_prepared(out->slabs, out->slabs_bytes)) return TEAL_ERR_ARG;  // a prepared workspace starts with the library's header
        rc = run_gemv(p, dtype, out->slabs, out->slabs_bytes, true, st, &used, true, out->slabs_interleaved != 0, false, nullptr, out->desc,
                      out->desc_bytes);
        if (rc == TEAL_OK && out->slabs_interleaved && !p.ws_il) return TEAL_ERR_CONFIG;  // > 8 slices cannot interleave
    } else if (out->mode == TEAL_OUT_SLAB_SUM) {
        // one fp32 vector [ncols]: the row slices' partial sums folded in slice order by the last slice of each tile to arrive
        // (arrival tickets in the caller's prepared workspace), NOT rounded — what tensor-parallel ranks all-reduce
        if (out->nseg != 1 || !out->slabs || out->slabs_bytes < (size_t)out->ncols[0] * sizeof(float)) return TEAL_ERR_ARG;
        if (ws_prepared(out->slabs, out->slabs_bytes)) return TEAL_ERR_ARG;
        p.sum32 = 1;
        p.seg[0].y = out->slabs;
        rc = run_gemv(p, dtype, ws, ws_bytes, false, st, &used, true, false, true, nullptr, out->desc, out->desc_bytes);
        if (rc == TEAL_OK && nslabs_out) { *nslabs_out = 1; return rc; }
    } else if (out->mode == TEAL_OUT_ROUNDED) {
        p.act0 = out->act_seg0 ? 1 : 0;
        rc = run_gemv(p, dtype, ws, ws_bytes, false, st, &used, false, false, true, nullptr, out->desc, out->desc_bytes);
    } else if (out->mode == TEAL_OUT_QKV_ROPE) {
        if (out->nseg != 3 || in->mode != TEAL_IN_RESID_NORM || !out->y[0] || !out->rope || !out->rope_pos || !out->k_cache ||
            !out->v_cache || out->rope_max_seq <= 0 || !nslabs_out)
            return TEAL_ERR_ARG;
        // the fused epilogue exists in the lean kernel for a launch without split-K; decide before launching
        int total_cols = 0;
        for (int i = 0; i < 3; ++i) total_cols += out->ncols[i];
        Params q = p;
        q.rope = reinterpret_cast<const uint16_t*>(out->rope);
        q.rope_pos = out->rope_pos;
        q.kc = reinterpret_cast<uint16_t*>(out->k_cache);
        q.vc = reinterpret_cast<uint16_t*>(out->v_cache);
        q.rope_hd = out->rope_head_dim;
        q.rope_max_seq = out->rope_max_seq;
        q.seg[1].y = reinterpret_cast<uint16_t*>(out->y[0]) + out->ncols[0];                   // (k, v land in the caches; the
        q.seg[2].y = reinterpret_cast<uint16_t*>(out->y[0]) + out->ncols[0] + out->ncols[1];   //  lean kernel wants one vector)
        const Config c0 = pick_config(Z, total_cols, 3);
        bool fused = false;
        // (a 70B-class projection hands over row-sliced slabs of 128-column tiles instead — run_gemv's wide_sliced geometry,
        //  measured faster there, profiles/r04_layer_experiments.txt)
        const bool sliced = (size_t)Z * total_cols >= (size_t)8192 * 8192;
        if (c0.split == 1 && !sliced && g_fast && !g_override.split && !g_override.lpr) {
            bool taken = false;
            rc = run_gemv(q, dtype, ws, ws_bytes, false, st, &used, false, false, true, &taken, out->desc, out->desc_bytes);
            if (rc != TEAL_OK && rc != TEAL_ERR_CONFIG) return rc;  // TEAL_ERR_CONFIG: not a lean-kernel shape, nothing was launched
            fused = rc == TEAL_OK && taken;
            if (rc == TEAL_OK && !fused) return TEAL_ERR_LAUNCH;    // (unreachable: run_gemv refuses to launch a rope request unrotated)
        }
        if (fused) { *nslabs_out = 0; return TEAL_OK; }
        if (!out->slabs) return TEAL_ERR_ARG;
        if (ws_prepared(out->slabs, out->slabs_bytes)) return TEAL_ERR_ARG;
        rc = run_gemv(p, dtype, out->slabs, out->slabs_bytes, true, st, &used, true, out->slabs_interleaved != 0, false, nullptr, out->desc,
                      out->desc_bytes);
        if (rc == TEAL_OK && out->slabs_interleaved && !p.ws_il) return TEAL_ERR_CONFIG;
    } else {
        return TEAL_ERR_ARG;
    }
    if (rc == TEAL_OK && nslabs_out) *nslabs_out = used.split;
    return rc;
}


}  // extern "C"
