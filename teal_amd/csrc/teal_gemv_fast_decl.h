// teal_gemv_fast_decl.h — argument blocks of the lean sparse GEMV kernel (teal_gemv_fast.h) shared with the host side.
#pragma once
#include "teal_common.h"

namespace teal {

struct FastArgs {
    const void* w0;                 // weight image [Z][ld0]
    const void* w1;                 // PAIR: the up projection's image [Z][ld1]
    void* y;                        // rounded output [ncols] (PAIR: h = silu(gate) * up)
    float* ws;                      // slab output, interleaved [ncols][ws_stride]
    unsigned long long* mask_out;   // PAIR: keep masks of h vs mask_tau (or null)
    void* resid_out;                // MODE 1: updated residual stream, written by workgroup (0, 0) (or null)
    unsigned long long* phase;      // PHASE instantiations: stamp buffer
    unsigned* ticket;               // split > 1 with a rounded output: per-tile arrival counters (zero between launches);
                                    // the last slice of a tile to arrive sums the slabs in slice order and stores y
    int ld0, ld1;                   // row strides in elements
    int seg_tile1, seg_tile2;       // first tile of threshold segments 1 and 2 (INT_MAX when absent)
    float tau0, tau1, tau2;         // thresholds (PAIR: tau0 = gate, tau1 = up)
    float mask_tau;
    int ws_stride;                  // (split + 3) & ~3, or 0: round and store y (split == 1)
    int att_hd, att_ns;             // MODE 4
    int cap;                        // list entries one wave can own
    const uint16_t* scale0;         // int8 weights: per-column scales of image 0 / image 1 (element 0 = first column of the image)
    const uint16_t* scale1;
    int w1_tile;                    // not PAIR: first tile that streams the second image w1 / ld1 (INT_MAX: one image)
    // ROPE instantiations (TEAL_OUT_QKV_ROPE: the fused wqkv projection with split == 1): RoPE of q and of the new k row and
    // the KV-cache append happen in the epilogue (gpt-fast/model.py:170-178), so the attention launch starts from finished rows
    const uint16_t* rope;           // (cos, sin) table [max_seq][head_dim / 2][2]
    const int* rope_pos;            // device int32: position of the token being decoded
    uint16_t* kc;                   // K cache [n_kv_head][max_seq][head_dim]
    uint16_t* vc;                   // V cache
    int rope_hd, rope_dim, rope_kv, rope_max_seq;  // head_dim, n_head * head_dim, n_kv_head * head_dim, cache rows
    int act0;                       // rounded output of threshold segment 0 goes through silu (the gate of gate | up)
    int gate_act;                   // MODE 2: the gate half already holds round(silu(gate))
    int sum32;                      // TEAL_OUT_SLAB_SUM: y is fp32 [ncols] and receives the unrounded sum over the row slices (slice order)
};

// Launch description filled by run_gemv when the shape qualifies (teal_kernels.hip: fast_eligible)
struct FastLaunch {
    const void* in0; const void* in1; const void* in2; const int* row_index;
    int Z, nslabs; float eps;
    FastArgs a;
    int mode, pair, lpr, kr, ntiles, split;
    int w8;  // int8 weight image(s): the W8 instantiations (teal_gemv_fast_w8_*.hip)
    size_t lds;
};

hipError_t launch_fast_f16(const FastLaunch& f, hipStream_t st);   // teal_gemv_fast_f16.hip
hipError_t launch_fast_bf16(const FastLaunch& f, hipStream_t st);  // teal_gemv_fast_bf16.hip
hipError_t launch_fast_w8_f16(const FastLaunch& f, hipStream_t st);   // teal_gemv_fast_w8_f16.hip
hipError_t launch_fast_w8_bf16(const FastLaunch& f, hipStream_t st);  // teal_gemv_fast_w8_bf16.hip

}  // namespace teal
