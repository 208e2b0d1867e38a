// csrc/avx_internal.h -- shared between the translation units of libavx.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/avx.h"

struct avx_ws {
    hipStream_t stream = nullptr;
    bool used = false;
    uint32_t* d_flags = nullptr;     // per-frame "any byte > 1" flags
    size_t flags_cap = 0;
    float* d_row_gain = nullptr;     // per-row gains (AVX_POST_ROWGAIN)
    size_t row_gain_cap = 0;
    void* h_row_tab = nullptr;       // host mirror of what d_row_gain holds: an unchanged table is not uploaded again
    size_t h_row_tab_bytes = 0, h_row_tab_cap = 0;
    // Small constant tables of a caller's pipeline (matrices, weights, coordinate rows): device copy + host mirror per slot,
    // uploaded only when the bytes change (avx_const_upload).  Slots: 0 UV matrix / weights, 1 band-major spectral weights (k_spectral_nhwc_h), 2-7 mantis tables,
    // 8-9 label stroke segments (labels.hip).
    struct const_slot { void* dev = nullptr; void* host = nullptr; size_t bytes = 0, cap = 0; };
    const_slot consts[10];
    void* uv_small = nullptr;        // UV path: partial statistics, histogram, select state, percentiles
    void* d_scratch = nullptr;       // scratch arena (UV path planes)
    size_t scratch_cap = 0;
    void* d_geom = nullptr;          // resampling coefficient tables (geom.hip)
    size_t geom_cap = 0;
    struct tab { uint64_t key; void* dev; };
    tab geom_tabs[64] = {};          // immutable resampling tables cached by (kind, component, source size, target size)
    int n_geom_tabs = 0;
    void* bee_small = nullptr;       // honeybee tail, recompute form (uv.hip): per-frame partials, statistics, histograms, select states, percentiles
    float* d_scan = nullptr;         // mantis.hip: the blurred scanline plane of the current (H, W, rows, taps) -- frame-independent, kept across frames
    size_t scan_cap = 0;
    uint64_t scan_key = 0;
    float* d_periph = nullptr;       // mantis.hip: the periphery blend weight 1 / (1 + exp(-softness (r - radius))) of the current (H, W, xx, yy, softness, radius): frame-independent too
    size_t periph_cap = 0;
    uint64_t periph_key = 0;
    void* d_ew = nullptr;            // per-block partial reductions of elementwise programs (ew.hip)
    size_t ew_cap = 0;
};

struct avx_ctx {
    int device = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    char err[512] = {0};
    // constant tables (reference outputs, csrc/srgb_tables.h), resident for the ctx lifetime
    float* d_decode_lut = nullptr;   // 256
    void* d_zero = nullptr;          // 256 zero bytes: where the LDS-direct loads of padding pixels point (mst_fused.hip)
    float* d_enc_thr_f32 = nullptr;  // 255 (+1 pad)
    double* d_enc_thr_f64 = nullptr; // 255 (+1 pad)
    // bucketed quantiser (csrc/dichromat.hip quantize_coarse): per top-bits bucket, #thresholds <= bucket start
    uint8_t* d_coarse_f32 = nullptr;
    uint8_t* d_coarse_f64 = nullptr;
    uint32_t coarse_lo_key[2] = {0, 0};  // [0] f32 (bits >> 16), [1] f64 (bits >> 45): dichromat_common.h
    uint32_t coarse_n_keys[2] = {0, 0};
    int coarse_n_fix[2] = {0, 0};
    // Per-stream workspaces: launches on different streams may be in flight together (pipeline.py keeps
    // `depth` frames in flight), so anything a launch writes and a later kernel of the same launch reads
    // is keyed by the stream it was enqueued on.
    static constexpr int kMaxWorkspaces = 32;
    avx_ws ws[kMaxWorkspaces];
    int n_ws = 0;
    int num_cus = 256;
    // Measured launch geometry of the marching kernels (dichromat_march.hip): rows-per-workgroup split that was
    // fastest for (kernel configuration, batch, frame size), found by timing the candidates on the first call.
    struct tuned { uint64_t key; int nchunks; };
    tuned march_tuned[64] = {};
    int n_march_tuned = 0;
    // Frame lanes (uv.hip): the UV tail of a frame is ~10 dependent launches of 7-25 us each, none of which fills the GPU at
    // 1080p; independent frames of a batch are therefore enqueued round-robin on these internal streams (forked from and
    // joined back into the caller's stream with events), each with its own workspace, so their kernels overlap.
    static constexpr int kMaxLanes = 8;
    hipStream_t lanes[kMaxLanes] = {};
    hipEvent_t lane_done[kMaxLanes] = {};
    hipEvent_t lane_fork = nullptr;
    static constexpr int kWaitEvents = 16;   // avx_stream_wait: a ring of timing-less events
    hipEvent_t wait_ev[kWaitEvents] = {};
    int wait_next = 0;
    int n_lanes = 0;
};

int avx_fail(avx_ctx* ctx, int code, const char* fmt, ...);
void avx_march_seed_tuned(avx_ctx* ctx);  // dichromat_march.hip: measured launch geometries of the standard workloads
avx_ws* avx_workspace(avx_ctx* ctx, hipStream_t stream);  // find or create; NULL when all slots are taken
int avx_ensure_scratch(avx_ctx* ctx, avx_ws* ws, size_t bytes);
// Per-row table (row gains / streak taps) -> ws->d_row_gain: grows the buffer, uploads only when the bytes differ from the last upload.
int avx_upload_row_table(avx_ctx* ctx, avx_ws* ws, const void* host, size_t bytes, hipStream_t s);
int avx_const_upload(avx_ctx* ctx, avx_ws* ws, int slot, const void* host, size_t bytes, hipStream_t s, void** dev_out);
int avx_lanes(avx_ctx* ctx, int want);  // creates the frame-lane streams on first use; returns how many exist (<= want), 0 on failure
// mst_mfma.hip: reduce the per-workgroup Gram partials ([blocks][heads][34][32]) of a Gram pass into gram / nq / nk (k_mst_qkv_final)
int avx_mst_qkv_final_launch(avx_ctx* ctx, const float* partial, int blocks, int heads, float* gram, float* nq, float* nk, hipStream_t s);

#define AVX_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return avx_fail((ctx), AVX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                    \
    } while (0)

#define AVX_REQUIRE(ctx, cond, ...)                                  \
    do {                                                             \
        if (!(cond)) return avx_fail((ctx), AVX_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// `stream` is a hipStream_t; NULL is HIP's null stream, exactly as in HIP (torch's default stream is 0).
static inline hipStream_t avx_pick_stream(avx_ctx* ctx, void* stream) {
    (void)ctx;
    return (hipStream_t)stream;
}
