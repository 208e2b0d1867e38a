// csrc/avx_core.hip -- context, device memory, streams, HIP-event stopwatch of libavx.so.
// Nothing here exists in the reference (pure Python, no device runtime: SURVEY.md 2.1); it is the
// plumbing under the C ABI of include/avx.h.
#include <cstdlib>
#include "avx_internal.h"
#include "dichromat_common.h"
#include "srgb_tables.h"

static char g_init_err[512] = "";

int avx_fail(avx_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

avx_ws* avx_workspace(avx_ctx* ctx, hipStream_t stream) {
    for (int i = 0; i < ctx->n_ws; ++i)
        if (ctx->ws[i].used && ctx->ws[i].stream == stream) return &ctx->ws[i];
    if (ctx->n_ws >= avx_ctx::kMaxWorkspaces) {
        avx_fail(ctx, AVX_ERR_NOMEM, "more than %d distinct streams used with one context", avx_ctx::kMaxWorkspaces);
        return nullptr;
    }
    avx_ws* w = &ctx->ws[ctx->n_ws++];
    w->stream = stream;
    w->used = true;
    return w;
}

static void avx_ws_release(avx_ws* w) {
    if (w->d_flags) (void)hipFree(w->d_flags);
    if (w->d_row_gain) (void)hipFree(w->d_row_gain);
    if (w->h_row_tab) free(w->h_row_tab);
    for (auto& c : w->consts) { if (c.dev) (void)hipFree(c.dev); if (c.host) free(c.host); }
    if (w->uv_small) (void)hipFree(w->uv_small);
    if (w->d_scratch) (void)hipFree(w->d_scratch);
    if (w->d_geom) (void)hipFree(w->d_geom);
    if (w->d_ew) (void)hipFree(w->d_ew);
    if (w->d_scan) (void)hipFree(w->d_scan);
    if (w->d_periph) (void)hipFree(w->d_periph);
    if (w->bee_small) (void)hipFree(w->bee_small);
    for (int i = 0; i < w->n_geom_tabs; ++i)
        if (((w->geom_tabs[i].key >> 54) & 0xf) < 8) (void)hipFree(w->geom_tabs[i].dev);  // component >= 8: a scalar, not a pointer
    *w = avx_ws();
}

// A species' per-row table is the same for every batch of a video, and a pageable host-to-device copy in front of every launch
// is a staging copy plus a DMA the kernel has to wait for (streak species, 207 KB per launch at 1080p): the workspace remembers
// the bytes it uploaded last and skips the copy when they have not changed.
int avx_upload_row_table(avx_ctx* ctx, avx_ws* ws, const void* host, size_t bytes, hipStream_t s) {
    if (bytes > ws->row_gain_cap * sizeof(float)) {
        if (ws->d_row_gain) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(ws->d_row_gain)); }
        ws->d_row_gain = nullptr;
        ws->row_gain_cap = 0;
        ws->h_row_tab_bytes = 0;
        AVX_HIP(ctx, hipMalloc((void**)&ws->d_row_gain, bytes));
        ws->row_gain_cap = bytes / sizeof(float);
    }
    if (ws->h_row_tab && ws->h_row_tab_bytes == bytes && memcmp(ws->h_row_tab, host, bytes) == 0) return AVX_OK;
    if (bytes > ws->h_row_tab_cap) {
        void* p = realloc(ws->h_row_tab, bytes);
        if (!p) return avx_fail(ctx, AVX_ERR_NOMEM, "row table mirror: out of host memory");
        ws->h_row_tab = p;
        ws->h_row_tab_cap = bytes;
    }
    memcpy(ws->h_row_tab, host, bytes);
    ws->h_row_tab_bytes = bytes;
    AVX_HIP(ctx, hipMemcpyAsync(ws->d_row_gain, ws->h_row_tab, bytes, hipMemcpyHostToDevice, s));
    return AVX_OK;
}

// Same idea for the small constant tables that used to be copied from pageable host memory in front of every frame (six per
// mantis frame): each slot owns its device buffer, so nothing else can overwrite what the mirror says is there.
int avx_const_upload(avx_ctx* ctx, avx_ws* ws, int slot, const void* host, size_t bytes, hipStream_t s, void** dev_out) {
    AVX_REQUIRE(ctx, slot >= 0 && slot < 10 && host && bytes > 0, "avx_const_upload: bad arguments");
    avx_ws::const_slot& c = ws->consts[slot];
    if (bytes > c.cap) {
        if (c.dev) { AVX_HIP(ctx, hipStreamSynchronize(s)); AVX_HIP(ctx, hipFree(c.dev)); }
        c.dev = nullptr;
        c.cap = 0;
        c.bytes = 0;
        const size_t cap = (bytes + 255) & ~(size_t)255;
        void* h = realloc(c.host, cap);
        if (!h) return avx_fail(ctx, AVX_ERR_NOMEM, "constant table mirror: out of host memory");
        c.host = h;
        AVX_HIP(ctx, hipMalloc(&c.dev, cap));
        c.cap = cap;
    }
    if (c.bytes != bytes || memcmp(c.host, host, bytes) != 0) {
        memcpy(c.host, host, bytes);
        c.bytes = bytes;
        AVX_HIP(ctx, hipMemcpyAsync(c.dev, c.host, bytes, hipMemcpyHostToDevice, s));
    }
    *dev_out = c.dev;
    return AVX_OK;
}

int avx_lanes(avx_ctx* ctx, int want) {
    if (want > avx_ctx::kMaxLanes) want = avx_ctx::kMaxLanes;
    if (!ctx->lane_fork && hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming) != hipSuccess) return 0;
    while (ctx->n_lanes < want) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(st); break; }
        ctx->lanes[ctx->n_lanes] = st;
        ctx->lane_done[ctx->n_lanes] = ev;
        ++ctx->n_lanes;
    }
    return ctx->n_lanes < want ? ctx->n_lanes : want;
}

int avx_ensure_scratch(avx_ctx* ctx, avx_ws* ws, size_t bytes) {
    if (bytes <= ws->scratch_cap) return AVX_OK;
    if (ws->d_scratch) {
        AVX_HIP(ctx, hipStreamSynchronize(ws->stream));  // launches still reading the old arena
        AVX_HIP(ctx, hipFree(ws->d_scratch));
    }
    ws->d_scratch = nullptr;
    ws->scratch_cap = 0;
    size_t cap = bytes + bytes / 4;
    AVX_HIP(ctx, hipMalloc(&ws->d_scratch, cap));
    ws->scratch_cap = cap;
    return AVX_OK;
}

extern "C" {

int avx_abi_version(void) { return AVX_ABI_VERSION; }

int avx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* avx_last_error(const avx_ctx* ctx) { return ctx ? ctx->err : g_init_err; }

int avx_init(int device, avx_ctx** out_ctx) {
    if (!out_ctx) return avx_fail(nullptr, AVX_ERR_INVALID, "avx_init: out_ctx is NULL");
    *out_ctx = nullptr;
    int n = avx_device_count();
    if (n <= 0) return avx_fail(nullptr, AVX_ERR_NO_DEVICE, "avx_init: no HIP device visible (libavx has no CPU path)");
    if (device < 0 || device >= n)
        return avx_fail(nullptr, AVX_ERR_NO_DEVICE, "avx_init: device %d out of range (0..%d)", device, n - 1);
    avx_ctx* ctx = new avx_ctx();
    ctx->device = device;
#define INIT_HIP(call)                                                                                      \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            avx_fail(nullptr, AVX_ERR_HIP, "avx_init: %s failed: %s", #call, hipGetErrorString(e__));      \
            delete ctx;                                                                                     \
            return AVX_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)
    INIT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    INIT_HIP(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    INIT_HIP(hipEventCreate(&ctx->t0));
    INIT_HIP(hipEventCreate(&ctx->t1));
    INIT_HIP(hipMalloc((void**)&ctx->d_decode_lut, 256 * sizeof(float)));
    INIT_HIP(hipMalloc(&ctx->d_zero, 256));
    INIT_HIP(hipMemset(ctx->d_zero, 0, 256));
    INIT_HIP(hipMalloc((void**)&ctx->d_enc_thr_f32, 256 * sizeof(float)));
    INIT_HIP(hipMalloc((void**)&ctx->d_enc_thr_f64, 256 * sizeof(double)));
    INIT_HIP(hipMemcpy(ctx->d_decode_lut, kDecodeLutBits, 256 * 4, hipMemcpyHostToDevice));
    INIT_HIP(hipMemset(ctx->d_enc_thr_f32, 0x7f, 256 * 4));  // pad entry: huge, never <= x
    INIT_HIP(hipMemset(ctx->d_enc_thr_f64, 0x7f, 256 * 8));
    INIT_HIP(hipMemcpy(ctx->d_enc_thr_f32, kEncThrF32Bits, 255 * 4, hipMemcpyHostToDevice));
    INIT_HIP(hipMemcpy(ctx->d_enc_thr_f64, kEncThrF64Bits, 255 * 8, hipMemcpyHostToDevice));
    // Bucket tables for the quantiser: pure bit logic on the threshold tables above.
    for (int which = 0; which < 2; ++which) {
        uint64_t tb[255];
        const int shift = which == 0 ? avxk::kCoarseShiftF32 : avxk::kCoarseShiftF64;
        for (int i = 0; i < 255; ++i) tb[i] = which == 0 ? (uint64_t)kEncThrF32Bits[i] : kEncThrF64Bits[i];
        const uint64_t one_bits = which == 0 ? 0x3f800000ull : 0x3ff0000000000000ull;
        const uint32_t lo = (uint32_t)(tb[0] >> shift), hi = (uint32_t)(one_bits >> shift);
        const uint32_t n = hi - lo + 2;  // + the leading "below the first bucket" entry
        if (n > (uint32_t)avxk::kCoarseTableBytes) {
            avx_fail(nullptr, AVX_ERR_INVALID, "avx_init: quantiser bucket table too large (%u)", n);
            delete ctx;
            return AVX_ERR_INVALID;
        }
        uint8_t table[avxk::kCoarseTableBytes];
        int n_fix = 0;
        table[0] = 0;
        for (uint32_t k = 0; k + 1 < n; ++k) {
            // non-negative IEEE floats order like their bit patterns
            const uint64_t start = (uint64_t)(lo + k) << shift, next = (uint64_t)(lo + k + 1) << shift;
            int le = 0, inside = 0;
            for (int i = 0; i < 255; ++i) {
                if (tb[i] <= start) ++le;
                else if (tb[i] < next) ++inside;
            }
            table[k + 1] = (uint8_t)le;
            if (inside > n_fix) n_fix = inside;
        }
        uint8_t** dst = which == 0 ? &ctx->d_coarse_f32 : &ctx->d_coarse_f64;
        INIT_HIP(hipMalloc((void**)dst, avxk::kCoarseTableBytes));
        INIT_HIP(hipMemset(*dst, 0, avxk::kCoarseTableBytes));
        INIT_HIP(hipMemcpy(*dst, table, n, hipMemcpyHostToDevice));
        ctx->coarse_lo_key[which] = lo - 1;  // index = max(key, lo - 1) - (lo - 1)
        ctx->coarse_n_keys[which] = n;
        ctx->coarse_n_fix[which] = n_fix;
    }
#undef INIT_HIP
    avx_march_seed_tuned(ctx);
    *out_ctx = ctx;
    return AVX_OK;
}

void avx_destroy(avx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_decode_lut) (void)hipFree(ctx->d_decode_lut);
    if (ctx->d_zero) (void)hipFree(ctx->d_zero);
    if (ctx->d_enc_thr_f32) (void)hipFree(ctx->d_enc_thr_f32);
    if (ctx->d_enc_thr_f64) (void)hipFree(ctx->d_enc_thr_f64);
    if (ctx->d_coarse_f32) (void)hipFree(ctx->d_coarse_f32);
    if (ctx->d_coarse_f64) (void)hipFree(ctx->d_coarse_f64);
    for (int i = 0; i < ctx->n_ws; ++i) avx_ws_release(&ctx->ws[i]);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    for (auto& e : ctx->wait_ev) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < ctx->n_lanes; ++i) {
        (void)hipStreamDestroy(ctx->lanes[i]);
        (void)hipEventDestroy(ctx->lane_done[i]);
    }
    if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
    delete ctx;
}

int avx_malloc(avx_ctx* ctx, size_t bytes, void** out_dptr) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, out_dptr != nullptr, "avx_malloc: out_dptr is NULL");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out_dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return avx_fail(ctx, AVX_ERR_NOMEM, "avx_malloc: out of device memory (%zu bytes)", bytes);
    AVX_HIP(ctx, e);
    return AVX_OK;
}

int avx_free(avx_ctx* ctx, void* dptr) {
    if (!ctx) return AVX_ERR_INVALID;
    if (dptr) AVX_HIP(ctx, hipFree(dptr));
    return AVX_OK;
}

int avx_host_alloc(avx_ctx* ctx, size_t bytes, void** out_hptr) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, out_hptr != nullptr, "avx_host_alloc: out_hptr is NULL");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    AVX_HIP(ctx, hipHostMalloc(out_hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return AVX_OK;
}

int avx_host_free(avx_ctx* ctx, void* hptr) {
    if (!ctx) return AVX_ERR_INVALID;
    if (hptr) AVX_HIP(ctx, hipHostFree(hptr));
    return AVX_OK;
}

int avx_memcpy_h2d(avx_ctx* ctx, void* dst, const void* src_host, size_t bytes, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, dst && src_host, "avx_memcpy_h2d: NULL pointer");
    AVX_HIP(ctx, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, avx_pick_stream(ctx, stream)));
    return AVX_OK;
}

int avx_memcpy_d2h(avx_ctx* ctx, void* dst_host, const void* src, size_t bytes, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, dst_host && src, "avx_memcpy_d2h: NULL pointer");
    AVX_HIP(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, avx_pick_stream(ctx, stream)));
    return AVX_OK;
}

int avx_memset(avx_ctx* ctx, void* dst, int value, size_t bytes, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, dst != nullptr, "avx_memset: NULL pointer");
    AVX_HIP(ctx, hipMemsetAsync(dst, value, bytes, avx_pick_stream(ctx, stream)));
    return AVX_OK;
}

int avx_stream_create(avx_ctx* ctx, void** out_stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, out_stream != nullptr, "avx_stream_create: out_stream is NULL");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s;
    AVX_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = (void*)s;
    return AVX_OK;
}

int avx_stream_destroy(avx_ctx* ctx, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    if (stream) {
        AVX_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
        for (int i = 0; i < ctx->n_ws; ++i)
            if (ctx->ws[i].used && ctx->ws[i].stream == (hipStream_t)stream) {  // free its workspace, compact the table
                avx_ws_release(&ctx->ws[i]);
                ctx->ws[i] = ctx->ws[ctx->n_ws - 1];
                ctx->ws[ctx->n_ws - 1] = avx_ws();
                --ctx->n_ws;
                break;
            }
        AVX_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    }
    return AVX_OK;
}

int avx_sync(avx_ctx* ctx, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    AVX_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return AVX_OK;
}

// `waiter` does not run past this point until everything enqueued on `signaler` so far has finished (no host wait): the
// fork / join of frames that run on several streams of one context (bench.py's UV-species steps, pipeline slots).
int avx_stream_wait(avx_ctx* ctx, void* waiter, void* signaler) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t& ev = ctx->wait_ev[ctx->wait_next];
    ctx->wait_next = (ctx->wait_next + 1) % avx_ctx::kWaitEvents;
    if (!ev) AVX_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    AVX_HIP(ctx, hipEventRecord(ev, avx_pick_stream(ctx, signaler)));
    AVX_HIP(ctx, hipStreamWaitEvent(avx_pick_stream(ctx, waiter), ev, 0));
    return AVX_OK;
}

int avx_device_sync(avx_ctx* ctx) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    AVX_HIP(ctx, hipDeviceSynchronize());
    return AVX_OK;
}

int avx_timer_start(avx_ctx* ctx, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_HIP(ctx, hipEventRecord(ctx->t0, avx_pick_stream(ctx, stream)));
    return AVX_OK;
}

int avx_timer_stop(avx_ctx* ctx, void* stream, float* out_ms) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, out_ms != nullptr, "avx_timer_stop: out_ms is NULL");
    AVX_HIP(ctx, hipEventRecord(ctx->t1, avx_pick_stream(ctx, stream)));
    AVX_HIP(ctx, hipEventSynchronize(ctx->t1));
    AVX_HIP(ctx, hipEventElapsedTime(out_ms, ctx->t0, ctx->t1));
    return AVX_OK;
}

int avx_get_table(int which, void* dst_host, size_t capacity) {
    const void* src = nullptr;
    size_t size = 0;
    switch (which) {
        case 0: src = kDecodeLutBits; size = sizeof(kDecodeLutBits); break;
        case 1: src = kEncThrF32Bits; size = sizeof(kEncThrF32Bits); break;
        case 2: src = kEncThrF64Bits; size = sizeof(kEncThrF64Bits); break;
        default: return AVX_ERR_INVALID;
    }
    if (dst_host) memcpy(dst_host, src, capacity < size ? capacity : size);
    return (int)size;
}

}  // extern "C"
