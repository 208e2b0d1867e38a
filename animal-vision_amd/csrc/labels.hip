// csrc/labels.hip -- the corner labels of VideoRenderer.make_split_frame (reference: renderers/video.py:160-196,
// :241-245) drawn on a device-resident uint8 frame.
//
// The reference draws a label with three OpenCV calls: a filled black rectangle blended in at 60 % (cv2.addWeighted of an
// overlay: inside the box out = saturate_cast<uchar>(0.4 * in), elsewhere unchanged), then the text twice with
// cv2.putText(FONT_HERSHEY_SIMPLEX, LINE_AA): black at thickness + 2, white at thickness.  Geometry (box, origin, scale rule)
// is computed on the host exactly as the reference computes it (renderers/labels.py); the strokes arrive as line segments
// in pixel coordinates.  Anti-aliasing is analytic -- coverage = clamp(thickness/2 + 0.5 - distance to the nearest
// segment, 0, 1), round caps and joins -- not OpenCV's fixed-point LINE_AA scan converter: label pixels are PARITY UNPINNED
// against OpenCV (it is not installed anywhere in the pipeline); they are bit-identical to oracle/cpu_ref.draw_label.
#include "avx_internal.h"

namespace {

constexpr int kLT = 256;

struct LabelArgs {
    uint8_t* img;
    int H, W;
    int bx0, by0, bx1, by1;   // dimmed box, inclusive corners (cv2.rectangle)
    int rx0, ry0, rx1, ry1;   // region to visit (box and strokes, clipped to the frame), inclusive
    const float* seg;          // [n][6]: ax, ay, dx, dy, 1 / (dx^2 + dy^2) (0 for a point), unused
    int nseg;
    float half_outline, half_text;  // thickness / 2 of the black outline pass and of the white pass
};

__device__ __forceinline__ float cover(float d, float half_t) {
    const float c = half_t + 0.5f - d;
    return c < 0.f ? 0.f : (c > 1.f ? 1.f : c);
}

__global__ __launch_bounds__(kLT) void k_draw_label(LabelArgs a) {
    const int rw = a.rx1 - a.rx0 + 1, rh = a.ry1 - a.ry0 + 1;
    for (int i = blockIdx.x * kLT + threadIdx.x; i < rw * rh; i += gridDim.x * kLT) {
        const int x = a.rx0 + i % rw, y = a.ry0 + i / rw;
        float d2 = 3.0e38f;
        const float px = (float)x, py = (float)y;
        for (int s = 0; s < a.nseg; ++s) {
            const float* g = a.seg + 6 * s;
            const float qx = px - g[0], qy = py - g[1];
            float t = (qx * g[2] + qy * g[3]) * g[4];
            t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
            const float ex = qx - t * g[2], ey = qy - t * g[3];
            const float e2 = ex * ex + ey * ey;
            d2 = e2 < d2 ? e2 : d2;
        }
        const float d = __fsqrt_rn(d2);
        const float co = cover(d, a.half_outline), ct = cover(d, a.half_text);
        const bool inbox = x >= a.bx0 && x <= a.bx1 && y >= a.by0 && y <= a.by1;
        uint8_t* p = a.img + ((size_t)y * a.W + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)p[c];
            if (inbox) v = rintf(v * 0.4f);       // addWeighted(overlay, 0.6, img, 0.4): the overlay is black inside the box
            v = rintf(v - v * co);                // black outline: v + (0 - v) * coverage
            v = rintf(v + (255.f - v) * ct);      // white text
            p[c] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
        }
    }
}

}  // namespace

extern "C" int avx_draw_label_u8(avx_ctx* ctx, uint8_t* img_hwc, int H, int W, const int box_xyxy[4], const float* segments_host, int n_segments,
                                 float outline_thickness, float text_thickness, int slot, void* stream) {
    if (!ctx) return AVX_ERR_INVALID;
    AVX_REQUIRE(ctx, img_hwc && box_xyxy && H > 0 && W > 0 && n_segments >= 0 && n_segments <= 4096 && (n_segments == 0 || segments_host),
                "avx_draw_label_u8: bad arguments");
    AVX_REQUIRE(ctx, slot == 0 || slot == 1, "avx_draw_label_u8: slot must be 0 or 1 (one cached segment table per label of a frame)");
    AVX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = avx_pick_stream(ctx, stream);
    avx_ws* ws = avx_workspace(ctx, s);
    if (!ws) return AVX_ERR_NOMEM;
    LabelArgs a{};
    a.img = img_hwc; a.H = H; a.W = W;
    a.bx0 = box_xyxy[0]; a.by0 = box_xyxy[1]; a.bx1 = box_xyxy[2]; a.by1 = box_xyxy[3];
    a.nseg = n_segments;
    a.half_outline = 0.5f * outline_thickness;
    a.half_text = 0.5f * text_thickness;
    // region = the box grown by the outline's reach (strokes stay inside the box by construction: it is the text box + pad)
    const int grow = (int)(a.half_outline + 2.f);
    a.rx0 = a.bx0 - grow < 0 ? 0 : a.bx0 - grow;
    a.ry0 = a.by0 - grow < 0 ? 0 : a.by0 - grow;
    a.rx1 = a.bx1 + grow > W - 1 ? W - 1 : a.bx1 + grow;
    a.ry1 = a.by1 + grow > H - 1 ? H - 1 : a.by1 + grow;
    if (a.rx1 < a.rx0 || a.ry1 < a.ry0) return AVX_OK;
    float* dseg = nullptr;
    if (n_segments > 0) {
        const int rc = avx_const_upload(ctx, ws, 8 + slot, segments_host, sizeof(float) * 6 * n_segments, s, (void**)&dseg);
        if (rc) return rc;
    }
    a.seg = dseg;
    const int n = (a.rx1 - a.rx0 + 1) * (a.ry1 - a.ry0 + 1);
    hipLaunchKernelGGL(k_draw_label, dim3((n + kLT - 1) / kLT), dim3(kLT), 0, s, a);
    AVX_HIP(ctx, hipGetLastError());
    return AVX_OK;
}
