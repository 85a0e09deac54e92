// Colour-frame remap with smoothed maps (deploy_bundle.py:136-146 warpRevBundle2), gfx950.
// The reference does this on the host with OpenCV after every sess.run; here the maps never leave the device:
//   map_shrink_kernel : cv2.resize(map, (W/rate, H/rate)) INTER_LINEAR (half-pixel centres, float32, h-pass then v-pass)
//   remap_color_kernel: cv2.resize back to (W, H) fused with (m+1)/2*size and cv2.remap(..., INTER_LINEAR) on the uint8
//                       BGR frame: OpenCV's FIXED-POINT bilinear path (1/32 px coordinates, 15-bit integer weight table,
//                       (sum + 16384) >> 15), BORDER_CONSTANT 0 -- integer arithmetic, bit-exact against the oracle
// Float32 op order of the two resizes as oracle/stabnet_oracle.py: cv_resize_linear_f32 (-ffp-contract=off).  HBM-bound:
// per frame 8*H*W (maps in) + 3*H*W (frame, gathered) + 3*H*W (out) bytes.
#include "common.h"

struct Taps1D { int i0, i1; float w0, w1; };

__device__ __forceinline__ Taps1D cv_taps(int d, int n_src, double scale) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    Taps1D t;
    t.i0 = s; t.i1 = min(s + 1, n_src - 1); t.w0 = 1.0f - f; t.w1 = f;
    return t;
}

__device__ __forceinline__ float cv_resize_at(const float* __restrict__ src, int sw, const Taps1D& tx, const Taps1D& ty) {
    const float r0 = src[(size_t)ty.i0 * sw + tx.i0] * tx.w0 + src[(size_t)ty.i0 * sw + tx.i1] * tx.w1;
    const float r1 = src[(size_t)ty.i1 * sw + tx.i0] * tx.w0 + src[(size_t)ty.i1 * sw + tx.i1] * tx.w1;
    return r0 * ty.w0 + r1 * ty.w1;
}

// small[n][0|1][h][w] <- x_map, y_map shrunk.  One thread per low-resolution pixel.
__global__ __launch_bounds__(256) void map_shrink_kernel(const float* __restrict__ x_map, const float* __restrict__ y_map, int H,
                                                         int W, int h, int w, float* __restrict__ small_maps) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (q >= h * w) return;
    const int dy = q / w, dx = q - dy * w;
    const Taps1D tx = cv_taps(dx, W, (double)W / w), ty = cv_taps(dy, H, (double)H / h);
    small_maps[((size_t)n * 2 + 0) * h * w + q] = cv_resize_at(x_map + (size_t)n * H * W, W, tx, ty);
    small_maps[((size_t)n * 2 + 1) * h * w + q] = cv_resize_at(y_map + (size_t)n * H * W, W, tx, ty);
}

__global__ __launch_bounds__(256) void remap_color_kernel(const unsigned char* __restrict__ img, const float* __restrict__ small_maps,
                                                          int H, int W, int C, int h, int w, unsigned char* __restrict__ out,
                                                          float* __restrict__ px_out, float* __restrict__ py_out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (q >= H * W) return;
    const int y = q / W, x = q - y * W;
    const Taps1D tx = cv_taps(x, w, (double)w / W), ty = cv_taps(y, h, (double)h / H);
    const float xs = cv_resize_at(small_maps + ((size_t)n * 2 + 0) * h * w, w, tx, ty);
    const float ys = cv_resize_at(small_maps + ((size_t)n * 2 + 1) * h * w, w, tx, ty);
    const float px = (xs + 1.0f) / 2.0f * (float)W;                 // deploy_bundle.py:142-143
    const float py = (ys + 1.0f) / 2.0f * (float)H;
    if (px_out != nullptr) { px_out[(size_t)n * H * W + q] = px; py_out[(size_t)n * H * W + q] = py; }
    // OpenCV's 8-bit bilinear remap (imgwarp.cpp: RemapInvoker + remapBilinear<FixedPtCast<int, uchar, 15>, RemapVec_8u, short>;
    // oracle cv_remap_linear_u8 / cv_bilinear_tab_i): coordinates quantised to 1/32 px (cvRound: half to even), integer part
    // saturated to int16, 15-bit integer weights wy[k1] * wx[k2] * 32768 -- all exact multiples of 32 except the table's entry
    // (0, 0), whose weight 1.0 saturates to 32767 and whose repair loop puts the missing 1 on tap [1][1] -- integer accumulate,
    // (sum + 16384) >> 15.  (A float beyond the int range is out of frame either way: clamped before the conversion.)
    const float qx = fminf(fmaxf(px * 32.0f, -2.0e9f), 2.0e9f), qy = fminf(fmaxf(py * 32.0f, -2.0e9f), 2.0e9f);
    const int sx = (qx == qx) ? (int)rintf(qx) : -2000000000, sy = (qy == qy) ? (int)rintf(qy) : -2000000000;
    const int ix = min(max(sx >> 5, -32768), 32767), iy = min(max(sy >> 5, -32768), 32767);
    const int fx = sx & 31, fy = sy & 31;
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
    const unsigned char* im = img + (size_t)n * H * W * C;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    for (int c = 0; c < C; ++c) {
        const int v00 = (x0 && y0) ? (int)im[((size_t)iy * W + ix) * C + c] : 0;
        const int v01 = (x1 && y0) ? (int)im[((size_t)iy * W + ix + 1) * C + c] : 0;
        const int v10 = (x0 && y1) ? (int)im[((size_t)(iy + 1) * W + ix) * C + c] : 0;
        const int v11 = (x1 && y1) ? (int)im[((size_t)(iy + 1) * W + ix + 1) * C + c] : 0;
        const int acc = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        out[((size_t)n * H * W + q) * C + c] = (unsigned char)min(max(acc, 0), 255);
    }
}

// Four consecutive pixels per thread (C == 3, W % 4 == 0: the colour frames of the path).  The byte traffic of the one-pixel kernel
// above -- 12 one-byte gathers and 3 one-byte stores per pixel -- becomes 4-byte traffic: the two taps of a row are 6 CONTIGUOUS bytes
// (BGR BGR), fetched as the three aligned dwords that cover them and funnel-shifted into place; the 12 output bytes of the four pixels
// leave as three 4-byte stores of one thread (12 contiguous bytes).  Same integer arithmetic, same results (tests/test_remap_gpu.py: exact against the oracle).
__device__ __forceinline__ unsigned long long remap_load6(const unsigned char* __restrict__ im, long b, long total) {
    // bytes b .. b+5 of the frame in the low 48 bits (b >= 0, b + 3 <= total: at least the first tap is in the buffer): the three
    // aligned dwords that cover them (b may sit at byte 3 of its dword), funnel-shifted; the last pixels of the frame, whose third
    // dword would lie past the buffer, take the bytes one by one
    const long al = b & ~3L;
    const int sh = 8 * (int)(b & 3);
    if (al + 12 <= total) {
        const unsigned* p = reinterpret_cast<const unsigned*>(im + al);
        const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
        const unsigned lo = (unsigned)(((((unsigned long long)d1) << 32) | d0) >> sh), hi = (unsigned)(((((unsigned long long)d2) << 32) | d1) >> sh);
        return (((unsigned long long)hi) << 32) | lo;
    }
    unsigned long long r = 0ull;
    for (int k = 0; k < 6; ++k)
        if (b + k < total) r |= ((unsigned long long)im[b + k]) << (8 * k);
    return r;
}

__global__ __launch_bounds__(256) void remap_color4_kernel(const unsigned char* __restrict__ img, const float* __restrict__ small_maps,
                                                           int H, int W, int h, int w, unsigned char* __restrict__ out,
                                                           float* __restrict__ px_out, float* __restrict__ py_out) {
    const int q4 = blockIdx.x * 256 + threadIdx.x;           // group of four pixels
    const int n = blockIdx.y;
    const int W4 = W >> 2;
    if (q4 >= H * W4) return;
    const int y = q4 / W4, x0 = (q4 - y * W4) * 4;
    const Taps1D ty = cv_taps(y, h, (double)h / H);
    const float* mx = small_maps + ((size_t)n * 2 + 0) * h * w;
    const float* my = small_maps + ((size_t)n * 2 + 1) * h * w;
    const unsigned char* im = img + (size_t)n * H * W * 3;
    const long total = (long)H * W * 3;
    unsigned ob[3] = {0u, 0u, 0u};                           // the 12 output bytes
    float pxs[4], pys[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const Taps1D tx = cv_taps(x0 + e, w, (double)w / W);
        const float xs = cv_resize_at(mx, w, tx, ty), ys = cv_resize_at(my, w, tx, ty);
        const float px = (xs + 1.0f) / 2.0f * (float)W;                 // deploy_bundle.py:142-143
        const float py = (ys + 1.0f) / 2.0f * (float)H;
        pxs[e] = px; pys[e] = py;
        const float qx = fminf(fmaxf(px * 32.0f, -2.0e9f), 2.0e9f), qy = fminf(fmaxf(py * 32.0f, -2.0e9f), 2.0e9f);
        const int sx = (qx == qx) ? (int)rintf(qx) : -2000000000, sy = (qy == qy) ? (int)rintf(qy) : -2000000000;
        const int ix = min(max(sx >> 5, -32768), 32767), iy = min(max(sy >> 5, -32768), 32767);
        const int fx = sx & 31, fy = sy & 31;
        int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
        if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
        const bool vx0 = ix >= 0 && ix < W, vx1 = ix + 1 >= 0 && ix + 1 < W, vy0 = iy >= 0 && iy < H, vy1 = iy + 1 >= 0 && iy + 1 < H;
        // rows iy and iy + 1: 6 bytes from pixel max(ix, 0) on (ix == -1: the first three bytes are tap 1)
        unsigned long long r0 = 0ull, r1 = 0ull;
        const int cx = max(ix, 0);
        if ((vx0 || vx1) && vy0) r0 = remap_load6(im, ((long)iy * W + cx) * 3, total);
        if ((vx0 || vx1) && vy1) r1 = remap_load6(im, ((long)(iy + 1) * W + cx) * 3, total);
        if (ix < 0) { r0 <<= 24; r1 <<= 24; }                  // tap 0 out of frame on the left: what was loaded is tap 1
        const unsigned long long m0 = vx0 ? 0xffffffull : 0ull, m1 = vx1 ? 0xffffff000000ull : 0ull;
        r0 &= (vy0 ? (m0 | m1) : 0ull);
        r1 &= (vy1 ? (m0 | m1) : 0ull);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v00 = (int)((r0 >> (8 * c)) & 0xff), v01 = (int)((r0 >> (24 + 8 * c)) & 0xff);
            const int v10 = (int)((r1 >> (8 * c)) & 0xff), v11 = (int)((r1 >> (24 + 8 * c)) & 0xff);
            const int acc = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
            const unsigned o = (unsigned)min(max(acc, 0), 255);
            const int byte = e * 3 + c;
            ob[byte >> 2] |= o << (8 * (byte & 3));
        }
    }
    const size_t pix = (size_t)n * H * W + (size_t)y * W + x0;
    unsigned* op = reinterpret_cast<unsigned*>(out + pix * 3);
    op[0] = ob[0]; op[1] = ob[1]; op[2] = ob[2];
    if (px_out != nullptr) {
        *reinterpret_cast<float4*>(px_out + pix) = make_float4(pxs[0], pxs[1], pxs[2], pxs[3]);
        *reinterpret_cast<float4*>(py_out + pix) = make_float4(pys[0], pys[1], pys[2], pys[3]);
    }
}

// cvt_train2img (deploy_bundle.py:75): ((x + 0.5) * 255).astype(uint8), clipped to [0, 255] first (the network's grey output is a
// bilinear blend of inputs in [-0.5, 0.5], so the clip only guards the cast).  4 pixels per thread, float4 in, one dword out.
__global__ __launch_bounds__(256) void cvt_train2img_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    auto cv = [](float v) -> unsigned int { return (unsigned int)fminf(fmaxf((v + 0.5f) * 255.0f, 0.0f), 255.0f); };
    if (i + 4 <= n && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        *reinterpret_cast<unsigned int*>(out + i) = cv(v.x) | (cv(v.y) << 8) | (cv(v.z) << 16) | (cv(v.w) << 24);
    } else {
        for (long j = i; j < n && j < i + 4; ++j) out[j] = (unsigned char)cv(x[j]);
    }
}

extern "C" {

/* warpRevBundle2(img, x_map, y_map) (deploy_bundle.py:136-146): img uint8 [N,H,W,C] (BGR, C = 3), x_map, y_map [N,H,W]
 * normalised (the warp's outputs) -> out uint8 [N,H,W,C].  workspace: 2*N*(H/rate)*(W/rate) floats.  px_out/py_out
 * (optional, [N,H,W]): the smoothed maps in pixel coordinates that cv2.remap would receive. */
int stabnet_warp_rev_bundle2(const unsigned char* img, const float* x_map, const float* y_map, int N, int H, int W, int C,
                             int rate, unsigned char* out, float* workspace, float* px_out, float* py_out, void* stream) {
    SN_REQUIRE(img && x_map && y_map && out && workspace, "warp_rev_bundle2: null pointer");
    SN_REQUIRE(N > 0 && N <= 65535 && C > 0 && rate >= 1 && H / rate >= 1 && W / rate >= 1, "warp_rev_bundle2: bad shape");
    SN_REQUIRE((px_out == nullptr) == (py_out == nullptr), "warp_rev_bundle2: px_out and py_out go together");
    const int h = H / rate, w = W / rate;
    hipStream_t st = (hipStream_t)stream;
    map_shrink_kernel<<<dim3(cdiv(h * w, 256), N), 256, 0, st>>>(x_map, y_map, H, W, h, w, workspace);
    SN_LAUNCH_CHECK("map_shrink_kernel");
    static const int v4 = []() { const char* v = getenv("STABNET_REMAP_VEC4"); return v ? atoi(v) : 1; }();
    const bool aligned = (((size_t)img | (size_t)out) & 3) == 0 && (px_out == nullptr || (((size_t)px_out | (size_t)py_out) & 15) == 0);
    if (v4 && C == 3 && W % 4 == 0 && (long)H * W * 3 >= 8 && aligned) {
        remap_color4_kernel<<<dim3(cdiv((long)H * W / 4, 256), N), 256, 0, st>>>(img, workspace, H, W, h, w, out, px_out, py_out);
        SN_LAUNCH_CHECK("remap_color4_kernel");
        return STABNET_OK;
    }
    remap_color_kernel<<<dim3(cdiv((long)H * W, 256), N), 256, 0, st>>>(img, workspace, H, W, C, h, w, out, px_out, py_out);
    SN_LAUNCH_CHECK("remap_color_kernel");
    return STABNET_OK;
}

/* cvt_train2img (deploy_bundle.py:75): out[i] = uint8((x[i] + 0.5) * 255), clipped to [0, 255].  x float [n], out uint8 [n]. */
int stabnet_cvt_train2img(const float* x, unsigned char* out, long n, void* stream) {
    SN_REQUIRE(x && out && n > 0, "cvt_train2img: bad arguments");
    cvt_train2img_kernel<<<cdiv(cdiv(n, 4), 256), 256, 0, (hipStream_t)stream>>>(x, out, n);
    SN_LAUNCH_CHECK("cvt_train2img_kernel");
    return STABNET_OK;
}

}  // extern "C"
