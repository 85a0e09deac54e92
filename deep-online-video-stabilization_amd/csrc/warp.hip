// Multi-grid homography warp, forward, for gfx950.
//
//   mesh_homography_kernel : theta|pts2 -> pts2, Hs        one 64-lane wave per (sample, cell); the 8x8
//                                                          system lives one element per lane (lane = 8*row+col)
//   warp_sample_kernel     : Hs, frame -> x_map,y_map,black,out   fused map + divide + black test + 4-tap gather
//                                                          (+ the online loop's feedback push, PUSH = 1)
//   interp_kernel          : frame, x, y -> out            the same sampler with caller-supplied maps
//
// Arithmetic follows the reference op for op in float32, one rounding per TF op (this file is built with
// -ffp-contract=off), so results are comparable bit for bit with oracle/stabnet_oracle.py:
//   get_4_pts   s_net_bundle_nobm.py:29-71        get_H/pinv  spatial_transformer3.py:144-175
//   get_Hs      spatial_transformer3.py:179-198   _transform3 spatial_transformer3.py:218-301
//   _interpolate spatial_transformer3.py:62-123   interpolate spatial_transformer.py:200-281
#include "warp.h"
#include "mesh_device.h"
#include <climits>

#define SN_MAX_CELLS 64

// tf.cast(float->int32) on the reference's x86 host: cvttss2si, out of range / NaN -> INT_MIN.
__device__ __forceinline__ int cvt_i32_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// ---------------------------------------------------------------------------------------------------------
// One wave per (n, cell).  in_is_theta: in = theta [N, (gh+1)(gw+1)*2] (vertex = regular grid + offset, clipped
// to +-lim); else in = pts2 [N, gh+1, gw+1, 2] used as is.  Writes pts2 (optional) and Hs [N, gh*gw, 9].
// The per-cell arithmetic lives in mesh_device.h (shared with the regressor's fused head).
__global__ __launch_bounds__(256) void mesh_homography_kernel(const float* __restrict__ in, int in_is_theta, int N,
                                                              int gh, int gw, float lim, float* __restrict__ pts2_out,
                                                              float* __restrict__ Hs_out, float* __restrict__ pts1_out,
                                                              int* __restrict__ head_adv, int depth) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    // online loop: nothing between the stack assembly and the sampler reads the ring head, and this kernel sits between them
    // in the stream, so ONE thread advances it here; the sampler pushes to (*head - 1) mod depth (no last-block ticket).
    if (head_adv != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *head_adv = (*head_adv + 1) % depth;
    const int cells = gh * gw;
    if (wave >= N * cells) return;                       // wave-uniform exit
    const int n = wave / cells, cell = wave % cells;
    const int nv = (gh + 1) * (gw + 1);
    sn_mesh_cell(in + (size_t)n * nv * 2, in_is_theta, gh, gw, lim, cell, lane,
                 pts2_out ? pts2_out + (size_t)n * nv * 2 : nullptr, Hs_out + (size_t)n * cells * 9,
                 pts1_out ? pts1_out + (size_t)n * cells * 8 : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
struct SampleTaps {
    int ia, ib, ic, id;            // flat pixel offsets inside one image: (y0,x0) (y1,x0) (y0,x1) (y1,x1)
    float wa, wb, wc, wd;
};

__device__ __forceinline__ SampleTaps make_taps(float xm, float ym, int H, int W) {
    const float xp = (xm + 1.0f) * (float)W / 2.0f;     // :81-82 (not align-corners)
    const float yp = (ym + 1.0f) * (float)H / 2.0f;
    int x0 = cvt_i32_x86(floorf(xp));
    int y0 = cvt_i32_x86(floorf(yp));
    int x1 = x0 + 1;
    int y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1);                        // corners clipped BEFORE the weights, :90-93
    x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1);
    y1 = min(max(y1, 0), H - 1);
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    SampleTaps t;
    t.wa = (x1f - xp) * (y1f - yp);
    t.wb = (x1f - xp) * (yp - y0f);
    t.wc = (xp - x0f) * (y1f - yp);
    t.wd = (xp - x0f) * (yp - y0f);
    t.ia = y0 * W + x0;
    t.ib = y1 * W + x0;
    t.ic = y0 * W + x1;
    t.id = y1 * W + x1;
    return t;
}

__device__ __forceinline__ float sample4(const float* __restrict__ img, const SampleTaps& t, int C, int ch) {
    const float Ia = img[(size_t)t.ia * C + ch], Ib = img[(size_t)t.ib * C + ch];
    const float Ic = img[(size_t)t.ic * C + ch], Id = img[(size_t)t.id * C + ch];
    return ((t.wa * Ia + t.wb * Ib) + t.wc * Ic) + t.wd * Id;          // tf.add_n order, :122
}

// Tiling (measured: tools/warp_probe.hip, profiles/r03_warp_probe.txt).  One wave = SN_WARP_PX * 64 CONSECUTIVE pixels of one
// row, lane-strided (pixel e of lane l = x0 + 64 e + l): every gather and every store instruction of a wave touches 64
// consecutive pixels (256 contiguous bytes), W needs no alignment.  A block = 4 consecutive wave segments; grid.y = sample.
// The whole segment usually lies in ONE cell column, so the cell's 9 homography entries are wave-uniform scalar loads (no LDS
// staging, no barrier); a segment that straddles a cell edge reads them per lane.  All 4 * PX taps of a thread are issued
// before the first blend.  x_map / y_map / black / out are written with non-temporal stores (nothing on the GPU re-reads them
// soon), the feedback planes with plain stores (the next frame's stack assembly reads them).
// 720p sampler + push: 16.8 us (4 pixels per thread + LDS staging + last-block ticket) -> 8.0 us; 7.3 us of the 16.8 were the
// ticket (one device-scope atomic per block on one address).
// PUSH = 1 (online loop, C == 1): the feedback of deploy_bundle.py:291-295,319-323 rides on the same pass --
//   frame = img + black * (-1) -> frames_ring[slot], black -> masks_ring[slot], frame -> frame_out, all_black += round(black)
//   -- with slot = (*head - 1) mod depth: the ring head has ALREADY been advanced by the mesh kernel that precedes this
//   launch in the stream (launch_mesh's head_adv; the kernel boundary orders it), so no block has to take a ticket.
#define SN_WARP_PX 2
template <int PUSH>
__global__ __launch_bounds__(256) void warp_sample_kernel(const float* __restrict__ Hs, const float* __restrict__ src,
                                                          int H, int W, int C, int gh, int gw,
                                                          float* __restrict__ out, float* __restrict__ black,
                                                          float* __restrict__ x_map, float* __restrict__ y_map,
                                                          const WarpPush push) {
    constexpr int PX = SN_WARP_PX;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = blockIdx.y;
    const int segs = (W + 64 * PX - 1) / (64 * PX);           // wave segments per row
    // workgroup ids are dealt round-robin over the 8 XCDs (own L2 each): ids with equal b % 8 take one CONTIGUOUS eighth of the
    // rows, so an XCD gathers from one band of the source frame instead of all of it
    const int wseg = (int)sn_xcd_band(blockIdx.x, gridDim.x) * 4 + wv;
    const int y = wseg / segs;
    if (y >= H) return;                                        // wave-uniform
    const int x0 = (wseg - y * segs) * (64 * PX);
    int slot = 0;
    if (PUSH) {
        slot = *push.head - 1;
        if (slot < 0) slot += push.depth;
    }
    const int cells = gh * gw;
    const int chh = H / gh, cww = W / gw;                      // floor, :227-228
    const int ci = min(y / chh, gh - 1);
    const float stepx = (W > 1) ? 2.0f / (float)(W - 1) : 0.0f;           // LinSpace: start + step*i
    const float stepy = (H > 1) ? 2.0f / (float)(H - 1) : 0.0f;
    const float gy = -1.0f + stepy * (float)y;
    const float* img = src + (size_t)n * H * W * C;
    const float* Hn = Hs + (size_t)n * cells * 9;
    const size_t rowoff = ((size_t)n * H + y) * W;

    float xm[PX], ym[PX], bl[PX];
    auto map_px = [&](const float* h, int e) {
        const int xx = min(x0 + e * 64 + lane, W - 1);
        const float gx = -1.0f + stepx * (float)xx;
        const float tx = (h[0] * gx + h[1] * gy) + h[2];
        const float ty = (h[3] * gx + h[4] * gy) + h[5];
        float tz = (h[6] * gx + h[7] * gy) + h[8];
        const float sgn = ((tz >= 0.0f) ? 1.0f : 0.0f) * 2.0f - 1.0f;  // :257
        tz = tz + sgn * 1e-8f;                                          // :258
        xm[e] = tx / tz;
        ym[e] = ty / tz;
        bl[e] = ((-1.0f > xm[e]) || (xm[e] > 1.0f) || (-1.0f > ym[e]) || (ym[e] > 1.0f)) ? 1.0f : 0.0f;
    };
    const int cj0 = min(x0 / cww, gw - 1), cj1 = min(min(x0 + 64 * PX - 1, W - 1) / cww, gw - 1);
    if (cj0 == cj1) {                                          // wave-uniform: the segment lies in one cell
        const float* hp = Hn + (ci * gw + cj0) * 9;
        float h[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) h[i] = hp[i];              // uniform address -> scalar loads
#pragma unroll
        for (int e = 0; e < PX; ++e) map_px(h, e);
    } else {
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int cj = min(min(x0 + e * 64 + lane, W - 1) / cww, gw - 1);
            const float* hp = Hn + (ci * gw + cj) * 9;
            float h[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = hp[i];
            map_px(h, e);
        }
    }
    if (C == 1) {
        SampleTaps t[PX];
        float Ia[PX], Ib[PX], Ic[PX], Id[PX];
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            t[e] = make_taps(xm[e], ym[e], H, W);
            Ia[e] = img[t[e].ia]; Ib[e] = img[t[e].ib]; Ic[e] = img[t[e].ic]; Id[e] = img[t[e].id];
        }
        const size_t hw = (size_t)H * W;
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int xx = x0 + e * 64 + lane;
            if (xx >= W) continue;
            const float o = ((t[e].wa * Ia[e] + t[e].wb * Ib[e]) + t[e].wc * Ic[e]) + t[e].wd * Id[e];   // tf.add_n order, :122
            const size_t pix = rowoff + xx;
            __builtin_nontemporal_store(xm[e], x_map + pix);
            __builtin_nontemporal_store(ym[e], y_map + pix);
            __builtin_nontemporal_store(bl[e], black + pix);
            __builtin_nontemporal_store(o, out + pix);
            if (PUSH) {
                const float f = o + bl[e] * -1.0f;                                                     // :293
                const size_t ipix = (size_t)y * W + xx;
                const size_t rp = ((size_t)n * push.depth + slot) * hw + ipix;
                push.frames[rp] = f;                                                                   // :322
                push.masks[rp] = bl[e];                                                                // :323
                push.frame_out[(size_t)n * hw + ipix] = f;
                if (push.all_black != nullptr) push.all_black[(size_t)n * hw + ipix] += (int)bl[e];    // :291
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int xx = x0 + e * 64 + lane;
            if (xx >= W) continue;
            const size_t pix = rowoff + xx;
            x_map[pix] = xm[e];
            y_map[pix] = ym[e];
            black[pix] = bl[e];
            const SampleTaps t = make_taps(xm[e], ym[e], H, W);
            for (int ch = 0; ch < C; ++ch) out[pix * C + ch] = sample4(img, t, C, ch);
        }
    }
}

// interpolate(im, x, y): caller-supplied maps (spatial_transformer.py:200-281).  Same tiling.
__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ im, const float* __restrict__ xs,
                                                     const float* __restrict__ ys, int H, int W, int C,
                                                     float* __restrict__ out) {
    constexpr int PX = SN_WARP_PX;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = blockIdx.y;
    const int segs = (W + 64 * PX - 1) / (64 * PX);
    const int wseg = blockIdx.x * 4 + wv;
    const int y = wseg / segs;
    if (y >= H) return;
    const int x0 = (wseg - y * segs) * (64 * PX);
    const float* img = im + (size_t)n * H * W * C;
    const size_t rowoff = ((size_t)n * H + y) * W;
    SampleTaps t[PX];
    float Ia[PX], Ib[PX], Ic[PX], Id[PX];
#pragma unroll
    for (int e = 0; e < PX; ++e) {
        const size_t pix = rowoff + min(x0 + e * 64 + lane, W - 1);
        t[e] = make_taps(xs[pix], ys[pix], H, W);
        if (C == 1) { Ia[e] = img[t[e].ia]; Ib[e] = img[t[e].ib]; Ic[e] = img[t[e].ic]; Id[e] = img[t[e].id]; }
    }
#pragma unroll
    for (int e = 0; e < PX; ++e) {
        const int xx = x0 + e * 64 + lane;
        if (xx >= W) continue;
        const size_t pix = rowoff + xx;
        if (C == 1) out[pix] = ((t[e].wa * Ia[e] + t[e].wb * Ib[e]) + t[e].wc * Ic[e]) + t[e].wd * Id[e];
        else for (int ch = 0; ch < C; ++ch) out[pix * C + ch] = sample4(img, t[e], C, ch);
    }
}

// ---------------------------------------------------------------------------------------------------------
int check_warp_args(int N, int H, int W, int C, int gh, int gw) {
    SN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "warp: N,H,W,C must be positive (got %d,%d,%d,%d)", N, H, W, C);
    SN_REQUIRE(gh > 0 && gw > 0 && gh * gw <= SN_MAX_CELLS, "warp: grid %dx%d unsupported (max %d cells)", gh, gw,
               SN_MAX_CELLS);
    SN_REQUIRE(H >= gh && W >= gw, "warp: frame %dx%d smaller than grid %dx%d", H, W, gh, gw);
    SN_REQUIRE((long)N * H * W * C < (1L << 31), "warp: tensor too large for int32 gather indices");
    SN_REQUIRE(N <= 65535, "warp: N > 65535");
    return STABNET_OK;
}

int launch_mesh(const float* in, int is_theta, int N, int gh, int gw, float lim, float* pts2, float* Hs,
                       hipStream_t st, float* pts1, int* head_adv, int depth) {
    const long waves = (long)N * gh * gw;
    mesh_homography_kernel<<<cdiv(waves * 64, 256), 256, 0, st>>>(in, is_theta, N, gh, gw, lim, pts2, Hs, pts1, head_adv, depth);
    SN_LAUNCH_CHECK("mesh_homography_kernel");
    return STABNET_OK;
}

static dim3 sample_grid(int N, int H, int W) {
    const long segs = cdiv(W, 64 * SN_WARP_PX);
    return dim3((unsigned)cdiv(segs * H, 4), (unsigned)N, 1);
}

int launch_sample(const float* Hs, const float* src, int N, int H, int W, int C, int gh, int gw, float* out,
                         float* black, float* x_map, float* y_map, hipStream_t st) {
    const WarpPush none{};
    warp_sample_kernel<0><<<sample_grid(N, H, W), 256, 0, st>>>(Hs, src, H, W, C, gh, gw, out, black, x_map, y_map, none);
    SN_LAUNCH_CHECK("warp_sample_kernel");
    return STABNET_OK;
}

// The online loop's last pass: sample + feedback push in one launch (C = 1).  *push.head must already be the ADVANCED head
// (launch_mesh(..., head_adv) earlier in the same stream): the push goes to slot (*head - 1) mod depth.
int launch_sample_push(const float* Hs, const float* src, int N, int H, int W, int gh, int gw, float* out, float* black,
                       float* x_map, float* y_map, const WarpPush& push, hipStream_t st) {
    SN_REQUIRE(push.frames && push.masks && push.frame_out && push.head && push.depth > 0, "sample_push: bad arguments");
    warp_sample_kernel<1><<<sample_grid(N, H, W), 256, 0, st>>>(Hs, src, H, W, 1, gh, gw, out, black, x_map, y_map, push);
    SN_LAUNCH_CHECK("warp_sample_kernel");
    return STABNET_OK;
}

extern "C" {

int stabnet_get_4_pts(const float* theta, int N, int grid_h, int grid_w, float do_crop_rate, float* pts1, float* pts2,
                      float* Hs, void* stream) {
    SN_REQUIRE(theta && pts2 && Hs, "get_4_pts: null pointer");
    SN_REQUIRE(N > 0 && grid_h > 0 && grid_w > 0 && grid_h * grid_w <= SN_MAX_CELLS, "get_4_pts: bad shape");
    SN_REQUIRE(do_crop_rate > 0.f, "get_4_pts: do_crop_rate must be > 0");
    return launch_mesh(theta, 1, N, grid_h, grid_w, 1.0f / do_crop_rate, pts2, Hs, (hipStream_t)stream, pts1);
}

int stabnet_transformer_fwd(const float* pts2, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                            float* out, float* black, float* x_map, float* y_map, float* Hs, void* stream) {
    SN_REQUIRE(pts2 && U && out && black && x_map && y_map && Hs, "transformer_fwd: null pointer");
    int rc = check_warp_args(N, H, W, C, grid_h, grid_w);
    if (rc) return rc;
    rc = launch_mesh(pts2, 0, N, grid_h, grid_w, 0.f, nullptr, Hs, (hipStream_t)stream);
    if (rc) return rc;
    return launch_sample(Hs, U, N, H, W, C, grid_h, grid_w, out, black, x_map, y_map, (hipStream_t)stream);
}

int stabnet_warp_fwd(const float* theta, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                     float do_crop_rate, float* out, float* black, float* x_map, float* y_map, float* Hs, float* pts2,
                     void* stream) {
    SN_REQUIRE(theta && U && out && black && x_map && y_map && Hs, "warp_fwd: null pointer");
    SN_REQUIRE(do_crop_rate > 0.f, "warp_fwd: do_crop_rate must be > 0");
    int rc = check_warp_args(N, H, W, C, grid_h, grid_w);
    if (rc) return rc;
    rc = launch_mesh(theta, 1, N, grid_h, grid_w, 1.0f / do_crop_rate, pts2, Hs, (hipStream_t)stream);
    if (rc) return rc;
    return launch_sample(Hs, U, N, H, W, C, grid_h, grid_w, out, black, x_map, y_map, (hipStream_t)stream);
}

int stabnet_maps_from_hs_fwd(const float* Hs, const float* U, int N, int H, int W, int C, int grid_h, int grid_w,
                             float* out, float* black, float* x_map, float* y_map, void* stream) {
    SN_REQUIRE(Hs && U && out && black && x_map && y_map, "maps_from_hs_fwd: null pointer");
    int rc = check_warp_args(N, H, W, C, grid_h, grid_w);
    if (rc) return rc;
    return launch_sample(Hs, U, N, H, W, C, grid_h, grid_w, out, black, x_map, y_map, (hipStream_t)stream);
}

int stabnet_interp_fwd(const float* im, const float* x, const float* y, int N, int H, int W, int C, float* out,
                       void* stream) {
    SN_REQUIRE(im && x && y && out, "interp_fwd: null pointer");
    SN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && N <= 65535, "interp_fwd: bad shape");
    SN_REQUIRE((long)N * H * W * C < (1L << 31), "interp_fwd: tensor too large for int32 gather indices");
    interp_kernel<<<sample_grid(N, H, W), 256, 0, (hipStream_t)stream>>>(im, x, y, H, W, C, out);
    SN_LAUNCH_CHECK("interp_kernel");
    return STABNET_OK;
}

}  // extern "C"
