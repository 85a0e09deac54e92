// Multi-grid homography warp, forward, for gfx950.
//
//   mesh_homography_kernel : theta|pts2 -> pts2, Hs        one 64-lane wave per (sample, cell); the 8x8
//                                                          system lives one element per lane (lane = 8*row+col)
//   warp_sample_kernel     : Hs, frame -> x_map,y_map,black,out   fused map + divide + black test + 4-tap gather
//   interp_kernel          : frame, x, y -> out            the same sampler with caller-supplied maps
//
// Arithmetic follows the reference op for op in float32, one rounding per TF op (this file is built with
// -ffp-contract=off), so results are comparable bit for bit with oracle/stabnet_oracle.py:
//   get_4_pts   s_net_bundle_nobm.py:29-71        get_H/pinv  spatial_transformer3.py:144-175
//   get_Hs      spatial_transformer3.py:179-198   _transform3 spatial_transformer3.py:218-301
//   _interpolate spatial_transformer3.py:62-123   interpolate spatial_transformer.py:200-281
#include "warp.h"
#include <climits>

#define SN_MAX_CELLS 64

__device__ __forceinline__ float wshfl(float v, int src) { return __shfl(v, src, 64); }

// tf.cast(float->int32) on the reference's x86 host: cvttss2si, out of range / NaN -> INT_MIN.
__device__ __forceinline__ int cvt_i32_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// ---------------------------------------------------------------------------------------------------------
// One wave per (n, cell).  in_is_theta: in = theta [N, (gh+1)(gw+1)*2] (vertex = regular grid + offset, clipped
// to +-lim); else in = pts2 [N, gh+1, gw+1, 2] used as is.  Writes pts2 (optional) and Hs [N, gh*gw, 9].
// LU with partial pivoting + two column-oriented triangular solves == Eigen PartialPivLU::inverse() for n=8.
__global__ __launch_bounds__(256) void mesh_homography_kernel(const float* __restrict__ in, int in_is_theta, int N,
                                                              int gh, int gw, float lim, float* __restrict__ pts2_out,
                                                              float* __restrict__ Hs_out, float* __restrict__ pts1_out) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const int cells = gh * gw;
    if (wave >= N * cells) return;                       // wave-uniform exit
    const int n = wave / cells, cell = wave % cells;
    const int ci = cell / gw, cj = cell % gw;
    const double hh = 2.0 / gh, ww = 2.0 / gw;
    const int nv = (gh + 1) * (gw + 1);

    // source corners (regular cell) and target corners (mesh vertices): order TL, TR, BL, BR
    float sx[4], sy[4], tu[4], tv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vi = ci + (q >> 1), vj = cj + (q & 1);
        const double bx = cj * ww - 1.0, by = ci * hh - 1.0;          // spatial_transformer3.py:187-189
        sx[q] = (float)((q & 1) ? bx + ww : bx);
        sy[q] = (float)((q >> 1) ? by + hh : by);
        const int v = vi * (gw + 1) + vj;
        float px = in[((size_t)n * nv + v) * 2 + 0];
        float py = in[((size_t)n * nv + v) * 2 + 1];
        if (in_is_theta) {                                             // s_net_bundle_nobm.py:44-58
            px = (float)(vj * ww - 1.0) + px;
            py = (float)(vi * hh - 1.0) + py;
            px = fminf(fmaxf(px, -lim), lim);
            py = fminf(fmaxf(py, -lim), lim);
        }
        tu[q] = px;
        tv[q] = py;
        if (pts2_out != nullptr && lane == q) {
            // each vertex is written by every cell that owns it with the same value
            pts2_out[((size_t)n * nv + v) * 2 + 0] = px;
            pts2_out[((size_t)n * nv + v) * 2 + 1] = py;
        }
    }
    if (pts1_out != nullptr && lane < 8)                 // [x_TL,x_TR,x_BL,x_BR,y_TL,y_TR,y_BL,y_BR], s_net_bundle_nobm.py:65-66
        pts1_out[((size_t)n * cells + cell) * 8 + lane] = (lane < 4) ? tu[lane & 3] : tv[lane & 3];

    const int r = lane >> 3, c = lane & 7;
    const int q = r & 3;
    const float tq = (r < 4) ? tu[q] : tv[q];
    float a;
    {   // A[r][c], spatial_transformer3.py:160-167
        const int cc = (r < 4) ? c : c - 3;             // u-rows use cols 0..2, v-rows cols 3..5
        float val = 0.0f;
        if (c < 6) {
            if (cc == 0) val = sx[q];
            else if (cc == 1) val = sy[q];
            else if (cc == 2) val = 1.0f;
            else val = 0.0f;
            if ((r < 4 && c >= 3) || (r >= 4 && c < 3)) val = 0.0f;
        } else if (c == 6) {
            val = (-sx[q]) * tq;
        } else {
            val = (-sy[q]) * tq;
        }
        a = val + ((r == c) ? 1e-4f : 0.0f);            // A + eye(8)*1e-4, :145
    }
    float x = (r == c) ? 1.0f : 0.0f;                   // right-hand side: identity, row-swapped with A (P*I)

    for (int k = 0; k < 8; ++k) {
        float best = -1.0f;
        int piv = k;
        for (int rr = k; rr < 8; ++rr) {                // first max |A[rr][k]|
            const float v = fabsf(wshfl(a, rr * 8 + k));
            if (v > best) { best = v; piv = rr; }
        }
        const int src = (r == k) ? piv * 8 + c : ((r == piv) ? k * 8 + c : lane);
        a = wshfl(a, src);
        x = wshfl(x, src);
        const float pivot = wshfl(a, k * 8 + k);
        if (r > k && c == k) a = a / pivot;
        const float l = wshfl(a, r * 8 + k);
        const float u = wshfl(a, k * 8 + c);
        if (r > k && c > k) a = a - l * u;
    }
    for (int i = 0; i < 8; ++i) {                       // unit-lower solve
        const float b = wshfl(x, i * 8 + c);
        const float l = wshfl(a, r * 8 + i);
        if (r > i) x = x - b * l;
    }
    for (int i = 7; i >= 0; --i) {                      // upper solve, reciprocal-diagonal form
        const float d = wshfl(a, i * 8 + i);
        const float inv = 1.0f / d;
        if (r == i) x = x * inv;
        const float b = wshfl(x, i * 8 + c);
        const float u = wshfl(a, r * 8 + i);
        if (r < i) x = x - b * u;
    }
    float acc = 0.0f;                                   // h = inv(A) @ b, k-sequential
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xk = wshfl(x, r * 8 + k);
        const float bk = (k < 4) ? tu[k & 3] : tv[k & 3];
        acc = acc + xk * bk;
    }
    float* Hout = Hs_out + ((size_t)n * cells + cell) * 9;
    if (c == 0) Hout[r] = acc;
    if (lane == 63) Hout[8] = 1.0f;
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

// Tile = 4 rows x 256 columns per 256-thread block; thread = 4 consecutive pixels of one row.
// grid = (ceil(W/256), ceil(H/4), N).
// PUSH = 1 (online loop, C == 1, W % 4 == 0): the feedback of deploy_bundle.py:291-295,319-323 rides on the same pass --
//   frame = img + black * (-1) -> frames_ring[head], black -> masks_ring[head], frame -> frame_out, all_black += round(black)
//   -- and the LAST block to finish (ticket counter) advances the device-side ring head, so the frame needs no push /
//   advance launches.  Every block reads `head` before it takes its ticket, so the advance cannot overtake a reader.
template <int VEC, int PUSH>
__global__ __launch_bounds__(256) void warp_sample_kernel(const float* __restrict__ Hs, const float* __restrict__ src,
                                                          int H, int W, int C, int gh, int gw,
                                                          float* __restrict__ out, float* __restrict__ black,
                                                          float* __restrict__ x_map, float* __restrict__ y_map,
                                                          const WarpPush push) {
    __shared__ float sH[SN_MAX_CELLS * 9];
    const int n = blockIdx.z;
    const int cells = gh * gw;
    for (int i = threadIdx.x; i < cells * 9; i += 256) sH[i] = Hs[(size_t)n * cells * 9 + i];
    int head = 0;
    if (PUSH) head = *push.head;
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wv;
    const int xb = blockIdx.x * 256 + lane * 4;
    if (y < H && xb < W) {
        const int chh = H / gh, cww = W / gw;               // floor, :227-228
        const int ci = min(y / chh, gh - 1);
        const float stepx = (W > 1) ? 2.0f / (float)(W - 1) : 0.0f;       // LinSpace: start + step*i
        const float stepy = (H > 1) ? 2.0f / (float)(H - 1) : 0.0f;
        const float gy = -1.0f + stepy * (float)y;
        const float* img = src + (size_t)n * H * W * C;
        const size_t rowoff = ((size_t)n * H + y) * W;

        float xm[4], ym[4], bl[4], o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = min(xb + e, W - 1);
            const int cj = min(xx / cww, gw - 1);
            const float* h = sH + (ci * gw + cj) * 9;
            const float gx = -1.0f + stepx * (float)xx;
            const float tx = (h[0] * gx + h[1] * gy) + h[2];
            const float ty = (h[3] * gx + h[4] * gy) + h[5];
            float tz = (h[6] * gx + h[7] * gy) + h[8];
            const float sgn = ((tz >= 0.0f) ? 1.0f : 0.0f) * 2.0f - 1.0f;  // :257
            tz = tz + sgn * 1e-8f;                                          // :258
            xm[e] = tx / tz;
            ym[e] = ty / tz;
            bl[e] = ((-1.0f > xm[e]) || (xm[e] > 1.0f) || (-1.0f > ym[e]) || (ym[e] > 1.0f)) ? 1.0f : 0.0f;
        }
        if (C == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const SampleTaps t = make_taps(xm[e], ym[e], H, W);
                o[e] = sample4(img, t, 1, 0);
            }
        }
        if (VEC == 4 && xb + 3 < W) {
            *reinterpret_cast<float4*>(x_map + rowoff + xb) = make_float4(xm[0], xm[1], xm[2], xm[3]);
            *reinterpret_cast<float4*>(y_map + rowoff + xb) = make_float4(ym[0], ym[1], ym[2], ym[3]);
            *reinterpret_cast<float4*>(black + rowoff + xb) = make_float4(bl[0], bl[1], bl[2], bl[3]);
            if (C == 1) *reinterpret_cast<float4*>(out + rowoff + xb) = make_float4(o[0], o[1], o[2], o[3]);
            if (PUSH) {
                const size_t hw = (size_t)H * W, pix = (size_t)y * W + xb;
                const float4 f = make_float4(o[0] + bl[0] * -1.0f, o[1] + bl[1] * -1.0f, o[2] + bl[2] * -1.0f,
                                             o[3] + bl[3] * -1.0f);                                    // :293
                const size_t slot = ((size_t)n * push.depth + head) * hw + pix;
                *reinterpret_cast<float4*>(push.frames + slot) = f;                                    // :322
                *reinterpret_cast<float4*>(push.masks + slot) = make_float4(bl[0], bl[1], bl[2], bl[3]);   // :323
                *reinterpret_cast<float4*>(push.frame_out + (size_t)n * hw + pix) = f;
                if (push.all_black != nullptr) {                                                       // :291
                    int4* ab = reinterpret_cast<int4*>(push.all_black + (size_t)n * hw + pix);
                    int4 v = *ab;
                    v.x += (int)bl[0]; v.y += (int)bl[1]; v.z += (int)bl[2]; v.w += (int)bl[3];
                    *ab = v;
                }
            }
        } else {
            for (int e = 0; e < 4 && xb + e < W; ++e) {
                x_map[rowoff + xb + e] = xm[e];
                y_map[rowoff + xb + e] = ym[e];
                black[rowoff + xb + e] = bl[e];
                if (C == 1) out[rowoff + xb + e] = o[e];
            }
        }
        if (C != 1) {
            for (int e = 0; e < 4 && xb + e < W; ++e) {
                const SampleTaps t = make_taps(xm[e], ym[e], H, W);
                for (int ch = 0; ch < C; ++ch) out[(rowoff + xb + e) * C + ch] = sample4(img, t, C, ch);
            }
        }
    }
    if (PUSH) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = gridDim.x * gridDim.y * gridDim.z;
            const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(push.ticket), 1u);
            if (old == total - 1) {                       // every block has read `head` (before its own ticket)
                *push.ticket = 0;
                *push.head_rw = (head + 1) % push.depth;
            }
        }
    }
}

// interpolate(im, x, y): caller-supplied maps (spatial_transformer.py:200-281).  Same tiling.
__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ im, const float* __restrict__ xs,
                                                     const float* __restrict__ ys, int H, int W, int C,
                                                     float* __restrict__ out) {
    const int n = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wv;
    const int xb = blockIdx.x * 256 + lane * 4;
    if (y >= H || xb >= W) return;
    const float* img = im + (size_t)n * H * W * C;
    const size_t rowoff = ((size_t)n * H + y) * W;
    if (C == 1 && (W & 3) == 0) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + rowoff + xb);
        const float4 yv = *reinterpret_cast<const float4*>(ys + rowoff + xb);
        float4 o;
        o.x = sample4(img, make_taps(xv.x, yv.x, H, W), 1, 0);
        o.y = sample4(img, make_taps(xv.y, yv.y, H, W), 1, 0);
        o.z = sample4(img, make_taps(xv.z, yv.z, H, W), 1, 0);
        o.w = sample4(img, make_taps(xv.w, yv.w, H, W), 1, 0);
        *reinterpret_cast<float4*>(out + rowoff + xb) = o;
    } else {
        for (int e = 0; e < 4 && xb + e < W; ++e) {
            const SampleTaps t = make_taps(xs[rowoff + xb + e], ys[rowoff + xb + e], H, W);
            for (int ch = 0; ch < C; ++ch) out[(rowoff + xb + e) * C + ch] = sample4(img, t, C, ch);
        }
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
                       hipStream_t st, float* pts1) {
    const long waves = (long)N * gh * gw;
    mesh_homography_kernel<<<cdiv(waves * 64, 256), 256, 0, st>>>(in, is_theta, N, gh, gw, lim, pts2, Hs, pts1);
    SN_LAUNCH_CHECK("mesh_homography_kernel");
    return STABNET_OK;
}

int launch_sample(const float* Hs, const float* src, int N, int H, int W, int C, int gh, int gw, float* out,
                         float* black, float* x_map, float* y_map, hipStream_t st) {
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    const WarpPush none{};
    if ((W & 3) == 0)
        warp_sample_kernel<4, 0><<<grid, 256, 0, st>>>(Hs, src, H, W, C, gh, gw, out, black, x_map, y_map, none);
    else
        warp_sample_kernel<1, 0><<<grid, 256, 0, st>>>(Hs, src, H, W, C, gh, gw, out, black, x_map, y_map, none);
    SN_LAUNCH_CHECK("warp_sample_kernel");
    return STABNET_OK;
}

// The online loop's last pass: sample + feedback push + ring-head advance in one launch (C = 1, W % 4 == 0).
int launch_sample_push(const float* Hs, const float* src, int N, int H, int W, int gh, int gw, float* out, float* black,
                       float* x_map, float* y_map, const WarpPush& push, hipStream_t st) {
    SN_REQUIRE((W & 3) == 0 && push.frames && push.masks && push.frame_out && push.head && push.head_rw && push.ticket &&
                   push.depth > 0, "sample_push: bad arguments");
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    warp_sample_kernel<4, 1><<<grid, 256, 0, st>>>(Hs, src, H, W, 1, gh, gw, out, black, x_map, y_map, push);
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
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    interp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(im, x, y, H, W, C, out);
    SN_LAUNCH_CHECK("interp_kernel");
    return STABNET_OK;
}

}  // extern "C"
