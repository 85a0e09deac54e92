// Backward of the multi-grid warp and of the flow sampler, and the per-tower / temporal loss kernels (gfx950).
// What TF autodiff does for `opt.minimize(total_loss)` (train_bundle_nobm.py:160) over
//   transformer(x, pts2)            spatial_transformer3.py:218-301   -> d theta           (sampler weights -> maps -> H -> ridge inverse -> vertices -> clip)
//   interpolate(out2, flow)         spatial_transformer.py:200-281    -> d im              (scatter-add of the 4 taps)
//   feature / img / temporal / mesh losses   s_net_bundle_nobm.py:139-210,335-352, train_bundle_nobm.py:110-125
// floor / casts / comparisons carry no gradient: corners, black_pix, the z sign and warp_pts indices are constants.
// All kernels are HBM-bound or tiny.  Every cross-thread reduction is ORDER-INDEPENDENT, so a training step is
// reproducible bit for bit: sums that many blocks contribute to are accumulated as 64-bit FIXED-POINT integers
// (integer addition is associative; scale 2^40: resolution 9.1e-13, range +-8.3e6 -- gradients of this objective are
// 1e-9 .. 1e2), or as block partials reduced in a fixed order by a second small kernel.  A contribution that is NaN, Inf or
// >= 2^22 in magnitude cannot be represented: it sets a POISON word behind the accumulators (atomic OR: still
// order-independent) and the read-out kernels then emit NaN, so a diverging step shows up as NaN gradients -- as it would
// with float atomics -- instead of finite garbage that Adam applies.
#include "common.h"
#include <algorithm>
#include <climits>

#define SN_MAX_CELLS 64
#define SN_FIX_SCALE 1099511627776.0            // 2^40
#define SN_FIX_INV (1.0 / 1099511627776.0)

typedef unsigned long long sn_u64;
__device__ __forceinline__ sn_u64 to_fix(double v, unsigned* poison) {
    if (!(fabs(v) < 4194304.0)) {                 // NaN, +-Inf, or outside what the 64-bit accumulator can hold
        atomicOr(poison, 1u);
        return 0ull;
    }
    return (sn_u64)(long long)__double2ll_rn(v * SN_FIX_SCALE);
}
__device__ __forceinline__ double from_fix(sn_u64 v) { return (double)(long long)v * SN_FIX_INV; }

__device__ __forceinline__ float wshfl_b(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int cvt_i32_x86_b(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// ---------------------------------------------------------------------------------------------------------
// Stage A: per pixel, d(out,x_map,y_map) -> contributions to dH[cell][0..7]; reduced per block, then atomics.
// Same tiling as the forward sampler: 4 rows x 256 columns per block, 4 pixels per thread.
__global__ __launch_bounds__(256) void warp_bwd_pixels_kernel(const float* __restrict__ Hs, const float* __restrict__ src,
                                                              const float* __restrict__ x_map,
                                                              const float* __restrict__ y_map,
                                                              const float* __restrict__ d_out,
                                                              const float* __restrict__ d_xmap,
                                                              const float* __restrict__ d_ymap, int H, int W, int C,
                                                              int gh, int gw, sn_u64* __restrict__ dHs, const float* __restrict__ dmap_scale) {
    __shared__ float sH[SN_MAX_CELLS * 9];
    __shared__ sn_u64 sAcc[SN_MAX_CELLS * 8];      // fixed point (the 8x8 adjoint solve amplifies rounding of these sums: no float32)
    const int n = blockIdx.z;
    const int cells = gh * gw;
    for (int i = threadIdx.x; i < cells * 9; i += 256) sH[i] = Hs[(size_t)n * cells * 9 + i];
    for (int i = threadIdx.x; i < cells * 8; i += 256) sAcc[i] = 0ull;
    __syncthreads();
    const float mscale = (dmap_scale != nullptr) ? dmap_scale[n] : 1.0f;   // d_xmap / d_ymap hold counts: gradient = count * scale[n]
    unsigned* poison = reinterpret_cast<unsigned*>(dHs + (size_t)gridDim.z * cells * 8);     // the word behind the accumulators

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wv;
    const int xb = blockIdx.x * 256 + lane * 4;
    const bool row_ok = y < H;
    const int chh = H / gh, cww = W / gw;
    const int ci = min(min(y, H - 1) / chh, gh - 1);
    const float stepx = (W > 1) ? 2.0f / (float)(W - 1) : 0.0f;
    const float stepy = (H > 1) ? 2.0f / (float)(H - 1) : 0.0f;
    const float gy = -1.0f + stepy * (float)min(y, H - 1);
    const float* img = src + (size_t)n * H * W * C;
    const size_t rowoff = ((size_t)n * H + min(y, H - 1)) * W;

    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int my_cell = -1;
    for (int e = 0; e < 4; ++e) {
        const int xx = xb + e;
        if (!row_ok || xx >= W) continue;
        const int cj = min(xx / cww, gw - 1);
        const int cell = ci * gw + cj;
        const float* h = sH + cell * 9;
        const float gx = -1.0f + stepx * (float)xx;
        const float xm = x_map[rowoff + xx], ym = y_map[rowoff + xx];
        float tz = (h[6] * gx + h[7] * gy) + h[8];
        tz = tz + (((tz >= 0.0f) ? 1.0f : 0.0f) * 2.0f - 1.0f) * 1e-8f;
        // sampler corners (constants of the graph) and d out / d (xp, yp)
        const float xp = (xm + 1.0f) * (float)W / 2.0f, yp = (ym + 1.0f) * (float)H / 2.0f;
        int x0 = cvt_i32_x86_b(floorf(xp)), y0 = cvt_i32_x86_b(floorf(yp));
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
        y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        float dxp = 0.f, dyp = 0.f;
        if (d_out != nullptr) {
            for (int ch = 0; ch < C; ++ch) {
                const float g = d_out[(rowoff + xx) * C + ch];
                const float Ia = img[((size_t)y0 * W + x0) * C + ch], Ib = img[((size_t)y1 * W + x0) * C + ch];
                const float Ic = img[((size_t)y0 * W + x1) * C + ch], Id = img[((size_t)y1 * W + x1) * C + ch];
                dxp += g * ((y1f - yp) * (Ic - Ia) + (yp - y0f) * (Id - Ib));
                dyp += g * ((x1f - xp) * (Ib - Ia) + (xp - x0f) * (Id - Ic));
            }
        }
        float dxm = dxp * (float)W / 2.0f, dym = dyp * (float)H / 2.0f;
        if (d_xmap != nullptr) dxm += d_xmap[rowoff + xx] * mscale;
        if (d_ymap != nullptr) dym += d_ymap[rowoff + xx] * mscale;
        const float dtx = dxm / tz, dty = dym / tz, dtz = -(dxm * xm + dym * ym) / tz;
        const float c[8] = {dtx * gx, dtx * gy, dtx, dty * gx, dty * gy, dty, dtz * gx, dtz * gy};
        if (my_cell == -1 || my_cell == cell) {
            my_cell = cell;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += c[k];
        } else {                                           // 4-pixel group straddles a cell seam (rare)
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&sAcc[cell * 8 + k], to_fix((double)c[k], poison));
        }
    }
    // wave reduction when the whole wave sits in one cell (the common case), LDS atomics otherwise
    int first = my_cell;                                   // largest cell id in the wave (-1 = lane has no pixel)
    for (int off = 32; off >= 1; off >>= 1) first = max(first, __shfl_xor(first, off, 64));
    const bool uniform = __all(my_cell == first || my_cell == -1) && first >= 0;
    if (uniform) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double v = (double)acc[k];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) atomicAdd(&sAcc[first * 8 + k], to_fix(v, poison));       // (the butterfly order is fixed)
        }
    } else if (my_cell >= 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&sAcc[my_cell * 8 + k], to_fix((double)acc[k], poison));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells * 8; i += 256) {
        const sn_u64 v = sAcc[i];
        if (v != 0ull) atomicAdd(&dHs[(size_t)n * cells * 8 + i], v);
    }
}

// Stage B: one block per sample, one wave per cell (waves loop when there are more than 16 cells):
// dH[0..7] -> d(target vertices) through h = inv(A + 1e-4 I) b:
//   lambda = inv(A)^T dH ;  dL/db = lambda ;  dL/dA = -lambda h^T ; b = [u,v], A[q][6..7] = -(x,y)_q u_q, A[4+q][6..7] = -(x,y)_q v_q.
// The 8 corner contributions of every cell go to LDS; each vertex then sums its (up to four) cells in the fixed order
// TL, TR, BL, BR of the cells around it and writes d_pts2 [N, gh+1, gw+1, 2] once: no atomics, no pre-zeroing.
__device__ __forceinline__ double wshfl_d(double v, int src) { return __shfl(v, src, 64); }

__global__ __launch_bounds__(1024) void warp_bwd_mesh_kernel(const float* __restrict__ pts2, const float* __restrict__ Hs,
                                                             const sn_u64* __restrict__ dHs, int N, int gh, int gw,
                                                             float* __restrict__ d_pts2) {
    __shared__ float sC[SN_MAX_CELLS * 8];               // [cell][u_TL,u_TR,u_BL,u_BR,v_TL,v_TR,v_BL,v_BR]
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int cells = gh * gw;
    const int nwaves = blockDim.x >> 6;
  for (int cell = threadIdx.x >> 6; cell < cells; cell += nwaves) {
    const int ci = cell / gw, cj = cell % gw;
    const double hh = 2.0 / gh, ww = 2.0 / gw;
    const int nv = (gh + 1) * (gw + 1);
    double sx[4], sy[4], tu[4], tv[4];
    int vid[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vi = ci + (q >> 1), vj = cj + (q & 1);
        const double bx = cj * ww - 1.0, by = ci * hh - 1.0;
        sx[q] = (double)(float)((q & 1) ? bx + ww : bx);
        sy[q] = (double)(float)((q >> 1) ? by + hh : by);
        vid[q] = vi * (gw + 1) + vj;
        tu[q] = (double)pts2[((size_t)n * nv + vid[q]) * 2 + 0];
        tv[q] = (double)pts2[((size_t)n * nv + vid[q]) * 2 + 1];
    }
    const int r = lane >> 3, c = lane & 7;
    const int q = r & 3;
    const double tq = (r < 4) ? tu[q] : tv[q];
    double a;
    {
        const int cc = (r < 4) ? c : c - 3;
        double val = 0.0;
        if (c < 6) {
            if (cc == 0) val = sx[q];
            else if (cc == 1) val = sy[q];
            else if (cc == 2) val = 1.0;
            if ((r < 4 && c >= 3) || (r >= 4 && c < 3)) val = 0.0;
        } else if (c == 6) {
            val = (-sx[q]) * tq;
        } else {
            val = (-sy[q]) * tq;
        }
        a = val + ((r == c) ? (double)1e-4f : 0.0);
    }
    double x = (r == c) ? 1.0 : 0.0;
    for (int k = 0; k < 8; ++k) {                        // inverse of (A + ridge), float64, partial pivoting
        double best = -1.0;
        int piv = k;
        for (int rr = k; rr < 8; ++rr) {
            const double v = fabs(wshfl_d(a, rr * 8 + k));
            if (v > best) { best = v; piv = rr; }
        }
        const int src = (r == k) ? piv * 8 + c : ((r == piv) ? k * 8 + c : lane);
        a = wshfl_d(a, src);
        x = wshfl_d(x, src);
        const double pivot = wshfl_d(a, k * 8 + k);
        if (r > k && c == k) a = a / pivot;
        const double l = wshfl_d(a, r * 8 + k);
        const double u = wshfl_d(a, k * 8 + c);
        if (r > k && c > k) a = a - l * u;
    }
    for (int i = 0; i < 8; ++i) {
        const double b = wshfl_d(x, i * 8 + c);
        const double l = wshfl_d(a, r * 8 + i);
        if (r > i) x = x - b * l;
    }
    for (int i = 7; i >= 0; --i) {
        const double d = wshfl_d(a, i * 8 + i);
        if (r == i) x = x / d;
        const double b = wshfl_d(x, i * 8 + c);
        const double u = wshfl_d(a, r * 8 + i);
        if (r < i) x = x - b * u;
    }
    // lambda_j = sum_k X[k][j] * g[k]   (X = inv(A), lane (k, j) holds X[k][j])
    const sn_u64* g = dHs + ((size_t)n * cells + cell) * 8;
    double lam = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) lam += wshfl_d(x, k * 8 + c) * from_fix(g[k]);     // valid in every lane with column c
    if (*reinterpret_cast<const unsigned*>(dHs + (size_t)N * cells * 8) != 0u) lam = __builtin_nan("");   // poisoned sums (see to_fix)
    const float* h = Hs + ((size_t)n * cells + cell) * 9;
    const double h6 = (double)h[6], h7 = (double)h[7];
    if (r == 0) {                                        // lanes 0..7: c = index j of lambda
        const int qq = c & 3;
        const double dv = lam + lam * (h6 * sx[qq] + h7 * sy[qq]);         // d/du_q (c<4) or d/dv_q (c>=4)
        sC[cell * 8 + c] = (float)dv;
    }
  }
    __syncthreads();
    const int nvv = (gh + 1) * (gw + 1);
    for (int t = threadIdx.x; t < nvv * 2; t += blockDim.x) {
        const int v = t >> 1, d = t & 1;
        const int vi = v / (gw + 1), vj = v % (gw + 1);
        float acc = 0.f;
        // cells around the vertex in the order TL, TR, BL, BR; in cell (ci, cj) the vertex is corner q = (vi-ci)*2 + (vj-cj)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = vi - 1 + (k >> 1), cj = vj - 1 + (k & 1);
            if (ci < 0 || cj < 0 || ci >= gh || cj >= gw) continue;
            const int q = (vi - ci) * 2 + (vj - cj);
            acc += sC[(ci * gw + cj) * 8 + d * 4 + q];
        }
        d_pts2[((size_t)n * nvv + v) * 2 + d] = acc;
    }
}

// interpolate() backward wrt the image: acc[tap] += w_tap * d_out as 64-bit fixed point (a scatter has no fixed order:
// integer adds make the sum order-independent), then d_im (+)= acc in a second pass.
// Tiling as the forward sampler (warp.hip): one wave = 128 consecutive pixels of a row, lane-strided, so the four atomic
// instructions of a pixel column each cover 64 neighbouring accumulators (the flow is close to the identity) instead of
// 16-B-strided ones: the atomics, not the arithmetic, set this kernel's time (180 us per step with 4 consecutive pixels per thread).
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                                         const float* __restrict__ d_out, int H, int W, int C,
                                                         sn_u64* __restrict__ acc_fix) {
    constexpr int PX = 2;
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int segs = (W + 64 * PX - 1) / (64 * PX);
    const int wseg = blockIdx.x * 4 + wv;
    const int y = wseg / segs;
    if (y >= H) return;
    const int x0s = (wseg - y * segs) * (64 * PX);
    const size_t rowoff = ((size_t)n * H + y) * W;
    sn_u64* dimg = acc_fix + (size_t)n * H * W * C;
    unsigned* poison = reinterpret_cast<unsigned*>(acc_fix + (size_t)gridDim.y * H * W * C);   // the word behind the accumulators
#pragma unroll
    for (int e = 0; e < PX; ++e) {
        const int xx = x0s + e * 64 + lane;
        if (xx >= W) continue;
        const float xm = xs[rowoff + xx], ym = ys[rowoff + xx];
        const float xp = (xm + 1.0f) * (float)W / 2.0f, yp = (ym + 1.0f) * (float)H / 2.0f;
        int x0 = cvt_i32_x86_b(floorf(xp)), y0 = cvt_i32_x86_b(floorf(yp));
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
        y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        const float wa = (x1f - xp) * (y1f - yp), wb = (x1f - xp) * (yp - y0f);
        const float wc = (xp - x0f) * (y1f - yp), wd = (xp - x0f) * (yp - y0f);
        for (int ch = 0; ch < C; ++ch) {
            const float g = d_out[(rowoff + xx) * C + ch];
            if (g == 0.f) continue;                          // (NaN != 0: a NaN gradient goes on and poisons the sums)
            atomicAdd(&dimg[((size_t)y0 * W + x0) * C + ch], to_fix((double)(wa * g), poison));
            atomicAdd(&dimg[((size_t)y1 * W + x0) * C + ch], to_fix((double)(wb * g), poison));
            atomicAdd(&dimg[((size_t)y0 * W + x1) * C + ch], to_fix((double)(wc * g), poison));
            atomicAdd(&dimg[((size_t)y1 * W + x1) * C + ch], to_fix((double)(wd * g), poison));
        }
    }
}
// The same scatter for one channel with the accumulators of a 16 x 128 pixel tile (+ a 4 x 8 pixel halo) in LDS: a flow close to
// the identity lands nearly every tap inside the window of its own tile, so the four 64-bit atomics of a pixel become LDS
// atomics and each window element goes to memory ONCE (3 456 atomics per tile instead of 8 192, none of them contended: in the
// direct form neighbouring pixels hit the same accumulators at the same time).  Taps outside the window go to memory directly:
// any flow gives the same sums (integer adds), only the speed depends on it.
constexpr int IB_TH = 16, IB_TW = 128, IB_HY = 4, IB_HX = 8, IB_WH = IB_TH + 2 * IB_HY, IB_WW = IB_TW + 2 * IB_HX;
__global__ __launch_bounds__(256) void interp_bwd_tiled_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                                               const float* __restrict__ d_out, int H, int W,
                                                               sn_u64* __restrict__ acc_fix) {
    __shared__ sn_u64 win[IB_WH * IB_WW];
    const int n = blockIdx.y;
    const int tiles_x = (W + IB_TW - 1) / IB_TW;
    const int ty0 = (int)(blockIdx.x / tiles_x) * IB_TH, tx0 = (int)(blockIdx.x % tiles_x) * IB_TW;
    const int wy0 = ty0 - IB_HY, wx0 = tx0 - IB_HX;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    sn_u64* dimg = acc_fix + (size_t)n * H * W;
    unsigned* poison = reinterpret_cast<unsigned*>(acc_fix + (size_t)gridDim.y * H * W);
    for (int i = tid; i < IB_WH * IB_WW; i += 256) win[i] = 0ull;
    __syncthreads();
    auto add = [&](int y, int x, float v) {
        const sn_u64 f = to_fix((double)v, poison);
        const int ly = y - wy0, lx = x - wx0;
        if ((unsigned)ly < (unsigned)IB_WH && (unsigned)lx < (unsigned)IB_WW) atomicAdd(&win[ly * IB_WW + lx], f);
        else atomicAdd(&dimg[(size_t)y * W + x], f);
    };
    for (int r = wv; r < IB_TH; r += 4) {
        const int y = ty0 + r;
        if (y >= H) break;
        const size_t rowoff = ((size_t)n * H + y) * W;
#pragma unroll
        for (int e = 0; e < IB_TW / 64; ++e) {
            const int xx = tx0 + e * 64 + lane;
            if (xx >= W) continue;
            const float g = d_out[rowoff + xx];
            if (g == 0.f) continue;                          // (NaN != 0: a NaN gradient goes on and poisons the sums)
            const float xm = xs[rowoff + xx], ym = ys[rowoff + xx];
            const float xp = (xm + 1.0f) * (float)W / 2.0f, yp = (ym + 1.0f) * (float)H / 2.0f;
            int x0 = cvt_i32_x86_b(floorf(xp)), y0 = cvt_i32_x86_b(floorf(yp));
            int x1 = x0 + 1, y1 = y0 + 1;
            x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
            y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
            const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
            const float wa = (x1f - xp) * (y1f - yp), wb = (x1f - xp) * (yp - y0f);
            const float wc = (xp - x0f) * (y1f - yp), wd = (xp - x0f) * (yp - y0f);
            add(y0, x0, wa * g);
            add(y1, x0, wb * g);
            add(y0, x1, wc * g);
            add(y1, x1, wd * g);
        }
    }
    __syncthreads();
    for (int i = tid; i < IB_WH * IB_WW; i += 256) {
        const sn_u64 v = win[i];
        if (v == 0ull) continue;
        const int y = wy0 + i / IB_WW, x = wx0 + i % IB_WW;  // (only in-frame taps were ever added)
        atomicAdd(&dimg[(size_t)y * W + x], v);
    }
}

__global__ __launch_bounds__(256) void fix_to_float_kernel(const sn_u64* __restrict__ acc_fix, long n, float* __restrict__ out,
                                                           int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (*reinterpret_cast<const unsigned*>(acc_fix + n) != 0u) ? __builtin_nanf("") : (float)from_fix(acc_fix[i]);
    out[i] = accumulate ? out[i] + v : v;
}

// ---------------------------------------------------------------------------------------------------------
// Per-sample sums for the masked-MSE losses.  mode 0 (img loss): a = out, b = y, m = 1 - black.
// mode 1 (temporal): a = out1, b = o2w, m = (1 - black1) * nb2w.   sums[n] = { sum((a-b)m)^2, sum m }: block partials
// (fixed in-block tree), then one wave per sample adds the partials in block order (float64) -- reproducible.
__global__ __launch_bounds__(256) void masked_mse_sums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ black,
                                                              const float* __restrict__ m2, long hw,
                                                              float* __restrict__ partial) {
    const int n = blockIdx.y;
    float s0 = 0.f, s1 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const size_t o = (size_t)n * hw + i;
        float m = 1.0f - black[o];
        if (m2 != nullptr) m = m * m2[o];
        const float e = (a[o] - b[o]) * m;
        s0 += e * e;
        s1 += m;
    }
    __shared__ float red[2][4];
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* p = partial + ((size_t)n * gridDim.x + blockIdx.x) * 2;
        p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(64) void masked_mse_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                                 float* __restrict__ sums) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int k = lane; k < nblocks; k += 64) {                 // lane-strided, then a fixed butterfly
        s0 += (double)partial[((size_t)n * nblocks + k) * 2];
        s1 += (double)partial[((size_t)n * nblocks + k) * 2 + 1];
    }
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane == 0) { sums[n * 2] = (float)s0; sums[n * 2 + 1] = (float)s1; }
}

// d/d a of  coef * sum_n [ sum((a-b)m)^2 / (sum m + 1e-8) ]:  g = coef * 2 (a-b) m^2 / den_n;  ga (+)= g, gb = -g.
__global__ __launch_bounds__(256) void masked_mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ black,
                                                              const float* __restrict__ m2,
                                                              const float* __restrict__ sums, float coef, long hw,
                                                              float* __restrict__ ga, int accumulate_a,
                                                              float* __restrict__ gb) {
    const int n = blockIdx.y;
    const float den = sums[n * 2 + 1] + 1e-8f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const size_t o = (size_t)n * hw + i;
        float m = 1.0f - black[o];
        if (m2 != nullptr) m = m * m2[o];
        const float g = coef * 2.0f * (a[o] - b[o]) * m * m / den;
        if (ga != nullptr) ga[o] = accumulate_a ? ga[o] + g : g;
        if (gb != nullptr) gb[o] = -g;
    }
}

// Feature loss (s_net_bundle_nobm.py:215-230,335-343): one block per sample.
// value[n] = sum_i mask_i (|xm - ux| + |ym - uy|) / max(sum mask, 1).  Gradient wrt the maps at the rounded pixel: the maps
// d_xmap / d_ymap receive SIGNED COUNTS (+-mask_i; sums of small integers are exact in float32 whatever the order of the
// atomics) and dscale[n] = gcoef / max(sum mask, 1) is the factor the consumer multiplies them with (stabnet_transformer_bwd).
// 1024 threads, FL_U matches per thread per round with all their loads issued together: a match is a chain of two dependent
// HBM round trips (the match, then the map gather at its pixel); one at a time, 12 rounds of 256 threads took 33 us per tower.
constexpr int FL_T = 1024, FL_U = 3;
__global__ __launch_bounds__(FL_T) void feature_loss_kernel(const float* __restrict__ matches, const float* __restrict__ mask,
                                                            const float* __restrict__ x_map,
                                                            const float* __restrict__ y_map, int H, int W, int Mx,
                                                            float gcoef, float* __restrict__ value,
                                                            float* __restrict__ d_xmap, float* __restrict__ d_ymap,
                                                            float* __restrict__ warped, float* __restrict__ dscale) {
    const int n = blockIdx.x;
    __shared__ float red[FL_T / 64];
    __shared__ float s_cnt;
    float c = 0.f, s = 0.f;
    for (int i0 = threadIdx.x; i0 < Mx; i0 += FL_T * FL_U) {
        float4 mt[FL_U];
        float mk[FL_U], xm[FL_U], ym[FL_U];
        size_t o[FL_U];
#pragma unroll
        for (int j = 0; j < FL_U; ++j) {
            const int i = min(i0 + j * FL_T, Mx - 1);
            mt[j] = *reinterpret_cast<const float4*>(matches + ((size_t)n * Mx + i) * 4);
            mk[j] = (i0 + j * FL_T < Mx) ? mask[(size_t)n * Mx + i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < FL_U; ++j) {
            float px = (mt[j].x + 1.0f) / 2.0f * (float)W;
            float py = (mt[j].y + 1.0f) / 2.0f * (float)H;
            px = fminf(fmaxf(px, 0.0f), (float)(W - 1));
            py = fminf(fmaxf(py, 0.0f), (float)(H - 1));
            const int xi = (int)rintf(px), yi = (int)rintf(py);        // tf.round: half to even
            o[j] = ((size_t)n * H + yi) * W + xi;
            xm[j] = x_map[o[j]];
            ym[j] = y_map[o[j]];
        }
#pragma unroll
        for (int j = 0; j < FL_U; ++j) {
            const int i = i0 + j * FL_T;
            if (i >= Mx) continue;
            if (warped != nullptr) { warped[((size_t)n * Mx + i) * 2] = xm[j]; warped[((size_t)n * Mx + i) * 2 + 1] = ym[j]; }
            const float ex = xm[j] - mt[j].z, ey = ym[j] - mt[j].w;
            c += mk[j];
            s += (fabsf(ex) + fabsf(ey)) * mk[j];
            if (d_xmap != nullptr && mk[j] != 0.f) {
                if (ex != 0.f) atomicAdd(&d_xmap[o[j]], ex > 0.f ? mk[j] : -mk[j]);
                if (ey != 0.f) atomicAdd(&d_ymap[o[j]], ey > 0.f ? mk[j] : -mk[j]);
            }
        }
    }
    // the two block sums in a fixed order: lanes, then the 16 waves in wave order
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); s += __shfl_xor(s, off, 64); }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int w = 1; w < FL_T / 64; ++w) t += red[w];
        s_cnt = fmaxf(t, 1.0f);
    }
    __syncthreads();
    const float cnt = s_cnt;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int w = 1; w < FL_T / 64; ++w) t += red[w];
        value[n] = t / cnt;
        if (dscale != nullptr) dscale[n] = gcoef / cnt;
    }
}

// Mesh losses and the final assembly of d theta (one thread per sample; N is small).
//   losses[0] id2 = mean|theta| * id_mul      (s_net_bundle_nobm.py:263)      losses[1] black_pos (mean hinge^2 * use_black)
//   losses[2] distortion (:166-181)           losses[3] consistency (:183-210)
// d_theta = clip_mask * (d_pts2_warp + w_dist * d dist + w_cons * d cons + w_black * d black) + w_id * sign(theta)/(N*nt)
// One block; a sample = 32 lanes, eight samples at a time.  Phases (barrier between them): vertices (clip + id term) -> LDS;
// cells (distortion triples + the black hinge, corner gradients in registers) and per-vertex second differences -> LDS;
// vertices GATHER their gradient from the (up to four) cells and the (up to twelve) second-difference terms that touch them, in a
// fixed order: no read-modify-write chains (the first form -- one thread per sample walking every term and accumulating into an
// LDS array -- spent 51 us per tower in ~1 600 dependent LDS updates; private arrays, indexed dynamically, land in scratch: 128 us).
#define SN_ML_LANES 32
#define SN_ML_SAMPLES 8
__global__ __launch_bounds__(SN_ML_LANES * SN_ML_SAMPLES) void mesh_losses_kernel(
    const float* __restrict__ theta, const float* __restrict__ d_pts2_warp, int N, int gh, int gw, float lim, float id_mul,
    float w_id, float w_dist, float w_cons, float use_black, float w_black, float* __restrict__ losses, float* __restrict__ d_theta) {
    constexpr int NVMAX = 81, NCMAX = 64;
    __shared__ float sp[SN_ML_SAMPLES][2 * NVMAX];            // clipped vertices
    __shared__ float sdc[SN_ML_SAMPLES][NCMAX][8];            // per cell: d/d corner q, axis d -> [2 q + d]
    __shared__ float se[SN_ML_SAMPLES][NVMAX][4][2];          // second differences e(v, t, d) (0 where the term does not exist)
    __shared__ float spart[SN_ML_SAMPLES][4][NVMAX];          // per vertex / per cell loss partials: id, black, dist, cons
    __shared__ float red[4][64];
    const int ls = threadIdx.x / SN_ML_LANES, l = threadIdx.x % SN_ML_LANES;
    const int nv = (gh + 1) * (gw + 1), nt = nv * 2, cells = gh * gw;
    const double hh = 2.0 / gh, ww = 2.0 / gw;
    const float cb = 2.0f * use_black / (float)(N * gh * gw * 8);
    const float kq0 = (float)(hh / ww), kq1 = (float)(ww / hh);
    const float cd = 1.0f / (8.0f * (float)(N * gh * gw) * 2.0f);   // mean over [N*cells, 2] then / 8
    int ncons = 0;
    for (int i = 0; i <= gh; ++i)
        for (int j = 0; j <= gw; ++j) ncons += (i > 1) + (j > 1) + (i < gh - 1) + (j < gw - 1);
    const float cc = (ncons > 0) ? 1.0f / ((float)N * 2.0f * (float)ncons) : 0.f;
    for (int n0 = 0; n0 < N; n0 += SN_ML_SAMPLES) {
        const int n = n0 + ls;
        const bool live = n < N;
        // ---- A: vertices
        if (live)
            for (int v = l; v < nv; v += SN_ML_LANES) {
                const int i = v / (gw + 1), j = v - i * (gw + 1);
                const float tx = theta[(size_t)n * nt + 2 * v], ty = theta[(size_t)n * nt + 2 * v + 1];
                sp[ls][2 * v] = fminf(fmaxf((float)(j * ww - 1.0) + tx, -lim), lim);
                sp[ls][2 * v + 1] = fminf(fmaxf((float)(i * hh - 1.0) + ty, -lim), lim);
                spart[ls][0][v] = fabsf(tx) + fabsf(ty);
            }
        __syncthreads();
        // ---- B: cells (8 ordered corner triples, R = 90 deg rotation scaled by the cell aspect k; black hinge on the corners)
        if (live)
            for (int c = l; c < cells; c += SN_ML_LANES) {
                const int i = c / gw, j = c - i * gw;
                const int vq[4] = {i * (gw + 1) + j, i * (gw + 1) + j + 1, (i + 1) * (gw + 1) + j, (i + 1) * (gw + 1) + j + 1};
                float px[4], py[4], gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) { px[q] = sp[ls][2 * vq[q]]; py[q] = sp[ls][2 * vq[q] + 1]; }
                float lb = 0.f, ld = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                 // identically 0 after the clip, kept for fidelity (s_net_bundle_nobm.py:139-146)
                    const float ex = (px[q] > lim) ? px[q] - lim : ((-lim > px[q]) ? -lim - px[q] : 0.f);
                    const float ey = (py[q] > lim) ? py[q] - lim : ((-lim > py[q]) ? -lim - py[q] : 0.f);
                    lb += ex * ex * use_black;
                    lb += ey * ey * use_black;
                    if (ex != 0.f) gx[q] += w_black * cb * ex * ((px[q] > lim) ? 1.f : -1.f);
                    if (ey != 0.f) gy[q] += w_black * cb * ey * ((py[q] > lim) ? 1.f : -1.f);
                }
                constexpr int trip[8][3] = {{0, 1, 3}, {1, 3, 2}, {3, 2, 0}, {2, 0, 1}, {1, 0, 2}, {0, 2, 3}, {2, 3, 1}, {3, 1, 0}};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int a0 = trip[t][0], a1 = trip[t][1], a2 = trip[t][2];
                    const bool clock = t >= 4;
                    const float k = (t & 1) ? kq1 : kq0;
                    const float r01 = clock ? k : -k, r10 = clock ? -k : k;      // R = [[0,r01],[r10,0]]
                    const float dx = px[a1] - px[a0], dy = py[a1] - py[a0];
                    const float ex = r01 * dy - (px[a2] - px[a1]);
                    const float ey = r10 * dx - (py[a2] - py[a1]);
                    ld += ex * ex + ey * ey;
                    const float gxx = w_dist * cd * 2.0f * ex, gyv = w_dist * cd * 2.0f * ey;
                    // ex = r01*(p1y - p0y) - p2x + p1x ; ey = r10*(p1x - p0x) - p2y + p1y
                    gy[a1] += gxx * r01; gy[a0] -= gxx * r01; gx[a2] -= gxx; gx[a1] += gxx;
                    gx[a1] += gyv * r10; gx[a0] -= gyv * r10; gy[a2] -= gyv; gy[a1] += gyv;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { sdc[ls][c][2 * q] = gx[q]; sdc[ls][c][2 * q + 1] = gy[q]; }
                spart[ls][1][c] = lb;
                spart[ls][2][c] = ld;
            }
        // ---- B': second differences 2 p[nb0] - p[v] - p[nb1] along both axes, from both ends (s_net_bundle_nobm.py:183-210)
        if (live)
            for (int v = l; v < nv; v += SN_ML_LANES) {
                const int i = v / (gw + 1), j = v - i * (gw + 1);
                const int st[4] = {-(gw + 1), -1, gw + 1, 1};
                const bool on[4] = {i > 1, j > 1, i < gh - 1, j < gw - 1};
                float lc = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float e = 0.f;
                        if (on[t]) e = 2.0f * sp[ls][2 * (v + st[t]) + d] - sp[ls][2 * v + d] - sp[ls][2 * (v + 2 * st[t]) + d];
                        se[ls][v][t][d] = e;
                        lc += e * e;
                    }
                spart[ls][3][v] = lc;
            }
        __syncthreads();
        // ---- C: every vertex gathers: cells around it (TL, TR, BL, BR of the vertex = that cell's corner BR, BL, TR, TL), then the
        //      second-difference terms in which it is the centre (-g), the near neighbour (+2g) or the far neighbour (-g)
        if (live)
            for (int v = l; v < nv; v += SN_ML_LANES) {
                const int i = v / (gw + 1), j = v - i * (gw + 1);
                float g[2] = {0.f, 0.f};
#pragma unroll
                for (int qc = 0; qc < 4; ++qc) {
                    const int ci = i - 1 + (qc >> 1), cj = j - 1 + (qc & 1);       // cell whose corner (3 - qc) is this vertex
                    if (ci < 0 || cj < 0 || ci >= gh || cj >= gw) continue;
                    const int corner = 3 - qc;
                    g[0] += sdc[ls][ci * gw + cj][2 * corner];
                    g[1] += sdc[ls][ci * gw + cj][2 * corner + 1];
                }
                const int st[4] = {-(gw + 1), -1, gw + 1, 1};
                const float kc = w_cons * cc * 2.0f;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) g[d] -= kc * se[ls][v][t][d];                        // centre of (v, t)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {                                                    // near neighbour of (v - st, t)
                        const int u = v - st[t];
                        const int ui = i - ((t == 0) ? -1 : (t == 2) ? 1 : 0), uj = j - ((t == 1) ? -1 : (t == 3) ? 1 : 0);
                        if (ui >= 0 && ui <= gh && uj >= 0 && uj <= gw) g[d] += 2.0f * (kc * se[ls][u][t][d]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {                                                    // far neighbour of (v - 2 st, t)
                        const int u = v - 2 * st[t];
                        const int ui = i - 2 * ((t == 0) ? -1 : (t == 2) ? 1 : 0), uj = j - 2 * ((t == 1) ? -1 : (t == 3) ? 1 : 0);
                        if (ui >= 0 && ui <= gh && uj >= 0 && uj <= gw) g[d] -= kc * se[ls][u][t][d];
                    }
                }
                // clip mask + id loss
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float th = theta[(size_t)n * nt + 2 * v + d];
                    const float raw = ((d == 0) ? (float)(j * ww - 1.0) : (float)(i * hh - 1.0)) + th;
                    const bool pass = (raw >= -lim) && (raw <= lim);
                    float gg = g[d] + (d_pts2_warp ? d_pts2_warp[((size_t)n * nv + v) * 2 + d] : 0.f);
                    gg = pass ? gg : 0.f;
                    gg += w_id * id_mul * ((th > 0.f) ? 1.f : ((th < 0.f) ? -1.f : 0.f)) / (float)(N * nt);
                    if (d_theta != nullptr) d_theta[(size_t)n * nt + 2 * v + d] = gg;
                }
            }
        // ---- per-sample loss sums in index order
        if (live && l < 4) {
            const int cnt = (l == 0 || l == 3) ? nv : cells;
            float sacc = 0.f;
            for (int k = 0; k < cnt; ++k) sacc += spart[ls][l][k];
            if (l == 2) sacc *= cd;
            if (l == 3) sacc *= cc;
            if (l == 1) sacc /= (float)(N * gh * gw * 8);
            red[l][n] = sacc;
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) s += red[threadIdx.x][i];
        if (threadIdx.x == 0) s = s / (float)(N * nt) * id_mul;
        losses[threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
extern "C" {

/* transformer backward (spatial_transformer3.py:218-301 under TF autodiff): d_pts2 [N,gh+1,gw+1,2] (pre-clip vertex
 * gradient).  d_out [N,H,W,C], d_xmap, d_ymap [N,H,W] may each be NULL; dmap_scale [N] (optional) multiplies d_xmap / d_ymap
 * per sample (the feature loss hands over signed counts + a scale).  workspace: N*gh*gw*8 8-byte words, 8-B aligned. */
int stabnet_transformer_bwd(const float* pts2, const float* Hs, const float* U, const float* x_map, const float* y_map,
                            const float* d_out, const float* d_xmap, const float* d_ymap, const float* dmap_scale, int N,
                            int H, int W, int C, int grid_h, int grid_w, float* d_pts2, void* workspace, void* stream) {
    SN_REQUIRE(pts2 && Hs && U && x_map && y_map && d_pts2 && workspace, "transformer_bwd: null pointer");
    SN_REQUIRE(N > 0 && N <= 65535 && H >= grid_h && W >= grid_w && C > 0 && grid_h * grid_w <= SN_MAX_CELLS,
               "transformer_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int cells = grid_h * grid_w;
    SN_REQUIRE(((uintptr_t)workspace & 7) == 0, "transformer_bwd: workspace must be 8-byte aligned");
    sn_u64* dHs = static_cast<sn_u64*>(workspace);
    if (hipMemsetAsync(workspace, 0, sizeof(sn_u64) * ((size_t)N * cells * 8 + 1), st) != hipSuccess) {      // + the poison word
        stabnet_set_error("transformer_bwd: memset failed");
        return STABNET_ERR_LAUNCH;
    }
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    warp_bwd_pixels_kernel<<<grid, 256, 0, st>>>(Hs, U, x_map, y_map, d_out, d_xmap, d_ymap, H, W, C, grid_h, grid_w, dHs,
                                                 dmap_scale);
    SN_LAUNCH_CHECK("warp_bwd_pixels_kernel");
    warp_bwd_mesh_kernel<<<N, std::min(1024, 64 * cells), 0, st>>>(pts2, Hs, dHs, N, grid_h, grid_w, d_pts2);
    SN_LAUNCH_CHECK("warp_bwd_mesh_kernel");
    return STABNET_OK;
}

/* interpolate(im, x, y) backward wrt im (train_bundle_nobm.py:117: the gradient that reaches tower 2's output).
 * workspace: N*H*W*C + 1 8-byte words (fixed-point accumulators + the poison word), 8-B aligned. */
int stabnet_interp_bwd(const float* x, const float* y, const float* d_out, int N, int H, int W, int C, float* d_im,
                       int accumulate, void* workspace, void* stream) {
    SN_REQUIRE(x && y && d_out && d_im && workspace, "interp_bwd: null pointer");
    SN_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && C > 0, "interp_bwd: bad shape");
    SN_REQUIRE(((uintptr_t)workspace & 7) == 0, "interp_bwd: workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * H * W * C;
    if (hipMemsetAsync(workspace, 0, sizeof(sn_u64) * ((size_t)total + 1), st) != hipSuccess) {               // + the poison word
        stabnet_set_error("interp_bwd: memset failed");
        return STABNET_ERR_LAUNCH;
    }
    static const bool tiled = getenv("STABNET_INTERP_BWD_TILED") == nullptr || atoi(getenv("STABNET_INTERP_BWD_TILED")) != 0;   // debug switch
    if (C == 1 && tiled) {
        interp_bwd_tiled_kernel<<<dim3(cdiv(W, IB_TW) * cdiv(H, IB_TH), N), 256, 0, st>>>(x, y, d_out, H, W, static_cast<sn_u64*>(workspace));
        SN_LAUNCH_CHECK("interp_bwd_tiled_kernel");
    } else {
        dim3 grid(cdiv((long)cdiv(W, 128) * H, 4), N, 1);
        interp_bwd_kernel<<<grid, 256, 0, st>>>(x, y, d_out, H, W, C, static_cast<sn_u64*>(workspace));
        SN_LAUNCH_CHECK("interp_bwd_kernel");
    }
    fix_to_float_kernel<<<cdiv(total, 256), 256, 0, st>>>(static_cast<const sn_u64*>(workspace), total, d_im, accumulate);
    SN_LAUNCH_CHECK("fix_to_float_kernel");
    return STABNET_OK;
}

__global__ __launch_bounds__(256) void axpb_kernel(const float* __restrict__ x, float a, float b, long n,
                                                   float* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a * x[i] + b;
}

/* y = a*x + b elementwise (e.g. 1 - black_pix, train_bundle_nobm.py:118). */
int stabnet_axpb(const float* x, float a, float b, long n, float* y, void* stream) {
    SN_REQUIRE(x && y && n > 0, "axpb: bad arguments");
    axpb_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, a, b, n, y);
    SN_LAUNCH_CHECK("axpb_kernel");
    return STABNET_OK;
}

/* Masked MSE used by img_loss (s_net_bundle_nobm.py:347-352; m2 = NULL) and temp_loss (train_bundle_nobm.py:110-125;
 * m2 = interp(1-black2)): sums [N,2] = {sum((a-b)m)^2, sum m}, m = (1-black)*m2.  value = sum_n s0/(s1+1e-8)/batch_size.
 * workspace: stabnet_masked_mse_workspace_bytes(N) (block partials). */
size_t stabnet_masked_mse_workspace_bytes(int N) { return (size_t)std::max(N, 0) * 512 * 2 * sizeof(float); }
int stabnet_masked_mse_sums(const float* a, const float* b, const float* black, const float* m2, int N, long hw,
                            float* sums, void* workspace, void* stream) {
    SN_REQUIRE(a && b && black && sums && workspace && N > 0 && N <= 65535 && hw > 0, "masked_mse_sums: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int bx = (int)std::min<long>(cdiv(hw, 256 * 8), 512);
    masked_mse_sums_kernel<<<dim3(bx, N), 256, 0, st>>>(a, b, black, m2, hw, static_cast<float*>(workspace));
    SN_LAUNCH_CHECK("masked_mse_sums_kernel");
    masked_mse_finalize_kernel<<<N, 64, 0, st>>>(static_cast<const float*>(workspace), bx, sums);
    SN_LAUNCH_CHECK("masked_mse_finalize_kernel");
    return STABNET_OK;
}

/* ga (+)= coef * d/da, gb = -that (either may be NULL). */
int stabnet_masked_mse_grad(const float* a, const float* b, const float* black, const float* m2, const float* sums,
                            float coef, int N, long hw, float* ga, int accumulate_a, float* gb, void* stream) {
    SN_REQUIRE(a && b && black && sums && N > 0 && N <= 65535 && hw > 0, "masked_mse_grad: bad arguments");
    const int bx = (int)std::min<long>(cdiv(hw, 256 * 4), 1024);
    masked_mse_grad_kernel<<<dim3(bx, N), 256, 0, (hipStream_t)stream>>>(a, b, black, m2, sums, coef, hw, ga, accumulate_a, gb);
    SN_LAUNCH_CHECK("masked_mse_grad_kernel");
    return STABNET_OK;
}

/* feature loss value per sample + its gradient wrt the maps as signed counts scattered into d_xmap/d_ymap (zeroed here
 * when given) and the per-sample factor dscale [N] = gcoef / max(sum mask, 1) (gcoef = dL/d(feature_loss) / N, the
 * reduce_mean over the batch): d loss / d x_map = d_xmap * dscale[n]. */
int stabnet_feature_loss(const float* matches, const float* mask, const float* x_map, const float* y_map, int N, int H,
                         int W, int max_matches, float gcoef, float* value, float* d_xmap, float* d_ymap, float* dscale,
                         float* warped, void* stream) {
    SN_REQUIRE(matches && mask && x_map && y_map && value && N > 0 && max_matches > 0, "feature_loss: bad arguments");
    SN_REQUIRE(((uintptr_t)matches & 15) == 0, "feature_loss: matches must be 16-byte aligned");
    SN_REQUIRE((d_xmap == nullptr) == (d_ymap == nullptr), "feature_loss: d_xmap and d_ymap go together");
    SN_REQUIRE(d_xmap == nullptr || dscale != nullptr, "feature_loss: the map gradient needs dscale");
    hipStream_t st = (hipStream_t)stream;
    if (d_xmap != nullptr) {
        if (hipMemsetAsync(d_xmap, 0, sizeof(float) * (size_t)N * H * W, st) != hipSuccess ||
            hipMemsetAsync(d_ymap, 0, sizeof(float) * (size_t)N * H * W, st) != hipSuccess) {
            stabnet_set_error("feature_loss: memset failed");
            return STABNET_ERR_LAUNCH;
        }
    }
    feature_loss_kernel<<<N, FL_T, 0, st>>>(matches, mask, x_map, y_map, H, W, max_matches, gcoef, value, d_xmap, d_ymap, warped,
                                          dscale);
    SN_LAUNCH_CHECK("feature_loss_kernel");
    return STABNET_OK;
}

/* id / black_pos / distortion / consistency losses (losses[0..3], un-weighted except id_mul) and d_theta:
 * clip(+-1/do_crop_rate)-masked vertex gradient (warp part given in d_pts2_warp, may be NULL) + the weighted mesh-loss
 * gradients + w_id * d id2_loss.  N <= 64, (gh+1)(gw+1) <= 81. */
int stabnet_mesh_losses(const float* theta, const float* d_pts2_warp, int N, int grid_h, int grid_w, float do_crop_rate,
                        float id_mul, float w_id, float w_dist, float w_cons, float use_black, float w_black,
                        float* losses4, float* d_theta, void* stream) {
    SN_REQUIRE(theta && losses4, "mesh_losses: null pointer");
    SN_REQUIRE(N > 0 && N <= 64 && (grid_h + 1) * (grid_w + 1) <= 81 && do_crop_rate > 0.f, "mesh_losses: bad shape");
    mesh_losses_kernel<<<1, SN_ML_LANES * SN_ML_SAMPLES, 0, (hipStream_t)stream>>>(theta, d_pts2_warp, N, grid_h, grid_w, 1.0f / do_crop_rate,
                                                                                id_mul, w_id, w_dist, w_cons, use_black, w_black, losses4, d_theta);
    SN_LAUNCH_CHECK("mesh_losses_kernel");
    return STABNET_OK;
}

}  // extern "C"
