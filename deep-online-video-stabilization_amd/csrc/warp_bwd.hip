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
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                                         const float* __restrict__ d_out, int H, int W, int C,
                                                         sn_u64* __restrict__ acc_fix) {
    const int n = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wv;
    const int xb = blockIdx.x * 256 + lane * 4;
    if (y >= H) return;
    const size_t rowoff = ((size_t)n * H + y) * W;
    sn_u64* dimg = acc_fix + (size_t)n * H * W * C;
    unsigned* poison = reinterpret_cast<unsigned*>(acc_fix + (size_t)gridDim.z * H * W * C);   // the word behind the accumulators
    for (int e = 0; e < 4 && xb + e < W; ++e) {
        const float xm = xs[rowoff + xb + e], ym = ys[rowoff + xb + e];
        const float xp = (xm + 1.0f) * (float)W / 2.0f, yp = (ym + 1.0f) * (float)H / 2.0f;
        int x0 = cvt_i32_x86_b(floorf(xp)), y0 = cvt_i32_x86_b(floorf(yp));
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
        y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
        const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
        const float wa = (x1f - xp) * (y1f - yp), wb = (x1f - xp) * (yp - y0f);
        const float wc = (xp - x0f) * (y1f - yp), wd = (xp - x0f) * (yp - y0f);
        for (int ch = 0; ch < C; ++ch) {
            const float g = d_out[(rowoff + xb + e) * C + ch];
            if (g == 0.f) continue;                          // (NaN != 0: a NaN gradient goes on and poisons the sums)
            atomicAdd(&dimg[((size_t)y0 * W + x0) * C + ch], to_fix((double)(wa * g), poison));
            atomicAdd(&dimg[((size_t)y1 * W + x0) * C + ch], to_fix((double)(wb * g), poison));
            atomicAdd(&dimg[((size_t)y0 * W + x1) * C + ch], to_fix((double)(wc * g), poison));
            atomicAdd(&dimg[((size_t)y1 * W + x1) * C + ch], to_fix((double)(wd * g), poison));
        }
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
__global__ __launch_bounds__(256) void feature_loss_kernel(const float* __restrict__ matches, const float* __restrict__ mask,
                                                           const float* __restrict__ x_map,
                                                           const float* __restrict__ y_map, int H, int W, int Mx,
                                                           float gcoef, float* __restrict__ value,
                                                           float* __restrict__ d_xmap, float* __restrict__ d_ymap,
                                                           float* __restrict__ warped, float* __restrict__ dscale) {
    const int n = blockIdx.x;
    __shared__ float red[4];
    __shared__ float s_cnt;
    float c = 0.f;
    for (int i = threadIdx.x; i < Mx; i += 256) c += mask[(size_t)n * Mx + i];
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) s_cnt = fmaxf((red[0] + red[1]) + (red[2] + red[3]), 1.0f);
    __syncthreads();
    const float cnt = s_cnt;
    float s = 0.f;
    for (int i = threadIdx.x; i < Mx; i += 256) {
        const float* mt = matches + ((size_t)n * Mx + i) * 4;
        float px = (mt[0] + 1.0f) / 2.0f * (float)W;
        float py = (mt[1] + 1.0f) / 2.0f * (float)H;
        px = fminf(fmaxf(px, 0.0f), (float)(W - 1));
        py = fminf(fmaxf(py, 0.0f), (float)(H - 1));
        const int xi = (int)rintf(px), yi = (int)rintf(py);            // tf.round: half to even
        const size_t o = ((size_t)n * H + yi) * W + xi;
        const float xm = x_map[o], ym = y_map[o];
        if (warped != nullptr) { warped[((size_t)n * Mx + i) * 2] = xm; warped[((size_t)n * Mx + i) * 2 + 1] = ym; }
        const float ex = xm - mt[2], ey = ym - mt[3];
        const float mk = mask[(size_t)n * Mx + i];
        s += (fabsf(ex) + fabsf(ey)) * mk;
        if (d_xmap != nullptr && mk != 0.f) {
            if (ex != 0.f) atomicAdd(&d_xmap[o], ex > 0.f ? mk : -mk);
            if (ey != 0.f) atomicAdd(&d_ymap[o], ey > 0.f ? mk : -mk);
        }
    }
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        value[n] = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
        if (dscale != nullptr) dscale[n] = gcoef / cnt;
    }
}

// Mesh losses and the final assembly of d theta (one thread per sample; N is small).
//   losses[0] id2 = mean|theta| * id_mul      (s_net_bundle_nobm.py:263)      losses[1] black_pos (mean hinge^2 * use_black)
//   losses[2] distortion (:166-181)           losses[3] consistency (:183-210)
// d_theta = clip_mask * (d_pts2_warp + w_dist * d dist + w_cons * d cons + w_black * d black) + w_id * sign(theta)/(N*nt)
// The per-sample vertex arrays are indexed dynamically: in LDS (template LDSV) they cost ~5 us, as private arrays they live
// in scratch memory (128 us for 8 samples).
template <bool LDSV>
__global__ __launch_bounds__(64) void mesh_losses_kernel(const float* __restrict__ theta, const float* __restrict__ d_pts2_warp,
                                                         int N, int gh, int gw, float lim, float id_mul, float w_id,
                                                         float w_dist, float w_cons, float use_black, float w_black,
                                                         float* __restrict__ losses, float* __restrict__ d_theta) {
    const int n = threadIdx.x;
    const int nv = (gh + 1) * (gw + 1), nt = nv * 2;
    __shared__ float red[4][64];
    float l_id = 0.f, l_black = 0.f, l_dist = 0.f, l_cons = 0.f;
    extern __shared__ float mesh_lds[];
    float p_priv[LDSV ? 1 : 2 * 81], dp_priv[LDSV ? 1 : 2 * 81];      // (gh+1)(gw+1) <= 81 vertices
    float* const p = LDSV ? mesh_lds + (size_t)n * 4 * nv : p_priv;
    float* const dp = LDSV ? p + 2 * nv : dp_priv;
    if (n < N) {
        const double hh = 2.0 / gh, ww = 2.0 / gw;
        for (int i = 0; i <= gh; ++i)
            for (int j = 0; j <= gw; ++j) {
                const int v = i * (gw + 1) + j;
                float px = (float)(j * ww - 1.0) + theta[(size_t)n * nt + 2 * v];
                float py = (float)(i * hh - 1.0) + theta[(size_t)n * nt + 2 * v + 1];
                p[2 * v] = fminf(fmaxf(px, -lim), lim);
                p[2 * v + 1] = fminf(fmaxf(py, -lim), lim);
                dp[2 * v] = dp[2 * v + 1] = 0.f;
                l_id += fabsf(theta[(size_t)n * nt + 2 * v]) + fabsf(theta[(size_t)n * nt + 2 * v + 1]);
            }
        // black_pos hinge on pts1 entries (each cell corner): identically 0 after the clip, kept for fidelity
        const float cb = 2.0f * use_black / (float)(N * gh * gw * 8);
        // distortion: 8 ordered corner triples per cell, R = 90deg rotation scaled by the cell aspect k
        const float kq0 = (float)(hh / ww), kq1 = (float)(ww / hh);
        const float cd = 1.0f / (8.0f * (float)(N * gh * gw) * 2.0f);   // mean over [N*cells, 2] then / 8
        const int trip[8][3] = {{0, 1, 3}, {1, 3, 2}, {3, 2, 0}, {2, 0, 1}, {1, 0, 2}, {0, 2, 3}, {2, 3, 1}, {3, 1, 0}};
        for (int i = 0; i < gh; ++i)
            for (int j = 0; j < gw; ++j) {
                const int vq[4] = {i * (gw + 1) + j, i * (gw + 1) + j + 1, (i + 1) * (gw + 1) + j, (i + 1) * (gw + 1) + j + 1};
                for (int q = 0; q < 4; ++q)
                    for (int d = 0; d < 2; ++d) {
                        const float v = p[2 * vq[q] + d];
                        const float e = (v > lim) ? v - lim : ((-lim > v) ? -lim - v : 0.f);
                        l_black += e * e * use_black;
                        if (e != 0.f) dp[2 * vq[q] + d] += w_black * cb * e * ((v > lim) ? 1.f : -1.f);
                    }
                for (int t = 0; t < 8; ++t) {
                    const int a0 = vq[trip[t][0]], a1 = vq[trip[t][1]], a2 = vq[trip[t][2]];
                    const bool clock = t >= 4;
                    const float k = (t & 1) ? kq1 : kq0;
                    const float r01 = clock ? k : -k, r10 = clock ? -k : k;      // R = [[0,r01],[r10,0]]
                    const float dx = p[2 * a1] - p[2 * a0], dy = p[2 * a1 + 1] - p[2 * a0 + 1];
                    const float ex = r01 * dy - (p[2 * a2] - p[2 * a1]);
                    const float ey = r10 * dx - (p[2 * a2 + 1] - p[2 * a1 + 1]);
                    l_dist += ex * ex + ey * ey;
                    const float gx = w_dist * cd * 2.0f * ex, gyv = w_dist * cd * 2.0f * ey;
                    // ex = r01*(p1y - p0y) - p2x + p1x ; ey = r10*(p1x - p0x) - p2y + p1y
                    dp[2 * a1 + 1] += gx * r01; dp[2 * a0 + 1] -= gx * r01; dp[2 * a2] -= gx; dp[2 * a1] += gx;
                    dp[2 * a1] += gyv * r10; dp[2 * a0] -= gyv * r10; dp[2 * a2 + 1] -= gyv; dp[2 * a1 + 1] += gyv;
                }
            }
        // consistency: squared second differences along both axes, from both ends
        int ncons = 0;
        for (int i = 0; i <= gh; ++i)
            for (int j = 0; j <= gw; ++j) ncons += (i > 1) + (j > 1) + (i < gh - 1) + (j < gw - 1);
        const float cc = (ncons > 0) ? 1.0f / ((float)N * 2.0f * (float)ncons) : 0.f;
        for (int i = 0; i <= gh; ++i)
            for (int j = 0; j <= gw; ++j) {
                const int v = i * (gw + 1) + j;
                const int nb[4][2] = {{(i - 1) * (gw + 1) + j, (i - 2) * (gw + 1) + j}, {i * (gw + 1) + j - 1, i * (gw + 1) + j - 2},
                                      {(i + 1) * (gw + 1) + j, (i + 2) * (gw + 1) + j}, {i * (gw + 1) + j + 1, i * (gw + 1) + j + 2}};
                const bool on[4] = {i > 1, j > 1, i < gh - 1, j < gw - 1};
                for (int t = 0; t < 4; ++t) {
                    if (!on[t]) continue;
                    for (int d = 0; d < 2; ++d) {
                        const float e = 2.0f * p[2 * nb[t][0] + d] - p[2 * v + d] - p[2 * nb[t][1] + d];
                        l_cons += e * e;
                        const float g = w_cons * cc * 2.0f * e;
                        dp[2 * nb[t][0] + d] += 2.0f * g; dp[2 * v + d] -= g; dp[2 * nb[t][1] + d] -= g;
                    }
                }
            }
        l_dist *= cd; l_cons *= cc; l_black /= (float)(N * gh * gw * 8);
        // clip mask + id loss
        for (int v = 0; v < nv; ++v)
            for (int d = 0; d < 2; ++d) {
                const float th = theta[(size_t)n * nt + 2 * v + d];
                const float raw = ((d == 0) ? (float)((v % (gw + 1)) * ww - 1.0) : (float)((v / (gw + 1)) * hh - 1.0)) + th;
                const bool pass = (raw >= -lim) && (raw <= lim);
                float g = dp[2 * v + d] + (d_pts2_warp ? d_pts2_warp[((size_t)n * nv + v) * 2 + d] : 0.f);
                g = pass ? g : 0.f;
                g += w_id * id_mul * ((th > 0.f) ? 1.f : ((th < 0.f) ? -1.f : 0.f)) / (float)(N * nt);
                if (d_theta != nullptr) d_theta[(size_t)n * nt + 2 * v + d] = g;
            }
    }
    red[0][threadIdx.x] = l_id; red[1][threadIdx.x] = l_black; red[2][threadIdx.x] = l_dist; red[3][threadIdx.x] = l_cons;
    __syncthreads();
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
    dim3 grid(cdiv(W, 256), cdiv(H, 4), N);
    interp_bwd_kernel<<<grid, 256, 0, st>>>(x, y, d_out, H, W, C, static_cast<sn_u64*>(workspace));
    SN_LAUNCH_CHECK("interp_bwd_kernel");
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
    feature_loss_kernel<<<N, 256, 0, st>>>(matches, mask, x_map, y_map, H, W, max_matches, gcoef, value, d_xmap, d_ymap, warped,
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
    const size_t lds = (size_t)N * 4 * (grid_h + 1) * (grid_w + 1) * sizeof(float);
    if (lds <= 60 * 1024)
        mesh_losses_kernel<true><<<1, 64, lds, (hipStream_t)stream>>>(theta, d_pts2_warp, N, grid_h, grid_w, 1.0f / do_crop_rate,
                                                                      id_mul, w_id, w_dist, w_cons, use_black, w_black, losses4, d_theta);
    else
        mesh_losses_kernel<false><<<1, 64, 0, (hipStream_t)stream>>>(theta, d_pts2_warp, N, grid_h, grid_w, 1.0f / do_crop_rate,
                                                                     id_mul, w_id, w_dist, w_cons, use_black, w_black, losses4, d_theta);
    SN_LAUNCH_CHECK("mesh_losses_kernel");
    return STABNET_OK;
}

}  // extern "C"
