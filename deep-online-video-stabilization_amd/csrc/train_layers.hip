// Training-mode layers of the regressor other than the convolutions (gfx950): batch-statistics BN forward and
// backward, bias gradients, max-pool / global-pool / FC backward, weight decay, Adam.  All HBM-bound.
//   slim batch_norm(is_training=True): tf.nn.moments (biased variance) + moving-average update
//   (decay 0.997, variable -= (variable - value) * (1 - decay)); the UPDATE_OPS run with the step (s_net_bundle_nobm.py:355-356).
#include "train_layers.h"
#include <algorithm>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------------
// Column reductions over an [M][C] tensor, two stages (deterministic order).  (Measured and dropped: ONE launch per reduction
// with a last-block-done ticket doing the finalize -- with agent-scope release/acquire fences each reduction took 22 us
// instead of 6 + 6, 512 blocks each paying a buffer_wbl2; in the fence-free sc1 store/load form the serial tail of the
// last block still cost more than the second launch: 22.9 ms vs 21.9 ms per step.)
//   MODE 0: (sum x, sum x^2)                           BN batch statistics
//   MODE 1: (sum dz, sum dz*xhat), dz = g*(a>0)        BN+ReLU backward;  a = x*scale+shift, xhat = (x-mean)*invstd
//   MODE 2: (sum g, -)                                 bias gradient
// block = 16 channel-quads x 16 row lanes; grid = (C/64, chunks, groups); partial [groups][chunks][2][C].
// GROUPS: the two siamese towers run in lockstep (train_bundle_nobm.py:107-108 builds two towers over the same weights), so
// the same reduction of both towers is ONE launch (blockIdx.z = tower) -- these kernels are launch-latency sized (about 6 us
// for either one or two towers' rows), and there are ~240 of them per tower per step.
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const ColGroups G, long M, int C, long rows_per_chunk,
                                                         float* __restrict__ partial_all) {
    __shared__ float4 s0[16][16], s1[16][16];
    const int grp = blockIdx.z;
    const float* __restrict__ x = G.x[grp];
    const float* __restrict__ g = G.g[grp];
    const float* __restrict__ scale = G.scale[grp];
    const float* __restrict__ shift = G.shift[grp];
    const float* __restrict__ mean = G.mean[grp];
    const float* __restrict__ invstd = G.invstd[grp];
    float* __restrict__ partial = partial_all + (size_t)grp * gridDim.y * 2 * C;
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + q * 4;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    const long r1 = min(M, r0 + rows_per_chunk);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (c < C) {
        float4 sc = a0, sh = a0, mu = a0, is = a0;
        if (MODE == 1) {
            sc = *reinterpret_cast<const float4*>(scale + c); sh = *reinterpret_cast<const float4*>(shift + c);
            mu = *reinterpret_cast<const float4*>(mean + c); is = *reinterpret_cast<const float4*>(invstd + c);
        }
        // four rows per trip, their loads issued together (one load per trip left the stream latency-bound: 3.8 TB/s for MODE 0);
        // the rows are still accumulated in the order r, r + 16, ... -- same sums, same bits
        for (long rb = r0 + rl; rb < r1; rb += 64) {
            float4 xv[4], gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long r = rb + 16 * j;
                const bool ok = r < r1;
                if (MODE != 2) xv[j] = ok ? *reinterpret_cast<const float4*>(x + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE != 0) gv[j] = ok ? *reinterpret_cast<const float4*>(g + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rb + 16 * j >= r1) break;
                if (MODE == 0) {
                    const float4 v = xv[j];
                    a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
                    a1.x += v.x * v.x; a1.y += v.y * v.y; a1.z += v.z * v.z; a1.w += v.w * v.w;
                } else if (MODE == 1) {
                    const float4 v = xv[j];
                    float4 d = gv[j];
                    d.x = (__builtin_fmaf(v.x, sc.x, sh.x) > 0.f) ? d.x : 0.f; d.y = (__builtin_fmaf(v.y, sc.y, sh.y) > 0.f) ? d.y : 0.f;   // the forward's own decision (fma + max in the conv prologue)
                    d.z = (__builtin_fmaf(v.z, sc.z, sh.z) > 0.f) ? d.z : 0.f; d.w = (__builtin_fmaf(v.w, sc.w, sh.w) > 0.f) ? d.w : 0.f;
                    a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
                    a1.x += d.x * (v.x - mu.x) * is.x; a1.y += d.y * (v.y - mu.y) * is.y;
                    a1.z += d.z * (v.z - mu.z) * is.z; a1.w += d.w * (v.w - mu.w) * is.w;
                } else {
                    const float4 d = gv[j];
                    a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
                }
            }
        }
    }
    s0[rl][q] = a0;
    s1[rl][q] = a1;
    __syncthreads();
    if (rl == 0 && c < C) {
        float4 t0 = s0[0][q], t1 = s1[0][q];
        for (int i = 1; i < 16; ++i) {
            const float4 u = s0[i][q], w = s1[i][q];
            t0.x += u.x; t0.y += u.y; t0.z += u.z; t0.w += u.w;
            t1.x += w.x; t1.y += w.y; t1.z += w.z; t1.w += w.w;
        }
        float* p = partial + (size_t)blockIdx.y * 2 * C;
        *reinterpret_cast<float4*>(p + c) = t0;
        *reinterpret_cast<float4*>(p + C + c) = t1;
    }
}

// Sum of the chunk partials for FIN_CH channels per block: FIN_LANES lanes per channel walk the chunks (float64), fixed-order
// combine through LDS (deterministic).  Returns the two sums in lane 0 of each channel.  These kernels are latency bound (a
// handful of blocks, each lane a serial chain of L2-latency loads): 16 channels x 64 lanes per 1024-thread block keeps the
// chain at chunks / 64 iterations (was 64 channels x 16 lanes: 9.9 us for two towers' 512 chunks, now a third of that).
// Measured and not kept (rocprofv3, 6.3-6.7 us per launch as it stands): eight chunk loads issued together (12-13 us: 120
// VGPRs at 1024 threads), the final 63-term sum by lane 0 only instead of by every thread (7.1-7.2 us).
constexpr int FIN_LANES = 64, FIN_CH = 16;
__device__ __forceinline__ void combine_partials(const float* __restrict__ partial, int chunks, int C, int c, int lane,
                                                 double& s0, double& s1) {
    __shared__ double sh0[FIN_LANES][FIN_CH], sh1[FIN_LANES][FIN_CH];
    double a0 = 0.0, a1 = 0.0;
    if (c < C)
        for (int k = lane; k < chunks; k += FIN_LANES) {
            a0 += (double)partial[(size_t)k * 2 * C + c];
            a1 += (double)partial[(size_t)k * 2 * C + C + c];
        }
    const int t = threadIdx.x % FIN_CH;
    sh0[lane][t] = a0;
    sh1[lane][t] = a1;
    __syncthreads();
    s0 = sh0[0][t]; s1 = sh1[0][t];
    for (int i = 1; i < FIN_LANES; ++i) { s0 += sh0[i][t]; s1 += sh1[i][t]; }
}

// BN statistics finalize: batch mean / biased variance (float64 combine) -> folded (scale, shift), saved (mean, invstd),
// moving averages.  grid = C/16 blocks of 1024 threads.  Groups (towers) are finalized one after the other by the same
// thread, so the moving averages receive tower 1's update and then tower 2's, as two sequential steps would give.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ partial, int chunks, int groups, long M,
                                                                int C, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float decay,
                                                                const BnStatOut o, float* __restrict__ mov_mean,
                                                                float* __restrict__ mov_var) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), lane = threadIdx.x / FIN_CH;
    for (int gi = 0; gi < groups; ++gi) {
        double s, ss;
        combine_partials(partial + (size_t)gi * chunks * 2 * C, chunks, C, c, lane, s, ss);
        if (lane == 0 && c < C) {
            const double mean = s / (double)M;
            const double var = fmax(ss / (double)M - mean * mean, 0.0);
            const float meanf = (float)mean, varf = (float)var;
            const float is = 1.0f / sqrtf(varf + eps);
            const float inv = is * gamma[c];
            o.scale[gi][c] = inv;
            o.shift[gi][c] = beta[c] - meanf * inv;
            o.save_mean[gi][c] = meanf;
            o.save_invstd[gi][c] = is;
            if (mov_mean != nullptr) {
                mov_mean[c] -= (mov_mean[c] - meanf) * (1.0f - decay);
                mov_var[c] -= (mov_var[c] - varf) * (1.0f - decay);
            }
        }
        __syncthreads();                                  // combine_partials' LDS is reused by the next group
    }
}

// BN backward finalize: d_gamma += sum dz*xhat, d_beta += sum dz (group after group, fixed order); coefficients for the apply
// pass per group: coef[0][c] = gamma*invstd, coef[1][c] = mean(dz), coef[2][c] = mean(dz*xhat).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int groups, long M,
                                                              int C, const float* __restrict__ gamma, const BnBwdFin f,
                                                              float* __restrict__ d_gamma, float* __restrict__ d_beta) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), lane = threadIdx.x / FIN_CH;
    for (int gi = 0; gi < groups; ++gi) {
        double s, sx;
        combine_partials(partial + (size_t)gi * chunks * 2 * C, chunks, C, c, lane, s, sx);
        if (lane == 0 && c < C) {
            d_beta[c] += (float)s;
            d_gamma[c] += (float)sx;
            float* coef = f.coef[gi];
            coef[c] = gamma[c] * f.invstd[gi][c];
            coef[C + c] = (float)(s / (double)M);
            coef[2 * C + c] = (float)(sx / (double)M);
        }
        __syncthreads();
    }
}

// (d_bias2, optional: a second bias fed by the same gradient -- conv3 and the projection shortcut of a unit both add into its output)
__global__ __launch_bounds__(1024) void bias_grad_finalize_kernel(const float* __restrict__ partial, int chunks, int groups, int C,
                                                                 float* __restrict__ d_bias, float* __restrict__ d_bias2) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), lane = threadIdx.x / FIN_CH;
    for (int gi = 0; gi < groups; ++gi) {
        double s, unused;
        combine_partials(partial + (size_t)gi * chunks * 2 * C, chunks, C, c, lane, s, unused);
        if (lane == 0 && c < C) {
            d_bias[c] += (float)s;
            if (d_bias2 != nullptr) d_bias2[c] += (float)s;
        }
        __syncthreads();
    }
}

// d_x = k1 * (dz - c1 - xhat * c2)  (+ addend[n, y/s, x/s, c] where y%s == 0 and x%s == 0: the identity-shortcut branch).
// g may alias d_x (in-place).  One thread per float4.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnApply A, int add_stride, int H, int W, long M, int C) {
    const int grp = blockIdx.y;
    const float* __restrict__ x = A.x[grp];
    const float* g = A.g[grp];
    const float* __restrict__ scale = A.scale[grp];
    const float* __restrict__ shift = A.shift[grp];
    const float* __restrict__ mean = A.mean[grp];
    const float* __restrict__ invstd = A.invstd[grp];
    const float* __restrict__ coef = A.coef[grp];
    const float* __restrict__ addend = A.addend[grp];
    float* d_x = A.d_x[grp];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = C / 4;
    if (i >= M * c4n) return;
    const long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
    float4 d = *reinterpret_cast<const float4*>(g + r * C + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + c), c1 = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 c2 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    d.x = (__builtin_fmaf(v.x, sc.x, sh.x) > 0.f) ? d.x : 0.f; d.y = (__builtin_fmaf(v.y, sc.y, sh.y) > 0.f) ? d.y : 0.f;   // the forward's own decision (fma + max in the conv prologue)
    d.z = (__builtin_fmaf(v.z, sc.z, sh.z) > 0.f) ? d.z : 0.f; d.w = (__builtin_fmaf(v.w, sc.w, sh.w) > 0.f) ? d.w : 0.f;
    float4 o;
    o.x = k1.x * (d.x - c1.x - (v.x - mu.x) * is.x * c2.x);
    o.y = k1.y * (d.y - c1.y - (v.y - mu.y) * is.y * c2.y);
    o.z = k1.z * (d.z - c1.z - (v.z - mu.z) * is.z * c2.z);
    o.w = k1.w * (d.w - c1.w - (v.w - mu.w) * is.w * c2.w);
    if (addend != nullptr) {
        bool take = true;
        long ar = r;
        if (add_stride > 1) {
            const int xw = (int)(r % W);
            const long t = r / W;
            const int yh = (int)(t % H);
            const long n = t / H;
            take = (xw % add_stride == 0) && (yh % add_stride == 0);
            const int Hs = (H + add_stride - 1) / add_stride, Ws = (W + add_stride - 1) / add_stride;
            ar = (n * Hs + yh / add_stride) * Ws + xw / add_stride;
        }
        if (take) {
            const float4 a = *reinterpret_cast<const float4*>(addend + ar * C + c);
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
    }
    *reinterpret_cast<float4*>(d_x + r * C + c) = o;
}

// max_pool2d backward, gather form (deterministic, no atomics): an input pixel receives the gradient of every window
// whose argmax (saved by the training forward as dy*k+dx, first maximum in scan order) points at it.
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const unsigned char* __restrict__ argmax, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                           int Wo, int k, int stride, int pt, int pl) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;           // one thread per (input pixel, 4 channels); < 2^32 (host-checked)
    const unsigned c4n = (unsigned)C / 4u;
    if (q >= (unsigned)N * H * W * c4n) return;
    const int c = (int)(q % c4n) * 4;
    unsigned r = q / c4n;
    const int ix = (int)(r % (unsigned)W); r /= (unsigned)W;
    const int iy = (int)(r % (unsigned)H);
    const int n = (int)(r / (unsigned)H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy_lo = max(0, (iy + pt - k + stride) / stride), oy_hi = min(Ho - 1, (iy + pt) / stride);
    const int ox_lo = max(0, (ix + pl - k + stride) / stride), ox_hi = min(Wo - 1, (ix + pl) / stride);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const long o = (((long)n * Ho + oy) * Wo + ox) * C + c;
            const uchar4 am = *reinterpret_cast<const uchar4*>(argmax + o);
            const float4 g = *reinterpret_cast<const float4*>(dy + o);
            const int want = (iy - (oy * stride - pt)) * k + (ix - (ox * stride - pl));     // the tap of this window that is (iy, ix)
            acc.x += (am.x == want) ? g.x : 0.f; acc.y += (am.y == want) ? g.y : 0.f;
            acc.z += (am.z == want) ? g.z : 0.f; acc.w += (am.w == want) ? g.w : 0.f;
        }
    *reinterpret_cast<float4*>(dx + (((long)n * H + iy) * W + ix) * C + c) = acc;
}

// max_pool2d forward that also records the argmax (training).  One thread per (output pixel, 4 channels): 16-B loads, one
// 4-byte argmax store (a thread per scalar channel moved the same 100 MB at 2.1 TB/s: 47.6 us per tower).
__global__ __launch_bounds__(256) void max_pool_argmax_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              unsigned char* __restrict__ argmax, int N, int H, int W, int C,
                                                              int Ho, int Wo, int k, int stride, int pt, int pl) {
    // (32-bit index math, host-checked.  XCD-banded ids -- sn_xcd_band, as the inference pool uses -- were measured on both
    //  training pool kernels at 8 x 144 x 256 x 64 per tower and lost: 22.9 -> 23.9 us here, 33.7 -> 37.3 us in the backward)
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    const unsigned c4n = (unsigned)C / 4u;
    if (q >= (unsigned)N * Ho * Wo * c4n) return;
    const int c = (int)(q % c4n) * 4;
    unsigned r = q / c4n;
    const int ox = (int)(r % (unsigned)Wo); r /= (unsigned)Wo;
    const int oy = (int)(r % (unsigned)Ho);
    const int n = (int)(r / (unsigned)Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int dyy = 0; dyy < k; ++dyy) {
        const int iy = oy * stride - pt + dyy;
        if (iy < 0 || iy >= H) continue;
        for (int dxx = 0; dxx < k; ++dxx) {
            const int ix = ox * stride - pl + dxx;
            if (ix < 0 || ix >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * C + c);
            const int t = dyy * k + dxx;                       // first maximum in scan order, per channel
            if (v.x > m.x) { m.x = v.x; a0 = t; }
            if (v.y > m.y) { m.y = v.y; a1 = t; }
            if (v.z > m.z) { m.z = v.z; a2 = t; }
            if (v.w > m.w) { m.w = v.w; a3 = t; }
        }
    }
    const long o = (((long)n * Ho + oy) * Wo + ox) * C + c;
    *reinterpret_cast<float4*>(y + o) = m;
    *reinterpret_cast<uchar4*>(argmax + o) = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
}

// d a[n,hw,c] = d g[n,c] / HW   (reduce_mean backward; the BN+ReLU before it is handled by the BN backward kernels)
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dg, int HW, int C, long total,
                                                      float* __restrict__ da) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int c = (int)(q % C);
    const long n = q / ((long)HW * C);
    da[q] = dg[n * C + c] / (float)HW;
}

// FC backward.  dyr = dy * (y > 0) when relu.  dW[n][k] += sum_m dyr[m][n] x[m][k];  db[n] += sum_m dyr[m][n].
// One thread per (n, 4 consecutive k): the x rows are L2-hot (<= 128 KB), dW is read-modify-written 16 B per lane.
__global__ __launch_bounds__(256) void fc_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ dy, int M, int K, int Nout, int relu,
                                                       float* __restrict__ dW, float* __restrict__ db) {
    const int n = blockIdx.y;
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float bacc = 0.f;
    const bool k_ok = k < K;
    for (int m0 = 0; m0 < M; m0 += 8) {                    // eight rows' loads in flight, added in row order
        float4 xv[8];
        float d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = min(m0 + j, M - 1);
            xv[j] = k_ok ? *reinterpret_cast<const float4*>(x + (size_t)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[j] = dy[(size_t)m * Nout + n];
            if (relu && !(y[(size_t)m * Nout + n] > 0.f)) d[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (m0 + j >= M) break;
            bacc += d[j];
            acc.x += d[j] * xv[j].x; acc.y += d[j] * xv[j].y; acc.z += d[j] * xv[j].z; acc.w += d[j] * xv[j].w;
        }
    }
    if (k < K) {
        float4* p = reinterpret_cast<float4*>(dW + (size_t)n * K + k);
        float4 o = *p;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *p = o;
    }
    if (k == 0) db[n] += bacc;
}

// dx[m][k] = sum_n dyr[m][n] W[n][k]: the n range is split over blockIdx.y into partial sums part[split][m][k] (plain
// stores), which fc_bwd_x_reduce_kernel adds in split order -- no atomics, reproducible.  A workgroup owns 64 columns k and
// n_per_block <= 256 rows n; thread (g = tid / 16, kq = tid % 16) takes the rows nb + g + 16 j against the float4 column kq for up
// to 16 samples m (the siamese pair): every weight row segment is read ONCE, 256 contiguous bytes per 16 lanes, eight rows in
// flight per thread; the block's [M][n_per_block] slice of dyr is staged in LDS.  The 16 row groups are then combined in a
// fixed order (two lane exchanges inside a wave, the four waves through LDS).  The first form had 128 workgroups of 32 rows x 1024
// columns walking their rows one load at a time: 0.4 TB/s on the 16.8 MB of fc_1.  M > 16: passes of 16 rows.
constexpr int FCX_NPB_MAX = 256;
__global__ __launch_bounds__(256) void fc_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                       const float* __restrict__ dy, int M, int K, int Nout, int relu,
                                                       int n_per_block, float* __restrict__ part) {
    __shared__ float sd[16][FCX_NPB_MAX];
    __shared__ __attribute__((aligned(16))) float red[4][16][64];
    const int tid = threadIdx.x, kq = tid & 15, g = tid >> 4, wave = tid >> 6;
    const int k = blockIdx.x * 64 + kq * 4;
    const bool k_ok = k < K;
    const int nb = blockIdx.y * n_per_block, ne = min(Nout, nb + n_per_block);
    for (int m0 = 0; m0 < M; m0 += 16) {
        __syncthreads();
        for (int i = tid; i < 16 * n_per_block; i += 256) {
            const int mi = i / n_per_block, ni = i - mi * n_per_block;
            const int m = m0 + mi, n = nb + ni;
            float d = 0.f;
            if (m < M && n < ne) {
                d = dy[(size_t)m * Nout + n];
                if (relu && !(y[(size_t)m * Nout + n] > 0.f)) d = 0.f;
            }
            sd[mi][ni] = d;
        }
        __syncthreads();
        float4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = 0; j0 * 16 < n_per_block; j0 += 8) {
            float4 wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = nb + g + 16 * (j0 + j);
                wv[j] = (k_ok && n < ne) ? *reinterpret_cast<const float4*>(w + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ni = g + 16 * (j0 + j);
                if (ni < n_per_block) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = sd[i][ni];
                        acc[i].x += d * wv[j].x; acc[i].y += d * wv[j].y; acc[i].z += d * wv[j].z; acc[i].w += d * wv[j].w;
                    }
                }
            }
        }
        // row groups: g = 4 wave + (lane >> 4)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] += __shfl_xor(v[c], 16, 64);
                v[c] += __shfl_xor(v[c], 32, 64);
            }
            if ((tid & 63) < 16) *reinterpret_cast<float4*>(&red[wave][i][kq * 4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        {   // 16 x 64 outputs, one float4 per thread: m = tid / 16, columns 4 kq
            const int mi = tid >> 4;
            const float4 a = *reinterpret_cast<const float4*>(&red[0][mi][kq * 4]), b = *reinterpret_cast<const float4*>(&red[1][mi][kq * 4]);
            const float4 c = *reinterpret_cast<const float4*>(&red[2][mi][kq * 4]), d = *reinterpret_cast<const float4*>(&red[3][mi][kq * 4]);
            const float4 o = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
            if (k_ok && m0 + mi < M) *reinterpret_cast<float4*>(part + ((size_t)blockIdx.y * M + m0 + mi) * K + k) = o;
        }
    }
}
// rows per workgroup: ~256 workgroups, 16..256 rows each (a multiple of 16)
static int fc_bwd_x_rows(int K, int Nout) {
    const int kblocks = cdiv(K, 64);
    int splits = std::max(1, std::min(cdiv(256, kblocks), cdiv(Nout, 16)));
    splits = std::max(splits, cdiv(Nout, FCX_NPB_MAX));
    return std::min(FCX_NPB_MAX, cdiv(cdiv(Nout, splits), 16) * 16);
}
__global__ __launch_bounds__(256) void fc_bwd_x_reduce_kernel(const float* __restrict__ part, int splits, long MK,
                                                              float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MK) return;
    float s = part[i];
    for (int z = 1; z < splits; ++z) s += part[(size_t)z * MK + i];
    dx[i] = s;
}

// Weight decay: loss += coef_seg * 0.5 * sum w^2 (block partials, summed in a fixed order by weight_decay_finalize_kernel),
// grad += gscale * coef_seg * w.
// seg table (device): [offset, length] as int64 pairs + coef as float, one block column per segment.
__global__ __launch_bounds__(256) void weight_decay_kernel(const float* __restrict__ params, float* __restrict__ grads,
                                                           const long* __restrict__ seg_off, const long* __restrict__ seg_len,
                                                           const float* __restrict__ seg_coef, float gscale,
                                                           float* __restrict__ partial) {
    const int sgi = blockIdx.y;
    const long off = seg_off[sgi], len = seg_len[sgi];
    const float coef = seg_coef[sgi];
    float s = 0.f;
    // scalar head up to the first 16-B boundary, float4 body, scalar tail (the fixed thread <-> element map keeps the sum reproducible)
    const long head = min(len, (long)((4 - ((off) & 3)) & 3));
    const long body = (len - head) >> 2;
    const float gc = gscale * coef;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < head + ((len - head) & 3)) {
        const long i = t < head ? t : head + 4 * body + (t - head);
        const float w = params[off + i];
        s += w * w;
        if (grads != nullptr) grads[off + i] += gc * w;
    }
    const float4* p4 = reinterpret_cast<const float4*>(params + off + head);
    float4* g4 = reinterpret_cast<float4*>(grads != nullptr ? grads + off + head : nullptr);
    for (long i = t; i < body; i += (long)gridDim.x * 256) {
        const float4 w = p4[i];
        s += (w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w);
        if (grads != nullptr) {
            float4 g = g4[i];
            g.x += gc * w.x; g.y += gc * w.y; g.z += gc * w.z; g.w += gc * w.w;
            g4[i] = g;
        }
    }
    if (partial != nullptr) {
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partial[(size_t)sgi * gridDim.x + blockIdx.x] = 0.5f * coef * ((red[0] + red[1]) + (red[2] + red[3]));
    }
}
// (1024 threads: the ~7 000 block partials are seven loads per thread, not 110 dependent ones; lanes, then waves in order)
__global__ __launch_bounds__(1024) void weight_decay_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss_out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int w = 1; w < 16; ++w) t += red[w];
        *loss_out += (float)t;
    }
}

// tf.train.AdamOptimizer step, the ApplyAdam kernel of TF 1.x op for op (one rounding per op, -ffp-contract=off):
//   m += (g - m) * (1 - b1);  v += (g*g - v) * (1 - b2);  w -= (m * alpha) / (sqrt(v) + eps)
// alpha = lr * sqrt(1 - b2^t) / (1 - b1^t) is formed on the host.  g is (g + g2) * gscale (g2 optional, gscale = 1/world).
__device__ __forceinline__ void adam_elem(float& w, float& m, float& v, float g, float alpha, float omb1, float omb2, float eps) {
    m = m + (g - m) * omb1;
    v = v + (g * g - v) * omb2;
    w = w - (m * alpha) / (sqrtf(v) + eps);
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   const float* __restrict__ g2, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float alpha, float b1, float b2,
                                                   float eps, float gscale) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    if (i + 3 < n) {
        float4 wv = *reinterpret_cast<float4*>(w + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        float4 gv = *reinterpret_cast<const float4*>(g + i);
        if (g2 != nullptr) {
            const float4 hv = *reinterpret_cast<const float4*>(g2 + i);
            gv.x += hv.x; gv.y += hv.y; gv.z += hv.z; gv.w += hv.w;
        }
        if (gscale != 1.f) { gv.x *= gscale; gv.y *= gscale; gv.z *= gscale; gv.w *= gscale; }
        adam_elem(wv.x, mv.x, vv.x, gv.x, alpha, omb1, omb2, eps);
        adam_elem(wv.y, mv.y, vv.y, gv.y, alpha, omb1, omb2, eps);
        adam_elem(wv.z, mv.z, vv.z, gv.z, alpha, omb1, omb2, eps);
        adam_elem(wv.w, mv.w, vv.w, gv.w, alpha, omb1, omb2, eps);
        *reinterpret_cast<float4*>(w + i) = wv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
    } else {
        for (long j = i; j < n; ++j) {
            float gg = g[j] + (g2 != nullptr ? g2[j] : 0.f);
            if (gscale != 1.f) gg *= gscale;
            adam_elem(w[j], m[j], v[j], gg, alpha, omb1, omb2, eps);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row chunks of a column reduction over [M][C]: about 256 blocks PER TOWER ((C/64) x chunks; both towers are one launch, so
// 512 blocks = two per CU in all.  Measured per tower, one tower per launch, round 1: 2048 / 1024 / 768 / 512 / 384 / 256 ->
// 389.6 / 394.7 / 398.0 / 400.0 / 399.5 / 395.4 pairs/s; with the towers paired, round 3: 2048 / 1024 / 512 / 384 / 256 / 192 /
// 128 -> 521.4 / 523.1 / 523.6 / 524.4 / 526.3 / 525.4 / 525.6: more chunks speed the reduction up but slow the
// single-block-per-16-channels finalize down), at least 64 rows per chunk.
static int reduce_chunks(long M, int C, long& rows_per_chunk) {
    static const int target = []() { const char* v = getenv("STABNET_REDUCE_BLOCKS"); return v ? atoi(v) : 256; }();
    const long want = std::max<long>(target / cdiv(C, 64), 16);
    int chunks = (int)std::min<long>(std::max<long>(M / 64, 1), want);
    rows_per_chunk = (M + chunks - 1) / chunks;
    return (int)((M + rows_per_chunk - 1) / rows_per_chunk);
}

size_t col_reduce_workspace_floats(long M, int C) {
    long rpc;
    const int chunks = reduce_chunks(M, C, rpc);
    return (size_t)chunks * 2 * C;
}

size_t col_reduce_workspace_floats(long M, int C, int groups) { return (size_t)groups * col_reduce_workspace_floats(M, C); }

// ---- grouped launchers: groups = 1 (one tower) or 2 (both siamese towers in one launch each) --------------------------
int launch_bn_stats_g(int groups, const float* const* x, long M, int C, const float* gamma, const float* beta, float eps,
                      float decay, float* const* scale, float* const* shift, float* const* save_mean, float* const* save_invstd,
                      float* mov_mean, float* mov_var, float* partial, hipStream_t st) {
    SN_REQUIRE(C % 4 == 0 && (groups == 1 || groups == 2), "bn_stats: C %% 4 != 0 or bad group count");
    long rpc;
    const int chunks = reduce_chunks(M, C, rpc);
    ColGroups G{};
    BnStatOut o{};
    for (int i = 0; i < groups; ++i) {
        G.x[i] = x[i];
        o.scale[i] = scale[i]; o.shift[i] = shift[i]; o.save_mean[i] = save_mean[i]; o.save_invstd[i] = save_invstd[i];
    }
    col_reduce_kernel<0><<<dim3(cdiv(C, 64), chunks, groups), 256, 0, st>>>(G, M, C, rpc, partial);
    SN_LAUNCH_CHECK("col_reduce_kernel<0>");
    bn_stats_finalize_kernel<<<cdiv(C, FIN_CH), FIN_CH * FIN_LANES, 0, st>>>(partial, chunks, groups, M, C, gamma, beta, eps, decay, o,
                                                                     mov_mean, mov_var);
    SN_LAUNCH_CHECK("bn_stats_finalize_kernel");
    return STABNET_OK;
}

int launch_bn_relu_bwd_g(int groups, const float* const* x, const float* const* g, const float* const* scale,
                         const float* const* shift, const float* const* mean, const float* const* invstd, const float* gamma,
                         long M, int C, const float* const* addend, int add_stride, int H, int W, float* d_gamma, float* d_beta,
                         float* const* d_x, float* partial, float* const* coef, hipStream_t st) {
    SN_REQUIRE(C % 4 == 0 && (groups == 1 || groups == 2), "bn_bwd: C %% 4 != 0 or bad group count");
    long rpc;
    const int chunks = reduce_chunks(M, C, rpc);
    ColGroups G{};
    BnBwdFin f{};
    BnApply A{};
    for (int i = 0; i < groups; ++i) {
        G.x[i] = x[i]; G.g[i] = g[i]; G.scale[i] = scale[i]; G.shift[i] = shift[i]; G.mean[i] = mean[i]; G.invstd[i] = invstd[i];
        f.invstd[i] = invstd[i]; f.coef[i] = coef[i];
        A.x[i] = x[i]; A.g[i] = g[i]; A.scale[i] = scale[i]; A.shift[i] = shift[i]; A.mean[i] = mean[i]; A.invstd[i] = invstd[i];
        A.coef[i] = coef[i]; A.addend[i] = addend ? addend[i] : nullptr; A.d_x[i] = d_x[i];
    }
    col_reduce_kernel<1><<<dim3(cdiv(C, 64), chunks, groups), 256, 0, st>>>(G, M, C, rpc, partial);
    SN_LAUNCH_CHECK("col_reduce_kernel<1>");
    bn_bwd_finalize_kernel<<<cdiv(C, FIN_CH), FIN_CH * FIN_LANES, 0, st>>>(partial, chunks, groups, M, C, gamma, f, d_gamma, d_beta);
    SN_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    bn_bwd_apply_kernel<<<dim3(cdiv(M * (C / 4), 256), groups), 256, 0, st>>>(A, add_stride, H, W, M, C);
    SN_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return STABNET_OK;
}

int launch_bias_grad_g(int groups, const float* const* g, long M, int C, float* d_bias, float* partial, hipStream_t st, float* d_bias2) {
    SN_REQUIRE(C % 4 == 0 && (groups == 1 || groups == 2), "bias_grad: C %% 4 != 0 or bad group count");
    long rpc;
    const int chunks = reduce_chunks(M, C, rpc);
    ColGroups G{};
    for (int i = 0; i < groups; ++i) G.g[i] = g[i];
    col_reduce_kernel<2><<<dim3(cdiv(C, 64), chunks, groups), 256, 0, st>>>(G, M, C, rpc, partial);
    SN_LAUNCH_CHECK("col_reduce_kernel<2>");
    bias_grad_finalize_kernel<<<cdiv(C, FIN_CH), FIN_CH * FIN_LANES, 0, st>>>(partial, chunks, groups, C, d_bias, d_bias2);
    SN_LAUNCH_CHECK("bias_grad_finalize_kernel");
    return STABNET_OK;
}

// ---- single-tower forms
int launch_bn_stats(const float* x, long M, int C, const float* gamma, const float* beta, float eps, float decay,
                    float* scale, float* shift, float* save_mean, float* save_invstd, float* mov_mean, float* mov_var,
                    float* partial, hipStream_t st) {
    return launch_bn_stats_g(1, &x, M, C, gamma, beta, eps, decay, &scale, &shift, &save_mean, &save_invstd, mov_mean, mov_var,
                             partial, st);
}

int launch_bn_relu_bwd(const float* x, const float* g, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* gamma, long M, int C, const float* addend, int add_stride, int H,
                       int W, float* d_gamma, float* d_beta, float* d_x, float* partial, float* coef, hipStream_t st) {
    return launch_bn_relu_bwd_g(1, &x, &g, &scale, &shift, &mean, &invstd, gamma, M, C, addend ? &addend : nullptr, add_stride, H, W,
                                d_gamma, d_beta, &d_x, partial, &coef, st);
}

int launch_bias_grad(const float* g, long M, int C, float* d_bias, float* partial, hipStream_t st) {
    return launch_bias_grad_g(1, &g, M, C, d_bias, partial, st, nullptr);
}

int launch_max_pool_bwd(const unsigned char* argmax, const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                        int k, int stride, int pt, int pl, hipStream_t st) {
    SN_REQUIRE(C % 4 == 0 && (long)N * H * W * (C / 4) < (1L << 32), "max_pool_bwd: C %% 4 != 0, or more than 2^32 channel quads");
    max_pool_bwd_kernel<<<cdiv((long)N * H * W * (C / 4), 256), 256, 0, st>>>(argmax, dy, dx, N, H, W, C, Ho, Wo, k, stride, pt, pl);
    SN_LAUNCH_CHECK("max_pool_bwd_kernel");
    return STABNET_OK;
}

int launch_max_pool_argmax(const float* x, float* y, unsigned char* argmax, int N, int H, int W, int C, int Ho, int Wo, int k,
                           int stride, int pt, int pl, hipStream_t st) {
    SN_REQUIRE(k * k <= 255 && C % 4 == 0 && (long)N * Ho * Wo * (C / 4) < (1L << 32), "max_pool: window too large for the argmax byte, C %% 4 != 0, or more than 2^32 channel quads");
    max_pool_argmax_kernel<<<cdiv((long)N * Ho * Wo * (C / 4), 256), 256, 0, st>>>(x, y, argmax, N, H, W, C, Ho, Wo, k, stride, pt, pl);
    SN_LAUNCH_CHECK("max_pool_argmax_kernel");
    return STABNET_OK;
}

int launch_gap_bwd(const float* dg, int N, int HW, int C, float* da, hipStream_t st) {
    const long total = (long)N * HW * C;
    gap_bwd_kernel<<<cdiv(total, 256), 256, 0, st>>>(dg, HW, C, total, da);
    SN_LAUNCH_CHECK("gap_bwd_kernel");
    return STABNET_OK;
}

int launch_fc_bwd(const float* x, const float* w, const float* y, const float* dy, int M, int K, int Nout, int relu,
                  float* dW, float* db, float* dx, float* scratch, size_t scratch_floats, hipStream_t st) {
    SN_REQUIRE(K % 4 == 0, "fc_bwd: K %% 4 != 0");
    fc_bwd_w_kernel<<<dim3(cdiv(K, 1024), Nout), 256, 0, st>>>(x, y, dy, M, K, Nout, relu, dW, db);
    SN_LAUNCH_CHECK("fc_bwd_w_kernel");
    if (dx != nullptr) {
        const int npb = fc_bwd_x_rows(K, Nout), splits = cdiv(Nout, npb);
        SN_REQUIRE(scratch != nullptr && scratch_floats >= (size_t)splits * M * K, "fc_bwd: scratch of %zu floats needed",
                   (size_t)splits * M * K);
        fc_bwd_x_kernel<<<dim3(cdiv(K, 64), splits), 256, 0, st>>>(w, y, dy, M, K, Nout, relu, npb, scratch);
        SN_LAUNCH_CHECK("fc_bwd_x_kernel");
        fc_bwd_x_reduce_kernel<<<cdiv((long)M * K, 256), 256, 0, st>>>(scratch, splits, (long)M * K, dx);
        SN_LAUNCH_CHECK("fc_bwd_x_reduce_kernel");
    }
    return STABNET_OK;
}

int launch_weight_decay(const float* params, float* grads, const long* seg_off, const long* seg_len, const float* seg_coef,
                        int nseg, float gscale, float* loss_out, float* partial, hipStream_t st) {
    weight_decay_kernel<<<dim3(64, nseg), 256, 0, st>>>(params, grads, seg_off, seg_len, seg_coef, gscale,
                                                        loss_out != nullptr ? partial : nullptr);
    SN_LAUNCH_CHECK("weight_decay_kernel");
    if (loss_out != nullptr) {
        weight_decay_finalize_kernel<<<1, 1024, 0, st>>>(partial, 64 * nseg, loss_out);
        SN_LAUNCH_CHECK("weight_decay_finalize_kernel");
    }
    return STABNET_OK;
}

int launch_adam(float* w, const float* g, const float* g2, float* m, float* v, long n, float alpha, float b1, float b2,
                float eps, float gscale, hipStream_t st) {
    adam_kernel<<<cdiv((n + 3) / 4, 256), 256, 0, st>>>(w, g, g2, m, v, n, alpha, b1, b2, eps, gscale);
    SN_LAUNCH_CHECK("adam_kernel");
    return STABNET_OK;
}
