// Convolution backward on the gfx950 matrix cores (exact float32 MFMA), the autodiff of slim conv2d that
// `opt.minimize` (train_bundle_nobm.py:160) runs for every conv of resnet_v2_50:
//
//   wgrad : dW[n][(kh,kw,c)] += sum_m dY[m][n] * act(x)[pix(m,kh,kw)][c]          (this file, its own kernel)
//   dgrad : dX = conv(dY dilated by the forward stride, W flipped and transposed)   (the forward kernel of conv.hip
//                                                                                    on weights re-packed here)
//
// wgrad tiling: the reduction runs over output pixels m, which is the slow index of both operands in memory
// (dY[m][n], x[pix][c]), so tiles are staged in LDS as [pixel][64 channels] exactly as loaded and the MFMA operands
// are read with ds_read_b32 (lane&31 -> consecutive channels: conflict-free).  Block = 64 output channels x 64
// k-columns, 4 waves (32x32 each), 32 pixels per step, double-buffered; the M range is split over blockIdx.z.  Each split
// writes its partial tile to its own SLAB with plain stores; wgrad_reduce_kernel then adds the slabs in split order and
// accumulates into dW (both towers of the siamese step add into it, in stream order).  No float atomics: the step is
// reproducible bit for bit, and plain stores run at ~5x the rate of memory-side float atomics (MI355X_MICROARCH.md).
#include "conv.h"
#include "prof.h"
#include "train_layers.h"
#include "ring.h"
#include <algorithm>
#include <cstdlib>

typedef float f32x16w __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float* x;          // forward input NHWC [N,H,W,Cin]
    const float* dy;         // output gradient NHWC [N,Ho,Wo,Cout]
    const float* x2;         // second siamese tower (or null): blockIdx.z >= splits_per_tower works on (x2, dy2, in_scale2,
    const float* dy2;        //   in_shift2) -- one launch per layer for both towers, slabs [tower][split]
    const float* in_scale2;
    const float* in_shift2;
    int splits_per_tower;
    float* dw;               // OHWI [Cout][KH][KW][Cin], accumulated into (only when slab == nullptr: one split)
    float* slab;             // [splits][Cout][K] partial sums, or nullptr
    const float* in_scale;   // forward prologue (folded BN + ReLU) or null
    const float* in_shift;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, K, rows_per_split;
    float* bias_slab;        // "same" 1x1 kernel with BIAS: column sums of dY (the bias gradient of the layer) per (tower, split):
    float *dbias, *dbias2;   //    bias_slab [T * splits][Cout] (reduced with the weight slabs), or += into dbias (and dbias2) when nothing is split
    int rowrun;              // 1 (general kernel only): x is the TIGHT zero-bordered image [N][H + 2 pad][W + 2 pad][Cin] (+ slack) the stem's
                             //    ring kernel reads (ConvArgs::rowrun): k = kh * Rp + j, Rp = roundup(KW * Cin, 32), column j of filter row
                             //    kh is float j of the run that starts at padded pixel (oy * stride + kh, ox * stride): no validity test,
                             //    loads at 4-byte alignment; dw / slab are [Cout][KH][Rp] (columns j >= KW * Cin hold garbage: dropped later)
};

__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const WgradArgs p) {
    constexpr int BR = 32, BT = 64;
    __shared__ __attribute__((aligned(16))) float sY[2][BR][BT];
    __shared__ __attribute__((aligned(16))) float sA[2][BR][BT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BT, k0 = blockIdx.y * BT;
    const int tower = (int)blockIdx.z >= p.splits_per_tower ? 1 : 0;
    const int zsplit = (int)blockIdx.z - tower * p.splits_per_tower;
    const float* __restrict__ px = tower ? p.x2 : p.x;
    const float* __restrict__ pdy = tower ? p.dy2 : p.dy;
    const float* __restrict__ pscale = tower ? p.in_scale2 : p.in_scale;
    const float* __restrict__ pshift = tower ? p.in_shift2 : p.in_shift;
    const int m_begin = zsplit * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    if (m_begin >= m_end) return;

    // loader mapping: thread -> (pixel row r = tid/16 and r+16, float4 column q = tid%16)
    const int lq = tid & 15, lr = tid >> 4;
    const int kcol = k0 + lq * 4;                          // this thread's k-columns: fixed tap and channels
    const bool k_ok = kcol < p.K;
    const int run = p.rowrun ? p.K / p.KH : p.Cin;         // row-run operand: a "tap" is a filter row, its "channels" the run
    const int tap = k_ok ? kcol / run : 0;
    const int c_t = k_ok ? kcol - tap * run : 0;
    const int kh = p.rowrun ? tap : tap / p.KW, kw = p.rowrun ? 0 : tap - kh * p.KW;
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;       // (row-run operand only)
    const int ncol = n0 + lq * 4;
    const bool n_ok = ncol < p.Cout;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_pro = pscale != nullptr;
    if (has_pro && k_ok) {
        sc = *reinterpret_cast<const float4*>(pscale + c_t);
        sh = *reinterpret_cast<const float4*>(pshift + c_t);
    }
    // pixel coordinates of the two rows this thread loads, advanced incrementally by BR per step
    int img[2], oy[2], ox[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m_begin + lr + 16 * t;
        img[t] = m / (p.Ho * p.Wo);
        const int r = m - img[t] * (p.Ho * p.Wo);
        oy[t] = r / p.Wo;
        ox[t] = r - oy[t] * p.Wo;
    }
    float4 ry[2], ra[2];
    unsigned ok = 0;
    int m_cur = m_begin;
    auto load_tiles = [&]() {
        ok = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int m = m_cur + lr + 16 * t;
            const bool row_ok = m < m_end;
            const int iy = oy[t] * p.stride - p.pad + kh, ix = ox[t] * p.stride - p.pad + kw;
            const bool a_ok = row_ok && k_ok && (p.rowrun || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W));
            const unsigned yoff = (row_ok && n_ok) ? (unsigned)(m * p.Cout + ncol) : 0u;
            const unsigned aoff = !a_ok ? 0u
                                  : p.rowrun ? (unsigned)(((img[t] * Hp + oy[t] * p.stride + kh) * Wp + ox[t] * p.stride) * p.Cin + c_t)
                                             : (unsigned)(((img[t] * p.H + iy) * p.W + ix) * p.Cin + c_t);
            ry[t] = *reinterpret_cast<const float4*>(pdy + yoff);
            if (p.rowrun) {                                    // runs start at any 4-byte offset (Cin = 13)
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const f4u v = *reinterpret_cast<const f4u*>(px + aoff);
                ra[t] = make_float4(v.x, v.y, v.z, v.w);
            } else
            ra[t] = *reinterpret_cast<const float4*>(px + aoff);
            ok |= ((row_ok && n_ok) ? 1u : 0u) << (2 * t);
            ok |= (a_ok ? 2u : 0u) << (2 * t);
            // advance this row by BR pixels
            ox[t] += BR;
            while (ox[t] >= p.Wo) {
                ox[t] -= p.Wo;
                if (++oy[t] == p.Ho) { oy[t] = 0; ++img[t]; }
            }
        }
        m_cur += BR;
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float4 y = ry[t], a = ra[t];
            if (has_pro) {
                a.x = fmaxf(__builtin_fmaf(a.x, sc.x, sh.x), 0.f);
                a.y = fmaxf(__builtin_fmaf(a.y, sc.y, sh.y), 0.f);
                a.z = fmaxf(__builtin_fmaf(a.z, sc.z, sh.z), 0.f);
                a.w = fmaxf(__builtin_fmaf(a.w, sc.w, sh.w), 0.f);
            }
            if (!((ok >> (2 * t)) & 1u)) y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!((ok >> (2 * t)) & 2u)) a = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&sY[buf][lr + 16 * t][lq * 4]) = y;
            *reinterpret_cast<float4*>(&sA[buf][lr + 16 * t][lq * 4]) = a;
        }
    };

    f32x16w acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;   // wave's 32x32 sub-tile: rows = out channels, cols = k
    const int li = lane & 31, lh = lane >> 5;

    load_tiles();
    store_tiles(0);
    __syncthreads();
    const int steps = (m_end - m_begin + BR - 1) / BR;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) load_tiles();
#pragma unroll
        for (int kk = 0; kk < BR / 2; ++kk) {
            const float a = sY[buf][2 * kk + lh][wn + li];
            const float b = sA[buf][2 * kk + lh][wk + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (s + 1 < steps) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // C/D map: col = lane&31 (k), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (out channel)
    const int k = k0 + wk + li;
    if (k < p.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n >= p.Cout) continue;
            if (p.slab != nullptr) p.slab[((size_t)blockIdx.z * p.Cout + n) * p.K + k] = acc[r];
            else p.dw[(size_t)n * p.K + k] += acc[r];          // single split: this lane owns the element
        }
    }
}

// The same product for "same"-geometry convolutions at stride 1 (1x1 pad 0, 3x3 pad 1: 50 of the 53 layers of the regressor),
// with the K loop's address work moved out of the vector ALU.  There the output pixel m and its input pixel share one linear
// index, so both operands advance by a CONSTANT per 32-pixel step: dY rows by 32*Cout floats, x rows by 32*Cin floats -- a
// scalar running base plus a loop-invariant per-lane byte offset (the `saddr + voffset` load form).  Only the 3x3 needs per-row
// state: the (iy, ix) of its tap, advanced by (32 / W, 32 % W) with carries, to know whether the tap is in frame (an out-of-frame
// tap loads the row's own centre pixel instead -- always in frame -- and is zeroed).  The general kernel above recomputes pixel
// coordinates with divisions-by-loop and 64-bit addresses every step: 103 VALU instructions per 16 MFMAs, and on gfx950 every
// VALU instruction costs ~3 cycles of f32-MFMA issue.  Requirements (host-checked): stride 1, Ho == H,
// Wo == W, M % 32 == 0, Cout % 64 == 0, K % 64 == 0, Cin % 4 == 0, 32 / W + 1 < H.
// BIAS: the workgroups of k-tile 0 also sum their dY tiles over the pixels (per thread over its rows, then the 16 row-threads of
// a column quad in a fixed order): the bias gradient of the layer comes out of the pass that reads dY anyway.
template <int K3, int PRO /* BN + ReLU prologue on x */, int BIAS = 0>
__global__ __launch_bounds__(256) void conv_wgrad_same_f32_kernel(const WgradArgs p) {
    constexpr int BR = 32, BT = 64;
    __shared__ __attribute__((aligned(16))) float sY[2][BR][BT];
    __shared__ __attribute__((aligned(16))) float sA[2][BR][BT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BT, k0 = blockIdx.y * BT;
    const int tower = (int)blockIdx.z >= p.splits_per_tower ? 1 : 0;
    const int zsplit = (int)blockIdx.z - tower * p.splits_per_tower;
    const float* __restrict__ px = tower ? p.x2 : p.x;
    const float* __restrict__ pdy = tower ? p.dy2 : p.dy;
    const float* __restrict__ pscale = tower ? p.in_scale2 : p.in_scale;
    const float* __restrict__ pshift = tower ? p.in_shift2 : p.in_shift;
    const int m_begin = zsplit * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    if (m_begin >= m_end) return;
    const int steps = (m_end - m_begin) / BR;                  // exact: M and rows_per_split are multiples of 32

    const int lq = tid & 15, lr = tid >> 4;                    // float4 column, pixel rows lr and lr + 16
    const int kcol = k0 + lq * 4;
    const int tap = K3 ? kcol / p.Cin : 0;
    const int c_t = kcol - tap * p.Cin;
    const int kh = K3 ? tap / 3 : 0, kw = K3 ? tap - kh * 3 : 0;
    const int pad = K3 ? 1 : 0;
    constexpr bool has_pro = PRO != 0;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_pro) {
        sc = *reinterpret_cast<const float4*>(pscale + c_t);
        sh = *reinterpret_cast<const float4*>(pshift + c_t);
    }
    // loop-invariant per-lane byte offsets from the running (wave-uniform) bases
    unsigned voff_y[2], voff_a[2], voff_c[2];
    int iy[2], ix[2];                                          // K3: input coordinates of this lane's tap for its two rows
    const int HW = p.H * p.W;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = lr + 16 * t;
        voff_y[t] = 4u * (unsigned)(row * p.Cout + n0 + lq * 4);
        voff_a[t] = 4u * (unsigned)((row + kh * p.W + kw) * p.Cin + c_t);
        voff_c[t] = 4u * (unsigned)((row + pad * p.W + pad) * p.Cin + c_t);
        const int m = m_begin + row;
        const int r = m - (m / HW) * HW;
        const int oy = r / p.W;
        iy[t] = oy + kh - pad;
        ix[t] = r - oy * p.W + kw - pad;
    }
    const char* sy = reinterpret_cast<const char*>(pdy + (size_t)m_begin * p.Cout);
    const char* sa = reinterpret_cast<const char*>(px + ((long)m_begin - pad * p.W - pad) * p.Cin);   // (may point in front of x)
    const size_t y_step = (size_t)BR * p.Cout * 4, a_step = (size_t)BR * p.Cin * 4;
    const int dq = BR / p.W, dr = BR - dq * p.W;               // a step advances a row by dq image rows and dr columns
    const int x_hi = p.W + kw - pad, y_hi = p.H + kh - pad;    // wrap bounds in the shifted (tap) coordinates

    // (scalars and macros, not arrays captured by lambdas: with arrays the compiler parked the loaded tile in scratch memory and
    //  waited for the loads right behind their issue)
    float4 ry0, ry1, ra0, ra1;
    bool ok0 = true, ok1 = true;
    int iy0 = iy[0], iy1 = iy[1], ix0 = ix[0], ix1 = ix[1];
    const unsigned vy0 = voff_y[0], vy1 = voff_y[1], va0 = voff_a[0], va1 = voff_a[1], vc0 = voff_c[0], vc1 = voff_c[1];
#define SN_WG_LOAD_ROW(RY, RA, OK, IY, IX, VY, VA, VC)                                            \
    do {                                                                                          \
        unsigned va_ = (VA);                                                                      \
        if (K3) {                                                                                 \
            OK = (unsigned)(IY) < (unsigned)p.H && (unsigned)(IX) < (unsigned)p.W;                \
            va_ = OK ? va_ : (VC);                                                                \
            IX += dr;                                                                             \
            const bool wrap_ = IX >= x_hi;                                                        \
            IX -= wrap_ ? p.W : 0;                                                                \
            IY += dq + (wrap_ ? 1 : 0);                                                           \
            IY -= (IY >= y_hi) ? p.H : 0;                                                         \
        }                                                                                         \
        RY = *reinterpret_cast<const float4*>(sy + (VY));                                         \
        RA = *reinterpret_cast<const float4*>(sa + va_);                                          \
    } while (0)
#define SN_WG_LOAD_TILES()                                                 \
    do {                                                                   \
        SN_WG_LOAD_ROW(ry0, ra0, ok0, iy0, ix0, vy0, va0, vc0);            \
        SN_WG_LOAD_ROW(ry1, ra1, ok1, iy1, ix1, vy1, va1, vc1);            \
        sy += y_step;                                                      \
        sa += a_step;                                                      \
    } while (0)
#define SN_WG_STORE_ROW(RY, RA, OK, ROW, BUF)                                                     \
    do {                                                                                          \
        float4 a_ = (RA);                                                                         \
        if (has_pro) {                                                                            \
            a_.x = fmaxf(__builtin_fmaf(a_.x, sc.x, sh.x), 0.f);                                  \
            a_.y = fmaxf(__builtin_fmaf(a_.y, sc.y, sh.y), 0.f);                                  \
            a_.z = fmaxf(__builtin_fmaf(a_.z, sc.z, sh.z), 0.f);                                  \
            a_.w = fmaxf(__builtin_fmaf(a_.w, sc.w, sh.w), 0.f);                                  \
        }                                                                                         \
        if (K3 && !(OK)) a_ = make_float4(0.f, 0.f, 0.f, 0.f);                                    \
        *reinterpret_cast<float4*>(&sY[BUF][ROW][lq * 4]) = (RY);                                 \
        *reinterpret_cast<float4*>(&sA[BUF][ROW][lq * 4]) = a_;                                   \
        if (BIAS && do_bias) { bsum.x += (RY).x; bsum.y += (RY).y; bsum.z += (RY).z; bsum.w += (RY).w; } \
    } while (0)
#define SN_WG_STORE_TILES(BUF)                                  \
    do {                                                        \
        SN_WG_STORE_ROW(ry0, ra0, ok0, lr, BUF);                \
        SN_WG_STORE_ROW(ry1, ra1, ok1, lr + 16, BUF);           \
    } while (0)
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = BIAS && blockIdx.y == 0;

    f32x16w acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int li = lane & 31, lh = lane >> 5;

    SN_WG_LOAD_TILES();
    SN_WG_STORE_TILES(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) SN_WG_LOAD_TILES();                 // tile s + 1 -> registers (with its ok bits), under the MFMAs of tile s
        // fragment reads run two MFMAs ahead of their use
        float fa[2], fb[2], ga[2], gb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { fa[j] = sY[buf][2 * j + lh][wn + li]; fb[j] = sA[buf][2 * j + lh][wk + li]; }
#pragma unroll
        for (int kk = 0; kk < BR / 2; kk += 2) {
            if (kk + 2 < BR / 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { ga[j] = sY[buf][2 * (kk + 2 + j) + lh][wn + li]; gb[j] = sA[buf][2 * (kk + 2 + j) + lh][wk + li]; }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) { fa[j] = ga[j]; fb[j] = gb[j]; }
        }
        if (s + 1 < steps) SN_WG_STORE_TILES(buf ^ 1);
        __syncthreads();
    }
#undef SN_WG_LOAD_ROW
#undef SN_WG_LOAD_TILES
#undef SN_WG_STORE_ROW
#undef SN_WG_STORE_TILES
    const int k = k0 + wk + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p.slab != nullptr) p.slab[((size_t)blockIdx.z * p.Cout + n) * p.K + k] = acc[r];
        else p.dw[(size_t)n * p.K + k] += acc[r];
    }
    if (BIAS && do_bias) {
        // rows lr = 4 * wave + (lane >> 4): the wave's four row-threads of a column quad by two exchanges, the four waves through
        // LDS (the tiles are dead: the loop ended on a barrier), added in wave order
        float* red = &sY[0][0][0];                              // [4 waves][64 columns]
        float v[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += __shfl_xor(v[j], 16, 64);
            v[j] += __shfl_xor(v[j], 32, 64);
        }
        if (lane < 16) *reinterpret_cast<float4*>(red + wave * 64 + lq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        __syncthreads();
        if (tid < 64) {
            const float t = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
            if (p.bias_slab != nullptr) p.bias_slab[(size_t)blockIdx.z * p.Cout + n0 + tid] = t;
            else {
                p.dbias[n0 + tid] += t;
                if (p.dbias2 != nullptr) p.dbias2[n0 + tid] += t;
            }
        }
    }
}

// Wt[ci][kh][kw][co] = W[co][KH-1-kh][KW-1-kw][ci]  : the weights of the dgrad convolution (OHWI in, OHWI out)
// kperm (3x3 filters of stride-2 layers): the tap rows are stored in the order (1, 0, 2), see conv_igemm_body.h (MODE 2).
__device__ __forceinline__ int dgrad_tap_row(int kh_stored, int kperm) {
    return kperm ? (kh_stored == 0 ? 1 : (kh_stored == 1 ? 0 : 2)) : kh_stored;
}
__global__ __launch_bounds__(256) void pack_dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                 int Cout, int KH, int KW, int Cin, int kperm) {
    const long total = (long)Cout * KH * KW * Cin;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int co = (int)(q % Cout);
    long r = q / Cout;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = dgrad_tap_row((int)(r % KH), kperm);
    const int ci = (int)(r / KH);
    wt[q] = w[(((size_t)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
}

// Split of the pixel range: ~1024 blocks in all, at least 256 pixels (8 steps) per block.
int wgrad_splits(int Cout, int K, int M, int* rows_per_split) {
    static const int target = []() { const char* v = getenv("STABNET_WGRAD_BLOCKS"); return v ? atoi(v) : 1024; }();   // (tuning switch)
    const int tiles = cdiv(Cout, 64) * cdiv(K, 64);
    int splits = std::max(1, std::min(cdiv(target, tiles), cdiv(M, 256)));
    splits = std::min(splits, 65535);
    const int rps = cdiv(cdiv(M, splits), 32) * 32;
    if (rows_per_split) *rows_per_split = rps;
    return cdiv(M, rps);
}
// slab floats one launch over T towers needs (none when a single block owns every element: T * splits == 1)
size_t wgrad_slab_floats(int Cout, int K, int M, int T, int with_bias) {
    const int s = T * wgrad_splits(Cout, K, M, nullptr);
    return s > 1 ? (size_t)s * Cout * (K + (with_bias ? 1 : 0)) : 0;
}

static bool wgrad_same_geometry(const WgradArgs& a, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    static const bool fast_ok = getenv("STABNET_WGRAD_SAME") == nullptr || atoi(getenv("STABNET_WGRAD_SAME")) != 0;     // debug switch
    return fast_ok && stride == 1 && a.Ho == H && a.Wo == W && KH == KW && (KH == 1 ? pad == 0 : (KH == 3 && pad == 1)) &&
           a.M % 32 == 0 && Cout % 64 == 0 && a.K % 64 == 0 && 32 / W + 1 < H &&
           ((long)a.M + 2L * W + 34) * Cin * 4 < (1L << 32) && ((long)32 * Cout + Cout) * 4 < (1L << 32);
}
// 1 if wgrad_launch_g can also produce the layer's bias gradient (1x1 layers on the "same" kernel)
int wgrad_bias_fusable(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    static const bool on = getenv("STABNET_WGRAD_BIAS") == nullptr || atoi(getenv("STABNET_WGRAD_BIAS")) != 0;          // debug switch
    WgradArgs a{};
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.M = N * a.Ho * a.Wo;
    a.K = KH * KW * Cin;
    return on && KH == 1 && Cin % 4 == 0 && Cout % 256 == 0 && wgrad_same_geometry(a, H, W, Cin, Cout, KH, KW, stride, pad);
}

// dw += sum_z slab[0][z] (+ sum_z slab[1][z]) (z ascending, tower 0 then tower 1) for every entry of the table; one float4 per thread.
__device__ __forceinline__ void slab_sum(const float* __restrict__ s, int splits, long elems, float4& acc) {
    int z = 0;
    for (; z + 3 < splits; z += 4) {                            // four slab loads in flight, added in slab order
        const float4 v0 = *reinterpret_cast<const float4*>(s + (size_t)z * elems);
        const float4 v1 = *reinterpret_cast<const float4*>(s + (size_t)(z + 1) * elems);
        const float4 v2 = *reinterpret_cast<const float4*>(s + (size_t)(z + 2) * elems);
        const float4 v3 = *reinterpret_cast<const float4*>(s + (size_t)(z + 3) * elems);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(s + (size_t)z * elems);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ grads, const WgradReduceTable t) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;          // float4 index over all entries
    if (q >= t.prefix[t.n]) return;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.prefix[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const WgradReduceEntry e = t.e[lo];
    const long i = (q - t.prefix[lo]) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    slab_sum(e.slab[0] + i, e.splits, e.elems, acc);
    if (e.slab[1] != nullptr) slab_sum(e.slab[1] + i, e.splits, e.elems, acc);
    float4* d = reinterpret_cast<float4*>(sn_float_at(grads, e.dw_off + i));
    float4 o = *d;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *d = o;
}

// The same reduction with the slab list of an element cut into FOUR contiguous runs summed by four threads and combined in run
// order through LDS (still one fixed order): a layer of 64 x 576 weights has 226 slabs of 9 216 float4 -- 36 blocks walking 226
// dependent-latency steps each in the flat form (the launch ran at 2.2 TB/s); here 144 blocks walk 57.  Needs every entry's
// element count to be a multiple of 256 (64 float4 columns per block never straddle two entries).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(float* __restrict__ grads, const WgradReduceTable t) {
    __shared__ float4 sh[4][64];
    const long q0 = (long)blockIdx.x * 64;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.prefix[mid] <= q0) lo = mid; else hi = mid - 1;
    }
    const WgradReduceEntry e = t.e[lo];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (q0 - t.prefix[lo] + col) * 4;
    const int per = (e.splits + 3) >> 2;
    const int z0 = min(e.splits, grp * per), z1 = min(e.splits, z0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    slab_sum(e.slab[0] + i + (size_t)z0 * e.elems, z1 - z0, e.elems, acc);
    if (e.slab[1] != nullptr) slab_sum(e.slab[1] + i + (size_t)z0 * e.elems, z1 - z0, e.elems, acc);
    sh[grp][col] = acc;
    __syncthreads();
    if (grp == 0) {
        float4 s = sh[0][col];
        for (int g = 1; g < 4; ++g) { const float4 u = sh[g][col]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
        float4* d = reinterpret_cast<float4*>(sn_float_at(grads, e.dw_off + i));
        float4 o = *d;
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *d = o;
    }
}

int wgrad_reduce_flush(float* grads, WgradReduceTable& t, hipStream_t st) {
    if (t.n == 0) return STABNET_OK;
    bool by4 = true;
    for (int i = 0; i < t.n; ++i) by4 = by4 && t.e[i].elems % 256 == 0 && t.e[i].slab[1] == nullptr;
    if (by4) wgrad_reduce4_kernel<<<(unsigned)(t.prefix[t.n] / 64), 256, 0, st>>>(grads, t);
    else wgrad_reduce_kernel<<<cdiv(t.prefix[t.n], 256), 256, 0, st>>>(grads, t);
    SN_LAUNCH_CHECK("wgrad_reduce_kernel");
    t.n = 0; t.prefix[0] = 0;
    return STABNET_OK;
}

// dw_base + dw_off: where the gradient accumulates.  T = 1 or 2 towers (x[t], dy[t], in_scale[t], in_shift[t]): ONE launch,
// grid.z = T * splits, slabs [tower][split] at slab_base + *slab_cursor (the cursor advances); the entry is appended to
// `table` and reduced (tower 0's splits, then tower 1's) by the next wgrad_reduce_flush.  table == nullptr (stand-alone use):
// the reduce is launched here.
int wgrad_launch_g(int T, const float* const* x, const float* const* dy, float* dw_base, long dw_off, const float* const* in_scale,
                   const float* const* in_shift, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   float* slab_base, size_t* slab_cursor, size_t slab_capacity, WgradReduceTable* table, hipStream_t st, Prof* prof,
                   int rowrun, long bias_off, long bias_off2) {
    WgradArgs a{};
    a.x = x[0]; a.dy = dy[0]; a.dw = sn_float_at(dw_base, dw_off); a.in_scale = in_scale ? in_scale[0] : nullptr; a.in_shift = in_shift ? in_shift[0] : nullptr;
    if (T == 2) { a.x2 = x[1]; a.dy2 = dy[1]; a.in_scale2 = in_scale ? in_scale[1] : nullptr; a.in_shift2 = in_shift ? in_shift[1] : nullptr; }
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.M = N * a.Ho * a.Wo;
    a.K = rowrun ? KH * 32 * cdiv(KW * Cin, 32) : KH * KW * Cin;
    a.rowrun = rowrun ? 1 : 0;
    SN_REQUIRE(T == 1 || T == 2, "wgrad: 1 or 2 towers");
    SN_REQUIRE((rowrun || Cin % 4 == 0) && Cout % 4 == 0, "wgrad: channel counts must be multiples of 4");
    // the slab reduction adds into the gradients 16 B per thread (float4 read-modify-write), weights and bias sums alike
    SN_REQUIRE(((size_t)sn_float_at(dw_base, dw_off) & 15) == 0 && (bias_off < 0 || ((size_t)sn_float_at(dw_base, bias_off) & 15) == 0) &&
                   (bias_off2 < 0 || ((size_t)sn_float_at(dw_base, bias_off2) & 15) == 0),
               "wgrad: dw and d_bias must be 16-byte aligned");
    SN_REQUIRE((long)a.M * Cout < (1L << 31) && ((long)N * (H + 2 * pad) + 1) * (W + 2 * pad) * Cin < (1L << 31),
               "wgrad: tensors must have < 2^31 elements");
    const int splits = wgrad_splits(Cout, a.K, a.M, &a.rows_per_split);
    a.splits_per_tower = splits;
    const int total = T * splits;
    const size_t elems = (size_t)Cout * a.K;
    const bool bias = bias_off >= 0;
    SN_REQUIRE(!bias || (!rowrun && wgrad_bias_fusable(N, H, W, Cin, Cout, KH, KW, stride, pad)), "wgrad: this layer cannot carry its bias gradient");
    SN_REQUIRE(bias || bias_off2 < 0, "wgrad: second bias without a first");
    a.dbias = bias ? sn_float_at(dw_base, bias_off) : nullptr;
    a.dbias2 = bias_off2 >= 0 ? sn_float_at(dw_base, bias_off2) : nullptr;
    WgradReduceTable local{};
    if (total > 1) {
        SN_REQUIRE(slab_base != nullptr && slab_cursor != nullptr, "wgrad: %d splits need a slab workspace", total);
        SN_REQUIRE(*slab_cursor + (size_t)total * elems <= slab_capacity, "wgrad: slab workspace overrun (%zu + %zu > %zu floats)",
                   *slab_cursor, (size_t)total * elems, slab_capacity);          // checked BEFORE anything is written
        a.slab = slab_base + *slab_cursor;
        WgradReduceTable* t = table ? table : &local;
        if (t->n == WGRAD_TABLE_MAX) {
            int rc = wgrad_reduce_flush(dw_base, *t, st);
            if (rc) return rc;
        }
        t->e[t->n] = {dw_off, (long)elems, {a.slab, nullptr}, total};
        t->prefix[t->n + 1] = t->prefix[t->n] + (long)(elems / 4);
        ++t->n;
        *slab_cursor += (size_t)total * elems;
        if (bias) {                                          // [tower][split][Cout] column sums of dY, reduced like a weight slab
            SN_REQUIRE(*slab_cursor + (size_t)total * Cout <= slab_capacity, "wgrad: slab workspace overrun (bias sums)");
            a.bias_slab = slab_base + *slab_cursor;
            const long offs[2] = {bias_off, bias_off2};
            for (int i = 0; i < 2 && offs[i] >= 0; ++i) {
                if (t->n == WGRAD_TABLE_MAX) {
                    int rc = wgrad_reduce_flush(dw_base, *t, st);
                    if (rc) return rc;
                }
                t->e[t->n] = {offs[i], (long)Cout, {a.bias_slab, nullptr}, total};
                t->prefix[t->n + 1] = t->prefix[t->n] + (long)(Cout / 4);
                ++t->n;
            }
            *slab_cursor += (size_t)total * Cout;
        }
    }
    const bool rec = prof != nullptr && prof->begin(st);
    const bool same = !rowrun && wgrad_same_geometry(a, H, W, Cin, Cout, KH, KW, stride, pad);
    const dim3 grid(cdiv(Cout, 64), cdiv(a.K, 64), total);
    const bool pro = a.in_scale != nullptr;
    if (same && KH == 1 && bias) {
        if (pro) conv_wgrad_same_f32_kernel<0, 1, 1><<<grid, 256, 0, st>>>(a);
        else conv_wgrad_same_f32_kernel<0, 0, 1><<<grid, 256, 0, st>>>(a);
    } else if (same && KH == 1) {
        if (pro) conv_wgrad_same_f32_kernel<0, 1><<<grid, 256, 0, st>>>(a);
        else conv_wgrad_same_f32_kernel<0, 0><<<grid, 256, 0, st>>>(a);
    } else if (same) {
        if (pro) conv_wgrad_same_f32_kernel<1, 1><<<grid, 256, 0, st>>>(a);
        else conv_wgrad_same_f32_kernel<1, 0><<<grid, 256, 0, st>>>(a);
    } else {
        conv_wgrad_f32_kernel<<<grid, 256, 0, st>>>(a);
    }
    if (rec) prof->end(st, same ? PK_KERNEL_WGRAD_SAME + 2 * (KH == 3 ? 1 : 0) + (pro ? 1 : 0) + ((KH == 1 && bias) ? 4 : 0) : PK_KERNEL_WGRAD, 2.0 * T * a.M * (double)a.K * Cout,
                       4.0 * T * ((double)a.M * Cout + (double)N * H * W * Cin + (double)a.K * Cout * splits), Cout, a.K, T * a.M, total);
    SN_LAUNCH_CHECK("conv_wgrad_f32_kernel");
    if (total > 1 && table == nullptr) return wgrad_reduce_flush(dw_base, local, st);
    return STABNET_OK;
}

// dw [Cout][KH][KW][CinPad] += tmp [Cout][KH][Rp] (the row-run wgrad's layout: column kw * Cin + c of filter row kh; columns
// beyond KW * Cin are not weights)
__global__ __launch_bounds__(256) void wgrad_rowrun_scatter_kernel(const float* __restrict__ tmp, float* __restrict__ dw, int Cout,
                                                                   int KH, int KW, int Cin, int CinPad, int Rp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * KH * KW * Cin) return;
    const int c = i % Cin, kw = (i / Cin) % KW, kh = (i / (Cin * KW)) % KH, n = i / (Cin * KW * KH);
    dw[(((size_t)n * KH + kh) * KW + kw) * CinPad + c] += tmp[((size_t)n * KH + kh) * Rp + kw * Cin + c];
}
int launch_wgrad_rowrun_scatter(const float* tmp, float* dw, int Cout, int KH, int KW, int Cin, int CinPad, hipStream_t st) {
    const int Rp = 32 * cdiv(KW * Cin, 32);
    wgrad_rowrun_scatter_kernel<<<cdiv((long)Cout * KH * KW * Cin, 256), 256, 0, st>>>(tmp, dw, Cout, KH, KW, Cin, CinPad, Rp);
    SN_LAUNCH_CHECK("wgrad_rowrun_scatter_kernel");
    return STABNET_OK;
}

int wgrad_launch(const float* x, const float* dy, float* dw_base, long dw_off, const float* in_scale, const float* in_shift,
                 int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* slab_base,
                 size_t* slab_cursor, size_t slab_capacity, WgradReduceTable* table, hipStream_t st, Prof* prof) {
    return wgrad_launch_g(1, &x, &dy, dw_base, dw_off, in_scale ? &in_scale : nullptr, in_shift ? &in_shift : nullptr, N, H, W, Cin,
                          Cout, KH, KW, stride, pad, slab_base, slab_cursor, slab_capacity, table, st, prof);
}

// All dgrad weight tensors of a net in ONE launch (the per-layer version cost 104 launches of ~6 us per training step).
// The layer table travels as a kernel argument; a thread finds its layer by binary search over the element prefix sums.
__global__ __launch_bounds__(256) void pack_dgrad_weights_all_kernel(const float* __restrict__ params, float* __restrict__ wt,
                                                                     const PackTable t) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= t.prefix[t.n]) return;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.prefix[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = t.d[lo];
    const long e = q - t.prefix[lo];
    const int co = (int)(e % d.Cout);
    long r = e / d.Cout;
    const int kw = (int)(r % d.K); r /= d.K;
    const int kh = dgrad_tap_row((int)(r % d.K), d.kperm);
    const int ci = (int)(r / d.K);
    wt[q] = params[d.w_off + (((size_t)co * d.K + (d.K - 1 - kh)) * d.K + (d.K - 1 - kw)) * d.Cin + ci];
}

// The same re-pack as 32 x 32 (Cout x Cin) tiles of one filter tap through LDS: 128-B runs on both sides (the element-wise
// form above reads with a stride of K*K*Cin floats).  Needs Cout % 32 == 0 and Cin % 32 == 0 for every layer, so that a
// layer's element prefix / 1024 is its tile prefix.
__global__ __launch_bounds__(256) void pack_dgrad_weights_tiled_kernel(const float* __restrict__ params, float* __restrict__ wt,
                                                                       const PackTable t) {
    __shared__ float tile[32][33];
    const long q = (long)blockIdx.x * 1024;
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.prefix[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = t.d[lo];
    long r = (q - t.prefix[lo]) >> 10;                    // tile index inside the layer: co tile fastest, then tap, then ci tile
    const int cot = (int)(r % (d.Cout >> 5)); r /= (d.Cout >> 5);
    const int kw = (int)(r % d.K); r /= d.K;
    const int kh = (int)(r % d.K);                         // stored tap row; its source row:
    const int khs = dgrad_tap_row(kh, d.kperm);
    const int cit = (int)(r / d.K);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float* src = params + d.w_off + ((size_t)(d.K - 1 - khs) * d.K + (d.K - 1 - kw)) * d.Cin + cit * 32 + lx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cot * 32 + ly + 8 * i;
        tile[ly + 8 * i][lx] = src[(size_t)co * d.K * d.K * d.Cin];
    }
    __syncthreads();
    float* dst = wt + t.prefix[lo] + cot * 32 + lx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = cit * 32 + ly + 8 * i;
        dst[(((size_t)ci * d.K + kh) * d.K + kw) * d.Cout] = tile[lx][ly + 8 * i];
    }
}

int pack_dgrad_weights_all(const float* params, float* wt, const PackTable& t, hipStream_t st) {
    if (t.n == 0) return STABNET_OK;
    bool tiled = true;
    for (int i = 0; i < t.n; ++i) tiled = tiled && t.d[i].Cout % 32 == 0 && t.d[i].Cin % 32 == 0;
    if (tiled) {
        pack_dgrad_weights_tiled_kernel<<<(unsigned)(t.prefix[t.n] >> 10), 256, 0, st>>>(params, wt, t);
        SN_LAUNCH_CHECK("pack_dgrad_weights_tiled_kernel");
        return STABNET_OK;
    }
    pack_dgrad_weights_all_kernel<<<cdiv(t.prefix[t.n], 256), 256, 0, st>>>(params, wt, t);
    SN_LAUNCH_CHECK("pack_dgrad_weights_all_kernel");
    return STABNET_OK;
}

int pack_dgrad_weights(const float* w, float* wt, int Cout, int KH, int KW, int Cin, int kperm, hipStream_t st) {
    const long total = (long)Cout * KH * KW * Cin;
    pack_dgrad_weights_kernel<<<cdiv(total, 256), 256, 0, st>>>(w, wt, Cout, KH, KW, Cin, kperm);
    SN_LAUNCH_CHECK("pack_dgrad_weights_kernel");
    return STABNET_OK;
}

// dX [N,H,W,Cin] = dgrad of a forward conv (x [N,H,W,Cin] -> y [N,Ho,Wo,Cout], KHxKW, stride, pad) from dY and the
// re-packed weights wt [Cin][KH][KW][Cout].  residual (optional, same shape as dX) is added (gradient accumulation).
int dgrad_launch(const float* dy, const float* wt, float* dx, const float* residual, int N, int H, int W, int Cin,
                 int Cout, int KH, int KW, int stride, int pad, float* splitk_ws, size_t splitk_bytes, hipStream_t st,
                 Prof* prof, const float* wt_img) {
    ConvArgs a{};
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    a.x = dy; a.w = wt; a.y = dx; a.residual = residual;
    a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin; a.KH = KH; a.KW = KW;
    a.stride = 1; a.pad = KH - 1 - pad; a.up = stride;
    a.Ho = H; a.Wo = W; a.res_H = H; a.res_W = W; a.res_stride = 1;
    SN_REQUIRE(a.pad >= 0 && Cout % 16 == 0, "dgrad: unsupported geometry (pad %d, Cout %d)", a.pad, Cout);
    const size_t need = conv_plan(a);
    if (need > splitk_bytes) {
        stabnet_set_error("dgrad: split-K workspace %zu B < %zu B", splitk_bytes, need);
        return STABNET_ERR_WORKSPACE;
    }
    a.partial = splitk_ws;
    return conv_launch(a, st, prof, wt_img != nullptr ? 4 : 0, wt_img);
}

extern "C" {

/* d conv2d / d weights, accumulated into dw (OHWI [Cout][KH][KW][Cin], NOT zeroed here).  x: the forward input,
 * (in_scale, in_shift): the forward's folded-BN + ReLU prologue (or NULL), dy [N,Ho,Wo,Cout].
 * workspace: stabnet_conv2d_wgrad_workspace_bytes() (per-split partial slabs; reduced in split order: reproducible). */
size_t stabnet_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    return wgrad_slab_floats(Cout, KH * KW * Cin, N * Ho * Wo, 1) * sizeof(float) + 16;
}
int stabnet_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* in_scale, const float* in_shift,
                         int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* workspace,
                         size_t workspace_bytes, void* stream) {
    SN_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
    SN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_wgrad: in_scale and in_shift go together");
    SN_REQUIRE(workspace != nullptr && workspace_bytes >= stabnet_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad),
               "conv2d_wgrad: workspace too small");
    size_t cursor = 0;
    return wgrad_launch(x, dy, dw, 0, in_scale, in_shift, N, H, W, Cin, Cout, KH, KW, stride, pad,
                        static_cast<float*>(workspace), &cursor, workspace_bytes / sizeof(float), nullptr, (hipStream_t)stream, nullptr);
}

/* 1x1 stride-1 layers with Cout % 256 == 0 (the unit-closing convolutions of the regressor): dW as above AND the bias gradient
 * d_bias [Cout] += column sums of dy, from the same pass over dy (the training step's form).  STABNET_ERR_BAD_ARG for a
 * geometry the stride-1 kernel does not take. */
size_t stabnet_conv2d_wgrad_bias_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    return wgrad_slab_floats(Cout, Cin, N * H * W, 1, 1) * sizeof(float) + 16;
}
int stabnet_conv2d_wgrad_bias(const float* x, const float* dy, float* dw, float* d_bias, const float* in_scale, const float* in_shift,
                              int N, int H, int W, int Cin, int Cout, void* workspace, size_t workspace_bytes, void* stream) {
    SN_REQUIRE(x && dy && dw && d_bias, "conv2d_wgrad_bias: null pointer");
    SN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_wgrad_bias: in_scale and in_shift go together");
    SN_REQUIRE(workspace != nullptr && workspace_bytes >= stabnet_conv2d_wgrad_bias_workspace_bytes(N, H, W, Cin, Cout),
               "conv2d_wgrad_bias: workspace too small");
    size_t cursor = 0;
    const float* xs[1] = {x};
    const float* dys[1] = {dy};
    const float* sc[1] = {in_scale};
    const float* sh[1] = {in_shift};
    // (one base for both gradients: the bias is addressed relative to dw, as the training step addresses its flat gradient buffer)
    return wgrad_launch_g(1, xs, dys, dw, 0, in_scale ? sc : nullptr, in_scale ? sh : nullptr, N, H, W, Cin, Cout, 1, 1, 1, 0,
                          static_cast<float*>(workspace), &cursor, workspace_bytes / sizeof(float), nullptr, (hipStream_t)stream, nullptr,
                          0, sn_float_distance(dw, d_bias), -1);
}

/* The same gradient for a layer whose channel count is NOT a multiple of 4 (the 13-channel stem, train_bundle_nobm.py's
 * x_tensor): x [N,H,W,Cin] tight, dw OHWI [Cout][KH][KW][CinPad] (ACCUMULATED into; the CinPad - Cin pad channels are left
 * alone).  x is embedded into a zero-bordered image in `workspace` and read as filter-row runs (WgradArgs::rowrun), the way the
 * training step's stem does it (there the bordered image is the forward's own operand). */
static size_t rowrun_x_floats(int N, int H, int W, int Cin, int pad) { return (((size_t)N * (H + 2 * pad) + 1) * (W + 2 * pad) * Cin + 63) & ~(size_t)63; }
size_t stabnet_conv2d_wgrad_rowrun_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1, Rp = 32 * cdiv(KW * Cin, 32);
    return (rowrun_x_floats(N, H, W, Cin, pad) + (size_t)Cout * KH * Rp + wgrad_slab_floats(Cout, KH * Rp, N * Ho * Wo, 1)) * sizeof(float) + 16;
}
int stabnet_conv2d_wgrad_rowrun(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int CinPad, int Cout,
                                int KH, int KW, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    SN_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad_rowrun: null pointer");
    SN_REQUIRE(CinPad >= Cin && Cin >= 1 && pad >= 0 && (W + 2 * pad) * Cin >= 64, "conv2d_wgrad_rowrun: bad geometry");
    SN_REQUIRE(workspace_bytes >= stabnet_conv2d_wgrad_rowrun_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad),
               "conv2d_wgrad_rowrun: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Rp = 32 * cdiv(KW * Cin, 32);
    float* xb = static_cast<float*>(workspace);
    float* tmp = xb + rowrun_x_floats(N, H, W, Cin, pad);
    float* slabs = tmp + (size_t)Cout * KH * Rp;
    int rc = launch_embed_border(x, N, H, W, Cin, pad, xb, st);
    if (rc) return rc;
    if (hipMemsetAsync(tmp, 0, (size_t)Cout * KH * Rp * sizeof(float), st) != hipSuccess) return STABNET_ERR_LAUNCH;
    size_t cursor = 0;
    const float* xs[1] = {xb};
    const float* dys[1] = {dy};
    rc = wgrad_launch_g(1, xs, dys, tmp, 0, nullptr, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, slabs, &cursor,
                        wgrad_slab_floats(Cout, KH * Rp, N * ((H + 2 * pad - KH) / stride + 1) * ((W + 2 * pad - KW) / stride + 1), 1),
                        nullptr, st, nullptr, 1);
    if (rc) return rc;
    return launch_wgrad_rowrun_scatter(tmp, dw, Cout, KH, KW, Cin, CinPad, st);
}

/* d conv2d / d input: dx [N,H,W,Cin] (+ residual if given) from dy [N,Ho,Wo,Cout] and the forward weights
 * w_ohwi [Cout][KH][KW][Cin].  workspace: Cout*KH*KW*Cin floats for the re-packed weights, followed by
 * stabnet_conv2d_dgrad_workspace_bytes() - that.  Cout % 16 == 0. */
size_t stabnet_conv2d_dgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    ConvArgs a{};
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = KH - 1 - pad;
    a.up = stride; a.Ho = H; a.Wo = W;
    return (size_t)Cout * KH * KW * Cin * sizeof(float) + conv_plan(a) + 256;
}
int stabnet_conv2d_dgrad(const float* dy, const float* w_ohwi, float* dx, const float* residual, int N, int H, int W,
                         int Cin, int Cout, int KH, int KW, int stride, int pad, void* workspace,
                         size_t workspace_bytes, void* stream) {
    SN_REQUIRE(dy && w_ohwi && dx && workspace, "conv2d_dgrad: null pointer");
    const size_t wbytes = (size_t)Cout * KH * KW * Cin * sizeof(float);
    SN_REQUIRE(workspace_bytes >= stabnet_conv2d_dgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad),
               "conv2d_dgrad: workspace too small");
    float* wt = static_cast<float*>(workspace);
    int rc = pack_dgrad_weights(w_ohwi, wt, Cout, KH, KW, Cin, dgrad_kperm(KH, KW, stride), (hipStream_t)stream);
    if (rc) return rc;
    const size_t off = (wbytes + 255) & ~(size_t)255;
    return dgrad_launch(dy, wt, dx, residual, N, H, W, Cin, Cout, KH, KW, stride, pad,
                        reinterpret_cast<float*>(static_cast<char*>(workspace) + off), workspace_bytes - off,
                        (hipStream_t)stream, nullptr);
}

}  // extern "C"
