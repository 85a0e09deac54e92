// Small layers of the regressor (everything that is not a convolution), gfx950.  All HBM-bound elementwise /
// reduction kernels: float4 accesses along the NHWC channel dimension.
//   pad_channels   : x_tensor [N,H,W,13] -> [N,H,W,16] (zeros in the pad) so the stem runs on aligned 16-B pixels
//   max_pool       : slim max_pool2d(3, stride 2, 'SAME')                       
//   bn_fold        : moving-average BN -> per-channel (scale, shift)            tf.nn.batch_normalization form
//   gap_bn_relu    : postnorm BN + ReLU + reduce_mean([1,2])                    s_net_bundle_nobm.py:254
//   fc             : slim.fully_connected / output_layer for small batches      s_net_bundle_nobm.py:256-259, resnet.py:44-56
#include "layers.h"
#include <algorithm>

__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long npix, int C, int Cp) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;      // one thread per (pixel, float4 of the padded pixel)
    const int f4 = Cp / 4;
    if (q >= npix * f4) return;
    const long pix = q / f4;
    const int c = (int)(q - pix * f4) * 4;
    const float* s = x + pix * C;
    float4 v;
    v.x = (c + 0 < C) ? s[c + 0] : 0.f;
    v.y = (c + 1 < C) ? s[c + 1] : 0.f;
    v.z = (c + 2 < C) ? s[c + 2] : 0.f;
    v.w = (c + 3 < C) ? s[c + 3] : 0.f;
    *reinterpret_cast<float4*>(y + pix * Cp + c) = v;
}

// (scale, shift) non-null: the consumer's folded BN + ReLU is applied to the pooled value (inference plan: the first unit of
// block1 has a projection shortcut, so nothing reads the raw pooled tensor).
__global__ __launch_bounds__(256) void max_pool_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                       int W, int C, int Ho, int Wo, int k, int stride, int pt, int pl,
                                                       const float* __restrict__ scale, const float* __restrict__ shift) {
    // (XCD-banded ids: the windows of neighbouring output rows share an input row; 32-bit index math, host-checked)
    const unsigned q = sn_xcd_band(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    const unsigned c4n = (unsigned)C / 4u;
    if (q >= (unsigned)N * Ho * Wo * c4n) return;
    const int c = (int)(q % c4n) * 4;
    unsigned r = q / c4n;
    const int ox = (int)(r % (unsigned)Wo); r /= (unsigned)Wo;
    const int oy = (int)(r % (unsigned)Ho);
    const int n = (int)(r / (unsigned)Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * stride - pt + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * stride - pl + dx;
            if (ix < 0 || ix >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * C + c);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    if (scale != nullptr) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        m.x = fmaxf(__builtin_fmaf(m.x, sc.x, sh.x), 0.f); m.y = fmaxf(__builtin_fmaf(m.y, sc.y, sh.y), 0.f);
        m.z = fmaxf(__builtin_fmaf(m.z, sc.z, sh.z), 0.f); m.w = fmaxf(__builtin_fmaf(m.w, sc.w, sh.w), 0.f);
    }
    *reinterpret_cast<float4*>(y + (((long)n * Ho + oy) * Wo + ox) * C + c) = m;
}

// scale = rsqrt(var + eps) * gamma ; shift = beta - mean * scale       ([external] tf.nn.batch_normalization)
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, int G, float* __restrict__ scale,
                                                      float* __restrict__ shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    const float inv = (1.0f / sqrtf(var[i] + eps)) * gamma[i];
    scale[i] = inv;
    shift[i] = beta[i] - mean[i] * inv;
}

// out[n][c] = mean over (h,w) of relu(x*scale[c] + shift[c]).  Two stages so that a single image still fills the chip:
// stage 1: grid (C/64, chunks, N), block = 16 channel-quads x 16 row lanes, partial sums [N][chunks][C];
// stage 2: fixed-order sum of the chunk partials (deterministic) and the division by H*W.
__global__ __launch_bounds__(256) void gap_bn_relu_partial_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int HW, int C,
                                                                  int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float4 part[16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + q * 4;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(HW, r0 + rows_per_chunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float* p = x + (size_t)n * HW * C + c;
        for (int ib = r0 + rl; ib < r1; ib += 128) {       // eight rows per trip, their loads issued together (same row order)
            float4 xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = ib + 16 * j;
                xv[j] = (i < r1) ? *reinterpret_cast<const float4*>(p + (size_t)i * C) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ib + 16 * j < r1) {
                    const float4 v = xv[j];
                    s.x += fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), 0.f); s.y += fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), 0.f);
                    s.z += fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), 0.f); s.w += fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), 0.f);
                }
            }
        }
    }
    part[rl][q] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float4 t = part[0][q];
        for (int i = 1; i < 16; ++i) { const float4 u = part[i][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        *reinterpret_cast<float4*>(partial + ((size_t)n * gridDim.y + blockIdx.y) * C + c) = t;
    }
}

__global__ __launch_bounds__(256) void gap_finalize_kernel(const float* __restrict__ partial, int chunks, int HW, int C,
                                                           float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (c >= C) return;
    float t = 0.f;
    for (int k = 0; k < chunks; ++k) t += partial[((size_t)n * chunks + k) * C + c];
    out[(size_t)n * C + c] = t / (float)HW;
}

// y[m][n] = act( sum_k x[m][k] * W[n][k] + b[n] ), M <= 16 rows per pass.  One wave per output channel n:
// the weight row is streamed once (float4 per lane), the M activations rows come from L1/L2.
template <int MR>
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ b, float* __restrict__ y, int M, int K,
                                                 int Nout, int relu, int m_begin) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= Nout) return;
    float acc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) acc[i] = 0.f;
    const float* wr = w + (size_t)n * K;
    // the weight row is the only HBM stream of this kernel: its (up to eight) 16-B loads per lane are issued together, not one
    // per loop trip (at batch 1 the trips were eight dependent HBM latencies); the sums keep their order
    for (int kb = 0; kb < K; kb += 8 * 256) {
        float4 wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb + j * 256 + lane * 4;
            wv[j] = (k < K) ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb + j * 256 + lane * 4;
            if (k < K) {
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const int m = m_begin + i;
                    if (m < M) {
                        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * K + k);
                        acc[i] += ((xv.x * wv[j].x + xv.y * wv[j].y) + xv.z * wv[j].z) + xv.w * wv[j].w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        float v = acc[i];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        const int m = m_begin + i;
        if (lane == 0 && m < M) {
            v += (b != nullptr) ? b[n] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            y[(size_t)m * Nout + n] = v;
        }
    }
}

// The same product for 9..16 rows (the siamese pair of a training step): the activations [M][K] are staged ONCE per workgroup in LDS
// (<= 128 KiB) and every wave then streams whole weight rows against them -- in fc_kernel each wave re-reads all M rows of x from
// L2 per weight row (17 load instructions per 4 weight floats), which left the 16.8 MB fc_1 at 0.5 TB/s.  Per output element the
// arithmetic (and its order) is that of fc_kernel: bitwise the same result.
__global__ __launch_bounds__(256) void fc_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int M, int K, int Nout,
                                                     int relu) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [M][K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int waves = gridDim.x * 4;
    int n = blockIdx.x * 4 + (tid >> 6);
    // the first weight row's loads go out BEFORE the activations are staged (the weights are the kernel's only HBM stream), and
    // every later row's before the products of the row in front of it
    float4 wv[8], wn[8];
    auto load_row = [&](float4 (&dst)[8], int row, int kb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb + j * 256 + lane * 4;
            dst[j] = (row < Nout && k < K) ? *reinterpret_cast<const float4*>(w + (size_t)row * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_row(wv, n, 0);
    // (eight loads in flight per thread: one at a time the 32 trips of this staging loop were 32 L2 round trips, most of the kernel)
    for (int i0 = tid * 4; i0 < M * K; i0 += 256 * 4 * 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 1024;
            v[j] = (i < M * K) ? *reinterpret_cast<const float4*>(x + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 1024;
            if (i < M * K) *reinterpret_cast<float4*>(xs + i) = v[j];
        }
    }
    __syncthreads();
    for (; n < Nout; n += waves) {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int kb = 0; kb < K; kb += 8 * 256) {
            // next chunk of this row, or the first chunk of the wave's next row, while this chunk is multiplied
            const bool last = kb + 8 * 256 >= K;
            load_row(wn, last ? n + waves : n, last ? 0 : kb + 8 * 256);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb + j * 256 + lane * 4;
                if (k < K) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (i < M) {
                            const float4 xv = *reinterpret_cast<const float4*>(xs + i * K + k);
                            acc[i] += ((xv.x * wv[j].x + xv.y * wv[j].y) + xv.z * wv[j].z) + xv.w * wv[j].w;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = wn[j];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = acc[i];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && i < M) {
                v += (b != nullptr) ? b[n] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)i * Nout + n] = v;
            }
        }
    }
}

// OHWI [Cout][KH][KW][CinPad] -> row-run layout [Cout][KH][Rp], Rp = roundup(KW*Cin, 32): element kw*Cin + c, zeros behind.
__global__ __launch_bounds__(256) void stem_repack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int KH,
                                                          int KW, int CinPad, int Cin, int Rp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * KH * Rp) return;
    const int e = i % Rp, kh = (i / Rp) % KH, n = i / (Rp * KH);
    float v = 0.f;
    if (e < KW * Cin) {
        const int kw = e / Cin, c = e - kw * Cin;
        v = w[(((size_t)n * KH + kh) * KW + kw) * CinPad + c];
    }
    out[i] = v;
}

int launch_stem_repack(const float* w, float* out, int Cout, int KH, int KW, int CinPad, int Cin, hipStream_t st) {
    const int Rp = (KW * Cin + 31) / 32 * 32;
    stem_repack_kernel<<<cdiv((long)Cout * KH * Rp, 256), 256, 0, st>>>(w, out, Cout, KH, KW, CinPad, Cin, Rp);
    SN_LAUNCH_CHECK("stem_repack_kernel");
    return STABNET_OK;
}

// Per-channel epilogue vectors of a merged (shortcut | conv1) launch, out = [bias | scale | shift | floor], each depth + dbn:
// shortcut channels: + bias, identity scale, no activation (floor = -inf); conv1 channels: folded bn1 + ReLU (floor = 0).
__global__ __launch_bounds__(256) void merge_vectors_kernel(const float* __restrict__ b_sc, const float* __restrict__ scale1,
                                                            const float* __restrict__ shift1, int depth, int dbn,
                                                            float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x, Ct = depth + dbn;
    if (c >= Ct) return;
    const bool s = c < depth;
    out[c] = s ? b_sc[c] : 0.f;
    out[Ct + c] = s ? 1.f : scale1[c - depth];
    out[2 * Ct + c] = s ? 0.f : shift1[c - depth];
    out[3 * Ct + c] = s ? -INFINITY : 0.f;
}

int launch_merge_vectors(const float* b_sc, const float* scale1, const float* shift1, int depth, int dbn, float* out,
                         hipStream_t st) {
    merge_vectors_kernel<<<cdiv(depth + dbn, 256), 256, 0, st>>>(b_sc, scale1, shift1, depth, dbn, out);
    SN_LAUNCH_CHECK("merge_vectors_kernel");
    return STABNET_OK;
}

int launch_pad_channels(const float* x, float* y, long npix, int C, int Cp, hipStream_t st) {
    SN_REQUIRE(Cp % 4 == 0 && Cp >= C, "pad_channels: bad channel counts %d -> %d", C, Cp);
    pad_channels_kernel<<<cdiv(npix * (Cp / 4), 256), 256, 0, st>>>(x, y, npix, C, Cp);
    SN_LAUNCH_CHECK("pad_channels_kernel");
    return STABNET_OK;
}

int launch_max_pool(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pt,
                    int pl, const float* scale, const float* shift, hipStream_t st) {
    SN_REQUIRE(C % 4 == 0 && (long)N * Ho * Wo * (C / 4) < (1L << 32), "max_pool: C %% 4 != 0, or more than 2^32 channel quads");
    max_pool_kernel<<<cdiv((long)N * Ho * Wo * (C / 4), 256), 256, 0, st>>>(x, y, N, H, W, C, Ho, Wo, k, stride, pt, pl, scale, shift);
    SN_LAUNCH_CHECK("max_pool_kernel");
    return STABNET_OK;
}

int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int G,
                   float* scale, float* shift, hipStream_t st) {
    bn_fold_kernel<<<cdiv(G, 256), 256, 0, st>>>(gamma, beta, mean, var, eps, G, scale, shift);
    SN_LAUNCH_CHECK("bn_fold_kernel");
    return STABNET_OK;
}

int gap_chunks(int HW) { return std::max(1, std::min(32, HW / 32)); }

int launch_gap_bn_relu(const float* x, const float* scale, const float* shift, int N, int HW, int C, float* out,
                       float* partial, hipStream_t st) {
    SN_REQUIRE(N <= 65535 && C % 4 == 0, "gap: bad shape");
    const int chunks = gap_chunks(HW);
    const int rpc = cdiv(HW, chunks);
    gap_bn_relu_partial_kernel<<<dim3(cdiv(C, 64), chunks, N), 256, 0, st>>>(x, scale, shift, HW, C, rpc, partial);
    SN_LAUNCH_CHECK("gap_bn_relu_partial_kernel");
    gap_finalize_kernel<<<dim3(cdiv(C, 256), N), 256, 0, st>>>(partial, chunks, HW, C, out);
    SN_LAUNCH_CHECK("gap_finalize_kernel");
    return STABNET_OK;
}

// partial[n][chunk][c] = sum over the chunk's rows of relu(bn(x)) (the first launch of launch_gap_bn_relu with a caller-chosen
// chunk count: the inference head sums the chunks inside fc_1's input staging, head.hip)
int launch_gap_partial(const float* x, const float* scale, const float* shift, int N, int HW, int C, int chunks, float* partial,
                       hipStream_t st) {
    SN_REQUIRE(N <= 65535 && C % 4 == 0 && chunks >= 1, "gap: bad shape");
    gap_bn_relu_partial_kernel<<<dim3(cdiv(C, 64), chunks, N), 256, 0, st>>>(x, scale, shift, HW, C, cdiv(HW, chunks), partial);
    SN_LAUNCH_CHECK("gap_bn_relu_partial_kernel");
    return STABNET_OK;
}

int launch_fc(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
              hipStream_t st) {
    SN_REQUIRE(K % 4 == 0, "fc: K %% 4 != 0");
    int m0 = 0;
    if (M > 8 && M <= 16 && (size_t)M * K * sizeof(float) <= 128 * 1024) {
        // the siamese pair at <= 8 samples per tower: activations staged in LDS, one workgroup per CU
        static bool configured = false;
        const size_t lds = (size_t)M * K * sizeof(float);
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fc_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) {
                stabnet_set_error("fc: hipFuncSetAttribute(128 KiB LDS) failed: %s", hipGetErrorString(e));
                return STABNET_ERR_LAUNCH;
            }
            configured = true;
        }
        fc_lds_kernel<<<std::min(256, cdiv(Nout, 4)), 256, lds, st>>>(x, w, b, y, M, K, Nout, relu);
        SN_LAUNCH_CHECK("fc_lds_kernel");
        return STABNET_OK;
    }
    for (; M - m0 > 8; m0 += 16) {          // 9..16 rows left: one pass
        fc_kernel<16><<<cdiv(Nout, 4), 256, 0, st>>>(x, w, b, y, M, K, Nout, relu, m0);
        SN_LAUNCH_CHECK("fc_kernel");
    }
    if (m0 < M) {
        fc_kernel<8><<<cdiv(Nout, 4), 256, 0, st>>>(x, w, b, y, M, K, Nout, relu, m0);
        SN_LAUNCH_CHECK("fc_kernel");
    }
    return STABNET_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Small re-packing helpers of the host mirror (so that no tensor operation of the path runs outside the library):
//   slice_channel : x [npix][C] -> out [npix] = x[:, c]        (x_tensor[..., 12:13], flow[..., 0])
//   interleave2   : a, b [n]   -> out [n][2]                   (img = [x_map, y_map], spatial_transformer3.py:295)
__global__ __launch_bounds__(256) void slice_channel_kernel(const float* __restrict__ x, long npix, int C, int c,
                                                            float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) out[i] = x[i * C + c];
}
__global__ __launch_bounds__(256) void interleave2_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) *reinterpret_cast<float2*>(out + 2 * i) = make_float2(a[i], b[i]);
}

extern "C" {
int stabnet_slice_channel(const float* x, long npix, int C, int c, float* out, void* stream) {
    SN_REQUIRE(x && out && npix > 0 && C > 0 && c >= 0 && c < C, "slice_channel: bad arguments");
    slice_channel_kernel<<<cdiv(npix, 256), 256, 0, (hipStream_t)stream>>>(x, npix, C, c, out);
    SN_LAUNCH_CHECK("slice_channel_kernel");
    return STABNET_OK;
}
int stabnet_interleave2(const float* a, const float* b, long n, float* out, void* stream) {
    SN_REQUIRE(a && b && out && n > 0, "interleave2: bad arguments");
    interleave2_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, out);
    SN_LAUNCH_CHECK("interleave2_kernel");
    return STABNET_OK;
}
}
