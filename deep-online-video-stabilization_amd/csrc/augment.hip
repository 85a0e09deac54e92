// Training sample assembly on the device: what the reference does with TF CPU ops inside its input
// queues, get_data_mini_after.py:7-147,229-253.  Every random quantity is an input (crop offsets, flip, contrast factor,
// brightness delta, mask homographies): the reference draws them from TF's Philox streams, which cannot be reproduced
// without TF; the arithmetic after the draw is restated here.  All kernels are HBM-bound elementwise passes:
//   pass A  aug_resize_crop_kernel : bilinear up-scale by 1/random_crop_rate (TF1 ResizeBilinear, align_corners = false),
//                                    crop, optional left-right flip of the 16 image channels of a pair -> planar scratch,
//                                    plus per-block partial sums (float64, fixed order) for the per-channel contrast mean
//   pass B  aug_finish_kernel      : contrast ((x - mean) * c + mean), brightness (+ d), clip to [-0.5, 0.5], the random-
//                                    homography black masks of the 6 history channels (black where H*grid leaves [-1, 1],
//                                    masked pixels = -1) and the channel interleave into x1 / y1 / x2 / y2 (NHWC)
//           aug_flow_kernel        : the same resize / crop / flip on the flow field + renormalisation
//           aug_points_kernel      : renormalisation / flip of the matched points and the in-frame mask
#include "common.h"

struct AugGeom {
    int N, H, W, h, w, bc;             // (h, w) = resized size; bc = before_ch
    float hs, ws;                      // H / h, W / w as float32 (ResizeBilinear's scales)
};

// value of the up-scaled image at (yb, xb), img = one [H][W] channel with pixel stride `cs` (channels-last source)
__device__ __forceinline__ float resize_at(const float* __restrict__ img, int cs, int H, int W, int yb, int xb, float hs, float ws) {
    const float iy = (float)yb * hs, ix = (float)xb * ws;
    const int y0 = (int)floorf(iy), x0 = (int)floorf(ix);
    const int y1 = min((int)ceilf(iy), H - 1), x1 = min((int)ceilf(ix), W - 1);
    const float ly = iy - (float)y0, lx = ix - (float)x0;
    const float tl = img[((size_t)y0 * W + x0) * cs], tr = img[((size_t)y0 * W + x1) * cs];
    const float bl = img[((size_t)y1 * W + x0) * cs], br = img[((size_t)y1 * W + x1) * cs];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    return top + (bot - top) * ly;
}

// bilinear taps of the up-scaled image at (yb, xb): the four source pixel indices and the two fractions (resize_at's arithmetic)
struct AugTaps {
    int i00, i01, i10, i11;
    float lx, ly;
};

__device__ __forceinline__ AugTaps aug_taps(int H, int W, int yb, int xb, float hs, float ws) {
    const float iy = (float)yb * hs, ix = (float)xb * ws;
    const int y0 = (int)floorf(iy), x0 = (int)floorf(ix);
    const int y1 = min((int)ceilf(iy), H - 1), x1 = min((int)ceilf(ix), W - 1);
    AugTaps t;
    t.i00 = y0 * W + x0; t.i01 = y0 * W + x1; t.i10 = y1 * W + x0; t.i11 = y1 * W + x1;
    t.ly = iy - (float)y0; t.lx = ix - (float)x0;
    return t;
}

__device__ __forceinline__ float aug_lerp(float tl, float tr, float bl, float br, float lx, float ly) {
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    return top + (bot - top) * ly;
}

// grid: (blocks over H*W, sample); 256 threads, one output pixel per thread and ALL its channels: the taps of a channels-last
// source pixel are one contiguous run (2 * (bc + 1) floats), read as float2s, so a source line is fetched once instead of once per
// channel.  The planar scratch is written one coalesced plane row per channel.  Channel sums: the block's values go through an LDS
// tile [channel][thread]; wave w reduces channels w, w + 4, ... in float64 (4 strided reads, xor-shuffle tree: a fixed order).
constexpr int AUG_PIX_PER_BLOCK = 256;
constexpr int AUG_MAX_CH = 32;               // 2 * (14 + 1) + 2

__global__ __launch_bounds__(256) void aug_resize_crop_kernel(const float* __restrict__ stable, const float* __restrict__ unstable,
                                                              const int* __restrict__ para, AugGeom g, float* __restrict__ tmp,
                                                              double* __restrict__ partial) {
    __shared__ float tile[AUG_MAX_CH][256];
    const int n = blockIdx.y, nst = 2 * (g.bc + 1), C = nst + 2;
    const int plane = g.H * g.W;
    const int q = blockIdx.x * AUG_PIX_PER_BLOCK + threadIdx.x;
    const bool live = q < plane;
    const int ph = para[3 * n], pw = para[3 * n + 1], flip = para[3 * n + 2];
    AugTaps t = {0, 0, 0, 0, 0.f, 0.f};
    if (live) {
        const int y = q / g.W, x = q - y * g.W;
        const int xs = flip ? g.W - 1 - x : x;
        t = aug_taps(g.H, g.W, ph + y, pw + xs, g.hs, g.ws);
    }
    float* dst = tmp + (size_t)n * C * plane + q;
    const float2* st = reinterpret_cast<const float2*>(stable + (size_t)n * plane * nst);
    const int np = nst >> 1;                 // float2s per stable pixel
    for (int j = 0; j < np; ++j) {
        float2 v = {0.f, 0.f};
        if (live) {
            const float2 tl = st[(size_t)t.i00 * np + j], tr = st[(size_t)t.i01 * np + j];
            const float2 bl = st[(size_t)t.i10 * np + j], br = st[(size_t)t.i11 * np + j];
            v.x = aug_lerp(tl.x, tr.x, bl.x, br.x, t.lx, t.ly);
            v.y = aug_lerp(tl.y, tr.y, bl.y, br.y, t.lx, t.ly);
            dst[(size_t)(2 * j) * plane] = v.x;
            dst[(size_t)(2 * j + 1) * plane] = v.y;
        }
        tile[2 * j][threadIdx.x] = v.x;
        tile[2 * j + 1][threadIdx.x] = v.y;
    }
    {
        const float2* un = reinterpret_cast<const float2*>(unstable + (size_t)n * plane * 2);
        float2 v = {0.f, 0.f};
        if (live) {
            const float2 tl = un[t.i00], tr = un[t.i01], bl = un[t.i10], br = un[t.i11];
            v.x = aug_lerp(tl.x, tr.x, bl.x, br.x, t.lx, t.ly);
            v.y = aug_lerp(tl.y, tr.y, bl.y, br.y, t.lx, t.ly);
            dst[(size_t)nst * plane] = v.x;
            dst[(size_t)(nst + 1) * plane] = v.y;
        }
        tile[nst][threadIdx.x] = v.x;
        tile[nst + 1][threadIdx.x] = v.y;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        double s = (((double)tile[c][lane] + (double)tile[c][lane + 64]) + (double)tile[c][lane + 128]) + (double)tile[c][lane + 192];
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) partial[((size_t)n * C + c) * gridDim.x + blockIdx.x] = s;
    }
}

// grid: (channel, sample); one wave: lane l adds partials l, l + 64, ... in order, then the xor-shuffle tree -> the float32 mean
__global__ __launch_bounds__(64) void aug_means_kernel(const double* __restrict__ partial, int nblk, int C, double count,
                                                       float* __restrict__ means) {
    const int n = blockIdx.y, c = blockIdx.x;
    const double* p = partial + ((size_t)n * C + c) * nblk;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) s += p[k];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) means[n * C + c] = (float)(s / count);
}

__device__ __forceinline__ float aug_jitter(float v, float mean, float contrast, float bright) {
    v = (v - mean) * contrast + mean;          // tf.image.adjust_contrast, per-channel mean
    v = v + bright;                            // tf.image.adjust_brightness
    return fminf(fmaxf(v, -0.5f), 0.5f);       // get_data_mini_after.py:31
}

// grid: (blocks over H*W, 2 towers, sample); one thread per pixel, all 2 * bc + 1 output channels of a tower.  The channels-last
// x rows of a block's 256 pixels are one contiguous run in the output: they are assembled in LDS (pixel stride 2 * bc + 1 is odd:
// conflict-free) and stored as coalesced dwords instead of 2 * bc + 1 strided scalar stores per thread.
constexpr int AUG_MAX_XCH = 29;              // 2 * 14 + 1

__global__ __launch_bounds__(256) void aug_finish_kernel(const float* __restrict__ tmp, const float* __restrict__ means_g,
                                                         const float* __restrict__ jitter, const float* __restrict__ Hs, AugGeom g,
                                                         float* __restrict__ x1, float* __restrict__ y1, float* __restrict__ x2,
                                                         float* __restrict__ y2) {
    __shared__ float rows[256 * AUG_MAX_XCH];
    __shared__ float means[16];
    __shared__ float Hm_s[14 * 9];
    const int n = blockIdx.z, tower = blockIdx.y, nst = 2 * (g.bc + 1), C = nst + 2;
    // channel means of this tower's bc+2 channels: label, bc history frames, current frame
    const int nch = g.bc + 2, nx = 2 * g.bc + 1;
    if (threadIdx.x < nch) {
        const int c = (threadIdx.x <= g.bc) ? tower * (g.bc + 1) + threadIdx.x : nst + tower;
        means[threadIdx.x] = means_g[n * C + c];
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + g.bc * 9) Hm_s[threadIdx.x - 64] = Hs[((size_t)n * 2 + tower) * g.bc * 9 + (threadIdx.x - 64)];
    __syncthreads();
    const int plane = g.H * g.W;
    const int q0 = blockIdx.x * 256, q = q0 + threadIdx.x;
    if (q < plane) {
        const int y = q / g.W, x = q - y * g.W;
        const float contrast = jitter[2 * n], bright = jitter[2 * n + 1];
        const float* t = tmp + (size_t)n * C * plane + q;
        float* yo = (tower ? y2 : y1) + (size_t)n * plane + q;
        float* xo = rows + threadIdx.x * nx;
        const int c0 = tower * (g.bc + 1);
        yo[0] = aug_jitter(t[(size_t)c0 * plane], means[0], contrast, bright);
        xo[2 * g.bc] = aug_jitter(t[(size_t)(nst + tower) * plane], means[g.bc + 1], contrast, bright);
        // grid point (tf.linspace: start + step * i)
        const float stepx = (g.W > 1) ? 2.0f / (float)(g.W - 1) : 0.0f, stepy = (g.H > 1) ? 2.0f / (float)(g.H - 1) : 0.0f;
        const float gx = -1.0f + stepx * (float)x, gy = -1.0f + stepy * (float)y;
        for (int k = 0; k < g.bc; ++k) {
            const float* Hm = Hm_s + k * 9;
            const float xs = (Hm[0] * gx + Hm[1] * gy) + Hm[2];
            const float ys = (Hm[3] * gx + Hm[4] * gy) + Hm[5];
            const float zs = (Hm[6] * gx + Hm[7] * gy) + Hm[8];
            const float u = xs / zs, v = ys / zs;
            const float m = (-1.0f > u || u > 1.0f || -1.0f > v || v > 1.0f) ? 1.0f : 0.0f;     // get_data_mini_after.py:104-107
            const float f = aug_jitter(t[(size_t)(c0 + 1 + k) * plane], means[1 + k], contrast, bright);
            xo[k] = m;
            xo[g.bc + k] = f * (1.0f - m) + m * -1.0f;                                            // :135
        }
    }
    __syncthreads();
    const int cnt = min(256, plane - q0) * nx;
    float* xg = (tower ? x2 : x1) + ((size_t)n * plane + q0) * nx;
    for (int i = threadIdx.x; i < cnt; i += 256) xg[i] = rows[i];
}

__global__ __launch_bounds__(256) void aug_flow_kernel(const float* __restrict__ flow, const int* __restrict__ para, AugGeom g,
                                                       float div_x, float div_y, float inv_w, float* __restrict__ out) {
    const int n = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.H * g.W) return;
    const int ph = para[3 * n], pw = para[3 * n + 1], flip = para[3 * n + 2];
    const int y = q / g.W, x = q - y * g.W;
    const int xs = flip ? g.W - 1 - x : x;
    const float* src = flow + (size_t)n * g.H * g.W * 2;
    float fx = resize_at(src, 2, g.H, g.W, ph + y, pw + xs, g.hs, g.ws);
    float fy = resize_at(src + 1, 2, g.H, g.W, ph + y, pw + xs, g.hs, g.ws);
    const float ox = 1.0f - ((float)pw / (float)g.w) * 2.0f;
    const float oy = 1.0f - ((float)ph / (float)g.h) * 2.0f;
    fx = (fx + ox) / div_x - 1.0f;              // get_data_mini_after.py:43 (x is divided by height / h there)
    fy = (fy + oy) / div_y - 1.0f;
    if (flip) fx = fx * -1.0f - inv_w;          // :47
    float* o = out + ((size_t)n * g.H * g.W + q) * 2;
    o[0] = fx;
    o[1] = fy;
}

__global__ __launch_bounds__(256) void aug_points_kernel(const float* __restrict__ pts, const int* __restrict__ count,
                                                         const int* __restrict__ para, AugGeom g, int M, float div_x, float div_y,
                                                         float inv_w, float* __restrict__ out, float* __restrict__ mask) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int ph = para[3 * n], pw = para[3 * n + 1], flip = para[3 * n + 2];
    const float ox = 1.0f - ((float)pw / (float)g.w) * 2.0f;
    const float oy = 1.0f - ((float)ph / (float)g.h) * 2.0f;
    const float* p = pts + ((size_t)n * M + i) * 4;
    float r[4];
    bool ok = i < count[n];
    for (int k = 0; k < 4; ++k) {
        float v;
        if ((k & 1) == 0) {
            v = (p[k] + ox) / div_x - 1.0f;
            if (flip) v = v * -1.0f - inv_w;
        } else {
            v = (p[k] + oy) / div_y - 1.0f;
        }
        r[k] = v;
        ok = ok && v >= -1.0f && v <= 1.0f;
    }
    float* o = out + ((size_t)n * M + i) * 4;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
    mask[(size_t)n * M + i] = ok ? 1.0f : 0.0f;
}

static int aug_geom(AugGeom& g, int N, int H, int W, int bc, float rate) {
    SN_REQUIRE(N > 0 && H > 1 && W > 1 && bc >= 1 && bc <= 14 && rate > 0.f && rate <= 1.f, "augment: bad geometry");
    g.N = N; g.H = H; g.W = W; g.bc = bc;
    g.h = (int)((double)H / (double)rate);      // int(height / random_crop_rate), get_data_mini_after.py:8-9
    g.w = (int)((double)W / (double)rate);
    g.hs = (float)H / (float)g.h;
    g.ws = (float)W / (float)g.w;
    SN_REQUIRE((long)N * H * W * 16 < (1L << 31), "augment: batch too large");
    return STABNET_OK;
}

extern "C" {

size_t stabnet_augment_workspace_bytes(int N, int H, int W, int before_ch) {
    if (N <= 0 || H <= 0 || W <= 0 || before_ch <= 0) return 0;
    const size_t C = 2 * ((size_t)before_ch + 1) + 2;
    const size_t nblk = (size_t)cdiv((long)H * W, AUG_PIX_PER_BLOCK);
    return (size_t)N * C * H * W * sizeof(float) + (size_t)N * C * nblk * sizeof(double) + (size_t)N * C * sizeof(float) + 512;
}

int stabnet_augment_pairs(const float* stable, const float* unstable, const float* flow_in, const float* matches1,
                          const int* n1, const float* matches2, const int* n2, const int* para, const float* jitter,
                          const float* Hs, int N, int H, int W, int before_ch, int max_matches, float random_crop_rate,
                          float* x1, float* y1, float* x2, float* y2, float* flow_out, float* fm1, float* mk1, float* fm2,
                          float* mk2, void* workspace, size_t workspace_bytes, void* stream) {
    SN_REQUIRE(stable && unstable && para && jitter && Hs && x1 && y1 && x2 && y2 && workspace, "augment_pairs: null pointer");
    AugGeom g;
    int rc = aug_geom(g, N, H, W, before_ch, random_crop_rate);
    if (rc) return rc;
    const size_t need = stabnet_augment_workspace_bytes(N, H, W, before_ch);
    if (workspace_bytes < need) {
        stabnet_set_error("augment_pairs: workspace %zu B < %zu B needed", workspace_bytes, need);
        return STABNET_ERR_WORKSPACE;
    }
    SN_REQUIRE(((uintptr_t)stable & 7) == 0 && ((uintptr_t)unstable & 7) == 0, "augment_pairs: stable / unstable must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int C = 2 * (before_ch + 1) + 2;
    const int nblk = cdiv((long)H * W, AUG_PIX_PER_BLOCK);
    float* tmp = static_cast<float*>(workspace);
    size_t off = ((size_t)N * C * H * W * sizeof(float) + 255) & ~(size_t)255;
    double* partial = reinterpret_cast<double*>(static_cast<char*>(workspace) + off);
    off = (off + (size_t)N * C * nblk * sizeof(double) + 255) & ~(size_t)255;
    float* means = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
    aug_resize_crop_kernel<<<dim3(nblk, N), 256, 0, st>>>(stable, unstable, para, g, tmp, partial);
    SN_LAUNCH_CHECK("aug_resize_crop_kernel");
    aug_means_kernel<<<dim3(C, N), 64, 0, st>>>(partial, nblk, C, (double)((size_t)H * W), means);
    SN_LAUNCH_CHECK("aug_means_kernel");
    aug_finish_kernel<<<dim3(cdiv((long)H * W, 256), 2, N), 256, 0, st>>>(tmp, means, jitter, Hs, g, x1, y1, x2, y2);
    SN_LAUNCH_CHECK("aug_finish_kernel");
    // the reference divides x by height / h and y by width / w (get_data_mini_after.py:43-44,62-63); both equal the crop rate
    const float div_x = (float)((double)H / (double)g.h), div_y = (float)((double)W / (double)g.w);
    const float inv_w = (float)(1.0 / (double)W);
    if (flow_in != nullptr) {
        SN_REQUIRE(flow_out != nullptr, "augment_pairs: flow_out missing");
        aug_flow_kernel<<<dim3(cdiv((long)H * W, 256), N), 256, 0, st>>>(flow_in, para, g, div_x, div_y, inv_w, flow_out);
        SN_LAUNCH_CHECK("aug_flow_kernel");
    }
    if (matches1 != nullptr) {
        SN_REQUIRE(n1 && matches2 && n2 && fm1 && mk1 && fm2 && mk2 && max_matches > 0, "augment_pairs: match arguments missing");
        aug_points_kernel<<<dim3(cdiv(max_matches, 256), N), 256, 0, st>>>(matches1, n1, para, g, max_matches, div_x, div_y, inv_w, fm1, mk1);
        SN_LAUNCH_CHECK("aug_points_kernel");
        aug_points_kernel<<<dim3(cdiv(max_matches, 256), N), 256, 0, st>>>(matches2, n2, para, g, max_matches, div_x, div_y, inv_w, fm2, mk2);
        SN_LAUNCH_CHECK("aug_points_kernel");
    }
    return STABNET_OK;
}

}  // extern "C"
