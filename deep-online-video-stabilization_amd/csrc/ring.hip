// On-device history ring of the online loop (deploy_bundle.py:216-232,259-274,291-295,
// 319-332).  The reference keeps Python lists of [1,H,W,1] NumPy frames and np.concatenate()s a 13-channel stack per
// frame on the host; here the ring is a circular buffer in HBM, [S streams][depth][H*W] for stabilised frames and for
// black masks, and the stack is assembled straight into the 16-channel padded NHWC tensor the stem conv reads.
//   slot of lag i at time t: (head - i) mod depth, head = slot the NEXT push writes.
#include "ring.h"

__global__ __launch_bounds__(256) void ring_init_kernel(float* __restrict__ frames, float* __restrict__ masks,
                                                        const float* __restrict__ first, int depth, long hw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // pixel
    const int s = blockIdx.y;
    if (i >= hw) return;
    const float v = first[(long)s * hw + i];
    for (int d = 0; d < depth; ++d) {                            // 32 x first frame, zero masks (:216-224)
        frames[((long)s * depth + d) * hw + i] = v;
        masks[((long)s * depth + d) * hw + i] = 0.f;
    }
}

// out[s][pix][c]: c in [0,n) masks at lags, [n,2n) frames at lags, 2n current frame, rest zero (channel padding).
__global__ __launch_bounds__(256) void stack_assemble_kernel(const float* __restrict__ frames,
                                                             const float* __restrict__ masks,
                                                             const float* __restrict__ cur, int depth,
                                                             const int* __restrict__ head_ptr,
                                                             const RingLags lags, long hw, int Cp,
                                                             float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= hw) return;
    const int head = *head_ptr;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.f;
    const int n = lags.n;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        if (l < n) {
            int slot = (head - lags.lag[l]) % depth;
            if (slot < 0) slot += depth;
            const long off = ((long)s * depth + slot) * hw + i;
            v[l] = masks[off];
            v[n + l] = frames[off];
        }
    }
    v[2 * n] = cur[(long)s * hw + i];
    float* o = out + ((long)s * hw + i) * Cp;
    if (Cp == 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
        for (int c = 0; c < Cp; ++c) o[c] = v[c];
    }
}

// The inference stem (ring kernel MODE 2) reads the stack as a tight 13-channel image WITH its zero border in memory:
// out[s][Hp][Wp][Cp], Hp = H + 2*border, Wp = W + 2*border, Cp = 2n+1.  One thread per bordered pixel; the block's 256 pixels
// x Cp floats are one contiguous run of the output, written coalesced through LDS.  The border is rewritten every frame
// (the arena may have reused the region in between).
__global__ __launch_bounds__(256) void stack_assemble_bordered_kernel(const float* __restrict__ frames,
                                                                      const float* __restrict__ masks,
                                                                      const float* __restrict__ cur, int depth,
                                                                      const int* __restrict__ head_ptr, const RingLags lags,
                                                                      int H, int W, int border, int Cp, float* __restrict__ out) {
    __shared__ float tile[256 * 16];
    const int Wp = W + 2 * border, Hp = H + 2 * border;
    const long hwp = (long)Hp * Wp, hw = (long)H * W;
    const long i0 = (long)blockIdx.x * 256;
    const long ip = i0 + threadIdx.x;
    const int s = blockIdx.y;
    const int head = *head_ptr;
    const int n = lags.n;
    if (ip < hwp) {
        const int yp = (int)(ip / Wp), xp = (int)(ip - (long)yp * Wp);
        const int y = yp - border, x = xp - border;
        float* t = tile + threadIdx.x * Cp;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const long i = (long)y * W + x;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                if (l < n) {
                    int slot = (head - lags.lag[l]) % depth;
                    if (slot < 0) slot += depth;
                    const long off = ((long)s * depth + slot) * hw + i;
                    t[l] = masks[off];
                    t[n + l] = frames[off];
                }
            }
            t[2 * n] = cur[(long)s * hw + i];
        } else {
            for (int c = 0; c < Cp; ++c) t[c] = 0.f;
        }
    }
    __syncthreads();
    const long npix = min((long)256, hwp - i0);
    float* o = out + ((long)s * hwp + i0) * Cp;
    for (long k = threadIdx.x; k < npix * Cp; k += 256) o[k] = tile[k];
    // slack behind the last image: the stem's last 32-float run reads up to 31 floats past its taps (they meet zero
    // weights, but 0 * NaN = NaN, and the workspace is not initialised)
    if (blockIdx.x == 0 && s == (int)gridDim.y - 1 && threadIdx.x < 64) out[(long)gridDim.y * hwp * Cp + threadIdx.x] = 0.f;
}

// x [N][H][W][C] -> out [N][H+2b][W+2b][C] with a zero border (the non-deploy entry of the same stem).  One workgroup row per
// padded image row: four floats per thread, border or interior by their offset in the row alone (no division per element; the first form
// -- one flat index, pixel = q / C in 64 bits -- took 53 us per 8 x 288 x 512 x 13 tower against 30 for the channel padding it replaced).
__global__ __launch_bounds__(256) void embed_border_kernel(const float* __restrict__ x, int H, int W, int C, int border,
                                                           float* __restrict__ out) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // rows start at any 4-byte offset (C = 13)
    const int Wp = W + 2 * border, Hp = H + 2 * border;
    const int q = (blockIdx.x * 256 + threadIdx.x) * 4;     // first of this thread's four floats of the padded row
    const int yp = blockIdx.y, n = blockIdx.z;
    const int row_floats = Wp * C, in_floats = W * C;
    if (q < row_floats) {
        const int y = yp - border, qi = q - border * C;
        const bool row_in = (unsigned)y < (unsigned)H;
        const float* src = x + ((size_t)n * H + y) * in_floats;
        float* dst = out + ((size_t)n * Hp + yp) * row_floats + q;
        f4u v = {0.f, 0.f, 0.f, 0.f};
        if (row_in && qi >= 0 && qi + 3 < in_floats) {
            v = *reinterpret_cast<const f4u*>(src + qi);
        } else if (row_in && qi + 3 >= 0 && qi < in_floats) {             // straddles the border: element by element
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((unsigned)(qi + e) < (unsigned)in_floats) v[e] = src[qi + e];
        }
        if (q + 3 < row_floats) {
            *reinterpret_cast<f4u*>(dst) = v;
        } else {
            for (int e = 0; e < 4 && q + e < row_floats; ++e) dst[e] = v[e];
        }
    }
    if (blockIdx.x == 0 && yp == 0 && n == (int)gridDim.z - 1 && threadIdx.x < 64)
        out[(size_t)gridDim.z * Hp * row_floats + threadIdx.x] = 0.f;                        // slack behind the last image (see above)
}

// frame = img + black * (-1) (:293); push frame and black into slot `head` (:322-323).
__global__ __launch_bounds__(256) void ring_push_kernel(float* __restrict__ frames, float* __restrict__ masks, int depth,
                                                        const int* __restrict__ head_ptr, const float* __restrict__ img,
                                                        const float* __restrict__ black, long hw,
                                                        float* __restrict__ frame_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= hw) return;
    const int head = (head_ptr != nullptr) ? *head_ptr : 0;
    const float b = black[(long)s * hw + i];
    const float f = img[(long)s * hw + i] + b * -1.0f;
    frames[((long)s * depth + head) * hw + i] = f;
    if (masks != nullptr) masks[((long)s * depth + head) * hw + i] = b;
    if (frame_out != nullptr) frame_out[(long)s * hw + i] = f;
}

int launch_ring_init(float* frames, float* masks, const float* first, int S, int depth, long hw, hipStream_t st) {
    ring_init_kernel<<<dim3(cdiv(hw, 256), S), 256, 0, st>>>(frames, masks, first, depth, hw);
    SN_LAUNCH_CHECK("ring_init_kernel");
    return STABNET_OK;
}

int launch_stack_assemble(const float* frames, const float* masks, const float* cur, int S, int depth, const int* head,
                          const RingLags& lags, long hw, int Cp, float* out, hipStream_t st) {
    SN_REQUIRE(lags.n >= 1 && lags.n <= 7 && 2 * lags.n + 1 <= Cp && Cp <= 16, "stack_assemble: %d lags do not fit %d channels",
               lags.n, Cp);
    stack_assemble_kernel<<<dim3(cdiv(hw, 256), S), 256, 0, st>>>(frames, masks, cur, depth, head, lags, hw, Cp, out);
    SN_LAUNCH_CHECK("stack_assemble_kernel");
    return STABNET_OK;
}

int launch_stack_assemble_bordered(const float* frames, const float* masks, const float* cur, int S, int depth, const int* head,
                                   const RingLags& lags, int H, int W, int border, float* out, hipStream_t st) {
    SN_REQUIRE(lags.n >= 1 && lags.n <= 7 && border >= 0, "stack_assemble: bad arguments");
    const long hwp = (long)(H + 2 * border) * (W + 2 * border);
    stack_assemble_bordered_kernel<<<dim3(cdiv(hwp, 256), S), 256, 0, st>>>(frames, masks, cur, depth, head, lags, H, W, border,
                                                                          2 * lags.n + 1, out);
    SN_LAUNCH_CHECK("stack_assemble_bordered_kernel");
    return STABNET_OK;
}

int launch_embed_border(const float* x, int N, int H, int W, int C, int border, float* out, hipStream_t st) {
    SN_REQUIRE(N >= 1 && N <= 65535 && H + 2 * border <= 65535 && (long)(W + 2 * border) * C < (1L << 31), "embed_border: bad shape");
    embed_border_kernel<<<dim3(cdiv((W + 2 * border) * C, 1024), H + 2 * border, N), 256, 0, st>>>(x, H, W, C, border, out);
    SN_LAUNCH_CHECK("embed_border_kernel");
    return STABNET_OK;
}

__global__ void ring_advance_kernel(int* head, int depth) { *head = (*head + 1) % depth; }

int launch_ring_advance(int* head, int depth, hipStream_t st) {
    ring_advance_kernel<<<1, 1, 0, st>>>(head, depth);
    SN_LAUNCH_CHECK("ring_advance_kernel");
    return STABNET_OK;
}

int launch_ring_push(float* frames, float* masks, int S, int depth, const int* head, const float* img, const float* black,
                     long hw, float* frame_out, hipStream_t st) {
    ring_push_kernel<<<dim3(cdiv(hw, 256), S), 256, 0, st>>>(frames, masks, depth, head, img, black, hw, frame_out);
    SN_LAUNCH_CHECK("ring_push_kernel");
    return STABNET_OK;
}
