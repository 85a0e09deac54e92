// Tiling probe of the fused map + gather + feedback-push sampler (warp_sample_kernel<4, 1> of csrc/warp.hip).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I deep-online-video-stabilization_amd/csrc -o /tmp/warp_probe tools/warp_probe.hip
//   /tmp/warp_probe [H W]      (default: 720x1280 and 1080x1920)
// Every variant must write bit-identical planes (checked against variant 0 = the shipped tiling); time = HIP events over
// 50 back-to-back launches (includes the ~1.5 us kernel boundary, like the frame's graph does).
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Push { float *frames, *masks, *frame_out; const int* head; int* ticket; int depth; };

__device__ __forceinline__ int cvt_i32_x86(float f) { return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN; }

struct Px { float xm, ym, bl, o; };

__device__ __forceinline__ void map_px(const float* h, float gx, float gy, float& xm, float& ym, float& bl) {
    const float tx = (h[0] * gx + h[1] * gy) + h[2];
    const float ty = (h[3] * gx + h[4] * gy) + h[5];
    float tz = (h[6] * gx + h[7] * gy) + h[8];
    const float sgn = ((tz >= 0.0f) ? 1.0f : 0.0f) * 2.0f - 1.0f;
    tz = tz + sgn * 1e-8f;
    xm = tx / tz;
    ym = ty / tz;
    bl = ((-1.0f > xm) || (xm > 1.0f) || (-1.0f > ym) || (ym > 1.0f)) ? 1.0f : 0.0f;
}

struct Taps { int ia, ib, ic, id; float wa, wb, wc, wd; };
__device__ __forceinline__ Taps make_taps(float xm, float ym, int H, int W) {
    const float xp = (xm + 1.0f) * (float)W / 2.0f;
    const float yp = (ym + 1.0f) * (float)H / 2.0f;
    int x0 = cvt_i32_x86(floorf(xp)), y0 = cvt_i32_x86(floorf(yp));
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    Taps t;
    t.wa = (x1f - xp) * (y1f - yp); t.wb = (x1f - xp) * (yp - y0f);
    t.wc = (xp - x0f) * (y1f - yp); t.wd = (xp - x0f) * (yp - y0f);
    t.ia = y0 * W + x0; t.ib = y1 * W + x0; t.ic = y0 * W + x1; t.id = y1 * W + x1;
    return t;
}
__device__ __forceinline__ float blend(const Taps& t, float Ia, float Ib, float Ic, float Id) {
    return ((t.wa * Ia + t.wb * Ib) + t.wc * Ic) + t.wd * Id;
}

// ---- variant 0: the shipped tiling (4 rows x 256 columns per block, 4 consecutive pixels per thread, Hs in LDS, ticket)
template <int TICKET>
__global__ __launch_bounds__(256) void v0_kernel(const float* __restrict__ Hs, const float* __restrict__ src, int H, int W, int gh, int gw,
                                                 float* __restrict__ out, float* __restrict__ black, float* __restrict__ x_map,
                                                 float* __restrict__ y_map, const Push push) {
    __shared__ float sH[64 * 9];
    const int cells = gh * gw;
    for (int i = threadIdx.x; i < cells * 9; i += 256) sH[i] = Hs[i];
    const int head = *push.head;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wv;
    const int xb = blockIdx.x * 256 + lane * 4;
    if (y < H && xb < W) {
        const int chh = H / gh, cww = W / gw;
        const int ci = min(y / chh, gh - 1);
        const float stepx = 2.0f / (float)(W - 1), stepy = 2.0f / (float)(H - 1);
        const float gy = -1.0f + stepy * (float)y;
        const size_t rowoff = (size_t)y * W;
        float xm[4], ym[4], bl[4], o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = min(xb + e, W - 1);
            const int cj = min(xx / cww, gw - 1);
            map_px(sH + (ci * gw + cj) * 9, -1.0f + stepx * (float)xx, gy, xm[e], ym[e], bl[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const Taps t = make_taps(xm[e], ym[e], H, W);
            o[e] = blend(t, src[t.ia], src[t.ib], src[t.ic], src[t.id]);
        }
        *reinterpret_cast<float4*>(x_map + rowoff + xb) = make_float4(xm[0], xm[1], xm[2], xm[3]);
        *reinterpret_cast<float4*>(y_map + rowoff + xb) = make_float4(ym[0], ym[1], ym[2], ym[3]);
        *reinterpret_cast<float4*>(black + rowoff + xb) = make_float4(bl[0], bl[1], bl[2], bl[3]);
        *reinterpret_cast<float4*>(out + rowoff + xb) = make_float4(o[0], o[1], o[2], o[3]);
        const size_t hw = (size_t)H * W, pix = rowoff + xb;
        const float4 f = make_float4(o[0] + bl[0] * -1.0f, o[1] + bl[1] * -1.0f, o[2] + bl[2] * -1.0f, o[3] + bl[3] * -1.0f);
        const size_t slot = (size_t)head * hw + pix;
        *reinterpret_cast<float4*>(push.frames + slot) = f;
        *reinterpret_cast<float4*>(push.masks + slot) = make_float4(bl[0], bl[1], bl[2], bl[3]);
        *reinterpret_cast<float4*>(push.frame_out + pix) = f;
    }
    if (TICKET) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = gridDim.x * gridDim.y * gridDim.z;
            const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(push.ticket), 1u);
            if (old == total - 1) *push.ticket = 0;
        }
    }
}

// ---- variant 1: PX pixels per thread, lane-strided (pixel e of lane l = base + e*64 + l: every gather / store instruction of a
// wave covers 64 CONSECUTIVE pixels), one wave = 64*PX consecutive pixels of one row, WAVES waves per block (consecutive
// row segments), homography of the wave's cell through scalar loads when the whole wave sits in one cell column, no LDS,
// no barrier, no ticket.  All taps of a thread's pixels are issued before the first blend.
template <int PX, int WAVES, int NT>
__global__ __launch_bounds__(64 * WAVES) void v1_kernel(const float* __restrict__ Hs, const float* __restrict__ src, int H, int W, int gh,
                                                        int gw, float* __restrict__ out, float* __restrict__ black,
                                                        float* __restrict__ x_map, float* __restrict__ y_map, const Push push) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int segs = (W + 64 * PX - 1) / (64 * PX);                 // wave segments per row
    const int wseg = blockIdx.x * WAVES + wv;                        // global wave id
    const int y = wseg / segs;
    if (y >= H) return;                                              // wave-uniform
    const int x0 = (wseg - y * segs) * (64 * PX);
    const int head = *push.head;
    const int chh = H / gh, cww = W / gw;
    const int ci = min(y / chh, gh - 1);
    const float stepx = 2.0f / (float)(W - 1), stepy = 2.0f / (float)(H - 1);
    const float gy = -1.0f + stepy * (float)y;
    const size_t rowoff = (size_t)y * W;
    float xm[PX], ym[PX], bl[PX], o[PX];
    const int xe = min(x0 + 64 * PX - 1, W - 1);
    const int cj0 = min(x0 / cww, gw - 1), cj1 = min(xe / cww, gw - 1);
    if (cj0 == cj1) {                                                // wave-uniform: one cell for the whole segment
        const float* hp = Hs + (ci * gw + cj0) * 9;
        float h[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) h[i] = hp[i];                   // wave-uniform address -> scalar loads
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int xx = min(x0 + e * 64 + lane, W - 1);
            map_px(h, -1.0f + stepx * (float)xx, gy, xm[e], ym[e], bl[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int xx = min(x0 + e * 64 + lane, W - 1);
            const int cj = min(xx / cww, gw - 1);
            float h[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = Hs[(ci * gw + cj) * 9 + i];
            map_px(h, -1.0f + stepx * (float)xx, gy, xm[e], ym[e], bl[e]);
        }
    }
    Taps t[PX];
    float Ia[PX], Ib[PX], Ic[PX], Id[PX];
#pragma unroll
    for (int e = 0; e < PX; ++e) {
        t[e] = make_taps(xm[e], ym[e], H, W);
        Ia[e] = src[t[e].ia]; Ib[e] = src[t[e].ib]; Ic[e] = src[t[e].ic]; Id[e] = src[t[e].id];
    }
#pragma unroll
    for (int e = 0; e < PX; ++e) o[e] = blend(t[e], Ia[e], Ib[e], Ic[e], Id[e]);
    const size_t hw = (size_t)H * W;
#pragma unroll
    for (int e = 0; e < PX; ++e) {
        const int xx = x0 + e * 64 + lane;
        if (xx < W) {
            const size_t pix = rowoff + xx;
            const float f = o[e] + bl[e] * -1.0f;
            if (NT) {
                __builtin_nontemporal_store(xm[e], x_map + pix); __builtin_nontemporal_store(ym[e], y_map + pix);
                __builtin_nontemporal_store(bl[e], black + pix); __builtin_nontemporal_store(o[e], out + pix);
            } else {
                x_map[pix] = xm[e]; y_map[pix] = ym[e]; black[pix] = bl[e]; out[pix] = o[e];
            }
            push.frames[(size_t)head * hw + pix] = f;
            push.masks[(size_t)head * hw + pix] = bl[e];
            push.frame_out[pix] = f;
        }
    }
}

// ---- variant 2: like variant 1 but a lane owns VEC consecutive pixels (16/8-byte stores), R such groups lane-strided
template <int VEC, int R, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void v2_kernel(const float* __restrict__ Hs, const float* __restrict__ src, int H, int W, int gh,
                                                        int gw, float* __restrict__ out, float* __restrict__ black,
                                                        float* __restrict__ x_map, float* __restrict__ y_map, const Push push) {
    constexpr int PXW = 64 * VEC * R;                                // pixels per wave
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int segs = (W + PXW - 1) / PXW;
    const int wseg = blockIdx.x * WAVES + wv;
    const int y = wseg / segs;
    if (y >= H) return;
    const int x0 = (wseg - y * segs) * PXW;
    const int head = *push.head;
    const int chh = H / gh, cww = W / gw;
    const int ci = min(y / chh, gh - 1);
    const float stepx = 2.0f / (float)(W - 1), stepy = 2.0f / (float)(H - 1);
    const float gy = -1.0f + stepy * (float)y;
    const size_t rowoff = (size_t)y * W;
    float xm[R][VEC], ym[R][VEC], bl[R][VEC], o[R][VEC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int xx = min(x0 + (r * 64 + lane) * VEC + e, W - 1);
            const int cj = min(xx / cww, gw - 1);
            const float* hp = Hs + (ci * gw + cj) * 9;
            float h[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = hp[i];
            map_px(h, -1.0f + stepx * (float)xx, gy, xm[r][e], ym[r][e], bl[r][e]);
        }
    Taps t[R][VEC];
    float Ia[R][VEC], Ib[R][VEC], Ic[R][VEC], Id[R][VEC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            t[r][e] = make_taps(xm[r][e], ym[r][e], H, W);
            Ia[r][e] = src[t[r][e].ia]; Ib[r][e] = src[t[r][e].ib]; Ic[r][e] = src[t[r][e].ic]; Id[r][e] = src[t[r][e].id];
        }
    const size_t hw = (size_t)H * W;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[r][e] = blend(t[r][e], Ia[r][e], Ib[r][e], Ic[r][e], Id[r][e]);
        const int xx = x0 + (r * 64 + lane) * VEC;
        if (xx < W) {                                                // W % VEC == 0
            const size_t pix = rowoff + xx;
            float f[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = o[r][e] + bl[r][e] * -1.0f;
            if (VEC == 4) {
                *reinterpret_cast<float4*>(x_map + pix) = make_float4(xm[r][0], xm[r][1], xm[r][2], xm[r][3]);
                *reinterpret_cast<float4*>(y_map + pix) = make_float4(ym[r][0], ym[r][1], ym[r][2], ym[r][3]);
                *reinterpret_cast<float4*>(black + pix) = make_float4(bl[r][0], bl[r][1], bl[r][2], bl[r][3]);
                *reinterpret_cast<float4*>(out + pix) = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
                *reinterpret_cast<float4*>(push.frames + (size_t)head * hw + pix) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(push.masks + (size_t)head * hw + pix) = make_float4(bl[r][0], bl[r][1], bl[r][2], bl[r][3]);
                *reinterpret_cast<float4*>(push.frame_out + pix) = make_float4(f[0], f[1], f[2], f[3]);
            } else {
                *reinterpret_cast<float2*>(x_map + pix) = make_float2(xm[r][0], xm[r][VEC - 1]);
                *reinterpret_cast<float2*>(y_map + pix) = make_float2(ym[r][0], ym[r][VEC - 1]);
                *reinterpret_cast<float2*>(black + pix) = make_float2(bl[r][0], bl[r][VEC - 1]);
                *reinterpret_cast<float2*>(out + pix) = make_float2(o[r][0], o[r][VEC - 1]);
                *reinterpret_cast<float2*>(push.frames + (size_t)head * hw + pix) = make_float2(f[0], f[VEC - 1]);
                *reinterpret_cast<float2*>(push.masks + (size_t)head * hw + pix) = make_float2(bl[r][0], bl[r][VEC - 1]);
                *reinterpret_cast<float2*>(push.frame_out + pix) = make_float2(f[0], f[VEC - 1]);
            }
        }
    }
}

// ---- ceiling: the same seven planes written + one read, no arithmetic (what the memory system does for this byte mix)
__global__ __launch_bounds__(256) void copy7_kernel(const float4* __restrict__ src, float4* o0, float4* o1, float4* o2, float4* o3, float4* o4,
                                                    float4* o5, float4* o6, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = src[i];
        o0[i] = v; o1[i] = v; o2[i] = v; o3[i] = v; o4[i] = v; o5[i] = v; o6[i] = v;
    }
}

struct Bufs { float *Hs, *src, *out, *black, *xm, *ym, *frames, *masks, *fout; int *head, *ticket; };

static void make_hs(std::vector<float>& h, int gh, int gw, unsigned seed) {
    h.resize((size_t)gh * gw * 9);
    for (int c = 0; c < gh * gw; ++c)
        for (int i = 0; i < 9; ++i) {
            seed = seed * 1664525u + 1013904223u;
            const float r = ((seed >> 8) & 0xffff) / 65536.0f - 0.5f;
            const float id = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
            h[(size_t)c * 9 + i] = id + ((i == 8) ? 0.0f : (i >= 6 ? 0.01f : 0.04f) * r);
        }
}

int main(int argc, char** argv) {
    const int sizes[2][2] = {{720, 1280}, {1080, 1920}};
    for (int s = 0; s < 2; ++s) {
        const int H = argc > 2 ? atoi(argv[1]) : sizes[s][0], W = argc > 2 ? atoi(argv[2]) : sizes[s][1];
        if (argc > 2 && s == 1) break;
        const int gh = 4, gw = 4, depth = 32;
        const size_t hw = (size_t)H * W;
        Bufs b{};
        CK(hipMalloc(&b.Hs, 16 * 9 * 4)); CK(hipMalloc(&b.src, hw * 4));
        float** planes[] = {&b.out, &b.black, &b.xm, &b.ym, &b.fout};
        for (float** p : planes) CK(hipMalloc(p, hw * 4));
        CK(hipMalloc(&b.frames, hw * 4 * depth)); CK(hipMalloc(&b.masks, hw * 4 * depth));
        CK(hipMalloc(&b.head, 8)); CK(hipMemset(b.head, 0, 8)); b.ticket = b.head + 1;
        std::vector<float> hs; make_hs(hs, gh, gw, 12345u);
        CK(hipMemcpy(b.Hs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> img(hw);
        unsigned sd = 7u;
        for (size_t i = 0; i < hw; ++i) { sd = sd * 1664525u + 1013904223u; img[i] = ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(b.src, img.data(), hw * 4, hipMemcpyHostToDevice));
        const int head_slot = 5;
        CK(hipMemcpy(b.head, &head_slot, 4, hipMemcpyHostToDevice));
        const Push push{b.frames, b.masks, b.fout, b.head, b.ticket, depth};
        std::vector<std::vector<float>> ref;
        auto snapshot = [&]() {
            std::vector<std::vector<float>> v(7, std::vector<float>(hw));
            const float* srcs[7] = {b.out, b.black, b.xm, b.ym, b.fout, b.frames + (size_t)head_slot * hw, b.masks + (size_t)head_slot * hw};
            for (int i = 0; i < 7; ++i) CK(hipMemcpy(v[i].data(), srcs[i], hw * 4, hipMemcpyDeviceToHost));
            return v;
        };
        auto clear = [&]() {
            float* dst[7] = {b.out, b.black, b.xm, b.ym, b.fout, b.frames + (size_t)head_slot * hw, b.masks + (size_t)head_slot * hw};
            for (float* d : dst) CK(hipMemset(d, 0xff, hw * 4));
        };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double bytes = 32.0 * hw + 776.0;
        auto run = [&](const char* name, auto&& launch) {
            clear();
            launch();
            CK(hipDeviceSynchronize());
            auto got = snapshot();
            long bad = 0;
            if (ref.empty()) ref = got;
            else for (int i = 0; i < 7; ++i) bad += memcmp(got[i].data(), ref[i].data(), hw * 4) != 0;
            for (int i = 0; i < 5; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 50;
            printf("%4dx%-4d %-44s %7.2f us  %6.0f GB/s  %s\n", W, H, name, us, bytes / us / 1e3, bad ? "MISMATCH" : "bit-identical");
            fflush(stdout);
        };
        run("v0 shipped (4px/thr, LDS Hs, ticket)", [&]() { v0_kernel<1><<<dim3((W + 255) / 256, (H + 3) / 4), 256>>>(b.Hs, b.src, H, W, gh, gw, b.out, b.black, b.xm, b.ym, push); });
        run("v0 without the ticket", [&]() { v0_kernel<0><<<dim3((W + 255) / 256, (H + 3) / 4), 256>>>(b.Hs, b.src, H, W, gh, gw, b.out, b.black, b.xm, b.ym, push); });
#define V1(PX, WAVES, NT) run("v1 lane-strided PX=" #PX " waves/blk=" #WAVES " nt=" #NT, [&]() { \
            const int segs = (W + 64 * PX - 1) / (64 * PX); const long waves = (long)segs * H;            \
            v1_kernel<PX, WAVES, NT><<<(unsigned)((waves + WAVES - 1) / WAVES), 64 * WAVES>>>(b.Hs, b.src, H, W, gh, gw, b.out, b.black, b.xm, b.ym, push); })
        V1(1, 4, 0); V1(2, 4, 0); V1(4, 4, 0); V1(1, 1, 0); V1(2, 1, 0); V1(2, 2, 0); V1(4, 1, 0); V1(2, 4, 1); V1(1, 4, 1); V1(1, 8, 0); V1(2, 8, 0);
#define V2(VEC, R, WAVES) run("v2 vec=" #VEC " R=" #R " waves/blk=" #WAVES, [&]() {                        \
            constexpr int PXW = 64 * VEC * R; const int segs = (W + PXW - 1) / PXW; const long waves = (long)segs * H; \
            v2_kernel<VEC, R, WAVES><<<(unsigned)((waves + WAVES - 1) / WAVES), 64 * WAVES>>>(b.Hs, b.src, H, W, gh, gw, b.out, b.black, b.xm, b.ym, push); })
        V2(4, 1, 4); V2(4, 1, 1); V2(2, 1, 4); V2(2, 2, 4); V2(2, 1, 1);
        {   // byte-mix ceiling
            const long n4 = (long)hw / 4;
            run("copy7 ceiling (1 read + 7 writes, no math)", [&]() {
                copy7_kernel<<<2048, 256>>>((const float4*)b.src, (float4*)b.out, (float4*)b.black, (float4*)b.xm, (float4*)b.ym, (float4*)b.fout,
                                             (float4*)(b.frames + (size_t)head_slot * hw), (float4*)(b.masks + (size_t)head_slot * hw), n4);
            });
        }
        CK(hipFree(b.Hs)); CK(hipFree(b.src)); for (float** p : planes) CK(hipFree(*p));
        CK(hipFree(b.frames)); CK(hipFree(b.masks)); CK(hipFree(b.head));
    }
    return 0;
}
