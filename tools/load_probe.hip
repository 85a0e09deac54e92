// Probe: per-CU global->register throughput under an MFMA loop, for different access shapes (gfx950).
// Each thread issues LD float4 loads per 16 MFMAs from a buffer of `span` bytes (L2/MALL resident or not).
//   pattern 0: wave-contiguous 1 KiB per instruction (lane*16 B)
//   pattern 1: 8 lanes x 128-B segments, segments 256 B apart      (Cin = 64 NHWC rows, BK = 32)
//   pattern 2: 8 lanes x 128-B segments, segments 2304 B apart     (weight rows, K = 576)
//   pattern 3: 16 lanes x 256-B segments, 1 KiB apart              (BK = 64 slices of Cin = 256 rows)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/load_probe tools/load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN, int LD, bool MFMA>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int iters, unsigned span_f4) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    unsigned base = (blockIdx.x * 7919u + wave * 131u) * 64u;       // float4 index, spread blocks over the buffer
    float4 sum = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        float4 v[LD];
#pragma unroll
        for (int l = 0; l < LD; ++l) {
            unsigned idx;
            const unsigned chunk = base + (unsigned)(it * LD + l) * 977u * 64u;
            if (PATTERN == 0) idx = chunk + lane;
            else if (PATTERN == 1) idx = chunk + (lane >> 3) * 16u + (lane & 7);
            else if (PATTERN == 2) idx = chunk + (lane >> 3) * 144u + (lane & 7);
            else idx = chunk + (lane >> 4) * 64u + (lane & 15);
            v[l] = g4[idx % span_f4];
        }
        if (MFMA) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sum.x, sum.y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int l = 0; l < LD; ++l) { sum.x += v[l].x; sum.y += v[l].y; sum.z += v[l].z; sum.w += v[l].w; }
    }
    float s = sum.x + sum.y + sum.z + sum.w;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)(blockIdx.x & 4095) * 256 + tid] = s;
}

template <int PATTERN, int LD, bool MFMA>
void run(const char* name, const float* g, float* out, unsigned span_bytes) {
    const int blocks = 1024, iters = 200;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<PATTERN, LD, MFMA><<<blocks, 256>>>(g, out, iters, span_bytes / 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) probe<PATTERN, LD, MFMA><<<blocks, 256>>>(g, out, iters, span_bytes / 16);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)blocks * 256 * iters * LD * 16.0;
    const double fl = MFMA ? (double)blocks * 4 * iters * 16 * 4096.0 : 0;
    printf("%-28s span=%4u MB LD=%d : %8.1f us  %7.2f TB/s  %5.1f B/clk/CU  %6.1f TFLOP/s\n", name, span_bytes >> 20, LD, ms * 1e3,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, fl / ms / 1e9);
}

int main() {
    float *g, *out;
    const size_t cap = (size_t)1 << 30;
    (void)hipMalloc(&g, cap); (void)hipMalloc(&out, (size_t)4096 * 256 * 4);
    (void)hipMemset(g, 0, cap);
    for (unsigned span : {8u << 20, 64u << 20, 1024u << 20}) {
        run<0, 4, false>("contig 1KiB, no mfma", g, out, span);
        run<1, 4, false>("128B seg /256B, no mfma", g, out, span);
        run<2, 4, false>("128B seg /2304B, no mfma", g, out, span);
        run<3, 4, false>("256B seg /1KiB, no mfma", g, out, span);
        run<0, 4, true>("contig 1KiB + mfma", g, out, span);
        run<1, 4, true>("128B seg /256B + mfma", g, out, span);
        run<2, 4, true>("128B seg /2304B + mfma", g, out, span);
        run<3, 4, true>("256B seg /1KiB + mfma", g, out, span);
        run<0, 8, true>("contig 1KiB + mfma", g, out, span);
        run<1, 8, true>("128B seg /256B + mfma", g, out, span);
    }
    return 0;
}
