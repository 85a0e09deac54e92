// Probe: does global_load_lds_dwordx4 accept a source that is only 4-byte aligned, and at what cost?
// Each lane copies 16 B from src + OFF floats (OFF = 0..3) + a 13-float "pixel" stride pattern (52-B rows).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/glds_unaligned tools/glds_unaligned_probe.hip && /tmp/glds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ dst, int off, int iters, size_t span) {
    __shared__ __attribute__((aligned(16))) float smem[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = src + (size_t)blockIdx.x * 4096 % span;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 2; ++i) {
            // 8 rows x 8 chunks per instruction; rows 104 floats apart (two 13-float pixels x 4 = arbitrary 4-B aligned rows)
            const float* g = base + (size_t)((wave * 2 + i) * 8 + (lane >> 3)) * 104 + (lane & 7) * 4 + off + (size_t)it * 832 % span;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(smem + (wave * 2 + i) * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0)
        for (int i = tid; i < 2048; i += 256) dst[i] = smem[i];
}

int main() {
    const size_t n = 64u << 20;
    std::vector<float> h(1 << 20), o(2048);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000003);
    float *s, *d;
    (void)hipMalloc(&s, n * 4 + 65536); (void)hipMalloc(&d, 8192);
    (void)hipMemset(s, 0, n * 4);
    (void)hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        probe<<<1, 256>>>(s, d, off, 1, n / 2);
        (void)hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int g = 0; g < 8; ++g)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const size_t srci = (size_t)(g * 8 + (l >> 3)) * 104 + (l & 7) * 4 + off + e;
                    if (o[(g * 64 + l) * 4 + e] != h[srci]) ++bad;
                }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<<<4096, 256>>>(s, d, off, 64, n / 2);
        (void)hipEventRecord(e0);
        probe<<<4096, 256>>>(s, d, off, 64, n / 2);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("source offset %d floats (%2d-byte aligned): %d mismatches; 4096 blocks x 64 x 8 KiB in %.1f us = %.2f TB/s\n", off,
               off == 0 ? 16 : (off == 2 ? 8 : 4), bad, ms * 1e3, 4096.0 * 64 * 8192 / ms / 1e9);
    }
    return 0;
}
