// Probe: semantics of __builtin_amdgcn_global_load_lds (16 B per lane) on gfx950 -- destination = wave-uniform LDS base +
// lane*16, source = per-lane global pointer -- and the counted-vmcnt + raw-barrier handshake.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/glds_probe tools/glds_probe.hip && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each wave copies 2 KiB: two instructions of 1 KiB; lane l of instruction i reads source chunk PERMUTED (l ^ 5)
    for (int i = 0; i < 2; ++i) {
        const float* g = src + ((wave * 2 + i) * 64 + (lane ^ 5)) * 4;
        float* l = smem + (wave * 2 + i) * 256;                       // wave-uniform base (floats)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < 2048; i += 256) dst[i] = smem[i];
}

int main() {
    std::vector<float> h(2048), o(2048);
    for (int i = 0; i < 2048; ++i) h[i] = (float)i;
    float *s, *d;
    (void)hipMalloc(&s, 8192); (void)hipMalloc(&d, 8192);
    (void)hipMemcpy(s, h.data(), 8192, hipMemcpyHostToDevice);
    probe<<<1, 256, 8192>>>(s, d);
    (void)hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 8; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const float want = (float)((w * 64 + (l ^ 5)) * 4 + e);          // LDS chunk l holds SOURCE chunk l^5
                if (o[(w * 64 + l) * 4 + e] != want) ++bad;
            }
    printf("glds probe: %d mismatches (0 expected): LDS[base + lane*16] <- *per-lane source\n", bad);
    return bad != 0;
}
