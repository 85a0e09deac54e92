// Which form of a float4 copy kernel reaches what rate on this box (the reference point of bench.py's `empirical_peaks.hbm_copy_gbps`).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT> __global__ __launch_bounds__(256) void one_per_thread(const f4* __restrict__ s, f4* __restrict__ d, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i); else d[i] = s[i]; }
}
template <int U, int NT> __global__ __launch_bounds__(256) void block_chunk(const f4* __restrict__ s, f4* __restrict__ d, long n4) {
    // a block owns U*256 consecutive float4 (U KiB x 4): U loads in flight per lane, each wave-instruction 1 KiB contiguous
    const long base = (long)blockIdx.x * (U * 256) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < n4) v[j] = NT ? __builtin_nontemporal_load(s + i) : s[i]; }
#pragma unroll
    for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < n4) { if (NT) __builtin_nontemporal_store(v[j], d + i); else d[i] = v[j]; } }
}
template <int U, int NT> __global__ __launch_bounds__(256) void grid_stride(const f4* __restrict__ s, f4* __restrict__ d, long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(s + i + j * stride) : s[i + j * stride];
#pragma unroll
        for (int j = 0; j < U; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + i + j * stride); else d[i + j * stride] = v[j]; }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
int main() {
    const long n = 1L << 28, n4 = n / 4;              // 1 GiB each way
    float *s, *d; CK(hipMalloc(&s, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMemset(s, 1, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %7.1f GB/s\n", name, 5 * 8.0 * n / (ms * 1e-3) / 1e9); fflush(stdout);
    };
    const f4* S = (const f4*)s; f4* D = (f4*)d;
    run("one float4 per thread, plain", [&] { one_per_thread<0><<<(unsigned)((n4 + 255) / 256), 256>>>(S, D, n4); });
    run("one float4 per thread, nt", [&] { one_per_thread<1><<<(unsigned)((n4 + 255) / 256), 256>>>(S, D, n4); });
    run("block chunk 4 x 256 float4, plain", [&] { block_chunk<4, 0><<<(unsigned)((n4 + 1023) / 1024), 256>>>(S, D, n4); });
    run("block chunk 4 x 256 float4, nt", [&] { block_chunk<4, 1><<<(unsigned)((n4 + 1023) / 1024), 256>>>(S, D, n4); });
    run("block chunk 8 x 256 float4, nt", [&] { block_chunk<8, 1><<<(unsigned)((n4 + 2047) / 2048), 256>>>(S, D, n4); });
    run("grid-stride 2048 blocks, 8 in flight, nt", [&] { grid_stride<8, 1><<<2048, 256>>>(S, D, n4); });
    run("grid-stride 2048 blocks, 8 in flight, plain", [&] { grid_stride<8, 0><<<2048, 256>>>(S, D, n4); });
    run("grid-stride 8192 blocks, 4 in flight, nt", [&] { grid_stride<4, 1><<<8192, 256>>>(S, D, n4); });
    run("grid-stride 1024 blocks, 16 in flight, nt", [&] { grid_stride<16, 1><<<1024, 256>>>(S, D, n4); });
    run("hipMemcpyDtoDAsync", [&] { CK(hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice)); });
    // read-only and write-only streams of the same size
    return 0;
}
