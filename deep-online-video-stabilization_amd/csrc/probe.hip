// Measurement helpers for bench.py ("fraction vs the vendor peak AND vs the empirical peak on the box"):
//   stabnet_probe_mfma_f32 : register-only v_mfma_f32_32x32x2_f32 chains, 4 independent accumulators per wave, one wave
//                            per SIMD and more -- the f32 matrix rate this chip sustains (no LDS, no memory)
//   stabnet_probe_hbm_copy : float4 streaming copy -- the HBM rate a plain kernel sustains (read + write bytes)
// Not on the product path; bench.py times them with events on the same stream.
#include "common.h"

typedef float probe_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(float* __restrict__ out, int iters, unsigned long long* __restrict__ stamps) {
    probe_f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // non-trivial operands (zeros / denormals would flatter the clock: MI355X_MICROARCH.md, DVFS give-back)
    float x = 0.001f * (float)((threadIdx.x * 37 + blockIdx.x * 11) % 97) - 0.05f;
    float y = 0.002f * (float)((threadIdx.x * 13 + blockIdx.x * 7) % 89) - 0.09f;
    // the clock the chip holds inside the loop: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime), MI355X_MICROARCH.md
    // "DVFS give-back" item 6
    unsigned long long c0 = 0, r0 = 0;
    if (stamps != nullptr) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        x = -x;
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (stamps != nullptr && threadIdx.x == 0) {
        // (s depends on every MFMA of the loop: reading it orders the stamps behind them)
        const unsigned long long c1 = __builtin_amdgcn_s_memtime() + (unsigned long long)(s != s), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// One float4 per thread, non-temporal, the grid covers the buffer once (tools/copy_probe.hip, profiles/r03_copy_probe.txt: 6.53 TB/s;
// plain loads / stores 6.20; a 2048-block grid-stride loop with 8 loads in flight per lane 4.4 -- its lanes revisit the same
// channel interleave every trip; the round-2 probe, one load per trip of such a loop, 4.9; hipMemcpyDtoD 4.8).
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe_copy_kernel(const probe_f32x4* __restrict__ src, probe_f32x4* __restrict__ dst, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// A stand-in for a collective's kernel on a one-GPU box: FEW long-lived workgroups (RCCL runs one per channel)
// that stream dst[i] += src[i] over a bucket with a grid-stride loop, each holding `lds` bytes of LDS so that it competes for the
// same CU resources a real collective kernel would.  Launched on a second stream beside the backward it answers two questions a
// one-rank RCCL group cannot (a sum over one rank moves nothing): does a small kernel get CU slots while the persistent conv
// kernels own every CU, and what does the step pay for it.
__global__ __launch_bounds__(256) void probe_comm_proxy_kernel(const probe_f32x4* __restrict__ src, probe_f32x4* __restrict__ dst, long n4,
                                                               unsigned long long* __restrict__ stamps) {
    extern __shared__ float proxy_lds[];
    if (threadIdx.x == 0 && stamps != nullptr) stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    float keep = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const probe_f32x4 a = src[i], b = dst[i];
        dst[i] = a + b;
        keep += a[0];
    }
    if (keep == 123.456f) proxy_lds[threadIdx.x] = keep;            // (keeps the LDS allocation alive)
    if (threadIdx.x == 0 && stamps != nullptr) stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

extern "C" {

/* dst[i] += src[i] by `workgroups` long-lived workgroups of 256 threads holding lds_bytes of LDS each (stand-in for a collective's
 * kernel; bytes moved = 12 * n_floats).  stamps (optional, device, 2 words per workgroup): 100 MHz ticks at entry and exit. */
int stabnet_probe_comm_proxy(const float* src, float* dst, long n_floats, int workgroups, int lds_bytes, unsigned long long* stamps,
                             void* stream) {
    SN_REQUIRE(src && dst && n_floats > 0 && (n_floats & 3) == 0 && workgroups > 0 && lds_bytes >= 0 && lds_bytes <= 64 * 1024,
               "probe_comm_proxy: bad arguments");
    probe_comm_proxy_kernel<<<workgroups, 256, (size_t)lds_bytes, (hipStream_t)stream>>>(reinterpret_cast<const probe_f32x4*>(src),
                                                                                      reinterpret_cast<probe_f32x4*>(dst), n_floats / 4, stamps);
    SN_LAUNCH_CHECK("probe_comm_proxy_kernel");
    return STABNET_OK;
}

/* out: blocks*256 floats.  FLOPs executed = stabnet_probe_mfma_f32_flops(blocks, iters).  stamps (optional, device, 2 words per
 * block): {shader cycles, 100 MHz ticks} spent inside the MFMA loop by wave 0 of each block -> the clock held under load. */
int stabnet_probe_mfma_f32(float* out, int blocks, int iters, unsigned long long* stamps, void* stream) {
    SN_REQUIRE(out && blocks > 0 && iters > 0, "probe_mfma_f32: bad arguments");
    probe_mfma_f32_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters, stamps);
    SN_LAUNCH_CHECK("probe_mfma_f32_kernel");
    return STABNET_OK;
}
double stabnet_probe_mfma_f32_flops(int blocks, int iters) {
    return (double)blocks * 4.0 * (double)iters * 16.0 * 4096.0;      // waves x iterations x MFMAs x (2*32*32*2) flops
}
/* dst[i] = src[i], n_floats % 4 == 0; bytes moved = 8 * n_floats. */
int stabnet_probe_hbm_copy(const float* src, float* dst, long n_floats, void* stream) {
    SN_REQUIRE(src && dst && n_floats > 0 && (n_floats & 3) == 0, "probe_hbm_copy: bad arguments");
    SN_REQUIRE(n_floats / 4 / 256 < (1L << 31), "probe_hbm_copy: buffer too large");
    probe_copy_kernel<<<(unsigned)cdiv(n_floats / 4, 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const probe_f32x4*>(src),
                                                                                         reinterpret_cast<probe_f32x4*>(dst), n_floats / 4);
    SN_LAUNCH_CHECK("probe_copy_kernel");
    return STABNET_OK;
}

}  // extern "C"
