// Measurement helpers for bench.py ("fraction vs the vendor peak AND vs the empirical peak on the box"):
//   stabnet_probe_mfma_f32 : register-only v_mfma_f32_32x32x2_f32 chains, 4 independent accumulators per wave, one wave
//                            per SIMD and more -- the f32 matrix rate this chip sustains (no LDS, no memory)
//   stabnet_probe_hbm_copy : float4 streaming copy -- the HBM rate a plain kernel sustains (read + write bytes)
// Not on the product path; bench.py times them with events on the same stream.
#include "common.h"

typedef float probe_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(float* __restrict__ out, int iters) {
    probe_f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // non-trivial operands (zeros / denormals would flatter the clock: MI355X_MICROARCH.md, DVFS give-back)
    float x = 0.001f * (float)((threadIdx.x * 37 + blockIdx.x * 11) % 97) - 0.05f;
    float y = 0.002f * (float)((threadIdx.x * 13 + blockIdx.x * 7) % 89) - 0.09f;
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
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = src[i];
}

extern "C" {

/* out: blocks*256 floats.  FLOPs executed = stabnet_probe_mfma_f32_flops(blocks, iters). */
int stabnet_probe_mfma_f32(float* out, int blocks, int iters, void* stream) {
    SN_REQUIRE(out && blocks > 0 && iters > 0, "probe_mfma_f32: bad arguments");
    probe_mfma_f32_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters);
    SN_LAUNCH_CHECK("probe_mfma_f32_kernel");
    return STABNET_OK;
}
double stabnet_probe_mfma_f32_flops(int blocks, int iters) {
    return (double)blocks * 4.0 * (double)iters * 16.0 * 4096.0;      // waves x iterations x MFMAs x (2*32*32*2) flops
}
/* dst[i] = src[i], n_floats % 4 == 0; bytes moved = 8 * n_floats. */
int stabnet_probe_hbm_copy(const float* src, float* dst, long n_floats, void* stream) {
    SN_REQUIRE(src && dst && n_floats > 0 && (n_floats & 3) == 0, "probe_hbm_copy: bad arguments");
    probe_copy_kernel<<<256 * 8, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4*>(src),
                                                               reinterpret_cast<float4*>(dst), n_floats / 4);
    SN_LAUNCH_CHECK("probe_copy_kernel");
    return STABNET_OK;
}

}  // extern "C"
