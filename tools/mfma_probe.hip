// Standalone probe: where does an LDS-fed f32-MFMA loop lose throughput on gfx950?
// hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PITCH 36

template <int MODE, int NACC, int NVALU = 0>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * PITCH; i += 256) smem[i] = g[i % 4096];
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int frag = (lane & 31) * PITCH + (lane >> 5) * 4;
    const float* Ab = smem + (wave & 1) * 32 * PITCH + frag;
    const float* Bb = smem + (64 + (wave >> 1) * 32) * PITCH + frag;
    float4 ra = make_float4(0, 0, 0, 0), rb = ra;
    const float* gp = g + (size_t)(blockIdx.x & 2047) * 8192 + tid * 4;   // g holds 4096*8192 floats: stay well inside
    float4 af = *(const float4*)Ab, bf = *(const float4*)Bb;
    for (int it = 0; it < iters; ++it) {
        if (NVALU > 0) {
            int t = it + tid;
#pragma unroll
            for (int q = 0; q < NVALU; ++q) t = (t ^ (t >> 3)) + q;      // dependent integer VALU chain, 2 ops per step
            asm volatile("" ::"v"(t));
        }
        if (MODE >= 3) { ra = *(const float4*)(gp + (it & 7) * 1024); rb = *(const float4*)(gp + 4096 + (it & 7) * 1024); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE >= 1) { af = *(const float4*)(Ab + kk * 8); bf = *(const float4*)(Bb + kk * 8); }
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[a], 0, 0, 0);
            }
        }
        if (MODE >= 3) {
            *(float4*)(smem + 128 * PITCH + (tid >> 3) * PITCH + (tid & 7) * 4) = ra;
            *(float4*)(smem + (128 + 64) * PITCH + (tid >> 3) * PITCH + (tid & 7) * 4) = rb;
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)(blockIdx.x & 4095) * 256 + tid] = s;
}

template <int MODE, int NACC, int NVALU = 0>
void run(const char* name, int blocks, int iters, float* g, float* out) {
    const size_t lds = 2 * 128 * PITCH * sizeof(float);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NACC, NVALU><<<blocks, 256, lds>>>(g, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<MODE, NACC, NVALU><<<blocks, 256, lds>>>(g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double fl = (double)blocks * 4 * iters * 16.0 * NACC * 4096.0;
    printf("%-34s blocks=%5d acc=%d : %8.1f us  %6.1f TFLOP/s\n", name, blocks, NACC, ms * 1e3, fl / ms / 1e9);
}

int main() {
    float *g, *out;
    hipMalloc(&g, (size_t)4096 * 8192 * 4); hipMalloc(&out, (size_t)4096 * 256 * 4);
    std::vector<float> h(4096 * 64, 0.5f); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    for (int i = 0; i < 128; ++i) hipMemcpy(g + (size_t)i * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {900, 1024, 4096}) {
        run<0, 1>("pure MFMA chain", blocks, 400, g, out);
        run<0, 4>("pure MFMA 4 acc", blocks, 100, g, out);
        run<1, 1>("+ds_read_b128", blocks, 400, g, out);
        run<1, 4>("+ds_read_b128 4acc", blocks, 100, g, out);
        run<2, 1>("+barrier/16mfma", blocks, 400, g, out);
        run<3, 1>("+global ld + lds st", blocks, 400, g, out);
        run<3, 4>("+global ld + lds st 4acc", blocks, 100, g, out);
        run<3, 1, 25>("+50 VALU/16mfma", blocks, 400, g, out);
        run<3, 1, 50>("+100 VALU/16mfma", blocks, 400, g, out);
        run<3, 1, 100>("+200 VALU/16mfma", blocks, 400, g, out);
        run<3, 4, 100>("+200 VALU/64mfma 4acc", blocks, 100, g, out);
    }
    return 0;
}
