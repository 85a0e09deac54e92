// The regressor's head as ONE launch (inference, batch <= 8): postnorm BN + ReLU + reduce_mean -> fc_1 -> fc_2 -> fc_3 ->
// output_layer (-> get_4_pts + get_Hs + the online loop's ring-head advance), s_net_bundle_nobm.py:254-262, resnet.py:44-56,
// spatial_transformer3.py:144-198.  Layer by layer these were 2 + 4 + 1 launches of 5-8 us each with one result vector
// between them; here 256 resident workgroups (one per CU) walk the phases with a grid barrier in between:
//
//   start : every wave issues the loads of ITS weight rows of fc_1 / fc_2 / fc_3 (27 MB over the chip, registers) -- the only
//           HBM stream of the head, in flight while phase 0 runs
//   P0    : partial sums of relu(bn(x)) over row chunks                -> partial [N][chunks][C]      (sc1 stores)
//   P1    : x = sum of the chunk partials / HW (each workgroup for itself, LDS); fc_1 rows w, w + 1024 -> fc1 [N][2048]
//   P2    : fc_2 row w                                                 -> fc2 [N][1024]
//   P3    : fc_3 row w (w < 512)                                       -> fc3 [N][512]
//   P4    : workgroup (n, cell group): the whole output_layer of sample n (512 x n_theta, 100 KB, prefetched while the
//           workgroup waits at the barrier) -> theta; one wave per cell of the group -> Hs
//
// Hand-off between phases (MI355X_MICROARCH.md, "inter-workgroup visibility"): every byte another workgroup reads is stored
// with agent-scope relaxed atomics (global_store sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)) before the
// workgroup barrier behind which ONE lane adds to the phase counter; the consumer polls that counter with sc1 loads from one
// lane, a workgroup barrier follows, and every load of handed-off bytes is an sc1 load.  No dependence on workgroup -> XCD
// placement.  The grid must be resident: 256 workgroups of 256 threads, one per CU (the launcher checks the CU count and
// falls back to the layer-by-layer launches otherwise); every spin is bounded and sets sync[7] on timeout.
// Per output element the arithmetic and its order are those of fc_kernel (layers.hip).
#include "layers.h"
#include "mesh_device.h"
#include <algorithm>
#include <cstdlib>

#define HC_BLOCKS 256
#define HC_WAVES (HC_BLOCKS * 4)

__device__ __forceinline__ void hc_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float hc_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void hc_arrive(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's hand-off stores have left the CU
    __syncthreads();                                          // ... and so have every other wave's
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hc_wait(unsigned* cnt, unsigned* timeout_flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)HC_BLOCKS) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {                       // never hang the GPU: give up, flag it, let the grid drain
                __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float hc_dot4(float4 x, float4 w) { return ((x.x * w.x + x.y * w.y) + x.z * w.z) + x.w * w.w; }
__device__ __forceinline__ float hc_wave_sum(float v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void head_chain_kernel(const HeadChainArgs p) {
    __shared__ __attribute__((aligned(16))) float xs[8 * 2048];        // the current layer's input, [N][K]
    __shared__ float4 part[16][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gwave = blockIdx.x * 4 + wave;                              // global wave id, 0 .. 1023
    const int N = p.N;

    // ---- the weight stream: rows of this wave, all loads issued now
    float4 w1[2][8], w2[8], w3[4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) w1[r][j] = *reinterpret_cast<const float4*>(p.w[0] + (size_t)(gwave + r * HC_WAVES) * 2048 + j * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) w2[j] = *reinterpret_cast<const float4*>(p.w[1] + (size_t)gwave * 2048 + j * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        w3[j] = (gwave < 512) ? *reinterpret_cast<const float4*>(p.w[2] + (size_t)gwave * 1024 + j * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- P0: chunked sums of relu(x * scale + shift): item = (sample, 64-channel group, row chunk); 16 channel quads x 16 row lanes
    {
        const int q = tid & 15, rl = tid >> 4;
        const int groups = p.C / 64, items = N * groups * p.chunks;
        for (int it = blockIdx.x; it < items; it += HC_BLOCKS) {
            const int n = it / (groups * p.chunks), rem = it - n * groups * p.chunks;
            const int chunk = rem / groups, c = (rem - chunk * groups) * 64 + q * 4;
            const int r0 = chunk * p.rows_per_chunk, r1 = min(p.HW, r0 + p.rows_per_chunk);
            const float4 sc = *reinterpret_cast<const float4*>(p.scale + c), sh = *reinterpret_cast<const float4*>(p.shift + c);
            const float* px = p.x + (size_t)n * p.HW * p.C + c;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = r0 + rl; i < r1; i += 16) {
                const float4 v = *reinterpret_cast<const float4*>(px + (size_t)i * p.C);
                s.x += fmaxf(v.x * sc.x + sh.x, 0.f); s.y += fmaxf(v.y * sc.y + sh.y, 0.f);
                s.z += fmaxf(v.z * sc.z + sh.z, 0.f); s.w += fmaxf(v.w * sc.w + sh.w, 0.f);
            }
            part[rl][q] = s;
            __syncthreads();
            if (rl == 0) {
                float4 t = part[0][q];
                for (int i = 1; i < 16; ++i) { const float4 u = part[i][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                float* o = p.partial + ((size_t)n * p.chunks + chunk) * p.C + c;
                hc_store(o, t.x); hc_store(o + 1, t.y); hc_store(o + 2, t.z); hc_store(o + 3, t.w);
            }
            __syncthreads();
        }
    }
    hc_arrive(p.sync + 0);
    hc_wait(p.sync + 0, p.sync + 7);

    // ---- P1: x = mean (chunk partials summed in chunk order), then fc_1
    for (int i = tid; i < N * p.C; i += 256) {
        const int n = i / p.C, c = i - n * p.C;
        float t = 0.f;
        for (int k = 0; k < p.chunks; ++k) t += hc_load(p.partial + ((size_t)n * p.chunks + k) * p.C + c);
        t = t / (float)p.HW;
        xs[i] = t;
        if (blockIdx.x == 0) p.gap[i] = t;                             // the "global_pool" tap
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = gwave + r * HC_WAVES;
        const float bias = p.b[0][row];
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += hc_dot4(*reinterpret_cast<const float4*>(xs + n * 2048 + j * 256 + lane * 4), w1[r][j]);
            acc = hc_wave_sum(acc);
            if (lane == 0) hc_store(p.fc[0] + (size_t)n * 2048 + row, fmaxf(acc + bias, 0.f));
        }
    }
    hc_arrive(p.sync + 1);
    hc_wait(p.sync + 1, p.sync + 7);

    // ---- P2: fc_2
    for (int i = tid; i < N * 2048; i += 256) xs[i] = hc_load(p.fc[0] + i);
    __syncthreads();
    {
        const float bias = p.b[1][gwave];
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += hc_dot4(*reinterpret_cast<const float4*>(xs + n * 2048 + j * 256 + lane * 4), w2[j]);
            acc = hc_wave_sum(acc);
            if (lane == 0) hc_store(p.fc[1] + (size_t)n * 1024 + gwave, fmaxf(acc + bias, 0.f));
        }
    }
    hc_arrive(p.sync + 2);
    hc_wait(p.sync + 2, p.sync + 7);

    // ---- P3: fc_3
    for (int i = tid; i < N * 1024; i += 256) xs[i] = hc_load(p.fc[1] + i);
    __syncthreads();
    if (gwave < 512) {
        const float bias = p.b[2][gwave];
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += hc_dot4(*reinterpret_cast<const float4*>(xs + n * 1024 + j * 256 + lane * 4), w3[j]);
            acc = hc_wave_sum(acc);
            if (lane == 0) hc_store(p.fc[2] + (size_t)n * 512 + gwave, fmaxf(acc + bias, 0.f));
        }
    }
    hc_arrive(p.sync + 3);

    // ---- P4: workgroup b < N * bps: sample n = b / bps, cells 4 (b % bps) + wave.  The output layer's rows of this wave
    //      (wave, wave + 4, ...) are loaded while the workgroup waits for fc_3.
    const int cells = p.gh * p.gw;
    const int bps = p.Hs != nullptr ? (cells + 3) / 4 : 1;             // workgroups per sample
    const bool p4 = (int)blockIdx.x < N * bps;
    constexpr int RMAX = 16;                                           // rows per wave: n_theta <= 64
    float4 w4[RMAX][2];
    if (p4) {
#pragma unroll
        for (int i = 0; i < RMAX; ++i) {
            const int row = wave + 4 * i;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                w4[i][j] = (row < p.n_theta) ? *reinterpret_cast<const float4*>(p.w[3] + (size_t)row * 512 + j * 256 + lane * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    hc_wait(p.sync + 3, p.sync + 7);
    if (!p4) return;
    const int n = blockIdx.x / bps, cg = blockIdx.x - n * bps;
    for (int i = tid; i < 512; i += 256) xs[i] = hc_load(p.fc[2] + (size_t)n * 512 + i);
    __syncthreads();
    float* th = xs + 1024;                                             // theta of sample n, LDS
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int row = wave + 4 * i;
        if (row < p.n_theta) {                                         // wave-uniform
            float acc = hc_dot4(*reinterpret_cast<const float4*>(xs + lane * 4), w4[i][0]);
            acc += hc_dot4(*reinterpret_cast<const float4*>(xs + 256 + lane * 4), w4[i][1]);
            acc = hc_wave_sum(acc);
            if (lane == 0) {
                float v = acc + p.b[3][row];                           // output_layer: no activation
                if (__hip_atomic_load(p.sync + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) v = __builtin_nanf("");   // a barrier timed out: fail loudly
                th[row] = v;
                if (cg == 0) p.theta[(size_t)n * p.n_theta + row] = v;
            }
        }
    }
    __syncthreads();
    if (p.Hs == nullptr) return;
    if (p.head_adv != nullptr && blockIdx.x == 0 && tid == 0) *p.head_adv = (*p.head_adv + 1) % p.depth;   // see launch_sample_push
    const int cell = cg * 4 + wave;
    if (cell < cells)
        sn_mesh_cell(th, 1, p.gh, p.gw, p.lim, cell, lane, nullptr, p.Hs + (size_t)n * cells * 9, nullptr);
}

// 1 when the fused head can run this shape on this device (else the caller launches the layers one by one).
int head_chain_supported(int N, int HW, int C, const int* fc_dims /* [5]: C, 2048, 1024, 512, n_theta */, int gh, int gw) {
    static const int cus = []() {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    }();
    static const int want = []() { const char* v = getenv("STABNET_HEAD_CHAIN"); return v ? atoi(v) : 1; }();
    return want && cus >= HC_BLOCKS && N >= 1 && N <= 8 && HW >= 1 && C == 2048 && fc_dims[0] == 2048 && fc_dims[1] == 2048 &&
           fc_dims[2] == 1024 && fc_dims[3] == 512 && fc_dims[4] >= 1 && fc_dims[4] <= 64 && gh * gw <= 64 &&
           N * ((gh * gw + 3) / 4) <= HC_BLOCKS;
}

int head_chain_chunks(int N, int HW) { return std::max(1, std::min(std::max(1, 8 / N), HW / 16)); }

int launch_head_chain(HeadChainArgs a, hipStream_t st) {
    SN_REQUIRE(a.x && a.scale && a.shift && a.partial && a.gap && a.theta && a.sync, "head_chain: null pointer");
    a.chunks = head_chain_chunks(a.N, a.HW);
    a.rows_per_chunk = cdiv(a.HW, a.chunks);
    head_chain_kernel<<<HC_BLOCKS, 256, 0, st>>>(a);
    SN_LAUNCH_CHECK("head_chain_kernel");
    return STABNET_OK;
}
