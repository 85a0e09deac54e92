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

#ifndef HC_STAMP
#define HC_STAMP 0                     // probe-only (tools/head_probe.hip): s_memrealtime stamps of the phases, per workgroup
#endif
#if HC_STAMP
__device__ unsigned long long g_hc_stamps[HC_BLOCKS * 16];
#define HC_STAMP_AT(i) do { if (threadIdx.x == 0) g_hc_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HC_STAMP_AT(i) do { } while (0)
#endif

__device__ __forceinline__ void hc_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float hc_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Four 16-B sc1 loads in flight, then ONE wait (relaxed atomic loads are issued one round trip at a time by the compiler: the
// first version of this kernel spent 64 dependent ~1 us trips per thread in P1 -- 69 us for the whole head).  The wait is inside
// the asm statement, so no register copy can land between a load and its wait.
__device__ __forceinline__ void hc_load8x4(const float* const (&p)[8], float4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}
__device__ __forceinline__ void hc_load4x4(const float* p0, const float* p1, const float* p2, const float* p3, float4& a, float4& b,
                                           float4& c, float4& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// xs[0 .. n4*4) = src[0 .. n4*4) (handed-off floats, 16-B aligned), every load sc1, four in flight per thread
__device__ __forceinline__ void hc_stage(float* xs, const float* src, int n4) {
    for (int i = threadIdx.x; i < n4; i += 4 * 256) {
        const int i1 = min(i + 256, n4 - 1), i2 = min(i + 512, n4 - 1), i3 = min(i + 768, n4 - 1);
        float4 a, b, c, d;
        hc_load4x4(src + 4 * (size_t)i, src + 4 * (size_t)i1, src + 4 * (size_t)i2, src + 4 * (size_t)i3, a, b, c, d);
        *reinterpret_cast<float4*>(xs + 4 * i) = a;
        if (i + 256 < n4) *reinterpret_cast<float4*>(xs + 4 * i1) = b;
        if (i + 512 < n4) *reinterpret_cast<float4*>(xs + 4 * i2) = c;
        if (i + 768 < n4) *reinterpret_cast<float4*>(xs + 4 * i3) = d;
    }
}

__device__ __forceinline__ void hc_arrive(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's hand-off stores have left the CU
    __syncthreads();                                          // ... and so have every other wave's
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hc_wait(unsigned* cnt, unsigned* timeout_flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)HC_BLOCKS) {
            __builtin_amdgcn_s_sleep(1);
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
    HC_STAMP_AT(0);
    const int gwave = blockIdx.x * 4 + wave;                              // global wave id, 0 .. 1023
    const int N = p.N;

    // ---- P0 (first item of this workgroup): chunked sums of relu(x * scale + shift); item = (sample, 64-channel group, row chunk),
    //      16 channel quads x 16 row lanes.  Its loads go out FIRST, eight rows per thread at once ...
    const int q = tid & 15, rl = tid >> 4;
    const int groups = p.C / 64, items = N * groups * p.chunks;
    auto item_geom = [&](int it, int& n, int& chunk, int& c, int& r0, int& r1) {
        n = it / (groups * p.chunks);
        const int rem = it - n * groups * p.chunks;
        chunk = rem / groups;
        c = (rem - chunk * groups) * 64 + q * 4;
        r0 = chunk * p.rows_per_chunk;
        r1 = min(p.HW, r0 + p.rows_per_chunk);
    };
    float4 xv[8];
    int n0 = 0, chunk0 = 0, c0 = 0, r00 = 0, r10 = 0;
    const bool have0 = (int)blockIdx.x < items;
    if (have0) {
        item_geom(blockIdx.x, n0, chunk0, c0, r00, r10);
        const float* px = p.x + (size_t)n0 * p.HW * p.C + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = r00 + rl + 16 * j;
            xv[j] = (i < r10) ? *reinterpret_cast<const float4*>(px + (size_t)i * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // ---- ... then the weight stream: the rows of this wave for fc_1 / fc_2 / fc_3, all loads issued now; they land while P0
    //      reduces, the workgroup waits at the first barrier and stages x
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

    for (int it = blockIdx.x; it < items; it += HC_BLOCKS) {
        int n, chunk, c, r0, r1;
        item_geom(it, n, chunk, c, r0, r1);
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + c), sh = *reinterpret_cast<const float4*>(p.shift + c);
        const float* px = p.x + (size_t)n * p.HW * p.C + c;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ib = r0 + rl; ib < r1; ib += 128) {                   // eight rows per trip (the first trip of the first item is in xv)
            if (!(it == (int)blockIdx.x && ib == r0 + rl)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = ib + 16 * j;
                    xv[j] = (i < r1) ? *reinterpret_cast<const float4*>(px + (size_t)i * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ib + 16 * j < r1) {                                 // same row order as a one-row-per-trip loop
                    const float4 v = xv[j];
                    s.x += fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), 0.f); s.y += fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), 0.f);
                    s.z += fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), 0.f); s.w += fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), 0.f);
                }
            }
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
    HC_STAMP_AT(1);
    hc_arrive(p.sync + 0);
    HC_STAMP_AT(2);
    hc_wait(p.sync + 0, p.sync + 7);
    HC_STAMP_AT(3);

    // ---- P1: x = mean (chunk partials summed in chunk order), then fc_1
    for (int i = tid; i < N * (p.C / 4); i += 256) {                   // one channel quad of one sample
        const int n = i / (p.C / 4), c = (i - n * (p.C / 4)) * 4;
        const float* base = p.partial + (size_t)n * p.chunks * p.C + c;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < p.chunks; k += 8) {                        // (chunks <= 8: one trip, eight loads in flight)
            const float* ptr[8];
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ptr[j] = base + (size_t)min(k + j, p.chunks - 1) * p.C;
            hc_load8x4(ptr, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k + j < p.chunks) { t.x += v[j].x; t.y += v[j].y; t.z += v[j].z; t.w += v[j].w; }
        }
        const float hw = (float)p.HW;
        t.x = t.x / hw; t.y = t.y / hw; t.z = t.z / hw; t.w = t.w / hw;
        *reinterpret_cast<float4*>(xs + n * p.C + c) = t;
        if (blockIdx.x == 0) *reinterpret_cast<float4*>(p.gap + (size_t)n * p.C + c) = t;   // the "global_pool" tap
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
    HC_STAMP_AT(4);
    hc_arrive(p.sync + 1);
    HC_STAMP_AT(5);
    hc_wait(p.sync + 1, p.sync + 7);
    HC_STAMP_AT(6);

    // ---- P2: fc_2
    hc_stage(xs, p.fc[0], N * 2048 / 4);
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
    HC_STAMP_AT(7);
    hc_arrive(p.sync + 2);
    hc_wait(p.sync + 2, p.sync + 7);
    HC_STAMP_AT(8);

    // ---- P3: fc_3
    hc_stage(xs, p.fc[1], N * 1024 / 4);
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
    HC_STAMP_AT(9);
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
    HC_STAMP_AT(10);
    if (!p4) return;
    const int n = blockIdx.x / bps, cg = blockIdx.x - n * bps;
    hc_stage(xs, p.fc[2] + (size_t)n * 512, 512 / 4);
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
    HC_STAMP_AT(11);
    if (p.Hs == nullptr) return;
    if (p.head_adv != nullptr && blockIdx.x == 0 && tid == 0) *p.head_adv = (*p.head_adv + 1) % p.depth;   // see launch_sample_push
    const int cell = cg * 4 + wave;
    if (cell < cells)
        sn_mesh_cell(th, 1, p.gh, p.gw, p.lim, cell, lane, nullptr, p.Hs + (size_t)n * cells * 9, nullptr);
    HC_STAMP_AT(12);
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
