// The regressor's head at inference (batch <= 8): postnorm BN + ReLU + reduce_mean -> fc_1 -> fc_2 -> fc_3 -> output_layer
// (-> get_4_pts + get_Hs + the online loop's ring-head advance), s_net_bundle_nobm.py:254-262, resnet.py:44-56,
// spatial_transformer3.py:144-198.  Was 2 + 4 + 1 launches; now 1 + 3 + 1:
//
//   gap_bn_relu_partial_kernel (layers.hip)  partial sums over 8 row chunks                     (unchanged kernel, fewer chunks)
//   fc_gap_kernel     fc_1 whose input staging IS the reduce_mean finalize: every workgroup sums the 8 chunk partials of
//                     every channel into LDS (/ HW) while its weight rows are already in flight       (gap_finalize launch gone)
//   fc_kernel x 2     fc_2, fc_3                                                                 (layers.hip)
//   theta_mesh_kernel output_layer + mesh: the workgroups that own a sample's cells each compute the WHOLE output layer of
//                     that sample (512 x n_theta, 100 KB of weights from L2) into LDS, then one wave per cell solves its
//                     homography; one thread advances the ring head                               (fc + mesh launches -> 1)
//
// What was measured and NOT kept (tools/scratch/head_chain_experiment.hip, profiles/r03_head_chain_probe.txt): the whole head as
// ONE persistent launch of 256 workgroups with four grid barriers (sc1 hand-off, counter + poll).  Every barrier costs
// 6.7-8 us after the last arrival on this chip (store drain ~2 us + device-scope atomic ~2 us + poll round trip ~2 us, all
// memory-side because the eight XCD L2s are not coherent), against ~1.5-2 us for a kernel boundary: 55-58 us for the chain in
// the frame (65 us from cold caches) vs 43.6 us for the seven launches it replaced; 534-538 vs 542 frames/s.
#include "layers.h"
#include "mesh_device.h"
#include <algorithm>

__device__ __forceinline__ float hd_dot4(float4 x, float4 w) { return ((x.x * w.x + x.y * w.y) + x.z * w.z) + x.w * w.w; }

// fc_1 on the pooled feature: y[m][n] = relu(sum_k x[m][k] W[n][k] + b[n]) with x[m][k] = (sum over chunks of partial[m][chunk][k]) / HW.
// One wave per output row (as fc_kernel: same products in the same order), K <= 2048, M <= 8.
__global__ __launch_bounds__(256) void fc_gap_kernel(const float* __restrict__ partial, int chunks, int HW, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, float* __restrict__ gap_out,
                                                     int M, int K, int Nout) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [M][K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.x * 4 + (tid >> 6);
    // the weight row first: the only HBM stream of the kernel is in flight while x is staged
    float4 wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = j * 256 + lane * 4;
        wv[j] = (n < Nout && k < K) ? *reinterpret_cast<const float4*>(w + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float hw = (float)HW;
    for (int i = tid; i < M * (K / 4); i += 256) {                       // one channel quad of one sample
        const int m = i / (K / 4), c = (i - m * (K / 4)) * 4;
        const float* base = partial + (size_t)m * chunks * K + c;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < chunks; k0 += 8) {                         // chunk order, eight loads in flight
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(base + (size_t)min(k0 + j, chunks - 1) * K);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < chunks) { t.x += v[j].x; t.y += v[j].y; t.z += v[j].z; t.w += v[j].w; }
        }
        t.x = t.x / hw; t.y = t.y / hw; t.z = t.z / hw; t.w = t.w / hw;
        *reinterpret_cast<float4*>(xs + m * K + c) = t;
        if (blockIdx.x == 0 && gap_out != nullptr) *reinterpret_cast<float4*>(gap_out + (size_t)m * K + c) = t;   // "global_pool" tap
    }
    __syncthreads();
    if (n >= Nout) return;
    const float bias = b[n];
    for (int m = 0; m < M; ++m) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = j * 256 + lane * 4;
            if (k < K) acc += hd_dot4(*reinterpret_cast<const float4*>(xs + m * K + k), wv[j]);
        }
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) y[(size_t)m * Nout + n] = fmaxf(acc + bias, 0.f);
    }
}

// output_layer (no activation) + mesh.  Workgroup b: sample n = b / bps, cells 4 (b % bps) + wave, bps = ceil(cells / 4).
// Each wave takes the rows wave, wave + 4, ... of the layer: all their weight loads first, all the dot products, then all the
// wave reductions together (one row at a time, the 6 dependent cross-lane steps of a reduction were the kernel's critical path).
// Warm-up of the sampler's gather (round 4): this launch is a latency chain on 4 workgroups while the other 252 CUs idle, and the
// launch behind it -- the sampler -- gathers from a frame that was last touched 1.8 ms and ~300 MB of traffic ago (cold in L2
// and mostly in the Infinity Cache).  The workgroups >= mesh_blocks do nothing but READ that frame: block ids are dealt round-robin
// over the 8 XCDs and the sampler gives the workgroups of XCD x the x-th eighth of the rows (sn_xcd_band), so the prefetching
// workgroups of XCD x read band x (+ a halo of rows for the warp's displacement) -- the lines land in the L2 the gathers will hit.
struct HeadPrefetch { const float* src; int N, H, W, mesh_blocks; };
__device__ __forceinline__ void head_prefetch_rows(const HeadPrefetch& pf, float* sink) {
    const int id = (int)blockIdx.x, P = (int)gridDim.x - pf.mesh_blocks;      // P % 8 == 0 (launcher)
    const int xcd = id & 7, rank = (id - pf.mesh_blocks) >> 3, per_xcd = P >> 3;
    constexpr int HALO = 24;
    const int r0 = max(0, xcd * pf.H / 8 - HALO), r1 = min(pf.H, (xcd + 1) * pf.H / 8 + HALO);
    const long band4 = (long)(r1 - r0) * pf.W / 4;                            // float4s of the band (W % 4 == 0: launcher)
    const long lo = band4 * rank / per_xcd, hi = band4 * (rank + 1) / per_xcd;
    float acc = 0.f;
    for (int n = 0; n < pf.N; ++n) {
        const float4* p = reinterpret_cast<const float4*>(pf.src + ((size_t)n * pf.H + r0) * pf.W);
        for (long i = lo + threadIdx.x; i < hi; i += 256) { const float4 v = p[i]; acc += v.x + v.w; }
    }
    if (acc == 1.2345e38f) *sink = acc;                                       // (never: keeps the loads)
}

template <int K /* 512 */>
__global__ __launch_bounds__(256) void theta_mesh_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         int n_theta, float* __restrict__ theta, int gh, int gw, float lim,
                                                         float* __restrict__ Hs, int* __restrict__ head_adv, int depth, const HeadPrefetch pf) {
    constexpr int RMAX = 16;                                             // rows per wave: n_theta <= 64
    __shared__ __attribute__((aligned(16))) float xs[K];
    __shared__ float th[64];
    if ((int)blockIdx.x >= pf.mesh_blocks) { head_prefetch_rows(pf, th); return; }   // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cells = gh * gw, bps = (cells + 3) / 4;
    const int n = blockIdx.x / bps, cg = blockIdx.x - n * bps;
    float4 wv[RMAX][K / 256];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int row = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < K / 256; ++j)
            wv[i][j] = (row < n_theta) ? *reinterpret_cast<const float4*>(w + (size_t)row * K + j * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < K / 4; i += 256) *reinterpret_cast<float4*>(xs + 4 * i) = *reinterpret_cast<const float4*>(x + (size_t)n * K + 4 * i);
    __syncthreads();
    float acc[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < K / 256; ++j) acc[i] += hd_dot4(*reinterpret_cast<const float4*>(xs + j * 256 + lane * 4), wv[i][j]);
    }
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int i = 0; i < RMAX; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int row = wave + 4 * i;
        if (row < n_theta && lane == 0) {
            const float v = acc[i] + b[row];
            th[row] = v;
            if (cg == 0) theta[(size_t)n * n_theta + row] = v;
        }
    }
    __syncthreads();
    // online loop: nothing between the stack assembly and the sampler reads the ring head (see launch_sample_push)
    if (head_adv != nullptr && blockIdx.x == 0 && tid == 0) *head_adv = (*head_adv + 1) % depth;
    const int cell = cg * 4 + wave;
    if (Hs != nullptr && cell < cells) sn_mesh_cell(th, 1, gh, gw, lim, cell, lane, nullptr, Hs + (size_t)n * cells * 9, nullptr);
}

// 1 when the shortened head can run this shape (else the caller launches the layers one by one).
int head_fused_supported(int N, int C, const int* fc_dims /* [5]: C, 2048, 1024, 512, n_theta */) {
    static const int want = []() { const char* v = getenv("STABNET_HEAD_FUSED"); return v ? atoi(v) : 1; }();
    return want && N >= 1 && N <= 8 && C % 64 == 0 && C <= 2048 && fc_dims[0] == C && fc_dims[3] == 512 && fc_dims[4] >= 1 && fc_dims[4] <= 64;
}
int head_gap_chunks(int N, int HW) { return std::max(1, std::min(std::max(1, 8 / N), HW / 16)); }

// partial: N * head_gap_chunks * C floats of scratch.  Enqueues gap partial sums + fc_1.
int launch_gap_fc1(const float* x, const float* scale, const float* shift, int N, int HW, int C, float* partial, float* gap_out,
                   const float* w, const float* b, float* y, int Nout, hipStream_t st) {
    const int chunks = head_gap_chunks(N, HW);
    int rc = launch_gap_partial(x, scale, shift, N, HW, C, chunks, partial, st);
    if (rc) return rc;
    const size_t lds = (size_t)N * C * sizeof(float);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fc_gap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) {
            stabnet_set_error("fc_gap: hipFuncSetAttribute(64 KiB LDS) failed: %s", hipGetErrorString(e));
            return STABNET_ERR_LAUNCH;
        }
        configured = true;
    }
    fc_gap_kernel<<<cdiv(Nout, 4), 256, lds, st>>>(partial, chunks, HW, w, b, y, gap_out, N, C, Nout);
    SN_LAUNCH_CHECK("fc_gap_kernel");
    return STABNET_OK;
}

int launch_theta_mesh(const float* x, const float* w, const float* b, int N, int n_theta, float* theta, int gh, int gw, float lim,
                      float* Hs, int* head_adv, int depth, hipStream_t st, const float* prefetch_src, int pf_H, int pf_W) {
    SN_REQUIRE(n_theta <= 64 && gh * gw <= 64, "theta_mesh: bad shape");
    const int bps = Hs != nullptr ? (gh * gw + 3) / 4 : 1;
    static const int pf_blocks = []() { const char* v = getenv("STABNET_HEAD_PREFETCH_BLOCKS"); return v ? atoi(v) / 8 * 8 : 256; }();
    const bool pf_on = prefetch_src != nullptr && pf_blocks > 0 && pf_W % 4 == 0 && pf_H >= 8 && ((size_t)prefetch_src & 15) == 0;
    const HeadPrefetch pf{prefetch_src, N, pf_H, pf_W, N * bps};
    theta_mesh_kernel<512><<<N * bps + (pf_on ? pf_blocks : 0), 256, 0, st>>>(x, w, b, n_theta, theta, Hs != nullptr ? gh : 1, Hs != nullptr ? gw : 1,
                                                                             lim, Hs, head_adv, depth, pf);
    SN_LAUNCH_CHECK("theta_mesh_kernel");
    return STABNET_OK;
}
