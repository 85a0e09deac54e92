// Ablation / timeline probe of the LDS-DMA ring conv kernel.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DRING_ABLATE=<mask> -I deep-online-video-stabilization_amd/csrc -o /tmp/ring_probe tools/ring_probe.hip
// mask bits: 1 no DMA, 2 no barrier, 4 no ds_read, 8 no MFMA.   args: H W Cin Cout KH [workgroups]   (stride 1, SAME padding)
#include "conv_ring_kernel.h"
#ifndef PROBE_BF16
#define PROBE_BF16 0          // 4: the packed split kernel (w is then a weight image: 1.5 x the floats, zeros here)
#endif
#include <cstdio>
#include <vector>
#include <algorithm>
void stabnet_set_error(const char*, ...) {}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 1, W = argc > 2 ? atoi(argv[2]) : 16384, Cin = argc > 3 ? atoi(argv[3]) : 1152;
    const int N = argc > 4 ? atoi(argv[4]) : 128, KH = argc > 5 ? atoi(argv[5]) : 1;
    const bool stem = argc > 7 && atoi(argv[7]) == 1;       // row-run stem geometry: stride 2, pad KH/2, K = KH*roundup(KH*Cin,32)
    const int st_ = stem ? 2 : 1;
    const int Ho = (H + 2 * (KH / 2) - KH) / st_ + 1, Wo = (W + 2 * (KH / 2) - KH) / st_ + 1;
    const int M = Ho * Wo, K = stem ? KH * ((KH * Cin + 31) / 32 * 32) : KH * KH * Cin;
    ConvArgs a{};
    float *x, *w, *y;
    (void)hipMalloc(&x, (size_t)(H + 8) * (W + 8) * Cin * 4 + 1024); (void)hipMalloc(&w, (size_t)(N + 64) * K * 6 + 4096); (void)hipMalloc(&y, (size_t)M * N * 4);
    (void)hipMemset(x, 0, (size_t)(H + 8) * (W + 8) * Cin * 4 + 1024); (void)hipMemset(w, 0, (size_t)(N + 64) * K * 6 + 4096);
    a.x = x; a.w = w; a.y = y; a.N = 1; a.H = H; a.W = W; a.Cin = Cin; a.Cout = N; a.KH = KH; a.KW = KH; a.stride = st_; a.pad = KH / 2; a.rowrun = stem ? 1 : 0;
    a.up = 1; a.Ho = Ho; a.Wo = Wo; a.res_H = Ho; a.res_W = Wo; a.res_stride = 1; a.M = M; a.K = K; a.splitk = 1;
    a.x_ld = a.Cin; a.res_ld = a.Cout; a.xcd_swizzle = getenv("NOSWZ") ? 0 : 1;
    sn_fastdiv_make((unsigned)(a.Ho * a.Wo), a.div_hw_mul, a.div_hw_shift); sn_fastdiv_make((unsigned)a.Wo, a.div_w_mul, a.div_w_shift);
    a.steps_per_split = K / 32;
    dim3 grid((M + 63) / 64, (N + 63) / 64, 1);
    const size_t nblk = (size_t)grid.x * grid.y;
    const int wgs = argc > 6 ? atoi(argv[6]) : 768;
    const int g1 = (int)std::min<size_t>(nblk, (size_t)wgs);
    auto launch = [&]() {
        if (stem) conv_ring_f32_kernel<2, PROBE_BF16><<<g1, 256>>>(a);
        else if (KH == 1) conv_ring_f32_kernel<0, PROBE_BF16><<<g1, 256>>>(a);
        else conv_ring_f32_kernel<1, PROBE_BF16><<<g1, 256>>>(a);
    };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("BF16=%d ABLATE=%2d M=%d K=%d N=%d blocks=%zu (%.2f/CU): %.1f us  %.1f TFLOP/s nominal\n", PROBE_BF16, RING_ABLATE, M, K, N, nblk, nblk / 256.0,
           ms * 1e3, 2.0 * M * K * N / ms / 1e9);
#if RING_STAMP
    {   // timeline of the LAST launch: per workgroup (wave 0) s_memtime at entry / first stage landed / before the epilogue of
        // its last tile / after its stores drained; s_memrealtime (100 MHz) at entry and exit for the clock
        std::vector<unsigned long long> st(4096 * 8);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_ring_stamps), st.size() * 8);
        const int n = std::min(g1, 4096);
        unsigned long long t0min = ~0ull, t3max = 0, r0min = ~0ull, r1max = 0;
        for (int b = 0; b < n; ++b) { t0min = std::min(t0min, st[b * 8]); t3max = std::max(t3max, st[b * 8 + 3]); r0min = std::min(r0min, st[b * 8 + 4]); r1max = std::max(r1max, st[b * 8 + 5]); }
        const double span_us = (double)(r1max - r0min) / 100.0, ghz = 0.1;   // stamps are s_memrealtime ticks (100 MHz)
        auto med = [&](int i0, int i1) { std::vector<double> v; for (int b = 0; b < n; ++b) v.push_back((double)(st[b * 8 + i1] - st[b * 8 + i0]) / ghz / 1e3); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [&](int i0, int i1) { double m = 0; for (int b = 0; b < n; ++b) m = std::max(m, (double)(st[b * 8 + i1] - st[b * 8 + i0]) / ghz / 1e3); return m; };
        std::vector<double> start; for (int b = 0; b < n; ++b) start.push_back((double)(st[b * 8] - t0min) / ghz / 1e3); std::sort(start.begin(), start.end());
        printf("  in-kernel span %.1f us (%.1f) | WG start skew median %.2f max %.2f us | entry->first stage landed med %.2f max %.2f | K loop med %.2f max %.2f | epilogue+store drain med %.2f max %.2f us\n",
               span_us, ghz, start[n / 2], start[n - 1], med(0, 1), mx(0, 1), med(1, 2), mx(1, 2), med(2, 3), mx(2, 3));
    }
#endif
    return 0;
}
