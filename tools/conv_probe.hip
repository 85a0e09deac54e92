// Ablation probe of the production conv kernel.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DCONV_ABLATE=<mask> -DPROBE_TILE=<0|2> -I deep-online-video-stabilization_amd/csrc
//         -o /tmp/conv_probe tools/conv_probe.hip
// mask bits: 1 no global loads, 2 no LDS stores, 4 no epilogue stores, 8 no MFMA
#define CONV_STAMP 1
#include "conv_kernel.h"
#include <cstdio>
#include <vector>
#include <algorithm>
void stabnet_set_error(const char*, ...) {}
#ifndef PROBE_TILE
#define PROBE_TILE 2
#endif
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1, H = argc > 2 ? atoi(argv[2]) : 180, W = argc > 3 ? atoi(argv[3]) : 320;
    const int Cin = argc > 4 ? atoi(argv[4]) : 64, Cout = argc > 5 ? atoi(argv[5]) : 64, KH = 3, pad = 1;
    ConvArgs a{};
    float *x, *w, *y;
    (void)hipMalloc(&x, (size_t)N * H * W * Cin * 4); (void)hipMalloc(&w, (size_t)Cout * KH * KH * Cin * 4); (void)hipMalloc(&y, (size_t)N * H * W * Cout * 4);
    (void)hipMemset(x, 0, (size_t)N * H * W * Cin * 4); (void)hipMemset(w, 0, (size_t)Cout * KH * KH * Cin * 4);
    a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KH; a.stride = 1; a.pad = pad;
    a.up = 1; a.Ho = H; a.Wo = W; a.res_H = H; a.res_W = W; a.res_stride = 1; a.M = N * H * W; a.K = KH * KH * Cin; a.splitk = 1;
    a.x_ld = a.Cin; a.res_ld = a.Cout;
    sn_fastdiv_make((unsigned)(a.Ho * a.Wo), a.div_hw_mul, a.div_hw_shift); sn_fastdiv_make((unsigned)a.Wo, a.div_w_mul, a.div_w_shift);
    a.steps_per_split = KH * KH * (Cin / 32);
#if PROBE_TILE == 0
    auto kern = conv_igemm_f32_kernel<128, 128, 32, 64, 64, 1, 2>;
    const int BM = 128, BN = 128;
#else
    auto kern = conv_igemm_f32_kernel<64, 64, 32, 32, 32, 1, 2>;
    const int BM = 64, BN = 64;
#endif
    const size_t lds = 2 * (size_t)(BM + BN) * 36 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((a.M + BM - 1) / BM, (Cout + BN - 1) / BN, 1);
    unsigned long long* stamps; const size_t nblk = (size_t)grid.x * grid.y;
    (void)hipMalloc(&stamps, nblk * 4 * 8); (void)hipMemset(stamps, 0, nblk * 4 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(a);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) kern<<<grid, 256, lds>>>(a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("tile %dx%d blocks=%d ABLATE=%d : %.1f us  %.1f TFLOP/s (nominal)\n", BM, BN, grid.x * grid.y, CONV_ABLATE, ms * 1e3,
           2.0 * a.M * a.K * a.Cout / ms / 1e9);
    {
        std::vector<unsigned long long> h(nblk * 4);
        (void)hipMemcpy(h.data(), stamps, nblk * 4 * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0; double pro = 0, loop = 0, epi = 0;
        for (size_t b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b*4]); t3 = std::max(t3, h[b*4+3]); pro += h[b*4+1]-h[b*4]; loop += h[b*4+2]-h[b*4+1]; epi += h[b*4+3]-h[b*4+2]; }
        double last_start = 0; for (size_t b = 0; b < nblk; ++b) last_start = std::max(last_start, (double)(h[b*4]-t0));
        printf("  stamps (100 MHz ticks? shader clk): kernel span %.0f, avg per block: prologue %.0f, loop %.0f, epilogue %.0f; last block start at +%.0f\n", (double)(t3-t0), pro/nblk, loop/nblk, epi/nblk, last_start);
    }
    return 0;
}
