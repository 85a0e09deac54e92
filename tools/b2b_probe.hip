// Ablation probe of conv_b2b_f32_kernel (conv2 3x3 -> conv3 1x1 as one launch), stand-alone launches on zero data.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Wno-inline-asm [-DB2B_ABLATE=<mask>] -I deep-online-video-stabilization_amd/csrc -o /tmp/b2b_probe tools/b2b_probe.hip
//   args: H W C [stride] [workgroups per CU]      C = 64 -> Cout 256 (block 1), C = 128 -> Cout 512 (block 2)
// ablate bits: 1 chunk epilogue without global loads / stores, 2 no chunk epilogue, 4 no park, 8 no MFMA
#include "conv_b2b_kernel.h"
#include <cstdio>
#include <vector>
#include <algorithm>
void stabnet_set_error(const char*, ...) {}
static void plan(ConvArgs& a) {
    a.M = a.N * a.Ho * a.Wo; a.K = a.KH * a.KW * a.Cin;
    if (!a.x_ld) a.x_ld = a.Cin;
    if (!a.res_ld) a.res_ld = a.Cout;
    sn_fastdiv_make((unsigned)(a.Ho * a.Wo), a.div_hw_mul, a.div_hw_shift); sn_fastdiv_make((unsigned)a.Wo, a.div_w_mul, a.div_w_shift);
    a.splitk = 1; a.steps_per_split = a.K / 32; a.up = 1; a.xcd_swizzle = getenv("NOSWZ") ? 0 : 1;
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 180, W = argc > 2 ? atoi(argv[2]) : 320, C = argc > 3 ? atoi(argv[3]) : 64;
    const int stride = argc > 4 ? atoi(argv[4]) : 1, wgs_cu = argc > 5 ? atoi(argv[5]) : (C == 64 ? 2 : 1);
    const int Cout = 4 * C, Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    B2bArgs P{};
    float *x, *w2, *w3, *y, *res, *vec;
    (void)hipMalloc(&x, (size_t)H * W * C * 4); (void)hipMalloc(&w2, (size_t)C * 9 * C * 4); (void)hipMalloc(&w3, (size_t)Cout * C * 4);
    (void)hipMalloc(&y, (size_t)Ho * Wo * Cout * 4); (void)hipMalloc(&res, (size_t)Ho * Wo * Cout * 4); (void)hipMalloc(&vec, 4096 * 4);
    (void)hipMemset(x, 0, (size_t)H * W * C * 4); (void)hipMemset(w2, 0, (size_t)C * 9 * C * 4); (void)hipMemset(w3, 0, (size_t)Cout * C * 4);
    (void)hipMemset(res, 0, (size_t)Ho * Wo * Cout * 4); (void)hipMemset(vec, 0, 4096 * 4);
    ConvArgs& a = P.c2; ConvArgs& b = P.c3;
    a.x = x; a.w = w2; a.N = 1; a.H = H; a.W = W; a.Cin = C; a.Cout = C; a.KH = a.KW = 3; a.stride = stride; a.pad = 1; a.Ho = Ho; a.Wo = Wo;
    a.res_H = Ho; a.res_W = Wo; a.res_stride = 1; a.out_scale = vec; a.out_shift = vec + 1024; a.relu_out = 1;
    plan(a);
    b.w = w3; b.y = y; b.bias = vec + 2048; b.residual = res; b.N = 1; b.H = Ho; b.W = Wo; b.Cin = C; b.Cout = Cout; b.KH = b.KW = 1; b.stride = 1; b.pad = 0;
    b.Ho = Ho; b.Wo = Wo; b.res_H = Ho; b.res_W = Wo; b.res_stride = 1;
    plan(b);
    const int tiles = (a.M + 63) / 64, grid = std::min(tiles, wgs_cu * 256);
    auto launch = [&]() {
        if (C == 64) conv_b2b_f32_kernel<2><<<grid, 256>>>(P);
        else conv_b2b_f32_kernel<4><<<grid, 512>>>(P);
    };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double fl = 2.0 * a.M * ((double)a.K * C + (double)C * Cout);
    printf("ABLATE=%2d  %dx%d C=%d s=%d  M=%d tiles=%d grid=%d (%.2f tiles/CU): %.1f us  %.1f TFLOP/s nominal\n", B2B_ABLATE, H, W, C, stride, a.M, tiles,
           grid, tiles / 256.0, ms * 1e3, fl / ms / 1e9);
    return 0;
}
