// Ablation probe of the production conv kernel: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DCONV_ABLATE=<mask>
//   -I deep-online-video-stabilization_amd/csrc -o /tmp/conv_probe tools/conv_probe.hip
#include "conv_kernel.h"
#include <cstdio>
void stabnet_set_error(const char*, ...) {}
int main() {
    const int N = 1, H = 180, W = 320, Cin = 64, Cout = 64, KH = 3, pad = 1;
    ConvArgs a{};
    float *x, *w, *y;
    hipMalloc(&x, (size_t)N * H * W * Cin * 4); hipMalloc(&w, (size_t)Cout * KH * KH * Cin * 4); hipMalloc(&y, (size_t)N * H * W * Cout * 4);
    hipMemset(x, 0, (size_t)N * H * W * Cin * 4); hipMemset(w, 0, (size_t)Cout * KH * KH * Cin * 4);
    a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KH; a.stride = 1; a.pad = pad;
    a.up = 1; a.Ho = H; a.Wo = W; a.res_H = H; a.res_W = W; a.res_stride = 1; a.M = N * H * W; a.K = KH * KH * Cin; a.splitk = 1;
    a.steps_per_split = KH * KH * (Cin / 32);
    auto kern = conv_igemm_f32_kernel<64, 64, 32, 32, 32, 1>;
    const size_t lds = 2 * (64 + 64) * 36 * 4;
    dim3 grid((a.M + 63) / 64, 1, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) kern<<<grid, 256, lds>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("ABLATE=%d : %.1f us  %.1f TFLOP/s (nominal)\n", CONV_ABLATE, ms * 1e3, 2.0 * a.M * a.K * a.Cout / ms / 1e9);
    return 0;
}
