// Launchers of the warp kernels (warp.hip) for use by the fused deploy path.
#pragma once
#include "common.h"

int launch_mesh(const float* in, int is_theta, int N, int gh, int gw, float lim, float* pts2, float* Hs, hipStream_t st,
                float* pts1 = nullptr);
// feedback of the online loop fused behind the sampler (warp_sample_kernel<4, 1>)
struct WarpPush {
    float* frames;        // ring [S][depth][H*W]: slot `*head` receives frame = img - black
    float* masks;         // ring [S][depth][H*W]: slot `*head` receives black
    float* frame_out;     // [S][H*W]
    int* all_black;       // [S][H*W] int32 += round(black), or null
    const int* head;      // device: slot of this frame's push
    int* head_rw;         // same word, advanced by the last block
    int* ticket;          // device counter, zero between launches
    int depth;
};
int launch_sample_push(const float* Hs, const float* src, int N, int H, int W, int gh, int gw, float* out, float* black,
                       float* x_map, float* y_map, const WarpPush& push, hipStream_t st);
int launch_sample(const float* Hs, const float* src, int N, int H, int W, int C, int gh, int gw, float* out,
                  float* black, float* x_map, float* y_map, hipStream_t st);
int check_warp_args(int N, int H, int W, int C, int gh, int gw);
