// Launchers of the warp kernels (warp.hip) for use by the fused deploy path.
#pragma once
#include "common.h"

// head_adv (device int*, optional): the online loop's ring head, advanced to (head + 1) % depth by one thread of the mesh
// kernel -- see launch_sample_push.
int launch_mesh(const float* in, int is_theta, int N, int gh, int gw, float lim, float* pts2, float* Hs, hipStream_t st,
                float* pts1 = nullptr, int* head_adv = nullptr, int depth = 1);
// feedback of the online loop fused behind the sampler (warp_sample_kernel<1>)
struct WarpPush {
    float* frames;        // ring [S][depth][H*W]: slot (*head - 1) mod depth receives frame = img - black
    float* masks;         // ring [S][depth][H*W]: the same slot receives black
    float* frame_out;     // [S][H*W]
    int* all_black;       // [S][H*W] int32 += round(black), or null
    const int* head;      // device: the ring head, ALREADY advanced past this frame's slot (launch_mesh's head_adv)
    int depth;
};
int launch_sample_push(const float* Hs, const float* src, int N, int H, int W, int gh, int gw, float* out, float* black,
                       float* x_map, float* y_map, const WarpPush& push, hipStream_t st);
int launch_sample(const float* Hs, const float* src, int N, int H, int W, int C, int gh, int gw, float* out,
                  float* black, float* x_map, float* y_map, hipStream_t st);
int check_warp_args(int N, int H, int W, int C, int gh, int gw);
