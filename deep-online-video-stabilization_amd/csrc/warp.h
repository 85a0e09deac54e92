// Launchers of the warp kernels (warp.hip) for use by the fused deploy path.
#pragma once
#include "common.h"

int launch_mesh(const float* in, int is_theta, int N, int gh, int gw, float lim, float* pts2, float* Hs, hipStream_t st);
int launch_sample(const float* Hs, const float* src, int N, int H, int W, int C, int gh, int gw, float* out,
                  float* black, float* x_map, float* y_map, hipStream_t st);
int check_warp_args(int N, int H, int W, int C, int gh, int gw);
