// Launchers of the non-convolution layers (layers.hip).
#pragma once
#include "common.h"

int launch_merge_vectors(const float* b_sc, const float* scale1, const float* shift1, int depth, int dbn, float* out,
                         hipStream_t st);
int launch_stem_repack(const float* w, float* out, int Cout, int KH, int KW, int CinPad, int Cin, hipStream_t st);
int launch_pad_channels(const float* x, float* y, long npix, int C, int Cp, hipStream_t st);
int launch_max_pool(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pt,
                    int pl, const float* scale, const float* shift, hipStream_t st);
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int G,
                   float* scale, float* shift, hipStream_t st);
int gap_chunks(int HW);   // partial buffer = N * gap_chunks(HW) * C floats
int launch_gap_bn_relu(const float* x, const float* scale, const float* shift, int N, int HW, int C, float* out,
                       float* partial, hipStream_t st);
int launch_fc(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
              hipStream_t st);

// The shortened inference head (head.hip): reduce_mean partials + fc_1 (finalize folded into its input staging); output_layer +
// mesh homographies (+ ring-head advance) as one launch.
int launch_gap_partial(const float* x, const float* scale, const float* shift, int N, int HW, int C, int chunks, float* partial,
                       hipStream_t st);
int head_fused_supported(int N, int C, const int* fc_dims);
int head_gap_chunks(int N, int HW);
int launch_gap_fc1(const float* x, const float* scale, const float* shift, int N, int HW, int C, float* partial, float* gap_out,
                   const float* w, const float* b, float* y, int Nout, hipStream_t st);
int launch_theta_mesh(const float* x, const float* w, const float* b, int N, int n_theta, float* theta, int gh, int gw, float lim,
                      float* Hs, int* head_adv, int depth, hipStream_t st, const float* prefetch_src = nullptr, int pf_H = 0, int pf_W = 0);
