// Launchers of the non-convolution layers (layers.hip).
#pragma once
#include "common.h"

int launch_merge_vectors(const float* b_sc, const float* scale1, const float* shift1, int depth, int dbn, float* out,
                         hipStream_t st);
int launch_stem_repack(const float* w, float* out, int Cout, int KH, int KW, int CinPad, int Cin, hipStream_t st);
int launch_pad_channels(const float* x, float* y, long npix, int C, int Cp, hipStream_t st);
// zero8 (optional): 8 words zeroed by one thread (the phase counters of the fused head, which runs later in the same forward)
int launch_max_pool(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pt,
                    int pl, const float* scale, const float* shift, hipStream_t st, unsigned* zero8 = nullptr);
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int G,
                   float* scale, float* shift, hipStream_t st);
int gap_chunks(int HW);   // partial buffer = N * gap_chunks(HW) * C floats
int launch_gap_bn_relu(const float* x, const float* scale, const float* shift, int N, int HW, int C, float* out,
                       float* partial, hipStream_t st);
int launch_fc(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
              hipStream_t st);

// The regressor's head as one launch (head_chain.hip): postnorm BN + ReLU + reduce_mean -> fc_1..3 -> output_layer
// (-> mesh homographies + ring-head advance when Hs != nullptr).
struct HeadChainArgs {
    const float* x;                    // last conv3 output [N][HW][C]
    const float *scale, *shift;        // folded postnorm BN [C]
    int N, HW, C, chunks, rows_per_chunk;   // (chunks / rows_per_chunk are filled by the launcher)
    const float* w[4];                 // fc_1, fc_2, fc_3, output_layer weights [out][in]
    const float* b[4];
    int n_theta;
    float* partial;                    // [N][chunks <= 8][C] scratch
    float* gap;                        // [N][C]      ("global_pool" tap)
    float* fc[3];                      // [N][2048], [N][1024], [N][512]
    float* theta;                      // [N][n_theta]
    int gh, gw;                        // mesh (Hs == nullptr: theta only)
    float lim;
    float* Hs;                         // [N][gh*gw][9]
    int* head_adv;                     // online loop: ring head advanced by one thread (or null)
    int depth;
    unsigned* sync;                    // 8 words, ZERO when the kernel starts: [0..3] phase counters, [7] timeout flag
};
int head_chain_supported(int N, int HW, int C, const int* fc_dims, int gh, int gw);
int head_chain_chunks(int N, int HW);
int launch_head_chain(HeadChainArgs a, hipStream_t st);
