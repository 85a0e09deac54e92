// Launchers of the training-mode layers (train_layers.hip) and of the convolution backward (conv_bwd.hip).
#pragma once
#include "common.h"

struct Prof;

size_t col_reduce_workspace_floats(long M, int C);
size_t col_reduce_workspace_floats(long M, int C, int groups);
// per-tower operands of the grouped (both siamese towers in one launch) reduction / BN kernels
struct ColGroups { const float* x[2]; const float* g[2]; const float* scale[2]; const float* shift[2]; const float* mean[2]; const float* invstd[2]; };
struct BnStatOut { float* scale[2]; float* shift[2]; float* save_mean[2]; float* save_invstd[2]; };
struct BnBwdFin { const float* invstd[2]; float* coef[2]; };
struct BnApply { const float* x[2]; const float* g[2]; const float* scale[2]; const float* shift[2]; const float* mean[2];
                 const float* invstd[2]; const float* coef[2]; const float* addend[2]; float* d_x[2]; };
int launch_bn_stats_g(int groups, const float* const* x, long M, int C, const float* gamma, const float* beta, float eps,
                      float decay, float* const* scale, float* const* shift, float* const* save_mean, float* const* save_invstd,
                      float* mov_mean, float* mov_var, float* partial, hipStream_t st);
int launch_bn_relu_bwd_g(int groups, const float* const* x, const float* const* g, const float* const* scale,
                         const float* const* shift, const float* const* mean, const float* const* invstd, const float* gamma,
                         long M, int C, const float* const* addend, int add_stride, int H, int W, float* d_gamma, float* d_beta,
                         float* const* d_x, float* partial, float* const* coef, hipStream_t st);
int launch_bias_grad_g(int groups, const float* const* g, long M, int C, float* d_bias, float* partial, hipStream_t st,
                       float* d_bias2 = nullptr);      // d_bias2: a second bias fed by the same gradient (+= the same sums)
int launch_bn_stats(const float* x, long M, int C, const float* gamma, const float* beta, float eps, float decay,
                    float* scale, float* shift, float* save_mean, float* save_invstd, float* mov_mean, float* mov_var,
                    float* partial, hipStream_t st);
int launch_bn_relu_bwd(const float* x, const float* g, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* gamma, long M, int C, const float* addend, int add_stride, int H,
                       int W, float* d_gamma, float* d_beta, float* d_x, float* partial, float* coef, hipStream_t st);
int launch_bias_grad(const float* g, long M, int C, float* d_bias, float* partial, hipStream_t st);
int launch_max_pool_bwd(const unsigned char* argmax, const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                        int k, int stride, int pt, int pl, hipStream_t st);
int launch_max_pool_argmax(const float* x, float* y, unsigned char* argmax, int N, int H, int W, int C, int Ho, int Wo, int k,
                           int stride, int pt, int pl, hipStream_t st);
int launch_gap_bwd(const float* dg, int N, int HW, int C, float* da, hipStream_t st);
int launch_fc_bwd(const float* x, const float* w, const float* y, const float* dy, int M, int K, int Nout, int relu,
                  float* dW, float* db, float* dx, float* scratch, size_t scratch_floats, hipStream_t st);
int launch_weight_decay(const float* params, float* grads, const long* seg_off, const long* seg_len, const float* seg_coef,
                        int nseg, float gscale, float* loss_out, float* partial, hipStream_t st);
int launch_adam(float* w, const float* g, const float* g2, float* m, float* v, long n, float lr_t, float b1, float b2,
                float eps, float gscale, hipStream_t st);

// wgrad: per-split partial slabs reduced in split order by wgrad_reduce_flush (conv_bwd.hip)
#define WGRAD_TABLE_MAX 64
struct WgradReduceEntry { long dw_off, elems; const float* slab[2]; int splits; };   // slab[t]: tower t's [splits][elems] partials (or null)
struct WgradReduceTable { WgradReduceEntry e[WGRAD_TABLE_MAX]; long prefix[WGRAD_TABLE_MAX + 1]; int n; };
int wgrad_splits(int Cout, int K, int M, int* rows_per_split);
// A reduce-table destination is `dw_off` floats from the table's base address.  Most lie inside the caller's gradient buffer; the
// row-run stem's scratch gradient and a stand-alone bias (stabnet_conv2d_wgrad_bias) are other allocations, so the distance is
// taken between ADDRESSES and applied to the address, never as pointer arithmetic across objects.
inline long sn_float_distance(const float* base, const float* p) {
    return (long)(((intptr_t)reinterpret_cast<uintptr_t>(p) - (intptr_t)reinterpret_cast<uintptr_t>(base)) / (intptr_t)sizeof(float));
}
#ifdef __HIPCC__
__host__ __device__
#endif
inline float* sn_float_at(float* base, long floats) {
    return reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(base) + (uintptr_t)(floats * (long)sizeof(float)));
}
size_t wgrad_slab_floats(int Cout, int K, int M, int T, int with_bias = 0);
int wgrad_reduce_flush(float* grads, WgradReduceTable& t, hipStream_t st);
// T towers (1 or 2) in ONE launch: grid.z = T * splits, slabs [tower][split]; the reduce adds them in that order into dw
int wgrad_launch_g(int T, const float* const* x, const float* const* dy, float* dw_base, long dw_off, const float* const* in_scale,
                   const float* const* in_shift, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   float* slab_base, size_t* slab_cursor, size_t slab_capacity, WgradReduceTable* table, hipStream_t st, Prof* prof,
                   int rowrun = 0, long bias_off = -1, long bias_off2 = -1);
// bias_off >= 0 (only where wgrad_bias_fusable(): 1x1 layers on the stride-1 kernel, Cout % 256 == 0): the launch also leaves the
// column sums of dy -- the gradient of the layer's bias -- at dw_base + bias_off (and + bias_off2: a second bias with the same
// gradient, the projection shortcut's), through the same slab reduction.
int wgrad_bias_fusable(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
// rowrun = 1 (the 13-channel stem): x is the zero-bordered tight image [N][H + 2 pad][W + 2 pad][Cin] (+ slack) of ConvArgs::rowrun and
// the gradient comes out as [Cout][KH][roundup(KW * Cin, 32)]; launch_wgrad_rowrun_scatter adds it into OHWI [Cout][KH][KW][CinPad]
int launch_wgrad_rowrun_scatter(const float* tmp, float* dw, int Cout, int KH, int KW, int Cin, int CinPad, hipStream_t st);
int wgrad_launch(const float* x, const float* dy, float* dw_base, long dw_off, const float* in_scale, const float* in_shift,
                 int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* slab_base,
                 size_t* slab_cursor, size_t slab_capacity, WgradReduceTable* table, hipStream_t st, Prof* prof);
// kperm: tap rows stored in the order (1, 0, 2) -- what the dgrad kernel expects for 3x3 filters of stride-2 layers (it skips the
// tap rows that only meet the zeros of the dilated dY); dgrad_kperm() says when.
inline int dgrad_kperm(int KH, int KW, int stride) { return (KH == 3 && KW == 3 && stride == 2) ? 1 : 0; }
int pack_dgrad_weights(const float* w, float* wt, int Cout, int KH, int KW, int Cin, int kperm, hipStream_t st);
// every dgrad weight tensor of a net, re-packed by one launch: layer i = params[w_off ...] OHWI [Cout][K][K][Cin] ->
// wt[prefix[i] ...] as [Cin][K][K][Cout] with both filter axes flipped
struct PackDesc { long w_off; int Cout, K, Cin, kperm; };
struct PackTable { PackDesc d[56]; long prefix[57]; int n; };
int pack_dgrad_weights_all(const float* params, float* wt, const PackTable& t, hipStream_t st);
int dgrad_launch(const float* dy, const float* wt, float* dx, const float* residual, int N, int H, int W, int Cin,
                 int Cout, int KH, int KW, int stride, int pad, float* splitk_ws, size_t splitk_bytes, hipStream_t st,
                 Prof* prof, const float* wt_img = nullptr /* pre-split image of wt (conv.h): the packed split kernels */);
