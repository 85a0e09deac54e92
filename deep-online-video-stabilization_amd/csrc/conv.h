// Internal launcher interface of the implicit-GEMM convolution kernels (conv.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    // tensors
    const float* x;          // input  NHWC [N,H,W,Cin]
    const float* w;          // weights OHWI [Cout][KH][KW][Cin]   (row n = one output channel, K contiguous)
    float* y;                // output NHWC [N,Ho,Wo,Cout]
    const float* bias;       // [Cout] or null
    const float* in_scale;   // [Cin] or null: A-operand prologue  a = relu(a*scale + shift)  (folded BN + ReLU)
    const float* in_shift;   // [Cin]
    const float* residual;   // NHWC [N,res_H,res_W,Cout] or null; read at (oy*res_stride, ox*res_stride)
    const float* out_scale;  // [Cout] or null: epilogue  y = y*out_scale + out_shift  (folded BN of the CONSUMER, inference)
    const float* out_shift;
    const float* out_floor;  // [Cout] or null: with out_scale, y = max(y, out_floor[n]) instead of the relu_out flag (0 = ReLU,
                             // -inf = none): lets ONE launch produce channels with and without an activation
    float* partial;          // split-K workspace [splitk][M][Cout] (only when splitk > 1)
    // geometry
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int up;                  // input dilation ("fractional stride") used by dgrad of strided convs; 1 otherwise
    int res_H, res_W, res_stride;
    int relu_out;
    int x_ld, res_ld;        // floats between consecutive pixels of x / of the residual (0 = Cin / Cout: dense tensors); the
                             // inference plan keeps two tensors side by side in one buffer (shortcut | conv1 of a projection unit)
    int M, K;                // M = N*Ho*Wo, K = KH*KW*Cin
    int splitk, steps_per_split;
    int cin_real;            // un-padded Cin (algorithmic flop accounting only); 0 = Cin
    unsigned div_hw_mul, div_hw_shift, div_w_mul, div_w_shift;   // exact m / (Ho*Wo) and r / Wo by multiply-high (conv_plan)
    int rowrun;              // 1: "row-run" A operand (ring kernel MODE 2, the 13-channel stem): for every filter row kh the KW*Cin
                             //    input floats of a tap row are contiguous; they are taken as ONE run, zero-padded to a multiple of 32,
                             //    so K = KH * roundup(KW*Cin, 32) and the weights are [Cout][KH][roundup(KW*Cin, 32)].  Cin is the
                             //    real (unpadded) pixel stride; no prologue.
    int in_scale_expected;   // plan-time hint: 1 if the launch will carry an input BN prologue (in_scale is bound later)
    int xcd_swizzle;         // 1: remap workgroup ids so each XCD (own 4 MiB L2) works on a contiguous run of M tiles
};

// Chooses tile shape / split-K and returns the workspace bytes the launch needs (0 if none).
size_t conv_plan(ConvArgs& a);
// Enqueues the convolution (and the split-K reduction when a.splitk > 1).
struct Prof;
// bf16_operands = 1: SECONDARY fast mode: A and B fragments are rounded to bf16 when they are read from
// LDS and multiplied by v_mfma_f32_32x32x16_bf16 (fp32 accumulate; tensors stay fp32 in memory).  Never the default: the
// reference is fp32 end to end.  (A launch argument, not a ConvArgs field: the fp32 kernels' argument block -- and with it their
// register allocation -- stays exactly what it was; an extra field cost the fp32 path 1 %.)
// bf16_operands = 2 / 3: split operands -- exact f32 products as six / nine bf16 MFMAs, both operands split when the fragments
// are read (conv_kernel.h, sn_split3).  bf16_operands = 4 with w_img: the weights come as a pre-split fragment-major image
// (conv_weight_image_floats / launch_weight_split_image), only the A fragments are split at run time; launches the packed ring
// kernel cannot take (register-staged path, in-workgroup split-K) run the exact f32 MFMA kernels on a.w as before.
int conv_launch(const ConvArgs& a, hipStream_t st, Prof* prof = nullptr, int bf16_operands = 0, const float* w_img = nullptr);
// Pre-split weight image of a convolution whose weights are [Cout][K] rows (K % 32 == 0): per (64-channel N tile, 32-deep K step)
// 3072 floats = [wave column 2][plane h, m, l][k group 2][lane 64][8 bf16]; lane (n = lane & 31, g = lane >> 5), element e holds
// k = 32 step + 16 group + 8 (e >> 2) + 4 g + (e & 3) -- the k order of the ring kernel's A fragments.
int conv_packed_variant();              // 4 or 5: the BF16 template value the packed launches run (conv.hip)
size_t conv_weight_image_floats(int Cout, int K);
int launch_weight_split_image(const float* w, int Cout, int K, float* img, hipStream_t st);
// The images of several weight matrices by ONE launch (the training step re-splits its dgrad weights once per step): matrix i is
// w_base[w_off[i] ...] as [Cout[i]][K[i]], its image goes to img_base[img_off[i] ...]; tprefix = running thread counts.
struct WeightImageTable { long w_off[64], img_off[64], tprefix[65]; int Cout[64], K[64]; int n; };
void weight_image_table_add(WeightImageTable& t, long w_off, long img_off, int Cout, int K);
int launch_weight_split_images(const float* w_base, float* img_base, const WeightImageTable& t, hipStream_t st);
// 1 if that launch is followed by a split-K reduce launch (0: no split, or the split runs inside the workgroups)
int conv_reduce_launches(const ConvArgs& a, int operand_mode = 0);      // (operand_mode 4: as conv_launch() with a weight image decides)

// The two siamese towers of a training step as ONE launch (forward convolutions, conv_kernel.h).  `a` describes the pair as one
// batch of 2N samples whose first N live where a.x / a.y / a.residual / a.in_scale / a.in_shift point; tiles of rows >= m_tower
// (the second tower: its tensors sit in another workspace, and its batch statistics differ) get these ELEMENT offsets added to
// the five pointers.  m_tower must be a multiple of the tile height (64).  conv_pair_supported(): the plan of `a` is a 64x64
// register-staged launch (anything else: launch the towers one after the other).
struct ConvPair { int m_tower, x_tower_floats /* N*H*W*Cin of one tower's input */; long dx, dy, dres, dscale; };
bool conv_pair_supported(const ConvArgs& a);
int conv_launch_pair(const ConvArgs& a, const ConvPair& pr, hipStream_t st, Prof* prof = nullptr,
                     const float* w_img = nullptr /* pre-split image of a.w: the prologue-carrying 1x1 pairs run the packed split kernel */);

// A bottleneck unit's 3x3 `conv2` and 1x1 `conv3` as ONE launch (conv_b2b_kernel.h): c2 / c3 are the two planned convolutions as
// conv_launch() would take them (c2.out_scale / out_shift = the folded BN between them, ReLU implied; c2.y is not written).
// conv_b2b_supported(): the geometry the kernel takes (3x3 / pad 1 over Cin == Cout == 64 or 128, no bias / residual / prologue;
// 1x1 stride-1 c3 over the same pixels with Cout a multiple of c2.Cout, no prologue) -- pointers are not looked at.
bool conv_b2b_supported(const ConvArgs& c2, const ConvArgs& c3);
int conv_b2b_launch(const ConvArgs& c2, const ConvArgs& c3, hipStream_t st, Prof* prof = nullptr);

// Exact unsigned division by an invariant divisor d >= 1 for numerators n < 2^31 (Granlund-Montgomery round-up form):
//   l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1, n / d = (mulhi(mul, n) + n) >> l.
// Three instructions instead of the ~40 of a software 32-bit division; the kernels' per-tile index setup is VALU work
// that competes with the MFMAs for issue slots.
static inline void sn_fastdiv_make(unsigned d, unsigned& mul, unsigned& shift) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shift = l;
}
#ifdef __HIPCC__
__device__ __forceinline__ int sn_fastdiv(int n, unsigned mul, unsigned shift) {
    return (int)((__umulhi(mul, (unsigned)n) + (unsigned)n) >> shift);
}
#endif
