// Implicit-GEMM convolution on the gfx950 matrix cores, exact float32 (v_mfma_f32_32x32x2_f32).
//
//   y[m][n] = sum_k A(m,k) * Wt[n][k]      m = (img, oy, ox)   k = (kh, kw, c)   n = output channel
//   A(m,k)  = act( x[img, oy*stride-pad+kh, ox*stride-pad+kw, c] )   act = optional folded BN + ReLU (prologue)
//
// Replaces the slim conv2d / conv2d_same (+ batch_norm + relu) clusters of resnet_v2_50 called at
// s_net_bundle_nobm.py:252-253 (SURVEY.md section 2.1 rows K1,K3,K4,K5).
//
// Tiling: 256 threads = 4 waves (64 lanes).  Block tile BM x BN, K-step BK; both operands are staged in LDS with
// the reduction index contiguous ([row][BK+4] floats, rows 16-B aligned, pitch 36/20 dwords = conflict-free
// ds_read_b128 for the 16-lane service groups).  Lane (i = lane&31, h = lane>>5) of a wave reads FOUR consecutive
// k (one ds_read_b128) of row i at k-offset 8*kk + 4*h and feeds them to four successive MFMAs: the MFMA's two
// k-slots are thus k = 8kk+e and 8kk+4+e for A and B alike -- a permutation of the reduction order only.
// Global->LDS staging goes through registers (the BN+ReLU prologue and the im2col zero padding need the VALU),
// issued one K-step ahead of the MFMAs (double-buffered LDS, one barrier per K-step).
#include "conv.h"
#include "prof.h"
#include <climits>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "conv_kernel.h"
#include "conv_ring_kernel.h"
#include "conv_b2b_kernel.h"

// Sums the split-K partial slabs in a fixed order and applies the epilogue.  One thread per 4 channels.
template <int PAIR>
__device__ __forceinline__ void conv_splitk_reduce_body(const ConvArgs& p, const ConvPair& pr) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = p.Cout / 4;
    if (q >= (size_t)p.M * c4n) return;
    const int m = (int)(q / c4n);
    const int n = (int)(q - (size_t)m * c4n) * 4;
    const bool t1 = PAIR && m >= pr.m_tower;              // row of the second tower: its y / residual live at an offset
    const size_t slab = (size_t)p.M * p.Cout;
    const float* src = p.partial + (size_t)m * p.Cout + n;
    float4 s = *reinterpret_cast<const float4*>(src);
    for (int z = 1; z < p.splitk; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(src + z * slab);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (p.bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (p.residual != nullptr) {
        size_t ri;
        if (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo) {
            ri = (size_t)m * p.res_ld + n;
        } else {
            const int img = sn_fastdiv(m, p.div_hw_mul, p.div_hw_shift);
            const int rr = m - img * (p.Ho * p.Wo);
            const int oy = sn_fastdiv(rr, p.div_w_mul, p.div_w_shift), ox = rr - oy * p.Wo;
            ri = (((size_t)img * p.res_H + oy * p.res_stride) * p.res_W + ox * p.res_stride) * p.res_ld + n;
        }
        const float4 t = *reinterpret_cast<const float4*>(p.residual + (t1 ? pr.dres : 0) + ri);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (p.out_scale != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(p.out_scale + n), b = *reinterpret_cast<const float4*>(p.out_shift + n);
        s.x = __builtin_fmaf(s.x, a.x, b.x); s.y = __builtin_fmaf(s.y, a.y, b.y);
        s.z = __builtin_fmaf(s.z, a.z, b.z); s.w = __builtin_fmaf(s.w, a.w, b.w);
    }
    if (p.out_scale != nullptr && p.out_floor != nullptr) {
        const float4 f = *reinterpret_cast<const float4*>(p.out_floor + n);
        s.x = fmaxf(s.x, f.x); s.y = fmaxf(s.y, f.y); s.z = fmaxf(s.z, f.z); s.w = fmaxf(s.w, f.w);
    } else if (p.relu_out) {
        s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    *reinterpret_cast<float4*>(p.y + (t1 ? pr.dy : 0) + (size_t)m * p.Cout + n) = s;
}
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs p) {
    conv_splitk_reduce_body<0>(p, ConvPair{});
}
__global__ __launch_bounds__(256) void conv_splitk_reduce_pair_kernel(const ConvArgs p, const ConvPair pr) {
    conv_splitk_reduce_body<1>(p, pr);
}

// ---------------------------------------------------------------------------------------------------------
enum TileId { T128x128 = 0, T128x64 = 1, T64x64 = 2 };

static void tile_dims(int t, int& bm, int& bn) {
    bm = (t == T64x64) ? 64 : 128;
    bn = (t == T128x128) ? 128 : 64;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int g_force_tile = -2, g_force_split = -1, g_force_bk16 = 0;

static int conv_bk(const ConvArgs& a) { return (a.rowrun || (a.Cin % 32 == 0 && !g_force_bk16)) ? 32 : 16; }
static int conv_total_steps(const ConvArgs& a) {
    if (a.rowrun) return a.KH * cdiv(a.KW * a.Cin, 32);
    return a.KH * a.KW * (a.Cin / conv_bk(a));
}
   // tuning hook (tools/autotune.py); -2 = read the environment once

// ---- measured split-K table ------------------------------------------------------------------------------------
struct TuneEntry { int M, Cout, K, KH, ring, splitk; };
#include "conv_tuning_table.h"          // static const TuneEntry g_tuning_builtin[]; generated by tools/tune_splitk.py
#include "conv_tuning_table_packed.h"   // static const TuneEntry g_tuning_packed[]: the same measurement with the packed split kernels (operand mode 4)
static int g_tuning_profile = 0;        // 1: plans are made for the packed split kernels (stabnet_conv_tuning_profile)
static std::vector<TuneEntry> g_tuning_runtime;      // set through stabnet_conv_tuning_table_set (the tuner itself)
static int g_tuning_use_builtin = -1;

static int tuning_lookup(int M, int Cout, int K, int KH, int ring) {
    for (const TuneEntry& e : g_tuning_runtime)
        if (e.M == M && e.Cout == Cout && e.K == K && e.KH == KH && e.ring == ring) return e.splitk;
    if (g_tuning_use_builtin < 0) g_tuning_use_builtin = env_int("STABNET_CONV_TUNING_TABLE", 1);
    if (g_tuning_use_builtin && g_tuning_profile == 1)
        for (const TuneEntry& e : g_tuning_packed)
            if (e.M == M && e.Cout == Cout && e.K == K && e.KH == KH && e.ring == ring) return e.splitk;
    if (g_tuning_use_builtin)
        for (const TuneEntry& e : g_tuning_builtin)
            if (e.M == M && e.Cout == Cout && e.K == K && e.KH == KH && e.ring == ring) return e.splitk;
    return 0;
}

static int pick_tile(const ConvArgs& a, int& splitk) {
    if (g_force_tile == -2) {
        g_force_tile = env_int("STABNET_CONV_TILE", -1);
        g_force_split = env_int("STABNET_CONV_SPLITK", -1);
        g_force_bk16 = env_int("STABNET_CONV_BK16", 0);
    }
    const int force_tile = g_force_tile, force_split = g_force_split;
    if (force_tile >= 0) {
        const int steps0 = conv_total_steps(a);
        splitk = force_split > 0 ? std::min(force_split, steps0) : 1;
        return force_tile;
    }
    // Tile: at every shape of the regressor -- batch-1 720p, batch-8 288x512 forward and dgrad -- the 64x64 tile is the
    // fastest (tools/autotune.py, profiles/r01_autotune_*.txt).  Split-K: (1) the measured table (conv_tuning_table.h,
    // generated by tools/tune_splitk.py from in-network per-layer timings, kernel + reduce + launch gap); (2) for shapes
    // the table does not know, the rule distilled from it: split only grids of fewer than ~400 tiles, into about 448 / tiles
    // slices of at least 8 K-steps (two resident workgroups per CU; a cost model built from probe constants did worse
    // than this rule: 484.7 vs 494.9 frames/s).
    const int total_steps = conv_total_steps(a);
    const long blocks = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64);
    const bool ring_path = a.rowrun || (a.in_scale_expected == 0 && a.up == 1 && a.Cin % 32 == 0);
    int s = tuning_lookup(a.M, a.Cout, a.K, a.KH, ring_path ? 1 : 0);
    static const int thr = env_int("STABNET_CONV_SPLIT_BELOW", 400), target = env_int("STABNET_CONV_SPLIT_TARGET", 448);
    if (s > 0) {
        s = std::min(s, std::max(1, total_steps / 2));
    } else if (blocks < thr && a.Cout % 4 == 0) {
        s = (int)((target + blocks / 2) / blocks);
        s = std::min(s, std::max(1, total_steps / 8));
        s = std::min(s, 32);
    } else {
        s = 1;
    }
    splitk = std::max(1, s);
    return T64x64;
}

size_t conv_plan(ConvArgs& a) {
    static const int swz = env_int("STABNET_CONV_XCD", 1);   // halves the memory-side traffic at ~0.5 % of the frame rate
    a.xcd_swizzle = swz;
    a.M = a.N * a.Ho * a.Wo;
    a.K = a.rowrun ? a.KH * 32 * cdiv(a.KW * a.Cin, 32) : a.KH * a.KW * a.Cin;
    if (a.x_ld == 0) a.x_ld = a.Cin;
    if (a.res_ld == 0) a.res_ld = a.Cout;
    sn_fastdiv_make((unsigned)(a.Ho * a.Wo), a.div_hw_mul, a.div_hw_shift);
    sn_fastdiv_make((unsigned)a.Wo, a.div_w_mul, a.div_w_shift);
    int splitk = 1;
    (void)pick_tile(a, splitk);
    const int total_steps = conv_total_steps(a);
    a.steps_per_split = cdiv(total_steps, splitk);
    a.splitk = cdiv(total_steps, a.steps_per_split);
    return a.splitk > 1 ? (size_t)a.splitk * a.M * a.Cout * sizeof(float) : 0;
}

template <int BM, int BN, int BK, int WM, int WN, int MODE, int NBUF, int BF16 = 0>
static int launch_one_nb(const ConvArgs& a, hipStream_t st) {
    constexpr size_t lds_op = NBUF * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    constexpr size_t lds_epi = 4 * (size_t)SN_EPI_WAVE_BYTES;        // the epilogue's transposition scratch
    constexpr size_t lds = lds_op > lds_epi ? lds_op : lds_epi;
    static bool configured = false;
    auto kern = conv_igemm_f32_kernel<BM, BN, BK, WM, WN, MODE, NBUF, BF16>;
    if (!configured) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                stabnet_set_error("conv: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
                return STABNET_ERR_LAUNCH;
            }
        }
        configured = true;
    }
    dim3 grid(cdiv(a.M, BM), cdiv(a.Cout, BN), a.splitk);
    kern<<<grid, 256, lds, st>>>(a);
    SN_LAUNCH_CHECK("conv_igemm_f32_kernel");
    return STABNET_OK;
}

static thread_local int g_bf16_operands = 0;      // set by conv_launch for the duration of one launch (see conv.h)

// Which instantiation a register-staged launch runs: (NBUF, BF16) -- one rule for the launcher and for the profiler's kernel name.
// 1x1 launches over the large maps (block 1 at 720p: M = 57 600) are bandwidth / epilogue shaped: ONE LDS stage (18 KB, twice
// the resident workgroups) beats the double-buffered loop there (30.4 -> 28.8 us, 33.7 -> 32.2 us); below that it loses 0.3 us.
// The bf16-operand variant exists for the inference tile (64 x 64 x 32) only.
static void igemm_variant(const ConvArgs& a, int bm, int bn, int bk, int mode, int& nbuf_out, int& bf16_out) {
    static const int nbuf = env_int("STABNET_CONV_NBUF", 0);          // 0: the rule; 1 / 2: forced
    bf16_out = (g_bf16_operands && bm == 64 && bn == 64 && bk == 32) ? g_bf16_operands : 0;
    const bool one_stage = nbuf == 1 || (nbuf == 0 && mode == 0 && bm == 64 && bn == 64 && a.M >= 32768);
    nbuf_out = (bf16_out != 1 && one_stage) ? 1 : 2;
}

template <int BM, int BN, int BK, int WM, int WN, int MODE>
static int launch_one_t(const ConvArgs& a, hipStream_t st) {
    int nbuf, bf16;
    igemm_variant(a, BM, BN, BK, MODE, nbuf, bf16);
    if constexpr (BM == 64 && BN == 64 && BK == 32) {
        if (bf16 == 1) return launch_one_nb<BM, BN, BK, WM, WN, MODE, 2, 1>(a, st);
        else if (bf16 == 2) return nbuf == 1 ? launch_one_nb<BM, BN, BK, WM, WN, MODE, 1, 2>(a, st) : launch_one_nb<BM, BN, BK, WM, WN, MODE, 2, 2>(a, st);
        else if (bf16 == 3) return nbuf == 1 ? launch_one_nb<BM, BN, BK, WM, WN, MODE, 1, 3>(a, st) : launch_one_nb<BM, BN, BK, WM, WN, MODE, 2, 3>(a, st);
    }
    return nbuf == 1 ? launch_one_nb<BM, BN, BK, WM, WN, MODE, 1>(a, st) : launch_one_nb<BM, BN, BK, WM, WN, MODE, 2>(a, st);
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_one(const ConvArgs& a, hipStream_t st) {
    if (a.up > 1) return launch_one_t<BM, BN, BK, WM, WN, 2>(a, st);
    if (a.pad == 0) return launch_one_t<BM, BN, BK, WM, WN, 0>(a, st);
    return launch_one_t<BM, BN, BK, WM, WN, 1>(a, st);
}

// LDS-DMA ring kernel (conv_ring_kernel.h): no A-operand prologue, stride-free addressing, Cin % 32 == 0, 64x64 tile.
static int g_ring = -1;

static bool ring_eligible(const ConvArgs& a, int tile, bool has_prologue) {
    if (g_ring < 0) {
        g_ring = env_int("STABNET_CONV_RING", 1);
    }
    if (a.rowrun) return true;                              // the row-run A operand exists only in the ring kernel
    // two-step tiles (1x1, K = 64) with a plain epilogue are 2 us faster per launch on the register-staged kernel (more resident
    // workgroups to overlap the 16 KB epilogues: 33.4 vs 35.3 us at M = 57 600, N = 256); the merged shortcut|conv1 launch stays here
    static const int lowk = env_int("STABNET_CONV_LOWK_IGEMM", 1);
    if (lowk && a.KH == 1 && a.KW == 1 && a.Cin == 64 && a.x_ld == a.Cin && a.out_floor == nullptr) return false;
    return g_ring && tile == T64x64 && !has_prologue && a.up == 1 && a.Cin % 32 == 0 && !g_force_bk16;
}
static bool ring_eligible(const ConvArgs& a, int tile) { return ring_eligible(a, tile, a.in_scale != nullptr); }

static int g_ring_wgs = 0;        // resident workgroups of the ring kernel on this device (3 per CU: 48 KiB LDS each)
static int g_ring_cus = 0;        // (the CU count g_ring_wgs was computed from)
static int g_reserved_cus = 0;    // CUs the persistent grids leave to a communication stream (stabnet_conv_reserve_cus)
// persistent grids are sized for the CUs that are NOT reserved (never fewer than an eighth of the chip)
static int usable_cus(int cus) { return std::max(cus / 8, cus - std::max(0, g_reserved_cus)); }
static int ring_grid_cap() { return g_ring_cus > 0 ? g_ring_wgs / g_ring_cus * usable_cus(g_ring_cus) : g_ring_wgs; }

template <int MODE>
static int launch_ring_mode(const ConvArgs& a, hipStream_t st) {
    if (g_ring_wgs == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0) {
            stabnet_set_error("conv: cannot read the CU count");
            return STABNET_ERR_LAUNCH;
        }
        g_ring_wgs = env_int("STABNET_CONV_RING_WGS_PER_CU", 3) * cus;
        g_ring_cus = cus;
    }
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64) * a.splitk;
    const int grid = (int)std::min<long>(ntiles, ring_grid_cap());
    if (g_bf16_operands == 1) conv_ring_f32_kernel<MODE, 1><<<grid, 256, 0, st>>>(a);
    else if (g_bf16_operands == 2) conv_ring_f32_kernel<MODE, 2><<<grid, 256, 0, st>>>(a);
    else if (g_bf16_operands == 3) conv_ring_f32_kernel<MODE, 3><<<grid, 256, 0, st>>>(a);
    else conv_ring_f32_kernel<MODE, 0><<<grid, 256, 0, st>>>(a);
    SN_LAUNCH_CHECK("conv_ring_f32_kernel");
    return STABNET_OK;
}

static int g_cus = 0;
static int device_cus() {
    if (g_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 0;
        g_cus = cus;
    }
    return g_cus;
}

// 4 (default): every wave splits both k groups of its A rows itself.  5 (STABNET_CONV_PACKED_SHARE=1): the two waves over the same A
// rows split one k group each and exchange the planes through LDS -- half the split VALU, a second barrier per step and an LDS
// round trip: measured SLOWER (stand-alone 35.0 / 38.4 / 131.7 us against 31.0 / 32.5 / 119.7 us, 627 vs 636 frames/s at 720p; even
// with the barriers compiled out 33.5 / 35.8 / 122.9 us): the split VALU is not what bounds the packed kernel.  Kept as a switch.
int conv_packed_variant() {
    static const int share = env_int("STABNET_CONV_PACKED_SHARE", 0);
    return share ? 5 : 4;
}

// The packed split kernel (conv_ring_kernel.h, BF16 = 4 / 5): 60 / 72 KiB of LDS per workgroup -> two per CU.
template <int MODE>
static int launch_ring_packed_mode(const ConvArgs& a, hipStream_t st) {
    const int cus = device_cus();
    if (cus <= 0) {
        stabnet_set_error("conv: cannot read the CU count");
        return STABNET_ERR_LAUNCH;
    }
    static const int per_cu = env_int("STABNET_CONV_PACKED_WGS_PER_CU", 2);
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64) * a.splitk;
    const int grid = (int)std::min<long>(ntiles, (long)per_cu * usable_cus(cus));
    if constexpr (MODE == 0) {
        if (a.in_scale != nullptr) {                          // BN + ReLU prologue on the A fragments (the inference conv1 layers)
            if (conv_packed_variant() == 5) conv_ring_f32_kernel<0, 5, 1, 1><<<grid, 256, 0, st>>>(a);
            else conv_ring_f32_kernel<0, 4, 1, 1><<<grid, 256, 0, st>>>(a);
            SN_LAUNCH_CHECK("conv_ring_f32_kernel<packed, PRO>");
            return STABNET_OK;
        }
    }
    if (conv_packed_variant() == 5) conv_ring_f32_kernel<MODE, 5><<<grid, 256, 0, st>>>(a);
    else conv_ring_f32_kernel<MODE, 4><<<grid, 256, 0, st>>>(a);
    SN_LAUNCH_CHECK("conv_ring_f32_kernel<packed>");
    return STABNET_OK;
}
// Two K groups inside the workgroup (p.splitk == 2, equal slices): 8 waves, 2 x 60 KiB of ring, one workgroup per CU, no reduce launch.
static int launch_ring_packed_kg2(const ConvArgs& a, hipStream_t st) {
    const int cus = device_cus();
    if (cus <= 0) {
        stabnet_set_error("conv: cannot read the CU count");
        return STABNET_ERR_LAUNCH;
    }
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64);
    const int grid = (int)std::min<long>(ntiles, usable_cus(cus));
    if (conv_packed_variant() == 5) {
        if (a.in_scale != nullptr) conv_ring_f32_kernel<0, 5, 2, 1><<<grid, 512, 0, st>>>(a);
        else if (a.pad == 0) conv_ring_f32_kernel<0, 5, 2, 0><<<grid, 512, 0, st>>>(a);
        else conv_ring_f32_kernel<1, 5, 2, 0><<<grid, 512, 0, st>>>(a);
    } else {
        if (a.in_scale != nullptr) conv_ring_f32_kernel<0, 4, 2, 1><<<grid, 512, 0, st>>>(a);
        else if (a.pad == 0) conv_ring_f32_kernel<0, 4, 2, 0><<<grid, 512, 0, st>>>(a);
        else conv_ring_f32_kernel<1, 4, 2, 0><<<grid, 512, 0, st>>>(a);
    }
    SN_LAUNCH_CHECK("conv_ring_f32_kernel<packed, KG 2>");
    return STABNET_OK;
}
static int launch_ring_packed(const ConvArgs& a, hipStream_t st) {
    if (a.rowrun) return launch_ring_packed_mode<2>(a, st);
    return a.pad == 0 ? launch_ring_packed_mode<0>(a, st) : launch_ring_packed_mode<1>(a, st);
}

static int launch_ring(const ConvArgs& a, hipStream_t st) {
    if (a.rowrun) return launch_ring_mode<2>(a, st);
    return a.pad == 0 ? launch_ring_mode<0>(a, st) : launch_ring_mode<1>(a, st);
}

// Split-K inside the workgroup (conv_ring_kernel.h, "KG"): a launch strategy for a given split count, chosen here for the
// planner (no reduce launch counted) and for the launcher alike.  Ring path only: 3 groups x 48 KiB of ring = one 12-wave workgroup
// per CU; the slices must be equal (steps % 3 == 0).  Measured at 720p (rocprofv3 inside the graph replay, block-3 conv2, M = 3600,
// N = 256, K = 2304): 45.3 us against 45.8 us + a 4.9 us reduce launch.  The register-staged kernel's two-group form lost
// (30.8 vs 24.3 + 4.9 us) and was removed.
// Two groups with the fragment prologue (conv_ring_f32_kernel<0, 0, 2, 1>): the 1x1 layers that carry a BN + ReLU prologue AND
// split K in two (the block-3 conv1 layers of a 720p frame, K = 1024: register-staged kernel x 2 slices + slabs + a reduce launch
// before) as one 8-wave workgroup per tile.
// The PRO form fetches a step's scales AND shifts with one DMA instruction: the shifts are addressed as an unsigned 32-bit byte
// offset from the (running) scale pointer.  Callers of the public operators pass two independent pointers: a shift vector below
// the scales, or 4 GiB or more above them, must take the register-staged kernel (which dereferences both pointers).  Plan time
// (pointers not bound yet): true -- the launch decides again with the real pointers.
static bool ring_pro_vectors_ok(const ConvArgs& a) {
    if (a.in_scale == nullptr || a.in_shift == nullptr) return true;
    const long d = (long)(a.in_shift - a.in_scale);                   // floats
    return d >= 0 && d < (1L << 30);
}
static bool ring_pro_geometry(const ConvArgs& a, bool has_prologue) {
    static const int on = env_int("STABNET_CONV_RING_PRO", 1);
    if (g_ring < 0) g_ring = env_int("STABNET_CONV_RING", 1);
    return on && g_ring && g_bf16_operands != 1 && !g_force_bk16 && has_prologue && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 &&
           a.up == 1 && a.Cin % 32 == 0 && !a.rowrun && a.x_ld == a.Cin;
}
static int conv_kgroups(const ConvArgs& a, bool ring, bool has_prologue) {
    static const int on = env_int("STABNET_CONV_KGROUPS", 1);
    static const int on2 = env_int("STABNET_CONV_KGROUPS_PRO", 1);
    if (!on || a.splitk < 2 || g_bf16_operands == 1) return 1;
    const int steps = conv_total_steps(a);
    if (a.steps_per_split * a.splitk != steps) return 1;
    if (ring && a.splitk == 3 && !a.rowrun) return 3;
    if (on2 && !ring && a.splitk == 2 && ring_pro_geometry(a, has_prologue) && ring_pro_vectors_ok(a)) return 2;
    return 1;
}


static int launch_ring_kg(const ConvArgs& a, int kg, hipStream_t st) {
    const int cus = device_cus();
    if (cus <= 0) {
        stabnet_set_error("conv: cannot read the CU count");
        return STABNET_ERR_LAUNCH;
    }
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64);
    const int grid = (int)std::min<long>(ntiles, usable_cus(cus));     // 144 (102) KiB of LDS: one workgroup per CU
#define SN_KG_LAUNCH(B)                                                                  \
    do {                                                                                  \
        if (kg == 2) conv_ring_f32_kernel<0, B, 2, 1><<<grid, 512, 0, st>>>(a);           \
        else if (a.pad == 0) conv_ring_f32_kernel<0, B, 3><<<grid, 768, 0, st>>>(a);      \
        else conv_ring_f32_kernel<1, B, 3><<<grid, 768, 0, st>>>(a);                      \
    } while (0)
    if (g_bf16_operands == 2) SN_KG_LAUNCH(2);
    else if (g_bf16_operands == 3) SN_KG_LAUNCH(3);
    else SN_KG_LAUNCH(0);
#undef SN_KG_LAUNCH
    SN_LAUNCH_CHECK("conv_ring_f32_kernel<KG>");
    return STABNET_OK;
}

// The ring kernel's PRO form (conv_ring_kernel.h): 1x1 / stride 1 convolutions whose input carries a BN + ReLU prologue and whose
// epilogue has no consumer BN (the training forward).  `delta` != 0: the pair of towers as one launch (ConvPair::dscale).
// Measured: the 36 paired 1x1 launches of the 8 x 288 x 512 step 61.2 -> 55.9 us (76 -> 82.8 TF), 494.7 -> 502.2 pairs/s.  The
// inference conv1 layers (prologue AND consumer BN) were tried on it too: 25.7 us against 23.5 / 28.7 us on the register-staged
// kernel, the same 310 us per frame in sum -- they stay where they were.
static bool ring_pro_eligible(const ConvArgs& a) {
    static const int on = env_int("STABNET_CONV_RING_PRO", 1);
    if (g_ring < 0) g_ring = env_int("STABNET_CONV_RING", 1);
    return on && g_ring && g_bf16_operands != 1 && !g_force_bk16 && a.in_scale != nullptr && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 &&
           a.up == 1 && a.Cin % 32 == 0 && a.out_scale == nullptr && a.out_floor == nullptr && !a.rowrun && a.x_ld == a.Cin &&
           ring_pro_vectors_ok(a);
}
static int launch_ring_pro(ConvArgs a, long delta, hipStream_t st) {
    if (g_ring_wgs == 0) {
        const int cus = device_cus();
        if (cus <= 0) {
            stabnet_set_error("conv: cannot read the CU count");
            return STABNET_ERR_LAUNCH;
        }
        g_ring_wgs = env_int("STABNET_CONV_RING_WGS_PER_CU", 3) * cus;
        g_ring_cus = cus;
    }
    a.out_floor = reinterpret_cast<const float*>((size_t)delta);      // the kernel's pair distance (not a pointer: see conv_ring_kernel.h PRO)
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64) * a.splitk;
    const int grid = (int)std::min<long>(ntiles, ring_grid_cap());
    if (g_bf16_operands == 2) conv_ring_f32_kernel<0, 2, 1, 1><<<grid, 256, 0, st>>>(a);
    else if (g_bf16_operands == 3) conv_ring_f32_kernel<0, 3, 1, 1><<<grid, 256, 0, st>>>(a);
    else conv_ring_f32_kernel<0, 0, 1, 1><<<grid, 256, 0, st>>>(a);
    SN_LAUNCH_CHECK("conv_ring_f32_kernel<PRO>");
    return STABNET_OK;
}

// 1 if conv_launch() of this (planned) convolution is followed by a split-K reduce launch (plan-time view: the prologue is known
// from in_scale_expected)
int conv_reduce_launches(const ConvArgs& a, int operand_mode) {
    if (a.splitk < 2) return 0;
    int splitk_unused = 1;
    const int t = pick_tile(a, splitk_unused);
    const bool pro = a.in_scale_expected != 0;
    const bool ring = ring_eligible(a, t, pro);
    if (operand_mode == 4 && t == T64x64 && a.K % 32 == 0 && conv_bk(a) == 32) {
        // conv_launch() with a weight image: the packed split kernel takes ring launches and the prologue-carrying 1x1 layers; only a
        // two-way split with equal halves runs inside the workgroup
        static const int kg2_on = env_int("STABNET_CONV_PACKED_KG2", 1), pro_on = env_int("STABNET_CONV_PACKED_PRO", 1),
                         kg3_on = env_int("STABNET_CONV_PACKED_KG3", 1);
        const int kgf = conv_kgroups(a, ring, pro);
        const bool takes = (ring && (kgf == 1 || (kgf == 3 && kg3_on))) || (!ring && pro && pro_on && ring_pro_geometry(a, true));
        if (takes) return (kg2_on && a.splitk == 2 && !a.rowrun && a.steps_per_split * 2 == conv_total_steps(a)) ? 0 : 1;
    }
    return (t == T64x64 && conv_kgroups(a, ring, pro) > 1) ? 0 : 1;
}

// ---- pre-split weight image (conv.h) -----------------------------------------------------------------------------------
size_t conv_weight_image_floats(int Cout, int K) { return (size_t)cdiv(Cout, 64) * (size_t)(K / 32) * 3072; }

// one thread per (N tile, K step, wave column, k group, lane): 8 weights -> three 16-byte plane entries
__device__ __forceinline__ void weight_split_image_thread(const float* __restrict__ w, int Cout, int K, uint4* __restrict__ img, long i) {
    const int steps = K / 32;
    const int lane = (int)(i & 63), j = (int)((i >> 6) & 1), wn = (int)((i >> 7) & 1);
    const long ts = i >> 8;                                  // nt * steps + ks
    const int ks = (int)(ts % steps), nt = (int)(ts / steps);
    const int n = nt * 64 + wn * 32 + (lane & 31), g = lane >> 5;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (n < Cout) {
        const float* r = w + (size_t)n * K + ks * 32 + 16 * j + 4 * g;
        lo = *reinterpret_cast<const f32x4*>(r);
        hi = *reinterpret_cast<const f32x4*>(r + 8);
    }
    const SnSplit3 a = sn_split3(lo), b = sn_split3(hi);
    uint4* out = img + ts * 768 + (size_t)(wn * 6 + j) * 64 + lane;       // 16-byte units: stage 768, chunk 64
    out[0] = __builtin_bit_cast(uint4, SN_CAT8(a.h, b.h));
    out[2 * 64] = __builtin_bit_cast(uint4, SN_CAT8(a.m, b.m));
    out[4 * 64] = __builtin_bit_cast(uint4, SN_CAT8(a.l, b.l));
}

__global__ __launch_bounds__(256) void weight_split_image_kernel(const float* __restrict__ w, int Cout, int K, uint4* __restrict__ img) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)((Cout + 63) / 64) * (K / 32) * 256) return;
    weight_split_image_thread(w, Cout, K, img, i);
}

__global__ __launch_bounds__(256) void weight_split_images_kernel(const float* __restrict__ w_base, float* __restrict__ img_base, const WeightImageTable t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= t.tprefix[t.n]) return;
    int e = 0;
    while (i >= t.tprefix[e + 1]) ++e;                       // (every matrix covers whole 256-thread blocks: uniform per block)
    weight_split_image_thread(w_base + t.w_off[e], t.Cout[e], t.K[e], reinterpret_cast<uint4*>(img_base + t.img_off[e]), i - t.tprefix[e]);
}

void weight_image_table_add(WeightImageTable& t, long w_off, long img_off, int Cout, int K) {
    if (t.n == 0) t.tprefix[0] = 0;
    t.w_off[t.n] = w_off; t.img_off[t.n] = img_off; t.Cout[t.n] = Cout; t.K[t.n] = K;
    t.tprefix[t.n + 1] = t.tprefix[t.n] + (long)cdiv(Cout, 64) * (K / 32) * 256;
    ++t.n;
}

int launch_weight_split_images(const float* w_base, float* img_base, const WeightImageTable& t, hipStream_t st) {
    if (t.n == 0) return STABNET_OK;
    SN_REQUIRE(w_base && img_base && t.n <= 64, "weight images: bad table");
    weight_split_images_kernel<<<(unsigned)(t.tprefix[t.n] >> 8), 256, 0, st>>>(w_base, img_base, t);
    SN_LAUNCH_CHECK("weight_split_images_kernel");
    return STABNET_OK;
}

int launch_weight_split_image(const float* w, int Cout, int K, float* img, hipStream_t st) {
    SN_REQUIRE(w && img && Cout > 0 && K > 0 && K % 32 == 0, "weight image: K=%d must be a multiple of 32", K);
    const long total = (long)cdiv(Cout, 64) * (K / 32) * 256;
    weight_split_image_kernel<<<cdiv(total, 256), 256, 0, st>>>(w, Cout, K, reinterpret_cast<uint4*>(img));
    SN_LAUNCH_CHECK("weight_split_image_kernel");
    return STABNET_OK;
}

int conv_launch(const ConvArgs& a_in, hipStream_t st, Prof* prof, int bf16_operands, const float* w_img) {
    const bool want_packed = bf16_operands == 4 && w_img != nullptr;
    if (bf16_operands == 4) bf16_operands = 0;               // every path but the packed ring kernel is the exact f32 one
    g_bf16_operands = bf16_operands;
    const ConvArgs& a = a_in;
    SN_REQUIRE(a.rowrun || a.Cin % 16 == 0, "conv: Cin=%d must be a multiple of 16 (pad the channels)", a.Cin);
    SN_REQUIRE(!a.rowrun || (a.in_scale == nullptr && a.up == 1 && a.KH <= 8 && cdiv(a.KW * a.Cin, 32) <= 4),
               "conv: row-run operand needs no prologue, KH <= 8 and KW*Cin <= 128");
    SN_REQUIRE(a.Cout % 4 == 0, "conv: Cout=%d must be a multiple of 4", a.Cout);
    SN_REQUIRE(a.splitk >= 1 && a.steps_per_split >= 1 && a.div_hw_mul != 0, "conv: conv_plan() not called");
    SN_REQUIRE(a.splitk == 1 || a.partial != nullptr, "conv: split-K needs a workspace");
    int splitk_unused = 1;
    const int t = pick_tile(a, splitk_unused);
    const bool bk32 = conv_bk(a) == 32;
    int rc;
    const bool rec = prof != nullptr && prof->begin(st);
    const bool ring = ring_eligible(a, t);
    SN_REQUIRE(ring || a.x_ld == a.Cin, "conv: a strided input (x_ld %d != Cin %d) needs the ring kernel", a.x_ld, a.Cin);
    int kg = (t == T64x64) ? conv_kgroups(a, ring, a.in_scale != nullptr) : 1;
    // the packed split kernel has no in-workgroup split-K (60 KiB of ring per group): a ring launch that would split K three ways
    // inside the workgroup goes through the slabs + reduce launch instead
    static const int packed_kg3 = env_int("STABNET_CONV_PACKED_KG3", 1);
    if (want_packed && packed_kg3 && ring && kg == 3 && a.K % 32 == 0 && a.partial != nullptr) kg = 1;
    const bool pro = kg == 1 && !ring && t == T64x64 && bk32 && ring_pro_eligible(a);
    // packed split kernel with the fragment prologue: 1x1 / stride 1 layers that carry an input BN + ReLU (register-staged kernel or
    // the two-group PRO ring form otherwise); a K split goes through the slabs + reduce launch
    static const int packed_pro_on = env_int("STABNET_CONV_PACKED_PRO", 1);
    const bool packed_pro = want_packed && packed_pro_on && !ring && t == T64x64 && bk32 && a.in_scale != nullptr && a.out_floor == nullptr &&
                            ring_pro_geometry(a, true) && ring_pro_vectors_ok(a) && (a.splitk == 1 || a.partial != nullptr);   // (out_floor: the kernel's pair distance)
    if (packed_pro) kg = 1;
    const bool packed = (want_packed && ring && kg == 1 && !pro && a.K % 32 == 0) || packed_pro;
    // a two-way K split runs inside the workgroup (no slabs, no reduce launch)
    static const int packed_kg2_on = env_int("STABNET_CONV_PACKED_KG2", 1);
    const bool packed_kg2 = packed && packed_kg2_on && a.splitk == 2 && !a.rowrun && a.steps_per_split * 2 == conv_total_steps(a);
    if (packed_kg2) kg = 2;
    if (packed) {
        ConvArgs b = a;
        b.w = w_img;
        rc = packed_kg2 ? launch_ring_packed_kg2(b, st) : launch_ring_packed(b, st);
    } else if (pro) {
        rc = launch_ring_pro(a, 0, st);
    } else if (kg > 1) {
        rc = launch_ring_kg(a, kg, st);
    } else if (ring) {
        rc = launch_ring(a, st);
    } else if (bk32) {
        if (t == T128x128) rc = launch_one<128, 128, 32, 64, 64>(a, st);
        else if (t == T128x64) rc = launch_one<128, 64, 32, 64, 32>(a, st);
        else rc = launch_one<64, 64, 32, 32, 32>(a, st);
    } else {
        if (t == T128x128) rc = launch_one<128, 128, 16, 64, 64>(a, st);
        else if (t == T128x64) rc = launch_one<128, 64, 16, 64, 32>(a, st);
        else rc = launch_one<64, 64, 16, 32, 32>(a, st);
    }
    const int mode = (a.up > 1 || a.rowrun) ? 2 : (a.pad == 0 ? 0 : 1);
    const bool b_has_pro = a.in_scale != nullptr;           // (the packed launchers pick the prologue form from the pointer)
    int kind = PK_KERNEL_CONV_RING + mode + (bf16_operands ? 3 : 0);
    if (packed) {
        kind = packed_kg2 ? (b_has_pro ? PK_KERNEL_CONV_PACKED + 6 : PK_KERNEL_CONV_PACKED + 4 + mode)
                          : (b_has_pro ? PK_KERNEL_CONV_PACKED + 3 : PK_KERNEL_CONV_PACKED + mode);
    } else if (pro) {
        kind = PK_KERNEL_CONV_KG + 2;
    } else if (kg > 1) {
        kind = kg == 2 ? PK_KERNEL_CONV_KG + 3 : PK_KERNEL_CONV_KG + mode;
    } else if (!ring) {                                                  // + 18 for the one-stage (NBUF = 1), + 36 for the bf16-operand instantiation
        int bm, bn, nbuf, bf16;
        tile_dims(t, bm, bn);
        igemm_variant(a, bm, bn, bk32 ? 32 : 16, mode, nbuf, bf16);
        kind = PK_KERNEL_CONV_BASE + mode * 6 + t * 2 + (bk32 ? 1 : 0) + (nbuf == 1 ? 18 : 0) + (bf16 ? 36 : 0);
    }
    if (rec) prof->end(st, kind, 2.0 * a.M * (double)(a.KH * a.KW * (a.cin_real ? a.cin_real : a.Cin)) * a.Cout,
                       // algorithmic bytes: input + weights + output (or the split-K slabs) + the residual read
                       4.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.K * a.Cout + (double)a.M * a.Cout * (kg > 1 ? 1 : a.splitk) +
                              ((a.residual != nullptr && (a.splitk == 1 || kg > 1)) ? (double)a.M * a.Cout : 0.0)),
                       a.M, a.Cout, a.K, a.splitk);
    if (rc) return rc;
    if (a.splitk > 1 && kg == 1) {
        const size_t q = (size_t)a.M * (a.Cout / 4);
        const bool rec2 = prof != nullptr && prof->begin(st);
        conv_splitk_reduce_kernel<<<cdiv((long)q, 256), 256, 0, st>>>(a);
        if (rec2) prof->end(st, PK_KERNEL_SPLITK_REDUCE, 0.0, 4.0 * (double)a.M * a.Cout * (a.splitk + 1 + (a.residual != nullptr ? 1 : 0)));
        SN_LAUNCH_CHECK("conv_splitk_reduce_kernel");
    }
    return STABNET_OK;
}

// The siamese pair as one launch (ConvPair, conv.h): the 64x64 register-staged tile, MODE 0 / 1, BK 32 or 16.
bool conv_pair_supported(const ConvArgs& a) {
    int splitk_unused = 1;
    const int t = pick_tile(a, splitk_unused);
    const bool pro = a.in_scale != nullptr || a.in_scale_expected;     // (at plan time the prologue pointers are not bound yet)
    return t == T64x64 && !ring_eligible(a, t, pro) && a.up == 1 && !a.rowrun && (a.x_ld == 0 || a.x_ld == a.Cin);
}

template <int BK, int MODE>
static int launch_pair_one(const ConvArgs& a, const ConvPair& pr, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(64 + 64) * (BK + 4) * sizeof(float);
    static_assert(lds >= 4 * (size_t)SN_EPI_WAVE_BYTES && lds <= 64 * 1024, "LDS of the 64x64 tile");
    dim3 grid(cdiv(a.M, 64), cdiv(a.Cout, 64), a.splitk);
    conv_igemm_f32_pair_kernel<64, 64, BK, 32, 32, MODE><<<grid, 256, lds, st>>>(a, pr);
    SN_LAUNCH_CHECK("conv_igemm_f32_pair_kernel");
    return STABNET_OK;
}

// The packed split kernel's prologue form over the PAIR of towers (training forward, opt-in): launch_ring_pro with a weight image.
static int launch_ring_packed_pro_pair(ConvArgs a, long delta, const float* w_img, hipStream_t st) {
    const int cus = device_cus();
    if (cus <= 0) {
        stabnet_set_error("conv: cannot read the CU count");
        return STABNET_ERR_LAUNCH;
    }
    a.w = w_img;
    a.out_floor = reinterpret_cast<const float*>((size_t)delta);      // the kernel's pair distance (conv_ring_kernel.h PRO)
    static const int per_cu = env_int("STABNET_CONV_PACKED_WGS_PER_CU", 2);
    const long ntiles = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64) * a.splitk;
    const int grid = (int)std::min<long>(ntiles, (long)per_cu * usable_cus(cus));
    if (conv_packed_variant() == 5) conv_ring_f32_kernel<0, 5, 1, 1><<<grid, 256, 0, st>>>(a);
    else conv_ring_f32_kernel<0, 4, 1, 1><<<grid, 256, 0, st>>>(a);
    SN_LAUNCH_CHECK("conv_ring_f32_kernel<packed, PRO, pair>");
    return STABNET_OK;
}

int conv_launch_pair(const ConvArgs& a, const ConvPair& pr, hipStream_t st, Prof* prof, const float* w_img) {
    g_bf16_operands = 0;
    SN_REQUIRE(a.Cin % 16 == 0 && a.Cout % 4 == 0, "conv pair: Cin %% 16 and Cout %% 4 must be 0");
    SN_REQUIRE(a.splitk >= 1 && a.steps_per_split >= 1 && a.div_hw_mul != 0, "conv pair: conv_plan() not called");
    SN_REQUIRE(a.splitk == 1 || a.partial != nullptr, "conv pair: split-K needs a workspace");
    SN_REQUIRE(conv_pair_supported(a), "conv pair: the plan of this shape is not the 64x64 register-staged launch");
    SN_REQUIRE(pr.m_tower > 0 && pr.m_tower % 64 == 0 && a.M == 2 * pr.m_tower, "conv pair: tower rows %d must be a multiple of 64 (M %d)",
               pr.m_tower, a.M);
    const bool bk32 = conv_bk(a) == 32;
    const int mode = a.pad == 0 ? 0 : 1;
    const bool rec = prof != nullptr && prof->begin(st);
    int rc;
    // 1x1 layers: the LDS-DMA ring kernel with the BN + ReLU prologue on the A fragments (both towers in one launch: the offsets of
    // the second tower all derive from the distance of the two workspaces, which must be what ConvPair describes)
    const bool pro = bk32 && ring_pro_eligible(a) && pr.dx == pr.dscale - (long)pr.m_tower * a.Cin && pr.dy == pr.dscale - (long)pr.m_tower * a.Cout &&
                     (a.residual == nullptr || pr.dres == pr.dscale - (long)(a.N / 2) * a.res_H * a.res_W * a.res_ld);
    const bool packed = pro && w_img != nullptr && a.K % 32 == 0 && a.splitk == 1;
    if (packed) rc = launch_ring_packed_pro_pair(a, pr.dscale, w_img, st);
    else if (pro) rc = launch_ring_pro(a, pr.dscale, st);
    else if (bk32) rc = mode == 0 ? launch_pair_one<32, 0>(a, pr, st) : launch_pair_one<32, 1>(a, pr, st);
    else rc = mode == 0 ? launch_pair_one<16, 0>(a, pr, st) : launch_pair_one<16, 1>(a, pr, st);
    if (rec) prof->end(st, packed ? PK_KERNEL_CONV_PACKED + 3 : pro ? PK_KERNEL_CONV_KG + 2 : PK_KERNEL_CONV_PAIR + mode * 2 + (bk32 ? 1 : 0), 2.0 * a.M * (double)(a.KH * a.KW * (a.cin_real ? a.cin_real : a.Cin)) * a.Cout,
                       4.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.K * a.Cout + (double)a.M * a.Cout * a.splitk),
                       a.M, a.Cout, a.K, a.splitk);
    if (rc) return rc;
    if (a.splitk > 1) {
        const size_t q = (size_t)a.M * (a.Cout / 4);
        const bool rec2 = prof != nullptr && prof->begin(st);
        conv_splitk_reduce_pair_kernel<<<cdiv((long)q, 256), 256, 0, st>>>(a, pr);
        if (rec2) prof->end(st, PK_KERNEL_SPLITK_REDUCE, 0.0, 4.0 * (double)a.M * a.Cout * (a.splitk + 1));
        SN_LAUNCH_CHECK("conv_splitk_reduce_pair_kernel");
    }
    return STABNET_OK;
}

// ---- conv2 (3x3) -> conv3 (1x1) back to back (conv_b2b_kernel.h) ------------------------------------------
bool conv_b2b_supported(const ConvArgs& c2, const ConvArgs& c3) {
    // A LAUNCH strategy the inference plan does not choose by default (STABNET_CONV_B2B_PLAN=1 turns it on, net.hip): measured
    // in the 720p frame it loses to the two launches at every unit it applies to.  The operator
    // itself (stabnet_conv3x3_conv1x1_fwd) is always available.
    static const int on = env_int("STABNET_CONV_B2B", 1);
    if (g_ring < 0) g_ring = env_int("STABNET_CONV_RING", 1);
    const bool g2 = c2.KH == 3 && c2.KW == 3 && c2.pad == 1 && c2.up == 1 && !c2.rowrun && (c2.stride == 1 || c2.stride == 2) &&
                    c2.Cin == c2.Cout && (c2.Cout == 64 || c2.Cout == 128) && c2.in_scale_expected == 0 && c2.in_scale == nullptr;
    const bool g3 = c3.KH == 1 && c3.KW == 1 && c3.pad == 0 && c3.stride == 1 && c3.up == 1 && !c3.rowrun && c3.Cin == c2.Cout &&
                    (c3.x_ld == 0 || c3.x_ld == c3.Cin) && c3.Cout % c2.Cout == 0 && c3.N == c2.N && c3.H == c2.Ho && c3.W == c2.Wo &&
                    c3.in_scale_expected == 0 && c3.in_scale == nullptr;
    return on && g_ring && g2 && g3;
}

int conv_b2b_launch(const ConvArgs& c2, const ConvArgs& c3, hipStream_t st, Prof* prof) {
    SN_REQUIRE(conv_b2b_supported(c2, c3), "conv b2b: unsupported geometry");
    SN_REQUIRE(c2.div_hw_mul != 0 && c3.div_hw_mul != 0 && c2.M == c3.M, "conv b2b: conv_plan() not called on both convolutions");
    SN_REQUIRE(c2.x && c2.w && c3.w && c3.y && c2.out_scale && c2.out_shift, "conv b2b: null pointer");
    SN_REQUIRE(c2.bias == nullptr && c2.residual == nullptr, "conv b2b: the 3x3 convolution takes no bias / residual");
    SN_REQUIRE(g_bf16_operands == 0, "conv b2b: fp32 only");
    const int cus = device_cus();
    if (cus <= 0) {
        stabnet_set_error("conv: cannot read the CU count");
        return STABNET_ERR_LAUNCH;
    }
    B2bArgs P;
    P.c2 = c2;
    P.c3 = c3;
    P.c3.splitk = 1;                                                  // the accumulators handed to the epilogue are the full sum
    P.c3.steps_per_split = conv_total_steps(c3);
    const int tiles_m = cdiv(c2.M, 64);
    const bool rec = prof != nullptr && prof->begin(st);
    if (c2.Cout == 64) {
        static const int wgs = env_int("STABNET_CONV_B2B_WGS_PER_CU", 2);     // 80 KB of LDS each
        conv_b2b_f32_kernel<2><<<std::min(tiles_m, wgs * usable_cus(cus)), 256, 0, st>>>(P);
    } else {
        conv_b2b_f32_kernel<4><<<std::min(tiles_m, usable_cus(cus)), 512, 0, st>>>(P);    // 136 KB of LDS
    }
    if (rec) prof->end(st, PK_KERNEL_CONV_B2B + (c2.Cout == 128 ? 1 : 0),
                       2.0 * c2.M * ((double)c2.K * c2.Cout + (double)c3.K * c3.Cout),
                       4.0 * ((double)c2.N * c2.H * c2.W * c2.Cin + (double)c2.K * c2.Cout + (double)c3.K * c3.Cout +
                              (double)c3.M * c3.Cout * (c3.residual ? 2 : 1)),
                       c2.M, c3.Cout, c2.K + c3.K, 1);
    SN_LAUNCH_CHECK("conv_b2b_f32_kernel");
    return STABNET_OK;
}

// ---------------------------------------------------------------------------------------------------------
static int fill_args(ConvArgs& a, const float* x, const float* w, const float* bias, const float* in_scale,
                     const float* in_shift, const float* residual, int res_H, int res_W, int res_stride, float* y,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu_out) {
    SN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "conv2d: bad geometry");
    SN_REQUIRE(Cin % 16 == 0, "conv2d: Cin=%d must be a multiple of 16", Cin);
    SN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d: in_scale and in_shift go together");
    a = ConvArgs{};
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.residual = residual;
    a.in_scale_expected = in_scale != nullptr ? 1 : 0;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.up = 1;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    SN_REQUIRE(a.Ho > 0 && a.Wo > 0, "conv2d: empty output");
    a.res_H = residual ? res_H : a.Ho;
    a.res_W = residual ? res_W : a.Wo;
    a.res_stride = residual ? res_stride : 1;
    a.relu_out = relu_out;
    SN_REQUIRE((long)N * a.Ho * a.Wo * Cout < (1L << 30) && (long)N * (H + 2 * pad) * (W + 2 * pad) * Cin < (1L << 30) &&
                   (long)Cout * KH * KW * Cin < (1L << 30), "conv2d: tensors must have < 2^30 elements");
    SN_REQUIRE(KH * KW <= 64, "conv2d: filter larger than 64 taps");
    return STABNET_OK;
}

extern "C" {

/* Tuning hook: force the tile (0 = 128x128, 1 = 128x64, 2 = 64x64) and split-K of every subsequent convolution;
 * tile < 0 restores the built-in choice.  Not thread-safe; used by tools/autotune.py only. */
int stabnet_conv_reserve_cus(int reserved) {
    const int prev = g_reserved_cus;
    g_reserved_cus = reserved < 0 ? 0 : reserved;
    return prev;
}

void stabnet_conv_tuning_override(int tile, int splitk) {
    g_force_tile = tile < 0 ? -1 : tile;
    g_force_split = splitk;
}

/* Tuning hook (tools/tune_splitk.py): split-K of the convolution with GEMM shape (M, Cout, K), filter height KH, on the
 * ring kernel (ring = 1: no input BN prologue) or the register-staged kernel (ring = 0); splitk <= 0 removes the entry,
 * M < 0 clears the table.  Takes effect for plans made afterwards (net_create / conv2d calls).  Not thread-safe. */
void stabnet_conv_tuning_profile(int profile) { g_tuning_profile = profile == 1 ? 1 : 0; }

void stabnet_conv_tuning_table_set(int M, int Cout, int K, int KH, int ring, int splitk) {
    if (M < 0) { g_tuning_runtime.clear(); return; }
    for (size_t i = 0; i < g_tuning_runtime.size(); ++i) {
        TuneEntry& e = g_tuning_runtime[i];
        if (e.M == M && e.Cout == Cout && e.K == K && e.KH == KH && e.ring == ring) {
            if (splitk > 0) e.splitk = splitk;
            else g_tuning_runtime.erase(g_tuning_runtime.begin() + i);
            return;
        }
    }
    if (splitk > 0) g_tuning_runtime.push_back({M, Cout, K, KH, ring, splitk});
}

size_t stabnet_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    ConvArgs a;
    if (fill_args(a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 1, nullptr, N, H, W, Cin, Cout, KH, KW,
                  stride, pad, 0))
        return 0;
    // the split-K choice (measured table) depends on whether the launch carries an input prologue, which this query does not
    // know: the larger of the two
    ConvArgs b = a;
    b.in_scale_expected = 1;
    return std::max(conv_plan(a), conv_plan(b));
}

int stabnet_conv2d_fwd_ex(const float* x, const float* w_ohwi, const float* bias, const float* in_scale,
                          const float* in_shift, const float* residual, int res_H, int res_W, int res_stride,
                          const float* out_scale, const float* out_shift, float* y, int N, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int relu_out, void* workspace,
                          size_t workspace_bytes, void* stream);

int stabnet_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, const float* in_scale,
                       const float* in_shift, const float* residual, int res_H, int res_W, int res_stride, float* y,
                       int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return stabnet_conv2d_fwd_ex(x, w_ohwi, bias, in_scale, in_shift, residual, res_H, res_W, res_stride, nullptr, nullptr,
                                 y, N, H, W, Cin, Cout, KH, KW, stride, pad, relu_out, workspace, workspace_bytes, stream);
}

int stabnet_conv2d_fwd_ex(const float* x, const float* w_ohwi, const float* bias, const float* in_scale,
                          const float* in_shift, const float* residual, int res_H, int res_W, int res_stride,
                          const float* out_scale, const float* out_shift, float* y, int N, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int relu_out, void* workspace,
                          size_t workspace_bytes, void* stream) {
    SN_REQUIRE(x && w_ohwi && y, "conv2d_fwd: null pointer");
    SN_REQUIRE((out_scale == nullptr) == (out_shift == nullptr), "conv2d_fwd: out_scale and out_shift go together");
    ConvArgs a;
    int rc = fill_args(a, x, w_ohwi, bias, in_scale, in_shift, residual, res_H, res_W, res_stride, y, N, H, W, Cin, Cout,
                       KH, KW, stride, pad, relu_out);
    if (rc) return rc;
    const size_t need = conv_plan(a);
    if (need > workspace_bytes || (need > 0 && workspace == nullptr)) {
        stabnet_set_error("conv2d_fwd: workspace %zu B < %zu B needed", workspace_bytes, need);
        return STABNET_ERR_WORKSPACE;
    }
    a.partial = static_cast<float*>(workspace);
    a.out_scale = out_scale;
    a.out_shift = out_shift;
    return conv_launch(a, (hipStream_t)stream);
}

/* stabnet_conv2d_fwd_ex through the packed split kernels (include/stabnet_hip.h) */
size_t stabnet_conv_weight_image_floats(int Cout, int KH, int KW, int Cin) {
    const int K = KH * KW * Cin;
    return (Cout > 0 && K > 0 && K % 32 == 0) ? conv_weight_image_floats(Cout, K) : 0;
}

int stabnet_conv_weight_split_image(const float* w_ohwi, int Cout, int KH, int KW, int Cin, float* w_img, void* stream) {
    SN_REQUIRE(w_ohwi && w_img, "conv_weight_split_image: null pointer");
    SN_REQUIRE(Cin % 32 == 0, "conv_weight_split_image: Cin=%d must be a multiple of 32", Cin);
    return launch_weight_split_image(w_ohwi, Cout, KH * KW * Cin, w_img, (hipStream_t)stream);
}

int stabnet_conv2d_fwd_packed(const float* x, const float* w_ohwi, const float* w_img, const float* bias, const float* in_scale,
                              const float* in_shift, const float* residual, int res_H, int res_W, int res_stride,
                              const float* out_scale, const float* out_shift, float* y, int N, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad, int relu_out, int splitk, void* workspace,
                              size_t workspace_bytes, void* stream) {
    SN_REQUIRE(x && w_ohwi && w_img && y, "conv2d_fwd_packed: null pointer");
    SN_REQUIRE((out_scale == nullptr) == (out_shift == nullptr), "conv2d_fwd_packed: out_scale and out_shift go together");
    ConvArgs a;
    int rc = fill_args(a, x, w_ohwi, bias, in_scale, in_shift, residual, res_H, res_W, res_stride, y, N, H, W, Cin, Cout,
                       KH, KW, stride, pad, relu_out);
    if (rc) return rc;
    size_t need = conv_plan(a);
    if (splitk > 0) {                                        // the caller's K split instead of the planned one
        const int total = conv_total_steps(a);
        a.steps_per_split = cdiv(total, std::min(splitk, total));
        a.splitk = cdiv(total, a.steps_per_split);
        need = a.splitk > 1 ? (size_t)a.splitk * a.M * a.Cout * sizeof(float) : 0;
    }
    if (need > workspace_bytes || (need > 0 && workspace == nullptr)) {
        stabnet_set_error("conv2d_fwd_packed: workspace %zu B < %zu B needed", workspace_bytes, need);
        return STABNET_ERR_WORKSPACE;
    }
    a.partial = static_cast<float*>(workspace);
    a.out_scale = out_scale;
    a.out_shift = out_shift;
    return conv_launch(a, (hipStream_t)stream, nullptr, 4, w_img);
}

/* conv2 (3x3, pad 1, stride 1 | 2, C -> C channels, C = 64 | 128, no bias) -> folded BN (mid_scale, mid_shift) + ReLU -> conv3
 * (1x1, C -> Cout, Cout % C == 0) with conv2d_fwd_ex's epilogue (bias, residual, out_scale / out_shift, relu_out) as ONE launch:
 * the tail of a slim bottleneck_v2 unit (s_net_bundle_nobm.py:252-253); the C-channel intermediate never reaches memory.
 * x [N,H,W,C] with x_ld floats between pixels (0 = C); y [N,Ho,Wo,Cout]; residual read at (oy*res_stride, ox*res_stride) with
 * res_ld floats between pixels (0 = Cout).  STABNET_ERR_BAD_ARG for other geometries. */
int stabnet_conv3x3_conv1x1_fwd(const float* x, int x_ld, const float* w2_ohwi, const float* mid_scale, const float* mid_shift,
                                const float* w3_ohwi, const float* bias3, const float* residual, int res_H, int res_W,
                                int res_stride, int res_ld, const float* out_scale, const float* out_shift, float* y, int N, int H,
                                int W, int C, int Cout, int stride, int relu_out, void* stream) {
    SN_REQUIRE(x && w2_ohwi && mid_scale && mid_shift && w3_ohwi && y, "conv3x3_conv1x1_fwd: null pointer");
    SN_REQUIRE((out_scale == nullptr) == (out_shift == nullptr), "conv3x3_conv1x1_fwd: out_scale and out_shift go together");
    ConvArgs c2, c3;
    int rc = fill_args(c2, x, w2_ohwi, nullptr, nullptr, nullptr, nullptr, 0, 0, 1, nullptr, N, H, W, C, C, 3, 3, stride, 1, 1);
    if (rc) return rc;
    c2.x_ld = x_ld;
    (void)conv_plan(c2);
    c2.out_scale = mid_scale;
    c2.out_shift = mid_shift;
    rc = fill_args(c3, nullptr, w3_ohwi, bias3, nullptr, nullptr, residual, res_H, res_W, res_stride, y, N, c2.Ho, c2.Wo, C, Cout,
                   1, 1, 1, 0, relu_out);
    if (rc) return rc;
    c3.res_ld = res_ld;
    (void)conv_plan(c3);
    c3.out_scale = out_scale;
    c3.out_shift = out_shift;
    SN_REQUIRE(conv_b2b_supported(c2, c3), "conv3x3_conv1x1_fwd: C must be 64 or 128, Cout a multiple of C, stride 1 or 2");
    g_bf16_operands = 0;
    return conv_b2b_launch(c2, c3, (hipStream_t)stream);
}

}  // extern "C"
