// The implicit-GEMM convolution kernel template (shared by conv.hip and tools/conv_probe.hip).
#pragma once
#include "conv.h"
#include <climits>

#ifdef CONV_STAMP                      // probe-only: s_memtime stamps per block (never defined in the library build)
__device__ unsigned long long* g_conv_stamps;
#define CONV_STAMP_AT(i)                                                                               \
    do {                                                                                               \
        if (threadIdx.x == 0) {                                                                        \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
            g_conv_stamps[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))) * 4 + (i)] = t_; \
        }                                                                                              \
    } while (0)
#else
#define CONV_STAMP_AT(i)
#endif

#ifndef CONV_ABLATE
#define CONV_ABLATE 0          // probe-only bit mask: 1 no global loads, 2 no LDS stores, 4 no epilogue stores, 8 no MFMA
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue shared by the convolution kernels.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Workgroup id -> (M tile, N tile, K slice).  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, Workgroup dispatch), each with a private 4 MiB L2: with the identity map every XCD streams the
// WHOLE input.  The remap gives the ids sharing an XCD (equal id % 8) one contiguous run of logical tiles, N tile fastest:
// the N tiles of an M tile and the next M tiles (halo rows of a 3x3) then hit in that XCD's L2.  Bijective for any grid
// size (cdna_hip_programming.md, "XCD swizzle must be bijective"); a speed choice only.
struct ConvTile { int mt, nt, z; };
__device__ __forceinline__ ConvTile conv_tile_of_block(const ConvArgs& p) {
    ConvTile t;
    if (!p.xcd_swizzle) {
        t.mt = blockIdx.x; t.nt = blockIdx.y; t.z = blockIdx.z;
        return t;
    }
    const unsigned gm = gridDim.x, gn = gridDim.y, nwg = gm * gn * gridDim.z;
    const unsigned w = blockIdx.x + gm * (blockIdx.y + gn * blockIdx.z);
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = w & 7;
    const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    const unsigned mn = L / gn;
    t.nt = (int)(L - mn * gn);
    t.z = (int)(mn / gm);
    t.mt = (int)(mn - (unsigned)t.z * gm);
    return t;
}

// y = acc + bias (+ residual) ; y = y*out_scale + out_shift (optional) ; relu (optional); split-K slices store raw partials.
//
// The accumulators go through a wave-private LDS scratch ([32][32] floats = 4 KiB per wave; the caller provides its LDS
// byte address and has made sure no wave still reads the LDS it overlays) so that every lane ends up with FOUR CONSECUTIVE
// CHANNELS of one pixel: residual / bias / BN operands are read and y is written 16 B per lane, 128 B per 8 lanes (the
// C/D map itself gives 4 B per lane; measured, that form left the store path -- not the MFMAs -- bounding the low-K 1x1
// layers).  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  DS operations of one wave
// execute in order, so the write -> read -> overwrite sequence on the private scratch needs no barrier.  The LDS accesses
// are inline asm: compiler-visible LDS accesses next to in-flight LDS-DMA get an `s_waitcnt vmcnt(0)` from the compiler
// (see conv_ring_kernel.h), which would drain the next tile's prefetch.  Requires Cout % 4 == 0.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SN_EPI_WAVE_BYTES = 32 * 32 * 4;

#define SN_EPI_W(r) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(wr), "v"(a[r]), "n"((((r) & 3) + 8 * ((r) >> 2)) * 128) : "memory")
#define SN_EPI_R(q, dst) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(rd), "n"((q) * 1024) : "memory")

// NOSPLIT: p.splitk > 1 describes a split INSIDE the workgroup (the accumulators handed in are already the full sum): full epilogue.
// OFFS (the ring kernel's PRO form, a tile of the second tower of a pair): y and the residual live y_off / res_off elements
// further on; out_floor is not a floor there.
template <int TM, int TN, bool NOSPLIT = false, bool OFFS = false>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[TM][TN], const ConvArgs& p, int mw0, int nw0, int lane, int z,
                                              unsigned scratch, long y_off = 0, long res_off = 0) {
    const bool split = !NOSPLIT && p.splitk > 1;
    float* const outp = split ? p.partial + (size_t)z * p.M * p.Cout : (OFFS ? p.y + y_off : p.y);
    const bool has_res = !split && p.residual != nullptr;
    const bool has_obn = !split && p.out_scale != nullptr;
    const bool has_bias = !split && p.bias != nullptr;
    const bool relu = !split && p.relu_out;
    const bool res_plain = p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo;
    const unsigned wr = scratch + (unsigned)(((4 * (lane >> 5)) * 32 + (lane & 31)) * 4);   // C/D role
    const int rrow = lane >> 3, rc4 = (lane & 7) * 4;                                        // row-major role: rows rrow + 8q
    const unsigned rd = scratch + (unsigned)((rrow * 32 + rc4) * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int mrow[4];
        unsigned roff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mrow[q] = mw0 + i * 32 + rrow + 8 * q;
            const int m = min(mrow[q], p.M - 1);
            if (!has_res || res_plain) {
                roff[q] = (unsigned)(m * p.res_ld);
            } else {
                const int img = sn_fastdiv(m, p.div_hw_mul, p.div_hw_shift);
                const int rr = m - img * (p.Ho * p.Wo);
                const int oy = sn_fastdiv(rr, p.div_w_mul, p.div_w_shift), ox = rr - oy * p.Wo;
                roff[q] = (unsigned)(((img * p.res_H + oy * p.res_stride) * p.res_W + ox * p.res_stride) * p.res_ld);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nw0 + j * 32 + rc4;
            const bool ncol = n < p.Cout;
            const int nc = min(n, p.Cout - 4);
            float4 rv[4];
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4*>((OFFS ? p.residual + res_off : p.residual) + (roff[q] + (unsigned)nc));
            }
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bv = has_bias ? *reinterpret_cast<const float4*>(p.bias + nc) : zero4;
            const float4 os = has_obn ? *reinterpret_cast<const float4*>(p.out_scale + nc) : zero4;
            const float4 ob = has_obn ? *reinterpret_cast<const float4*>(p.out_shift + nc) : zero4;
            const bool has_floor = !OFFS && has_obn && p.out_floor != nullptr;
            const float4 fl = has_floor ? *reinterpret_cast<const float4*>(p.out_floor + nc) : zero4;
            {
                const f32x16 a = acc[i][j];
                SN_EPI_W(0); SN_EPI_W(1); SN_EPI_W(2); SN_EPI_W(3); SN_EPI_W(4); SN_EPI_W(5); SN_EPI_W(6); SN_EPI_W(7);
                SN_EPI_W(8); SN_EPI_W(9); SN_EPI_W(10); SN_EPI_W(11); SN_EPI_W(12); SN_EPI_W(13); SN_EPI_W(14); SN_EPI_W(15);
            }
            f32x4 t[4];
            SN_EPI_R(0, t[0]); SN_EPI_R(1, t[1]); SN_EPI_R(2, t[2]); SN_EPI_R(3, t[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(t[q].x, t[q].y, t[q].z, t[q].w);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (has_res) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
                if (has_obn) {
                    v.x = __builtin_fmaf(v.x, os.x, ob.x); v.y = __builtin_fmaf(v.y, os.y, ob.y);
                    v.z = __builtin_fmaf(v.z, os.z, ob.z); v.w = __builtin_fmaf(v.w, os.w, ob.w);
                }
                if (has_floor) { v.x = fmaxf(v.x, fl.x); v.y = fmaxf(v.y, fl.y); v.z = fmaxf(v.z, fl.z); v.w = fmaxf(v.w, fl.w); }
                else if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (!(CONV_ABLATE & 4)) {
                    if (ncol && mrow[q] < p.M) *reinterpret_cast<float4*>(outp + ((size_t)mrow[q] * p.Cout + n)) = v;
                } else if (v.x == 123.456f) outp[0] = v.x;
            }
        }
    }
}

// ---- exact float32 products on the bf16 matrix pipe (BF16 template value 2: six partial products, 3: all nine) ------------------
// gfx950 multiplies f32 operands at 1/16 of its bf16 MFMA rate (v_mfma_f32_32x32x2_f32: 64 cycles for 2 k; v_mfma_f32_32x32x16_bf16:
// 32 cycles for 16 k).  A float32 is EXACTLY the sum of three bf16 numbers: h = bf16(x) (round to nearest: 8 significant bits),
// m = bf16(x - h) (the difference is exact in f32 and has at most 16 significant bits), l = x - h - m (exact, at most 8 bits, so its
// conversion is exact too) -- barring underflow of l below 2^-126 and non-finite x.  Every bf16 x bf16 product is exact in the f32
// accumulator's input precision, so a.b = sum over the nine (a_i, b_j) pairs with f32 accumulation is the f32 product; the three pairs
// (m,l), (l,m), (l,l) are below 2^-23 |a.b| together and are dropped in the six-product form (same order as ONE f32 rounding of the
// product).  Small terms are accumulated first.  Cost: 5.5 VALU per operand element and 6 (9) x 32 cycles of matrix pipe per 32-deep
// stage instead of 1024.
typedef __bf16 sn_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sn_bf16x8 __attribute__((ext_vector_type(8)));
struct SnSplit3 { sn_bf16x4 h, m, l; };
typedef __bf16 sn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// One level for a pair of floats: p = (bf16(x0), bf16(x1)) packed (one v_cvt_pk_bf16_f32), r = x - float(p): the two floats of the
// packed pair are one shift and one mask (the generic bf16 -> f32 conversion of the pair costs two more conversions).
__device__ __forceinline__ float sn_sub_f32(float a, float b) {
    // plain v_sub_f32: the compiler pairs neighbouring f32 subtractions into v_pk_add_f32, which costs the matrix pipe more issue
    // slots than the two scalar forms it replaces (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever beside MFMAs")
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned sn_split_level(float& x0, float& x1) {
    const f32x2 v = {x0, x1};
    const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, sn_bf16x2));
    x0 = sn_sub_f32(x0, __builtin_bit_cast(float, p << 16));
    x1 = sn_sub_f32(x1, __builtin_bit_cast(float, p & 0xffff0000u));
    return p;
}
__device__ __forceinline__ unsigned sn_pack_bf16(float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sn_bf16x2));
}
// The three planes of the lane's 8 k values of one 16-deep k group (lo = k 4g.., hi = k 8 + 4g..), each as four packed registers.
typedef unsigned sn_u32x4 __attribute__((ext_vector_type(4)));
struct SnPlanes { sn_u32x4 h, m, l; };
__device__ __forceinline__ SnPlanes sn_split3_pair(const f32x4 lo, const f32x4 hi) {
    float a = lo.x, b = lo.y, c = lo.z, d = lo.w, e = hi.x, f = hi.y, g = hi.z, i = hi.w;
    SnPlanes s;
    s.h.x = sn_split_level(a, b); s.h.y = sn_split_level(c, d); s.h.z = sn_split_level(e, f); s.h.w = sn_split_level(g, i);
    s.m.x = sn_split_level(a, b); s.m.y = sn_split_level(c, d); s.m.z = sn_split_level(e, f); s.m.w = sn_split_level(g, i);
    s.l.x = sn_pack_bf16(a, b); s.l.y = sn_pack_bf16(c, d); s.l.z = sn_pack_bf16(e, f); s.l.w = sn_pack_bf16(g, i);
    return s;
}
__device__ __forceinline__ SnSplit3 sn_split3(const f32x4 x) {
    float a = x.x, b = x.y, c = x.z, d = x.w;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 h, m, l;
    h.x = sn_split_level(a, b); h.y = sn_split_level(c, d);
    m.x = sn_split_level(a, b); m.y = sn_split_level(c, d);
    const f32x2 v0 = {a, b}, v1 = {c, d};
    l.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, sn_bf16x2));
    l.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, sn_bf16x2));
    SnSplit3 s;
    s.h = __builtin_bit_cast(sn_bf16x4, h); s.m = __builtin_bit_cast(sn_bf16x4, m); s.l = __builtin_bit_cast(sn_bf16x4, l);
    return s;
}
#define SN_CAT8(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
// acc += A . B for one 16-deep k group: (a0, a1) / (b0, b1) are the two 4-deep halves of the lane's 8 k values.
template <int NPROD>
__device__ __forceinline__ void sn_mfma_split3(f32x16& acc, const SnSplit3& a0, const SnSplit3& a1, const SnSplit3& b0, const SnSplit3& b1) {
    const sn_bf16x8 ah = SN_CAT8(a0.h, a1.h), am = SN_CAT8(a0.m, a1.m), al = SN_CAT8(a0.l, a1.l);
    const sn_bf16x8 bh = SN_CAT8(b0.h, b1.h), bm = SN_CAT8(b0.m, b1.m), bl = SN_CAT8(b0.l, b1.l);
    if constexpr (NPROD == 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}
// The same six products with B already split (planes as 8 x bf16 in four registers each: the packed weight image).
__device__ __forceinline__ void sn_mfma_split3_pk(f32x16& acc, const SnPlanes& a, const f32x4 bh_, const f32x4 bm_, const f32x4 bl_) {
    const sn_bf16x8 ah = __builtin_bit_cast(sn_bf16x8, a.h), am = __builtin_bit_cast(sn_bf16x8, a.m), al = __builtin_bit_cast(sn_bf16x8, a.l);
    const sn_bf16x8 bh = __builtin_bit_cast(sn_bf16x8, bh_), bm = __builtin_bit_cast(sn_bf16x8, bm_), bl = __builtin_bit_cast(sn_bf16x8, bl_);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// MODE 0: no padding and no dilation (1x1 convs, any stride): every tap of every row < M is in frame -> no masks.
// MODE 1: zero padding: a per-row bit mask (one bit per filter tap, built once) says which taps are in frame.
// MODE 2: dilated input (dgrad of a strided conv): validity and address are recomputed per tap (slow path).
template <int BM, int BN, int BK, int WM, int WN, int MODE, int NBUF = 2, int BF16 = 0>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs p) {
#define SN_IGEMM_SAFE_BASE 0u
#define SN_IGEMM_VALID_PTR p.x
#include "conv_igemm_body.h"
#undef SN_IGEMM_SAFE_BASE
#undef SN_IGEMM_VALID_PTR
}

// The siamese pair of a training step as one launch (ConvPair, conv.h): workgroups whose tile lies in the second tower shift the
// five tower-owned pointers (wave-uniform: scalar adds on the kernel arguments) and run the same body.
template <int BM, int BN, int BK, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_f32_pair_kernel(const ConvArgs p0, const ConvPair pr) {
    constexpr int NBUF = 2, BF16 = 0;
    ConvArgs p = p0;
    unsigned pair_safe_base = 0u;        // out-of-frame taps read (and discard) a "safe" address: it must lie in THIS tower's tensor
    if (conv_tile_of_block(p0).mt * BM >= pr.m_tower) {
        p.x += pr.dx;
        p.y += pr.dy;
        if (p.residual != nullptr) p.residual += pr.dres;
        if (p.in_scale != nullptr) { p.in_scale += pr.dscale; p.in_shift += pr.dscale; }
        pair_safe_base = (unsigned)pr.x_tower_floats;
    }
#define SN_IGEMM_SAFE_BASE pair_safe_base
#define SN_IGEMM_VALID_PTR p.w               /* (p.x of a second-tower tile points N images in front of that tower's tensor) */
#include "conv_igemm_body.h"
#undef SN_IGEMM_SAFE_BASE
#undef SN_IGEMM_VALID_PTR
}
