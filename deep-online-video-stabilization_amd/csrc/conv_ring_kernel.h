// Implicit-GEMM convolution, LDS-DMA ring variant (gfx950, exact float32 MFMA).
//
// For convolutions whose A operand needs no prologue (the input is already activated, or is a raw gradient): tiles go
// HBM/L2 -> LDS directly with `global_load_lds` (16 B per lane, no VGPR staging, no ds_write, no VALU on the data),
// through a ring of NS stages.  Measured motivation: with register staging the K-loop of the v1
// kernel is capped by (a) VALU work competing with the f32 MFMAs and (b) one K-step of load latency hiding.
//
//   block  : 64 x 64 output tile, 256 threads = 4 waves (2 x 2 of 32 x 32), BK = 32, NS = 3 stages of 16 KiB
//   stage  : A [64 rows][32 floats] then B [64 rows][32 floats], rows UNPADDED (a DMA instruction writes 1 KiB = 8 rows
//            contiguously); bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle of the 16-B
//            chunk index, chunk_phys = chunk ^ ((row >> 1) & 7), applied on the SOURCE address of the DMA lane and on
//            the read address (both sides or neither: the LDS image itself is lane-linear)
//   DMA    : per stage each wave issues 2 A + 2 B instructions (8 rows each); lane l -> row 8g + (l>>3), chunk l&7
//   sync   : counted `s_waitcnt vmcnt(4)` (the next stage stays in flight) + ONE raw s_barrier per K-step, which both
//            publishes stage k and retires every wave's reads of stage k-1 (the slot the new DMA overwrites)
//   padding: out-of-frame taps read a 16-B zero page (no prologue => zeros stay zeros, nothing to mask afterwards)
#pragma once
#include "conv_kernel.h"
#include <type_traits>

#ifndef RING_ABLATE
#define RING_ABLATE 0          // probe-only bit mask (tools/ring_probe.hip): 1 no DMA, 2 no barrier, 4 no ds_read, 8 no MFMA
#endif

typedef __attribute__((address_space(3))) void* sn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* sn_gbl_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_conv_zero_page[8];      // zero-initialised device storage

template <int MODE /* 0: no padding, 1: zero padding */, int NS>
__global__ __launch_bounds__(256) void conv_ring_f32_kernel(const ConvArgs p) {
    constexpr int BM = 64, BN = 64, BK = 32;
    constexpr int STAGE = (BM + BN) * BK;                  // floats
    static_assert(NS == 3, "the K loop is unrolled over a ring of exactly 3 stages");
    __shared__ __attribute__((aligned(16))) float ring[NS * STAGE];

    CONV_STAMP_AT(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const ConvTile tile = conv_tile_of_block(p);
    const int m0 = tile.mt * BM, n0 = tile.nt * BN;
    const int cin_steps = p.Cin / BK;
    const int total_steps = p.KH * p.KW * cin_steps;
    const int ks_begin = tile.z * p.steps_per_split;
    const int ks_end = min(total_steps, ks_begin + p.steps_per_split);
    const int nsteps = ks_end - ks_begin;

    // ---- DMA lane mapping: instruction g in {wave, wave + 4} covers tile rows 8g .. 8g+7
    const int rr = lane >> 3, cp = lane & 7;
    const int pad_off = (p.pad * p.W + p.pad) * p.Cin;
    unsigned a_voff[2], w_voff[2];
    unsigned long long a_mask[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 8 * (wave + 4 * t) + rr;
        const int cl = cp ^ ((row >> 1) & 7);              // logical chunk this lane fetches (swizzle on the source side)
        const int m = m0 + row;
        const int mc = min(m, p.M - 1);
        const int img = mc / (p.Ho * p.Wo);
        const int r = mc - img * (p.Ho * p.Wo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_voff[t] = (unsigned)(((img * p.H + iy0 + p.pad) * p.W + ix0 + p.pad) * p.Cin + cl * 4);
        unsigned long long mk = 0;
        if (MODE == 1 && m < p.M) {
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W)
                        mk |= 1ull << (kh * p.KW + kw);
        }
        a_mask[t] = mk;
        const int n = min(n0 + row, p.Cout - 1);           // rows >= Cout are never stored
        w_voff[t] = (unsigned)(n * p.K + cl * 4);
    }
    // ---- fragment read addresses (LDS bytes, stage 0) of the 4 kk sub-steps: lane (i = lane&31, h = lane>>5) reads
    //      logical chunk 2kk+h; the stage is selected by the immediate offset of the ds_read
    unsigned a_frag[4], b_frag[4];
    {
        const unsigned lds_base = (unsigned)(size_t)(sn_lds_ptr_t)ring;
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a_frag[kk] = lds_base + 4 * (ra * BK + (((2 * kk + h) ^ ((ra >> 1) & 7)) << 2));
            b_frag[kk] = lds_base + 4 * (BM * BK + rb * BK + (((2 * kk + h) ^ ((rb >> 1) & 7)) << 2));
        }
    }

    // ---- K-iteration state of the next stage to issue (wave-uniform running pointers, see conv_kernel.h)
    int l_tap = ks_begin / cin_steps;
    int l_c0 = (ks_begin - l_tap * cin_steps) * BK;
    int l_kh = l_tap / p.KW, l_kw = l_tap - (l_tap / p.KW) * p.KW;
    const float* xb = p.x + ((l_kh * p.W + l_kw) * p.Cin + l_c0 - pad_off);
    const float* wb = p.w + (l_tap * p.Cin + l_c0);
    const int row_jump = (p.W - p.KW) * p.Cin;
    const float* const zero_page = g_conv_zero_page;

    auto issue = [&](float* const st) {
        if (RING_ABLATE & 1) return;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* src = xb + a_voff[t];
            if (MODE == 1) src = ((a_mask[t] >> l_tap) & 1ull) ? src : zero_page;
            __builtin_amdgcn_global_load_lds((sn_gbl_ptr_t)src, (sn_lds_ptr_t)(st + (wave + 4 * t) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            __builtin_amdgcn_global_load_lds((sn_gbl_ptr_t)(wb + w_voff[t]),
                                             (sn_lds_ptr_t)(st + BM * BK + (wave + 4 * t) * 256), 16, 0, 0);
        xb += BK;
        wb += BK;
        l_c0 += BK;
        if (l_c0 == p.Cin) {
            l_c0 = 0;
            ++l_tap;
            if (++l_kw == p.KW) { l_kw = 0; ++l_kh; xb += row_jump; }
        }
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

    // prologue: two stages in flight
    if (0 < nsteps) issue(ring);
    if (1 < nsteps) issue(ring + STAGE);
    CONV_STAMP_AT(1);

    // The fragment reads are inline asm on purpose: the compiler's waitcnt insertion cannot tell which DMA a ds_read
    // depends on and puts `s_waitcnt vmcnt(0)` in front of compiler-visible LDS loads, draining the stage that was just
    // put in flight.  Waits are therefore explicit: vmcnt before the barrier, lgkmcnt (tied to the fragment registers
    // through "+v" so the MFMAs cannot move above them) before each MFMA group.
#define SN_DS_READ(dst, addr, off)                                                                        \
    do { if (!(RING_ABLATE & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off)); } while (0)
#define SN_LGKM_WAIT(n, a, b) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b))
#define SN_MFMA4(a, b)                                                                        \
    if (!(RING_ABLATE & 8)) {                                                                 \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[0][0], 0, 0, 0);       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[0][0], 0, 0, 0);       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[0][0], 0, 0, 0);       \
    } else {                                                                                  \
        acc[0][0][0] += a.x * b.x;                                                            \
    }

    auto step = [&](auto steady, auto slot_c, int k) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int ISSUE_SLOT = (SLOT + 2) % 3;
        constexpr int OFF = SLOT * STAGE * 4;
        // stage k has landed for THIS wave when at most the next stage's DMAs are outstanding (4 per stage per wave)
        if (decltype(steady)::value || k + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(RING_ABLATE & 2)) __builtin_amdgcn_s_barrier();   // stage k complete for all waves; stage k-1 no longer read
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(steady)::value || k + 2 < nsteps) issue(ring + ISSUE_SLOT * STAGE);   // overwrites stage k-1's slot
        f32x4 a0, b0, a1, b1;
        if (RING_ABLATE & 4) { a0 = b0 = a1 = b1 = f32x4{1.f, 2.f, 3.f, 4.f}; }
        SN_DS_READ(a0, a_frag[0], OFF);
        SN_DS_READ(b0, b_frag[0], OFF);
        SN_DS_READ(a1, a_frag[1], OFF);
        SN_DS_READ(b1, b_frag[1], OFF);
        SN_LGKM_WAIT(2, a0, b0);
        SN_MFMA4(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        SN_DS_READ(a0, a_frag[2], OFF);
        SN_DS_READ(b0, b_frag[2], OFF);
        SN_LGKM_WAIT(2, a1, b1);
        SN_MFMA4(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        SN_DS_READ(a1, a_frag[3], OFF);
        SN_DS_READ(b1, b_frag[3], OFF);
        SN_LGKM_WAIT(2, a0, b0);
        SN_MFMA4(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        SN_LGKM_WAIT(0, a1, b1);
        SN_MFMA4(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };
    using steady_t = std::integral_constant<bool, true>;
    using tail_t = std::integral_constant<bool, false>;
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    using s2 = std::integral_constant<int, 2>;
    int k = 0;
    for (; k + 4 < nsteps; k += 3) {
        step(steady_t{}, s0{}, k);
        step(steady_t{}, s1{}, k + 1);
        step(steady_t{}, s2{}, k + 2);
    }
    // tail: the last 1..4 steps (at most two of them still issue)
    if (k < nsteps) step(tail_t{}, s0{}, k);
    if (k + 1 < nsteps) step(tail_t{}, s1{}, k + 1);
    if (k + 2 < nsteps) step(tail_t{}, s2{}, k + 2);
    if (k + 3 < nsteps) step(tail_t{}, s0{}, k + 3);
#undef SN_DS_READ
#undef SN_LGKM_WAIT
#undef SN_MFMA4
    CONV_STAMP_AT(2);
    __syncthreads();                                       // every wave is done with the last stage
    conv_epilogue<1, 1>(acc, p, m0 + wm * 32, n0 + wn * 32, lane, tile.z, ring + wave * SN_EPI_WAVE_FLOATS);
    CONV_STAMP_AT(3);
}
