// Back-to-back GEMM: a bottleneck unit's 3x3 `conv2` and its 1x1 `conv3` as ONE launch (gfx950, exact float32 MFMA).
//
// slim's bottleneck_v2 (called at s_net_bundle_nobm.py:252-253): residual = conv3(relu(bn2(conv2(a1)))) with conv2 3x3 over
// d_b = 64 / 128 channels and conv3 1x1 to 4 d_b channels (+ bias + shortcut).  Launched one after the other (conv_ring_kernel.h)
// the `conv2` output makes a round trip through memory (14.7 / 7.4 MB at 720p) and the `conv3` launch -- two or four K-steps
// of matrix work per 16 KB of epilogue -- is bounded by its stores and by launch fill / drain.  Here a
// workgroup owns a 64-row M tile over the FULL width of conv2 (N = d_b = 32 NWN), so the activated conv2 tile never leaves the CU:
//
//   phase 0 (S2 = 9 NWN K-steps): the ring kernel's MODE 1 loop -- A [64][32] and B [32 NWN][32] stages by LDS-DMA, 3 stages in
//            flight -- with the MFMA operands SWAPPED: acc^T = W2 * A^T.  In the transposed C/D map a lane holds pixel row
//            m = lane & 31 and, in registers 4q .. 4q+3, the FOUR CONSECUTIVE channels n = 8q + 4h .. (h = lane >> 5): exactly one
//            16-byte chunk of the A-operand stage layout.  The tile's end applies bn2 + ReLU on the accumulators and PARKS them
//            with four ds_write_b128 per lane as NWN ready-made A stages ([ks][64 rows][32 floats], same XOR swizzle): no
//            transposition pass, no scratch.
//   phase 1 (S3 = (Cout3 / 32 NWN) * NWN steps): conv3 walks its N in chunks of 32 NWN channels, NWN K-steps each.  A fragments
//            come from the park, B stages (W3 rows of the chunk, [32 NWN][32]) from the SAME ring -- they do not depend on phase
//            0, so the producer side streams them three stages ahead like any other stage, across the phase and tile boundaries.
//            conv3 runs with swapped operands too (one step body for both phases); each chunk ends with the ring kernel's full
//            epilogue (bias + residual + the next unit's folded BN + ReLU / floor, 16 B per lane through a wave-private LDS
//            transpose -- from the transposed layout that is four ds_write_b128 instead of sixteen ds_write_b32).
//
// Differences from conv_ring_kernel.h that matter for speed:
//   * the epilogue scratch is NOT a ring slot (dedicated 4 KB per wave): a chunk's end neither stalls the producer nor needs the
//     extra barrier, which matters when a "tile" is only NWN = 2 / 4 steps long;
//   * a chunk's residual rows and per-channel vectors are fetched into (pinned) registers at the chunk's FIRST step by inline-asm
//     loads: the epilogue itself has no load -> use latency (1.2 of the 1.5 us it took with the loads inside).
//   NWN = 2: d_b = 64 (block 1), 4 waves, 80 KB of LDS (ring 48 + park 16 + scratch 16): two workgroups per CU.
//   NWN = 4: d_b = 128 (block 2), 8 waves (2 x 4 of 32 x 32), 136 KB (72 + 32 + 32): one workgroup per CU.
#pragma once
#include "conv_ring_kernel.h"

struct B2bArgs {
    ConvArgs c2;     // the 3x3 convolution (pad 1, stride 1 or 2, Cin == Cout == 32 NWN, no bias, no residual, no prologue):
                     //   x / x_ld / w / geometry as for conv_ring_f32_kernel<1>; out_scale / out_shift = folded bn2 (ReLU implied); y unused
    ConvArgs c3;     // the 1x1 convolution of the parked tile (Cin == c2.Cout, M == c2.M): w, bias, residual (+ res_*), out_scale /
                     //   out_shift / out_floor / relu_out, y; planned by conv_plan() (M, div_*, res_ld), splitk == 1
};

#ifndef B2B_ABLATE
#define B2B_ABLATE 0           // probe-only bit mask (tools/b2b_probe.hip): 1 chunk epilogue without its global loads / stores, 2 no chunk
#endif                         //   epilogue at all, 4 no park, 8 no MFMA
// conv3's chunk epilogue: conv_epilogue<1, 1> (conv_kernel.h) -- same arithmetic, same order: + bias, + residual, * out_scale +
// out_shift, floor / ReLU -- restated for TRANSPOSED accumulators (conv3 runs with swapped MFMA operands like conv2, so that one
// step body serves both phases): lane (m = lane & 31, h = lane >> 5) holds, in registers 4q .. 4q+3, channels 8q + 4h .. + 3 of
// pixel row m.  Four ds_write_b128 put them into the wave's [32 pixels][32 channels] scratch (16-byte chunk index XOR (m & 7):
// conflict-free for the writes and for the row-major ds_read_b128 that follow), then every lane owns FOUR CONSECUTIVE CHANNELS of
// four pixels as in conv_epilogue: 16-byte stores, 128 B per 8 lanes.  The residual rows and the per-channel vectors are NOT
// loaded here: the kernel fetched them into registers a chunk earlier (inline-asm loads it waits for with a counted vmcnt), so
// the epilogue has no memory latency of its own -- measured, the load -> use wait was 1.2 of the 1.5 us a chunk epilogue took.
// Vector-memory operations issued here: the 4 stores of an interior tile (a ragged tile may skip some: the caller counts 0).
struct B2bEpiOperands { f32x4 rv[4], bv, os, ob, fl; };
// part 1: transposition + arithmetic -> out[q] = the four channels n .. n+3 of pixel row rrow + 8q (no vector-memory operation)
__device__ __forceinline__ void b2b_chunk_compute(const f32x16& a, const ConvArgs& p, int lane, unsigned scratch, const B2bEpiOperands& o,
                                                  f32x4 (&out)[4]) {
    const bool has_res = p.residual != nullptr, has_obn = p.out_scale != nullptr, has_bias = p.bias != nullptr;
    const bool has_floor = has_obn && p.out_floor != nullptr, relu = p.relu_out != 0;
    const int tm = lane & 31, th = lane >> 5;                                                // transposed C/D role
    const int rrow = lane >> 3, rc = lane & 7;                                               // row-major role: rows rrow + 8q, chunk rc
    const unsigned rd = scratch + (unsigned)(rrow * 128 + ((rc ^ rrow) << 4));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        const unsigned wr = scratch + (unsigned)(tm * 128 + (((2 * q + th) ^ (tm & 7)) << 4));
        asm volatile("ds_write_b128 %0, %1" ::"v"(wr), "v"(v) : "memory");
    }
    f32x4 t[4];
    SN_EPI_R(0, t[0]); SN_EPI_R(1, t[1]); SN_EPI_R(2, t[2]); SN_EPI_R(3, t[3]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
    // The launch's flags pick ONE of six straight-line bodies (folded BN or not) x (floor / ReLU / nothing); inside a body there
    // is no branch.  (With the flags tested per element the compiler emitted 32 scalar branches per chunk: 0.85 us of the 1.1 us a
    // chunk epilogue took.)  An absent bias / residual is a zero operand: x + 0 is exact.
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = has_bias ? o.bv : zero4;
    f32x4 rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rv[q] = has_res ? o.rv[q] : zero4;
    auto body = [&](auto obn_c, auto act_c) {
        constexpr bool OBN = decltype(obn_c)::value;
        constexpr int ACT = decltype(act_c)::value;                  // 0 none, 1 ReLU, 2 per-channel floor
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = t[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = (v[e] + bv[e]) + rv[q][e];
                if (OBN) x = __builtin_fmaf(x, o.os[e], o.ob[e]);
                if (ACT == 2) x = fmaxf(x, o.fl[e]);
                else if (ACT == 1) x = fmaxf(x, 0.f);
                v[e] = x;
            }
            out[q] = v;
        }
    };
    using T = std::true_type; using F = std::false_type;
    using A0 = std::integral_constant<int, 0>; using A1 = std::integral_constant<int, 1>; using A2 = std::integral_constant<int, 2>;
    if (has_obn) { if (has_floor) body(T{}, A2{}); else if (relu) body(T{}, A1{}); else body(T{}, A0{}); }
    else { if (relu) body(F{}, A1{}); else body(F{}, A0{}); }
}
// part 2: the four 16-byte stores (128 B per 8 lanes)
__device__ __forceinline__ void b2b_chunk_store(const f32x4 (&out)[4], const ConvArgs& p, int mw0, int nw0, int lane) {
    const int rrow = lane >> 3, n = nw0 + (lane & 7) * 4;                                   // (n < Cout: chunks are whole)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mrow = mw0 + rrow + 8 * q;
        if (!(B2B_ABLATE & 1)) { if (mrow < p.M) *reinterpret_cast<f32x4*>(p.y + ((size_t)mrow * p.Cout + n)) = out[q]; }
        else if (out[q][0] == 123.456f) p.y[0] = out[q][1];
    }
}

template <int NWN /* 32-column wave tiles across conv2's N: 2 (d_b = 64) or 4 (d_b = 128) */>
__global__ __launch_bounds__(128 * NWN) void conv_b2b_f32_kernel(const B2bArgs P) {
    static_assert(NWN == 2 || NWN == 4, "d_b = 64 or 128");
    constexpr int NW = 2 * NWN;                            // waves: 2 (M) x NWN (N) of 32 x 32
    constexpr int BM = 64, BK = 32, BN2 = 32 * NWN;
    constexpr int A_BYTES = BM * BK * 4;                   // 8 KiB
    constexpr int B_BYTES = BN2 * BK * 4;                  // 4 KiB x NWN
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PARK_BYTES = NWN * A_BYTES;              // the activated conv2 tile as NWN A stages
    constexpr int SCR_BYTES = NW * SN_EPI_WAVE_BYTES;
    constexpr int A_PER_WAVE = 8 / NW;                     // A DMA instructions (8 rows each) per wave and conv2 stage: 2 / 1
    constexpr int CNT2 = A_PER_WAVE + 2;                   // DMAs per wave in a conv2 stage (a conv3 stage: 2)
    __shared__ __attribute__((aligned(16))) float lds[(3 * STAGE_BYTES + PARK_BYTES + SCR_BYTES) / 4];

    const ConvArgs& p = P.c2;
    const ConvArgs& q3 = P.c3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int t_first = blockIdx.x, t_stride = gridDim.x, t_end = tiles_m;
    if (p.xcd_swizzle && gridDim.x >= 8) {                 // workgroups of one XCD take one contiguous run of M tiles (conv_ring_kernel.h)
        const int x = blockIdx.x & 7, Gt = (int)gridDim.x;
        t_stride = (Gt - x + 7) >> 3;
        const int before = x * (Gt >> 3) + min(x, Gt & 7);
        const int lo = (int)((long)tiles_m * before / Gt), hi = (int)((long)tiles_m * (before + t_stride) / Gt);
        t_first = lo + (blockIdx.x >> 3);
        t_end = hi;
    }
    const int S2 = p.KH * p.KW * NWN;                      // conv2 K-steps (Cin == BN2: NWN per tap)
    const int S3 = (q3.Cout / BN2) * NWN;                  // conv3: chunks of BN2 output channels x NWN K-steps
    const int T = S2 + S3;
    int my_tiles = 0;
    for (int t = t_first; t < t_end; t += t_stride) ++my_tiles;
    int remaining = my_tiles * T;
    if (remaining == 0) return;

    const unsigned lds_base = (unsigned)(size_t)(sn_lds_ptr_t)lds;
    const unsigned park_base = lds_base + 3u * STAGE_BYTES;
    const unsigned scr_base = park_base + (unsigned)PARK_BYTES + (unsigned)wave * (unsigned)SN_EPI_WAVE_BYTES;

    // ---- bn2 (+ ReLU) of the transposed accumulators: channel n = wn*32 + 8q + 4h + e lives in register 4q + e
    f32x4 sc[4], sh[4];
    {
        const int n_base = wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4 a = *reinterpret_cast<const float4*>(p.out_scale + n_base + 8 * qq);
            const float4 b = *reinterpret_cast<const float4*>(p.out_shift + n_base + 8 * qq);
            sc[qq] = f32x4{a.x, a.y, a.z, a.w};
            sh[qq] = f32x4{b.x, b.y, b.z, b.w};
        }
        // (landed before the first DMA is issued: the compiler does not see the inline-asm DMAs and must not carry these loads as
        //  pending into the loop, where its own wait would drain the ring)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) asm volatile("" : "+v"(sc[qq]), "+v"(sh[qq]));
    }

    // ---- DMA lane roles: one instruction = 8 tile rows x 128 B; lane l -> row 8g + (l >> 3), 16-B chunk l & 7
    const int rr = lane >> 3, cp = lane & 7;
    const int ld = p.x_ld;
    const int pad_off = (p.pad * p.W + p.pad) * ld;
    const int row_jump = (p.W - p.KW) * ld;
    const float* const zero_page = g_conv_zero_page;

    // ---- producer state
    int pt = t_first;                                      // tile whose stages are being issued; >= t_end: exhausted
    int ps = 0;                                            // stage index inside the tile: [0, S2) conv2, [S2, T) conv3
    unsigned a_voff[A_PER_WAVE];
    unsigned long long a_mask[A_PER_WAVE];
    unsigned w2_voff[2], w3_voff[2];
    int l_tap = 0, l_c0 = 0, l_kw = 0;
    const float *xb = p.x, *wb = p.w;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                          // B rows of this wave: 8 (wave + NW t) + rr, the same in every tile
        const int row = 8 * (wave + NW * t) + rr;
        const int cl = cp ^ ((row >> 1) & 7);
        w2_voff[t] = 4u * (unsigned)(row * p.K + cl * 4);
        w3_voff[t] = 4u * (unsigned)(row * BN2 + cl * 4);
    }
    auto producer_setup = [&]() {
        const int m0 = pt * BM;
#pragma unroll
        for (int t = 0; t < A_PER_WAVE; ++t) {
            const int row = 8 * (wave + NW * t) + rr;
            const int cl = cp ^ ((row >> 1) & 7);
            const int m = m0 + row;
            const int mc = min(m, p.M - 1);
            const int img = sn_fastdiv(mc, p.div_hw_mul, p.div_hw_shift);
            const int r = mc - img * (p.Ho * p.Wo);
            const int oy = sn_fastdiv(r, p.div_w_mul, p.div_w_shift), ox = r - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_voff[t] = 4u * (unsigned)(((img * p.H + iy0 + p.pad) * p.W + ix0 + p.pad) * ld + cl * 4);
            unsigned long long mk = 0;
            if (m < p.M) {
                const int kh_lo = max(0, -iy0), kh_hi = min(p.KH, p.H - iy0);
                const int kw_lo = max(0, -ix0), kw_hi = min(p.KW, p.W - ix0);
                if (kh_hi > kh_lo && kw_hi > kw_lo) {
                    const unsigned long long rowbits = ((1ull << (kw_hi - kw_lo)) - 1ull) << kw_lo;
                    for (int kh = kh_lo; kh < kh_hi; ++kh) mk |= rowbits << (kh * p.KW);
                }
            }
            a_mask[t] = mk;
        }
        ps = 0; l_tap = 0; l_c0 = 0; l_kw = 0;
        xb = p.x - pad_off;
        wb = p.w;
    };
    const unsigned dma_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)wave * 1024u));
    auto dma16 = [&](const float* base_, unsigned voff_bytes, unsigned lds_byte) {
        const unsigned long long b64 = (unsigned long long)(size_t)base_;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const float* base = reinterpret_cast<const float*>((size_t)(((unsigned long long)hi << 32) | lo));
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(lds_byte) : "memory", "m0");
    };
    auto issue_a = [&](int slot, int t) {                  // conv2 stages only
        if (pt >= t_end || ps >= S2) return;
        const unsigned lds_byte = dma_base + (unsigned)(slot * STAGE_BYTES + t * NW * 1024);
        // out-of-frame taps read the zero page: a per-lane choice between two unrelated addresses needs the 64-bit form
        const float* src = ((a_mask[t] >> l_tap) & 1ull) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + a_voff[t]) : zero_page;
        __builtin_amdgcn_global_load_lds((sn_gbl_ptr_t)src, (sn_lds_ptr_t)(size_t)lds_byte, 16, 0, 0);
    };
    auto issue_b = [&](int slot, int t) {
        if (pt >= t_end) return;
        const unsigned lds_byte = dma_base + (unsigned)(slot * STAGE_BYTES + A_BYTES + t * NW * 1024);
        dma16(wb, ps < S2 ? w2_voff[t] : w3_voff[t], lds_byte);
    };
    auto issue_advance = [&]() {
        if (pt >= t_end) return;
        ++ps;
        if (ps == T) {
            pt += t_stride;
            if (pt < t_end) producer_setup();
            return;
        }
        if (ps > S2) {                                     // conv3: next K-step of the chunk, or the next chunk's rows
            const int j = ps - S2;
            wb = q3.w + (size_t)(j / NWN) * (BN2 * BN2) + (j % NWN) * BK;
            return;
        }
        if (ps == S2) { wb = q3.w; return; }               // conv2 -> conv3: chunk 0, K-step 0
        xb += BK;
        wb += BK;
        l_c0 += BK;
        if (l_c0 == p.Cin) {
            l_c0 = 0;
            ++l_tap;
            xb += ld - p.Cin;
            if (++l_kw == p.KW) { l_kw = 0; xb += row_jump; }
        }
    };
    auto issue_part = [&](int slot, int part) {
        if (part == 0) issue_a(slot, 0);
        else if (part == 1) { if constexpr (A_PER_WAVE == 2) issue_a(slot, 1); issue_b(slot, 0); }
        else if (part == 2) issue_b(slot, 1);
        else issue_advance();
    };
    auto issue = [&](int slot) { issue_part(slot, 0); issue_part(slot, 1); issue_part(slot, 2); issue_part(slot, 3); };

    // ---- consumer state
    int ct = t_first;                                      // tile being computed
    int cs = 0;                                            // its step index: [0, S2) conv2, [S2, T) conv3

    // ---- fragment read addresses (LDS bytes): lane (i = lane & 31, h = lane >> 5) reads logical chunk 2kk + h of its row
    unsigned a_frag[4], b_frag[4], park_wr[4];
    {
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned rel = 4u * (unsigned)(ra * BK + (((2 * kk + h) ^ ((ra >> 1) & 7)) << 2));
            a_frag[kk] = lds_base + rel;
            park_wr[kk] = park_base + (unsigned)(wn * A_BYTES) + rel;     // accumulator quad q = kk: K-step wn of the parked tile
            b_frag[kk] = lds_base + (unsigned)A_BYTES + 4u * (unsigned)(rb * BK + (((2 * kk + h) ^ ((rb >> 1) & 7)) << 2));
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    producer_setup();

    // Fragment reads / waits / MFMAs: the ring kernel's discipline (inline asm LDS reads on pinned registers, explicit lgkmcnt
    // waits tied to the fragment registers, every non-MFMA instruction right behind one MFMA).
    // (SN_RA0 .. SN_RB1: the pinned fragment registers v[100:115], defined by conv_ring_kernel.h)
#define B2B_RD(dst, reg, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" reg(dst) : "v"(addr), "n"(off))
#define B2B_WAIT0() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA0(a0), "+" SN_RB0(b0))
#define B2B_WAIT1() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA1(a1), "+" SN_RB1(b1))
#define B2B_MFT(a, b, c)                                                                                     \
    do {                                                                                                     \
        if (!(B2B_ABLATE & 8)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.c, a.c, acc, 0, 0, 0);           \
        else acc[0] += a.c * b.c;                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define B2B_BARRIER() do { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    // the epilogue operands fetched a chunk ahead: fixed registers, like the fragments (a copy between a load and its wait would
    // copy stale contents; tools/check_pinned_regs.py looks for such copies in the ISA)
#define B2B_RV0 "{v[132:135]}"
#define B2B_RV1 "{v[136:139]}"
#define B2B_RV2 "{v[140:143]}"
#define B2B_RV3 "{v[144:147]}"
#define B2B_RBV "{v[148:151]}"
#define B2B_ROS "{v[152:155]}"
#define B2B_ROB "{v[156:159]}"
#define B2B_RFL "{v[160:163]}"
#define B2B_GLD(dst, reg, voff, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=" reg(dst) : "v"(voff), "s"(base) : "memory")

    // ---- waits.  `s_waitcnt vmcnt(N)` waits until all but the N youngest vector-memory operations of the wave are done (loads,
    // stores and LDS-DMA count together, in issue order) and only exists with an immediate N.  Before the barrier of step j the
    // wave waits for EVERYTHING it has issued (vmcnt(0)): stage j+1 and, one step early, stage j+2.  Counting exactly -- a per-wave
    // ledger of every DMA, operand load and store, so that only stage j+1 is waited for and the stores of a chunk drain under
    // the next chunk's MFMAs -- was built and measured: 88.6 us against 86.7 us for vmcnt(0) at the 720p block-1 shape (the
    // stores are not what the barrier waits for; the variable count needs a tree of scalar branches in front of every barrier).
    // The epilogue operands of a chunk are waited for with a count that leaves the DMAs of the chunk's own NWN steps in flight.
    // ---- conv3's epilogue operands: residual rows + per-channel vectors of the NEXT chunk, loaded a chunk ahead (chunk 0: at the park)
    const bool has_res = q3.residual != nullptr, has_obn = q3.out_scale != nullptr, has_bias = q3.bias != nullptr;
    const bool has_floor = has_obn && q3.out_floor != nullptr;
    unsigned res_voff[4] = {0u, 0u, 0u, 0u};               // byte offsets of this wave's four residual rows (per tile)
    const unsigned vec_voff = (unsigned)((wn * 32 + (lane & 7) * 4) * 4);
    auto epi_rows_setup = [&](int m0) {                    // once per tile (the residual may be strided: two divisions per row)
        if (!has_res) return;
        const bool res_plain = q3.res_stride == 1 && q3.res_H == q3.Ho && q3.res_W == q3.Wo;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int m = min(m0 + wm * 32 + (lane >> 3) + 8 * qq, q3.M - 1);
            unsigned roff;
            if (res_plain) {
                roff = (unsigned)(m * q3.res_ld);
            } else {
                const int img = sn_fastdiv(m, q3.div_hw_mul, q3.div_hw_shift);
                const int rr2 = m - img * (q3.Ho * q3.Wo);
                const int oy = sn_fastdiv(rr2, q3.div_w_mul, q3.div_w_shift), ox = rr2 - oy * q3.Wo;
                roff = (unsigned)(((img * q3.res_H + oy * q3.res_stride) * q3.res_W + ox * q3.res_stride) * q3.res_ld);
            }
            res_voff[qq] = roff * 4u + vec_voff;
        }
    };
    B2bEpiOperands eo;
    eo.bv = eo.os = eo.ob = eo.fl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) eo.rv[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto epi_prefetch = [&](int n0) {                      // (uniform branches; every load is one inline-asm instruction: counted exactly)
        if (B2B_ABLATE & 1) return;
        if (has_res) {
            const float* rb = q3.residual + n0;
            B2B_GLD(eo.rv[0], B2B_RV0, res_voff[0], rb); B2B_GLD(eo.rv[1], B2B_RV1, res_voff[1], rb);
            B2B_GLD(eo.rv[2], B2B_RV2, res_voff[2], rb); B2B_GLD(eo.rv[3], B2B_RV3, res_voff[3], rb);
        }
        if (has_bias) { const float* b = q3.bias + n0; B2B_GLD(eo.bv, B2B_RBV, vec_voff, b); }
        if (has_obn) {
            const float* b1 = q3.out_scale + n0; const float* b2 = q3.out_shift + n0;
            B2B_GLD(eo.os, B2B_ROS, vec_voff, b1); B2B_GLD(eo.ob, B2B_ROB, vec_voff, b2);
        }
        if (has_floor) { const float* b = q3.out_floor + n0; B2B_GLD(eo.fl, B2B_RFL, vec_voff, b); }
    };

    f32x4 a0, b0, a1, b1;
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT2) : "memory");        // stage 0 landed (stage 1's DMAs may be in flight)
    B2B_BARRIER();
    issue(2);
    B2B_RD(a0, SN_RA0, a_frag[0], 0);
    B2B_RD(b0, SN_RB0, b_frag[0], 0);
    // A fragments of the CURRENT step live at a_frag[kk] + aoff: a ring slot (conv2) or a K-step of the park (conv3).  One step
    // body serves both phases (conv3 runs transposed as well); every fragment read below is UNCONDITIONAL: a read that is only
    // issued on some paths makes the compiler merge the pinned fragment registers with their previous contents through
    // register copies -- copies of registers whose ds_read is still in flight (seen in the ISA of a first version).
    constexpr unsigned PARK_REL = 3u * STAGE_BYTES;
    unsigned aoff = 0;

    auto step = [&](auto slot_c) -> bool {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int OFF = SLOT * STAGE_BYTES, OFF_NEXT = ((SLOT + 1) % 3) * STAGE_BYTES;
        B2B_WAIT0();
        B2B_MFT(a0, b0, x);
        { const unsigned ad = a_frag[1] + aoff; B2B_RD(a1, SN_RA1, ad, 0); } B2B_RD(b1, SN_RB1, b_frag[1], OFF);
        B2B_MFT(a0, b0, y);
        B2B_MFT(a0, b0, z);
        B2B_MFT(a0, b0, w);
        B2B_WAIT1();
        B2B_MFT(a1, b1, x);
        { const unsigned ad = a_frag[2] + aoff; B2B_RD(a0, SN_RA0, ad, 0); } B2B_RD(b0, SN_RB0, b_frag[2], OFF);
        B2B_MFT(a1, b1, y);
        B2B_MFT(a1, b1, z);
        B2B_MFT(a1, b1, w);
        B2B_WAIT0();
        B2B_MFT(a0, b0, x);
        { const unsigned ad = a_frag[3] + aoff; B2B_RD(a1, SN_RA1, ad, 0); } B2B_RD(b1, SN_RB1, b_frag[3], OFF);
        B2B_MFT(a0, b0, y);
        B2B_MFT(a0, b0, z);
        B2B_MFT(a0, b0, w);
        B2B_WAIT1();                                       // every fragment of this stage is in registers
        B2B_MFT(a1, b1, x);
        --remaining;
        ++cs;
        const int j3 = cs - S2;                            // conv3 steps of this tile done so far (<= 0: still in conv2)
        const bool park_now = j3 == 0;                     // that was conv2's last step
        const bool chunk_done = j3 > 0 && (j3 % NWN) == 0;
        const int n0 = ((j3 - 1) / NWN) * BN2;             // chunk_done: the chunk's first output channel
        const int m0 = ct * BM;
        if (cs == T) { cs = 0; ct += t_stride; }
        const bool more = remaining > 0;
        const unsigned aoff_next = (cs < S2) ? (unsigned)OFF_NEXT : PARK_REL + (unsigned)(((cs - S2) % NWN) * A_BYTES);
        const unsigned ad0 = a_frag[0] + aoff_next;
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of stage j+1 (and j+2) has landed
            B2B_BARRIER();                                 // publishes stage j+1, retires slot SLOT
        }
        B2B_RD(b0, SN_RB0, b_frag[0], OFF_NEXT);           // (harmless after the last step and, for a0, before the park is written)
        B2B_RD(a0, SN_RA0, ad0, 0);
        aoff = aoff_next;
        const bool fed = pt < t_end;                       // stage j+3 into the slot just retired (nothing once the producer is through)
        issue_part(SLOT, 0); __builtin_amdgcn_sched_barrier(0);
        B2B_MFT(a1, b1, y);
        issue_part(SLOT, 1); __builtin_amdgcn_sched_barrier(0);
        B2B_MFT(a1, b1, z);
        issue_part(SLOT, 2); __builtin_amdgcn_sched_barrier(0);
        B2B_MFT(a1, b1, w);
        issue_part(SLOT, 3); __builtin_amdgcn_sched_barrier(0);
        if (park_now && !(B2B_ABLATE & 4)) {
            B2B_WAIT0();                                   // the reads above: landed before the code below
            // park: relu(bn2(acc^T)) as NWN A stages.  Lane (m, h), quad q -> row wm*32 + m, chunk 2q + h of K-step wn.
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(acc[4 * qq + e], sc[qq][e], sh[qq][e]), 0.f);
                asm volatile("ds_write_b128 %0, %1" ::"v"(park_wr[qq]), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            B2B_BARRIER();                                 // the whole tile is parked
            B2B_RD(a0, SN_RA0, ad0, 0);                    // conv3's first A fragment, now that it exists
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            epi_rows_setup(m0);
            epi_prefetch(0);                               // chunk 0's epilogue operands: NWN steps ahead of their use
        }
        if (chunk_done && !(B2B_ABLATE & 2)) {
            B2B_WAIT0();                                   // the next step's first fragments: landed before the epilogue code
            // the operands fetched a chunk ago: younger than their loads are at least the 2 DMAs of each of this chunk's NWN
            // steps (as long as the producer had stages left) -- those stay in flight
            if (fed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+" B2B_RV0(eo.rv[0]), "+" B2B_RV1(eo.rv[1]), "+" B2B_RV2(eo.rv[2]), "+" B2B_RV3(eo.rv[3]),
                              "+" B2B_RBV(eo.bv), "+" B2B_ROS(eo.os), "+" B2B_ROB(eo.ob), "+" B2B_RFL(eo.fl));
            f32x4 outv[4];
            b2b_chunk_compute(acc, q3, lane, scr_base, eo, outv);
            // the NEXT chunk's operands go out BEFORE this chunk's stores (the operand registers are free again): vmcnt retires
            // in issue order, so a later wait for those loads does not have to sit out the stores
            if (j3 < S3) epi_prefetch(n0 + BN2);
            b2b_chunk_store(outv, q3, m0 + wm * 32, n0 + wn * 32, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        return more;
    };

    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    using s2 = std::integral_constant<int, 2>;
    while (true) {
        if (!step(s0{})) break;
        if (!step(s1{})) break;
        if (!step(s2{})) break;
    }
#undef B2B_RD
#undef B2B_WAIT0
#undef B2B_WAIT1
#undef B2B_MFT
#undef B2B_BARRIER
#undef B2B_GLD
}
