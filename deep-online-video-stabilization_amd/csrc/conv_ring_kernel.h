// Implicit-GEMM convolution, persistent LDS-DMA ring variant (gfx950, exact float32 MFMA).
//
// For convolutions whose A operand needs no prologue (the input is already activated, or is a raw gradient): tiles go
// HBM/L2 -> LDS directly with `global_load_lds` (16 B per lane, no VGPR staging, no ds_write, no VALU on the data),
// through a ring of 3 stages.  Measured motivation: with register staging the K-loop of the v1
// kernel is capped by (a) VALU work competing with the f32 MFMAs and (b) one K-step of load latency hiding.
//
//   block  : 64 x 64 output tile, 256 threads = 4 waves (2 x 2 of 32 x 32), BK = 32, 3 stages of 16 KiB
//   stage  : A [64 rows][32 floats] then B [64 rows][32 floats], rows UNPADDED (a DMA instruction writes 1 KiB = 8 rows
//            contiguously); bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle of the 16-B
//            chunk index, chunk_phys = chunk ^ ((row >> 1) & 7), applied on the SOURCE address of the DMA lane and on
//            the read address (both sides or neither: the LDS image itself is lane-linear)
//   DMA    : per stage each wave issues 2 A + 2 B instructions (8 rows each); lane l -> row 8g + (l>>3), chunk l&7
//   sync   : counted `s_waitcnt vmcnt(4)` (later stages stay in flight) + ONE raw s_barrier per K-step, which both
//            publishes the next stage and retires every wave's reads of the current one (the slot the new DMA overwrites)
//   padding: out-of-frame taps read a 16-B zero page (no prologue => zeros stay zeros, nothing to mask afterwards)
//   persist: the grid is at most 3 workgroups per CU; workgroup b walks tiles b, b+G, b+2G, ... and the ring runs ACROSS
//            tile boundaries: the producer side (DMA issue) is three K-steps ahead of the consumer side (MFMA), so the
//            first stages of the next tile are in flight while the current tile's epilogue runs.  The epilogue's
//            transposition scratch overlays the stage that was consumed last (free until the step after next issues
//            into it, behind a barrier).
//   KG > 1 : split-K INSIDE the workgroup.  KG groups of 4 waves, each with its own 3-stage ring (KG x 48 KiB of LDS), walk the
//            same tiles in lockstep (the per-step barrier is workgroup-wide), group g accumulating K slice g (p.splitk == KG,
//            every slice p.steps_per_split steps).  At a tile's end groups 1.. park their accumulators in the ring slot they
//            have just retired, one barrier, group 0 adds them in group order and runs the FULL epilogue: no partial slabs in
//            HBM and no reduce launch (4.6 us + a 12 MB round trip per split-K layer of the 720p frame).  One workgroup per
//            CU keeps 12 waves resident, as three co-resident split-K workgroups did.
//   PRO = 1: the A operand gets a per-channel prologue a = relu(a * scale[k] + shift[k]) (batch-statistics BN + ReLU of the training
//            forward, s_net_bundle_nobm.py:301) WITHOUT leaving the LDS-DMA path: 1x1 convolutions only (k = input channel).  Every
//            wave DMAs the step's 32 scales + 32 shifts (256 B) into its own corner of LDS with the stage (5 DMAs per wave and
//            stage instead of 4), reads the 4 + 4 values of its fragment next to the fragment itself and applies fma + max on
//            the fragment registers (8 VALU per 4 MFMAs; same arithmetic as the register-staged kernel's prologue).  The PAIR of
//            siamese towers runs as one launch (p.M = both towers' rows): `p.out_floor` -- unused here, there is no consumer BN in
//            training -- carries the element distance between the two towers' workspaces as an integer; tiles of the second tower
//            shift x / y / residual / scale / shift by it (minus the rows the pair index has already advanced by), exactly as
//            conv_igemm_f32_pair_kernel does with its ConvPair argument.
#pragma once
#include "conv_kernel.h"
#include <type_traits>

#ifndef RING_ABLATE
#define RING_ABLATE 0          // probe-only bit mask (tools/ring_probe.hip): 1 no DMA, 2 no barrier, 4 no ds_read, 8 no MFMA, 16 no vmcnt wait
#endif

typedef __attribute__((address_space(3))) void* sn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* sn_gbl_ptr_t;

__device__ __attribute__((aligned(16))) float g_conv_zero_page[8];      // zero-initialised device storage

#ifndef RING_STAMP
#define RING_STAMP 0           // probe-only (tools/ring_probe.hip): per-workgroup s_memtime stamps of the launch's phases
#endif
#if RING_STAMP
__device__ unsigned long long g_ring_stamps[4096 * 8];
#define SN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ring_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SN_STAMP_RT(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ring_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SN_STAMP(i) do { } while (0)
#define SN_STAMP_RT(i) do { } while (0)
#endif

template <int MODE /* 0: no padding, 1: zero padding, 2: row-run A operand on a zero-BORDERED image (ConvArgs::rowrun) */,
          int BF16 = 0 /* 1: fragments rounded to bf16 at read time, v_mfma_f32_32x32x16_bf16 (conv_launch's bf16_operands) */,
          int KG = 1 /* K groups of 4 waves: split-K inside the workgroup (p.splitk == KG), see the header comment */,
          int PRO = 0 /* 1: BN + ReLU prologue on the A fragments (MODE 0 only); KG = 1: the pair of towers as one launch */>
__global__ __launch_bounds__(256 * KG) void conv_ring_f32_kernel(const ConvArgs p) {
    constexpr int BM = 64, BN = 64, BK = 32;
    // BF16 == 4 ("packed split"): the B operand is a pre-split, fragment-major IMAGE of the weights (conv.hip,
    // weight_split_image_kernel): per (N tile, K step) 12 KiB = [wave column 2][plane h/m/l 3][k group 2][lane 64][8 bf16], copied
    // linearly by three DMAs per wave and read lane-linearly (conflict-free), so only the A fragments are split at run time.
    constexpr bool PK = (BF16 >= 4);
    // BF16 == 5 ("shared split"): the two waves that multiply the same 32 A rows (wn = 0, 1) split ONE k group each and exchange the
    // planes through LDS (12 KiB per group behind the ring: [wm][k group][plane][lane] 16-B entries, written and read lane-linearly):
    // 44 split VALU per wave and stage instead of 88 -- the packed kernel is bound by VALU issue (SQ counters), not by
    // the matrix pipe.  Costs a second barrier per step (plane hand-over) and 3 ds_write + 2 extra ds_read per wave and stage.
    constexpr bool PS = (BF16 == 5);

    constexpr int B_STAGE = PK ? 3072 : BN * BK;           // floats
    constexpr int STAGE = BM * BK + B_STAGE;               // floats
    constexpr int PLANE_FLOATS = (BF16 == 5) ? 3072 : 0;   // BF16 == 5: plane exchange buffer per K group, behind its ring
    constexpr int GROUP_FLOATS = 3 * STAGE + PLANE_FLOATS; // one K group's LDS
    static_assert(!PRO || (MODE == 0 && BF16 != 1), "the fragment prologue exists for the 1x1 fp32 / split kernels");
    static_assert(!PK || KG <= 2, "the packed split kernel splits K inside the workgroup two ways at most (60 KiB of ring per group)");
    constexpr int SC_FLOATS = PRO ? 3 * 4 * 64 : 0;        // per group, stage and wave: 32 scales + 32 shifts, behind the rings
    __shared__ __attribute__((aligned(16))) float ring[KG * GROUP_FLOATS + KG * SC_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = (KG > 1) ? ((tid >> 6) & 3) : (tid >> 6);
    const int grp = (KG > 1) ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;      // K group = K slice of this wave
    SN_STAMP(0); SN_STAMP_RT(4);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_mn = tiles_m * tiles_n;
    const int ntiles = (KG > 1) ? tiles_mn : tiles_mn * p.splitk;   // K slice slowest, then M tile, N tile fastest (KG > 1: the slice is the group)
    // Tiles of this workgroup: t_first, t_first + t_stride, ... < t_end.  Plain: b, b + G, ...  XCD-aware (xcd_swizzle):
    // the dispatcher deals workgroup ids round-robin over the 8 XCDs (own 4 MiB L2 each), so the workgroups with equal
    // b % 8 share an L2; they take one CONTIGUOUS eighth of the tile list (N tile fastest, then M): that XCD then reads
    // one slice of the input instead of all of it (a 3x3 layer re-reads every input row for 3 tap rows and all N tiles).
    int t_first = blockIdx.x, t_stride = gridDim.x, t_end = ntiles;
    if (p.xcd_swizzle && gridDim.x >= 8) {
        const int x = blockIdx.x & 7, Gt = (int)gridDim.x;
        t_stride = (Gt - x + 7) >> 3;                      // workgroups of this XCD (ids x, x+8, ...)
        const int before = x * (Gt >> 3) + min(x, Gt & 7); // workgroups of the XCDs in front of it
        const int lo = (int)((long)ntiles * before / Gt), hi = (int)((long)ntiles * (before + t_stride) / Gt);
        t_first = lo + (blockIdx.x >> 3);                  // tile ranges proportional to the workgroup counts: balanced
        t_end = hi;
    }
    const int cin_steps = (MODE == 2) ? 1 : p.Cin / BK;
    const int run_steps = (MODE == 2) ? p.K / (p.KH * BK) : 0;     // K-steps per filter row (MODE 2)
    const int total_steps = (MODE == 2) ? p.K / BK : p.KH * p.KW * cin_steps;
    const int last_slice_steps = total_steps - (p.splitk - 1) * p.steps_per_split;
    const unsigned lds_base = (unsigned)(size_t)(sn_lds_ptr_t)ring + (KG > 1 ? (unsigned)grp * (unsigned)(GROUP_FLOATS * 4) : 0u);   // this group's ring

    // PRO: the pair of towers (see the header comment)
    // (KG > 1 with PRO -- the inference conv1 layers that split K -- is one tower and out_floor is an ordinary floor vector)
    // (the packed split form runs both: the inference conv1 layers -- one tower, out_floor null, so the distance is 0 -- and the
    //  training forward's pairs)
    const long pair_delta = (PRO && KG == 1) ? (long)(size_t)p.out_floor : 0L;
    const int m_tower = (PRO && KG == 1 && pair_delta != 0) ? (p.M >> 1) : 0x7fffffff;
    const unsigned sc_base = (KG > 1) ? (unsigned)(size_t)(sn_lds_ptr_t)ring + (unsigned)(KG * GROUP_FLOATS * 4) + (unsigned)grp * (unsigned)(SC_FLOATS * 4)
                                      : lds_base + (unsigned)(GROUP_FLOATS * 4);   // [slot][wave][scale 32 | shift 32]
    // consumer steps of this workgroup (every slice has steps_per_split steps except the last one)
    int remaining = 0;
    for (int t = t_first; t < t_end; t += t_stride) remaining += (t / tiles_mn == p.splitk - 1) ? last_slice_steps : p.steps_per_split;

    // ---- DMA lane roles: instruction g in {wave, wave + 4} covers tile rows 8g .. 8g+7
    const int rr = lane >> 3, cp = lane & 7;
    const int ld = p.x_ld;                                  // floats between consecutive pixels of x (>= Cin)
    const int pad_off = (p.pad * p.W + p.pad) * ld;
    const int row_jump = (p.W - p.KW) * ld;
    const float* const zero_page = g_conv_zero_page;

    // ---- producer state: the tile whose stages are being issued
    int pt = t_first;                                      // tile id; >= t_end: exhausted
    int p_left = 0;                                        // stages of tile `pt` still to issue
    unsigned a_voff[2], w_voff[2];                         // BYTE offsets from the running pointers xb / wb
    unsigned long long a_mask[2];                          // MODE 1: tap bits per row
    int l_tap = 0, l_c0 = 0, l_kw = 0;                     // MODE 2: l_tap = filter row kh, l_kw = 32-float step inside its run
    const float *xb = p.x, *wb = p.w;
    const float* scb = p.in_scale;                         // PRO: running pointer of the step's 32 scales (shifts at a constant byte distance)
    const unsigned sc_voff = PRO ? (unsigned)((lane & 31) * 4) + ((lane >> 5) ? (unsigned)((p.in_shift - p.in_scale) * 4) : 0u) : 0u;
    auto producer_setup = [&]() {
        const int z = (KG > 1) ? grp : pt / tiles_mn;
        const int mn = (KG > 1) ? pt : pt - z * tiles_mn;
        const int mt = mn / tiles_n, nt = mn - mt * tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
        const int ks_begin = z * p.steps_per_split;
        p_left = (z == p.splitk - 1) ? last_slice_steps : p.steps_per_split;
        if constexpr (MODE == 2) {
            // The image carries its zero border in memory ([N][H+2pad][W+2pad][Cin], written by the stack assembly): every
            // tap of every output pixel is in the buffer, and for one filter row the KW*Cin floats of the taps are ONE
            // contiguous run starting at padded pixel (oy*stride + kh, ox*stride).  Rows start at arbitrary 4-byte offsets
            // (Cin = 13): global_load_lds_dwordx4 takes them (tools/glds_unaligned_probe.hip: correct, 87 % of the aligned
            // rate).  The run is read 32 floats at a time up to roundup(KW*Cin, 32): the excess multiplies zero weights.
            const int Wp = p.W + 2 * p.pad, Hp = p.H + 2 * p.pad;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 8 * (wave + 4 * t) + rr;
                const int cl = cp ^ ((row >> 1) & 7);
                const int mc = min(m0 + row, p.M - 1);
                const int img = sn_fastdiv(mc, p.div_hw_mul, p.div_hw_shift);
                const int r = mc - img * (p.Ho * p.Wo);
                const int oy = sn_fastdiv(r, p.div_w_mul, p.div_w_shift), ox = r - oy * p.Wo;
                a_voff[t] = 4u * (unsigned)(((img * Hp + oy * p.stride) * Wp + ox * p.stride) * p.Cin + cl * 4);
                a_mask[t] = 0;
                const int n = min(n0 + row, p.Cout - 1);
                w_voff[t] = 4u * (unsigned)(n * p.K + cl * 4);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 8 * (wave + 4 * t) + rr;
                const int cl = cp ^ ((row >> 1) & 7);          // logical chunk this lane fetches (swizzle on the source side)
                const int m = m0 + row;
                const int mc = min(m, p.M - 1);
                const int img = sn_fastdiv(mc, p.div_hw_mul, p.div_hw_shift);
                const int r = mc - img * (p.Ho * p.Wo);
                const int oy = sn_fastdiv(r, p.div_w_mul, p.div_w_shift), ox = r - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                a_voff[t] = 4u * (unsigned)(((img * p.H + iy0 + p.pad) * p.W + ix0 + p.pad) * ld + cl * 4);
                unsigned long long mk = 0;
                if (MODE == 1 && m < p.M) {
                    // taps (kh, kw) in frame: kh in [kh_lo, kh_hi), kw in [kw_lo, kw_hi): a run of KW-bit row patterns
                    const int kh_lo = max(0, -iy0), kh_hi = min(p.KH, p.H - iy0);
                    const int kw_lo = max(0, -ix0), kw_hi = min(p.KW, p.W - ix0);
                    if (kh_hi > kh_lo && kw_hi > kw_lo) {
                        const unsigned long long rowbits = ((1ull << (kw_hi - kw_lo)) - 1ull) << kw_lo;
                        for (int kh = kh_lo; kh < kh_hi; ++kh) mk |= rowbits << (kh * p.KW);
                    }
                }
                a_mask[t] = mk;
                const int n = min(n0 + row, p.Cout - 1);       // rows >= Cout are never stored
                w_voff[t] = 4u * (unsigned)(n * p.K + cl * 4);
            }
        }
        if constexpr (MODE == 2) {
            l_tap = ks_begin / run_steps;                  // filter row kh
            l_kw = ks_begin - l_tap * run_steps;           // run step s within the row
            xb = p.x + ((long)l_tap * (p.W + 2 * p.pad) * p.Cin + 32 * l_kw);
            wb = PK ? p.w + ((size_t)nt * total_steps + ks_begin) * B_STAGE : p.w + (size_t)ks_begin * BK;
        } else {
            l_tap = ks_begin / cin_steps;
            l_c0 = (ks_begin - l_tap * cin_steps) * BK;
            const int l_kh = l_tap / p.KW;
            l_kw = l_tap - l_kh * p.KW;
            xb = p.x + ((l_kh * p.W + l_kw) * ld + l_c0 - pad_off);
            wb = PK ? p.w + ((size_t)nt * total_steps + ks_begin) * B_STAGE : p.w + (l_tap * p.Cin + l_c0);
            if constexpr (PRO) {
                const bool t2 = m0 >= m_tower;             // tile of the second tower: its tensors live `pair_delta` floats further on
                if (t2) xb += pair_delta - (long)m_tower * ld;
                scb = p.in_scale + l_c0 + (t2 ? pair_delta : 0L);
            }
        }
    };
    // Issue one stage into ring slot `slot` (no-op when every tile of this workgroup has been issued), in four parts so
    // that the K loop can place each DMA (and its address arithmetic) in the shadow of an MFMA.
    const unsigned dma_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)wave * 1024u));
    // The DMA itself is inline asm in the SGPR-base + 32-bit lane-offset form (`global_load_lds_dwordx4 voff, s[base]`):
    // the builtin materialises a 64-bit per-lane address with vector ALU instructions, and every VALU instruction takes
    // issue slots from the f32 MFMAs.  M0 carries the wave-uniform LDS destination.
    auto dma16 = [&](const float* base_, unsigned voff_bytes, unsigned lds_byte) {
        // (the running pointers are wave-uniform; readfirstlane says so to the register allocator and folds away when the
        //  value already lives in SGPRs)
        const unsigned long long b64 = (unsigned long long)(size_t)base_;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const float* base = reinterpret_cast<const float*>((size_t)(((unsigned long long)hi << 32) | lo));
        // (PK: under the packed kernel's scalar-register pressure the wave-uniform LDS address can live in a vector register)
        if constexpr (PK) lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(lds_byte) : "memory", "m0");
    };
    auto issue_s = [&](int slot) {                         // PRO: this wave's copy of the step's scales | shifts (4 B per lane)
        if (pt >= t_end) return;
        const unsigned long long b64 = (unsigned long long)(size_t)scb;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const float* base = reinterpret_cast<const float*>((size_t)(((unsigned long long)hi << 32) | lo));
        const unsigned lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)(sc_base + (unsigned)(slot * 1024 + wave * 256)));
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" ::"v"(sc_voff), "s"(base), "s"(lds_byte) : "memory", "m0");
    };
    auto issue_a = [&](int slot, int t) {
        if (pt >= t_end || (RING_ABLATE & 1)) return;
        const unsigned lds_byte = dma_base + (unsigned)(slot * STAGE * 4 + t * 4096);
        if constexpr (MODE == 1) {
            // out-of-frame taps read the zero page: a per-lane choice between two unrelated addresses needs the 64-bit form
            const float* src = ((a_mask[t] >> l_tap) & 1ull) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + a_voff[t]) : zero_page;
            __builtin_amdgcn_global_load_lds((sn_gbl_ptr_t)src, (sn_lds_ptr_t)(size_t)lds_byte, 16, 0, 0);
        } else {
            dma16(xb, a_voff[t], lds_byte);
        }
    };
    const unsigned wimg_voff = (unsigned)(wave * 1024 + lane * 16);          // PK: lane's byte offset inside a DMA chunk pair (+ 4096 t)
    auto issue_b = [&](int slot, int t) {
        if (pt >= t_end || (RING_ABLATE & 1)) return;
        if constexpr (PK) dma16(wb, wimg_voff + (unsigned)(t * 4096), dma_base + (unsigned)(slot * STAGE * 4 + BM * BK * 4 + t * 4096));
        else dma16(wb, w_voff[t], dma_base + (unsigned)(slot * STAGE * 4 + BM * BK * 4 + t * 4096));
    };
    auto issue_advance = [&]() {
        if (pt >= t_end) return;
        if (--p_left == 0) {
            pt += t_stride;
            if (pt < t_end) producer_setup();
            return;
        }
        xb += BK;
        wb += PK ? B_STAGE : BK;
        if constexpr (PRO) scb += BK;
        if constexpr (MODE == 2) {
            if (++l_kw == run_steps) { l_kw = 0; ++l_tap; xb += (p.W + 2 * p.pad) * p.Cin - BK * run_steps; }
            return;
        }
        l_c0 += BK;
        if (l_c0 == p.Cin) {
            l_c0 = 0;
            ++l_tap;
            xb += ld - p.Cin;                              // next tap = next pixel (0 for a dense tensor)
            if (++l_kw == p.KW) { l_kw = 0; xb += row_jump; }
        }
    };
    // the stage's DMAs in four parts (each is placed behind one MFMA of the K loop)
    auto issue_part = [&](int slot, int part) {
        if (part == 0) issue_a(slot, 0);
        else if (part == 1) { issue_a(slot, 1); issue_b(slot, 0); }
        else if (part == 2) { issue_b(slot, 1); if constexpr (PK) issue_b(slot, 2); if constexpr (PRO) issue_s(slot); }
        else issue_advance();
    };
    auto issue = [&](int slot) {
        issue_part(slot, 0); issue_part(slot, 1); issue_part(slot, 2); issue_part(slot, 3);
    };

    // ---- consumer state: the tile being accumulated
    int ct = t_first, c_left = 0, c_m0 = 0, c_n0 = 0, c_z = 0;
    long c_yoff = 0, c_roff = 0;                           // PRO: element offsets of y / residual for a tile of the second tower
    auto consumer_setup = [&]() {
        c_z = (KG > 1) ? grp : ct / tiles_mn;
        const int mn = (KG > 1) ? ct : ct - c_z * tiles_mn;
        const int mt = mn / tiles_n;
        c_m0 = mt * BM;
        c_n0 = (mn - mt * tiles_n) * BN;
        c_left = (c_z == p.splitk - 1) ? last_slice_steps : p.steps_per_split;
        if constexpr (PRO) {
            const bool t2 = c_m0 >= m_tower;
            c_yoff = t2 ? pair_delta - (long)m_tower * p.Cout : 0L;
            c_roff = t2 ? pair_delta - (long)(p.N >> 1) * p.res_H * p.res_W * p.res_ld : 0L;
        }
    };

    // ---- fragment read addresses (LDS bytes, stage 0) of the 4 kk sub-steps: lane (i = lane&31, h = lane>>5) reads
    //      logical chunk 2kk+h; the stage is selected by the immediate offset of the ds_read
    unsigned a_frag[4], b_frag[4];
    {
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a_frag[kk] = lds_base + 4 * (ra * BK + (((2 * kk + h) ^ ((ra >> 1) & 7)) << 2));
            b_frag[kk] = lds_base + 4 * (BM * BK + rb * BK + (((2 * kk + h) ^ ((rb >> 1) & 7)) << 2));
        }
    }

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

    if (remaining == 0) return;                            // (the grid never exceeds the tile count; kept for safety)
    producer_setup();
    consumer_setup();

    // The fragment reads are inline asm on purpose: the compiler's waitcnt insertion cannot tell which DMA a ds_read
    // depends on and puts `s_waitcnt vmcnt(0)` in front of compiler-visible LDS loads, draining the stages in flight.
    // Waits are therefore explicit: vmcnt before the barrier, lgkmcnt (tied to the fragment registers through "+v" so the
    // MFMAs cannot move above them) before each MFMA group.
    // Between a read and its wait the compiler believes the destination already holds the data; a register copy there
    // would copy stale contents.  The fragment variables are therefore bound to FIXED physical registers at every asm
    // boundary (no phi copies at the loop back-edge or around the epilogue branch), and the one place where a fragment
    // pair stays in flight across a long code region (the epilogue) waits for it first.
#define SN_RA0 "{v[100:103]}"
#define SN_RB0 "{v[104:107]}"
#define SN_RA1 "{v[108:111]}"
#define SN_RB1 "{v[112:115]}"
#define SN_DS_READ(dst, reg, addr, off)                                                                   \
    do { if (!(RING_ABLATE & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" reg(dst) : "v"(addr), "n"(off)); } while (0)
#define SN_READ0(kk, off) do { SN_DS_READ(a0, SN_RA0, a_frag[kk], off); SN_DS_READ(b0, SN_RB0, b_frag[kk], off); } while (0)
#define SN_READ1(kk, off) do { SN_DS_READ(a1, SN_RA1, a_frag[kk], off); SN_DS_READ(b1, SN_RB1, b_frag[kk], off); } while (0)
#define SN_WAIT0() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA0(a0), "+" SN_RB0(b0))
#define SN_WAIT1() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA1(a1), "+" SN_RB1(b1))
    // PK: A fragments alone, and the six pre-split B fragments of a stage (k group j: planes h, m, l), lane-linear
#define SN_PH0 "{v[132:135]}"
#define SN_PM0 "{v[136:139]}"
#define SN_PL0 "{v[140:143]}"
#define SN_PH1 "{v[144:147]}"
#define SN_PM1 "{v[148:151]}"
#define SN_PL1 "{v[152:155]}"
#define SN_READA(dst, reg, kk, off) SN_DS_READ(dst, reg, a_frag[kk], off)
#define SN_READB0(off) do { SN_DS_READ(ph0, SN_PH0, bimg_frag, (off) + 0); SN_DS_READ(pm0, SN_PM0, bimg_frag, (off) + 2048); SN_DS_READ(pl0, SN_PL0, bimg_frag, (off) + 4096); } while (0)
#define SN_READB1(off) do { SN_DS_READ(ph1, SN_PH1, bimg_frag, (off) + 1024); SN_DS_READ(pm1, SN_PM1, bimg_frag, (off) + 3072); SN_DS_READ(pl1, SN_PL1, bimg_frag, (off) + 5120); } while (0)
#define SN_RA2 "{v[156:159]}"
#define SN_RA3 "{v[160:163]}"
    // waits of the packed schedule: LDS operations return in order, so lgkmcnt(N) says "everything but the youngest N reads has landed"
    // (a scalar load in flight can only make such a wait stricter)
#define SN_RS2 "{v[164:167]}"
#define SN_RH2 "{v[168:171]}"
#define SN_RS3 "{v[172:175]}"
#define SN_RH3 "{v[176:179]}"
    // (PRO: the scale / shift fragments of the four A fragments are read IN FRONT of them, so the same counts cover them)
#define SN_WAIT_A01(n)                                                                                                     \
    do {                                                                                                                   \
        if constexpr (PRO) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+" SN_RA0(a0), "+" SN_RA1(a1), "+" SN_RS0(sc0), "+" SN_RH0(sh0), "+" SN_RS1(sc1), "+" SN_RH1(sh1)); \
        else asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+" SN_RA0(a0), "+" SN_RA1(a1));                                  \
    } while (0)
#define SN_WAIT_A23B0(n)                                                                                                   \
    do {                                                                                                                   \
        if constexpr (PRO) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+" SN_RA2(a2), "+" SN_RA3(a3), "+" SN_PH0(ph0), "+" SN_PM0(pm0), "+" SN_PL0(pl0), "+" SN_RS2(sc2), "+" SN_RH2(sh2), "+" SN_RS3(sc3), "+" SN_RH3(sh3)); \
        else asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+" SN_RA2(a2), "+" SN_RA3(a3), "+" SN_PH0(ph0), "+" SN_PM0(pm0), "+" SN_PL0(pl0)); \
    } while (0)
#define SN_READS8(soff)                                                                                                    \
    do {                                                                                                                   \
        if constexpr (PRO) {                                                                                               \
            SN_DS_READ(sc0, SN_RS0, s_frag[0], soff); SN_DS_READ(sh0, SN_RH0, s_frag[0], (soff) + 128);                    \
            SN_DS_READ(sc1, SN_RS1, s_frag[1], soff); SN_DS_READ(sh1, SN_RH1, s_frag[1], (soff) + 128);                    \
            SN_DS_READ(sc2, SN_RS2, s_frag[2], soff); SN_DS_READ(sh2, SN_RH2, s_frag[2], (soff) + 128);                    \
            SN_DS_READ(sc3, SN_RS3, s_frag[3], soff); SN_DS_READ(sh3, SN_RH3, s_frag[3], (soff) + 128);                    \
        }                                                                                                                  \
    } while (0)
#define SN_WAIT_B1() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_PH1(ph1), "+" SN_PM1(pm1), "+" SN_PL1(pl1))
#define SN_READA4(off) do { SN_READA(a0, SN_RA0, 0, off); SN_READA(a1, SN_RA1, 1, off); SN_READA(a2, SN_RA2, 2, off); SN_READA(a3, SN_RA3, 3, off); } while (0)
#define SN_SB() __builtin_amdgcn_sched_barrier(0)
#define SN_PMFMA(ap, breg) do { if (!(RING_ABLATE & 8)) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sn_bf16x8, ap), __builtin_bit_cast(sn_bf16x8, breg), acc[0][0], 0, 0, 0); else acc[0][0][0] += __builtin_bit_cast(float, ap.x) + breg.x; SN_SB(); } while (0)
    // one pair of a fragment's floats through one split level (conv_kernel.h): L1 -> plane h, L2 -> plane m, L3 -> plane l
#define SN_L1(P, c, x0, x1) do { if (!(RING_ABLATE & 32)) P.h.c = sn_split_level(x0, x1); else P.h.c = __builtin_bit_cast(unsigned, x0); } while (0)
#define SN_L2(P, c, x0, x1) do { if (!(RING_ABLATE & 32)) P.m.c = sn_split_level(x0, x1); else P.m.c = __builtin_bit_cast(unsigned, x1); } while (0)
#define SN_L3(P, c, x0, x1) do { if (!(RING_ABLATE & 32)) P.l.c = sn_pack_bf16(x0, x1); else P.l.c = __builtin_bit_cast(unsigned, x0); } while (0)
    // ---- shared split (PS): plane registers (8 bf16 each), the exchange buffer's reads / writes, the own-half reads
#define SN_QH0 "{v[156:159]}"
#define SN_QM0 "{v[160:163]}"
#define SN_QL0 "{v[164:167]}"
#define SN_QH1 "{v[168:171]}"
#define SN_QM1 "{v[172:175]}"
#define SN_QL1 "{v[176:179]}"
#define SN_READQ()                                                                                                        \
    do {                                                                                                                  \
        SN_DS_READ(qh0, SN_QH0, pl_rd0, 0); SN_DS_READ(qm0, SN_QM0, pl_rd0, 1024); SN_DS_READ(ql0, SN_QL0, pl_rd0, 2048);  \
        SN_DS_READ(qh1, SN_QH1, pl_rd1, 0); SN_DS_READ(qm1, SN_QM1, pl_rd1, 1024); SN_DS_READ(ql1, SN_QL1, pl_rd1, 2048);  \
    } while (0)
#define SN_WAIT_Q0B0(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+" SN_QH0(qh0), "+" SN_QM0(qm0), "+" SN_QL0(ql0), "+" SN_PH0(ph0), "+" SN_PM0(pm0), "+" SN_PL0(pl0))
#define SN_WAIT_Q1B1() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_QH1(qh1), "+" SN_QM1(qm1), "+" SN_QL1(ql1), "+" SN_PH1(ph1), "+" SN_PM1(pm1), "+" SN_PL1(pl1))
#define SN_READOWN(off, soff)                                                                                             \
    do {                                                                                                                  \
        if constexpr (PRO) {                                                                                              \
            SN_DS_READ(sc0, SN_RS0, s_own0, soff); SN_DS_READ(sh0, SN_RH0, s_own0, (soff) + 128);                          \
            SN_DS_READ(sc1, SN_RS1, s_own1, soff); SN_DS_READ(sh1, SN_RH1, s_own1, (soff) + 128);                          \
        }                                                                                                                 \
        SN_DS_READ(a0, SN_RA0, a_own0, off); SN_DS_READ(a1, SN_RA1, a_own1, off);                                          \
    } while (0)
#define SN_WRITEQ(P)                                                                                                      \
    do {                                                                                                                  \
        asm volatile("ds_write_b128 %0, %1" ::"v"(pl_wr), "v"(P.h) : "memory");                                           \
        asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(pl_wr), "v"(P.m) : "memory");                               \
        asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(pl_wr), "v"(P.l) : "memory");                               \
    } while (0)
    // shared split, first half of a step: hand-over barrier, both k groups' planes + the B planes of k group 1, six MFMAs
#define SN_PS_GROUP0(OFF_)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* this wave's plane writes have reached LDS */                \
    SN_BARRIER();                                          /* ... and so have its partner's */                            \
    SN_READQ(); SN_READB1(OFF_);                                                                                         \
    SN_WAIT_Q0B0(6);                                       /* all but the youngest six reads (k group 1 planes, B1) */     \
    SN_SB();                                                                                                             \
    SN_PMFMA(ql0, ph0); SN_PMFMA(qh0, pl0); SN_PMFMA(qm0, pm0); SN_PMFMA(qm0, ph0); SN_PMFMA(qh0, pm0); SN_PMFMA(qh0, ph0)
    // ... last four MFMAs of the second half with the split of this wave's OWN k group of the next stage between them, planes out
#define SN_PS_GROUP1_TAIL()                                                                                              \
    SN_WAIT_A01(3);                                        /* own fragments (+ their scales); the next B planes may fly */ \
    if constexpr (PRO) { SN_PRO(a0, sc0, sh0); SN_PRO(a1, sc1, sh1); }                                                   \
    float y0 = a0.x, y1 = a0.y, y2 = a0.z, y3 = a0.w, y4 = a1.x, y5 = a1.y, y6 = a1.z, y7 = a1.w;                        \
    SnPlanes NP;                                                                                                         \
    SN_SB();                                                                                                             \
    SN_PMFMA(qm1, pm1); SN_L1(NP, x, y0, y1); SN_L1(NP, y, y2, y3); SN_SB();                                             \
    SN_PMFMA(qm1, ph1); SN_L1(NP, z, y4, y5); SN_L1(NP, w, y6, y7); SN_SB();                                             \
    SN_PMFMA(qh1, pm1); SN_L2(NP, x, y0, y1); SN_L2(NP, y, y2, y3); SN_SB();                                             \
    SN_PMFMA(qh1, ph1); SN_L2(NP, z, y4, y5); SN_L2(NP, w, y6, y7);                                                      \
    SN_L3(NP, x, y0, y1); SN_L3(NP, y, y2, y3); SN_L3(NP, z, y4, y5); SN_L3(NP, w, y6, y7); SN_SB();                     \
    SN_WRITEQ(NP); SN_SB()
    // first MFMA group of a packed step (planes AP x B planes of k group 0) with the split of k group 1 (a2, a3 -> CP) in its gaps
#define SN_PK_GROUP0(OFF_)                                                                                               \
    SN_READB1(OFF_);                                       /* (their registers were the last MFMA's operands) */          \
    SN_WAIT_A23B0(3);                                                                                                    \
    if constexpr (PRO) { SN_PRO(a2, sc2, sh2); SN_PRO(a3, sc3, sh3); }                                                   \
    float x0 = a2.x, x1 = a2.y, x2 = a2.z, x3 = a2.w, x4 = a3.x, x5 = a3.y, x6 = a3.z, x7 = a3.w;                        \
    SN_SB();                                                                                                             \
    SN_PMFMA(AP.l, ph0); SN_L1(CP, x, x0, x1); SN_L1(CP, y, x2, x3); SN_SB();                                            \
    SN_PMFMA(AP.h, pl0); SN_L1(CP, z, x4, x5); SN_L1(CP, w, x6, x7); SN_SB();                                            \
    SN_PMFMA(AP.m, pm0); SN_L2(CP, x, x0, x1); SN_L2(CP, y, x2, x3); SN_SB();                                            \
    SN_PMFMA(AP.m, ph0); SN_L2(CP, z, x4, x5); SN_L2(CP, w, x6, x7); SN_SB();                                            \
    SN_PMFMA(AP.h, pm0); SN_L3(CP, x, x0, x1); SN_L3(CP, y, x2, x3); SN_L3(CP, z, x4, x5); SN_L3(CP, w, x6, x7); SN_SB(); \
    SN_PMFMA(AP.h, ph0)
    // last four MFMAs of the second group (planes CP x B planes of k group 1) with the split of the NEXT stage's k group 0 (-> AP)
#define SN_PK_GROUP1_TAIL()                                                                                              \
    SN_WAIT_A01(5);                                        /* (a2, a3 and the B planes of the next stage may still fly) */ \
    if constexpr (PRO) { SN_PRO(a0, sc0, sh0); SN_PRO(a1, sc1, sh1); }                                                   \
    float y0 = a0.x, y1 = a0.y, y2 = a0.z, y3 = a0.w, y4 = a1.x, y5 = a1.y, y6 = a1.z, y7 = a1.w;                        \
    SN_SB();                                                                                                             \
    SN_PMFMA(CP.m, pm1); SN_L1(AP, x, y0, y1); SN_L1(AP, y, y2, y3); SN_SB();                                            \
    SN_PMFMA(CP.m, ph1); SN_L1(AP, z, y4, y5); SN_L1(AP, w, y6, y7); SN_SB();                                            \
    SN_PMFMA(CP.h, pm1); SN_L2(AP, x, y0, y1); SN_L2(AP, y, y2, y3); SN_SB();                                            \
    SN_PMFMA(CP.h, ph1); SN_L2(AP, z, y4, y5); SN_L2(AP, w, y6, y7);                                                     \
    SN_L3(AP, x, y0, y1); SN_L3(AP, y, y2, y3); SN_L3(AP, z, y4, y5); SN_L3(AP, w, y6, y7); SN_SB()
    // PRO: the fragment's 4 scales + 4 shifts ride with it (this wave's own copy: offset slot * 1024 selects the stage)
#define SN_RS0 "{v[116:119]}"
#define SN_RH0 "{v[120:123]}"
#define SN_RS1 "{v[124:127]}"
#define SN_RH1 "{v[128:131]}"
#define SN_READ0P(kk, off, soff)                                                                      \
    do { SN_READ0(kk, off); SN_DS_READ(sc0, SN_RS0, s_frag[kk], soff); SN_DS_READ(sh0, SN_RH0, s_frag[kk], (soff) + 128); } while (0)
#define SN_READ1P(kk, off, soff)                                                                      \
    do { SN_READ1(kk, off); SN_DS_READ(sc1, SN_RS1, s_frag[kk], soff); SN_DS_READ(sh1, SN_RH1, s_frag[kk], (soff) + 128); } while (0)
#define SN_WAIT0P() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA0(a0), "+" SN_RB0(b0), "+" SN_RS0(sc0), "+" SN_RH0(sh0))
#define SN_WAIT1P() asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_RA1(a1), "+" SN_RB1(b1), "+" SN_RS1(sc1), "+" SN_RH1(sh1))
#define SN_PRO(a, sc, sh)                                                                             \
    do {                                                                                              \
        a.x = fmaxf(__builtin_fmaf(a.x, sc.x, sh.x), 0.f); a.y = fmaxf(__builtin_fmaf(a.y, sc.y, sh.y), 0.f); \
        a.z = fmaxf(__builtin_fmaf(a.z, sc.z, sh.z), 0.f); a.w = fmaxf(__builtin_fmaf(a.w, sc.w, sh.w), 0.f); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#define SN_MFMA1(a, b, c)                                                                                      \
    do {                                                                                                       \
        if (!(RING_ABLATE & 8)) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.c, b.c, acc[0][0], 0, 0, 0); \
        else acc[0][0][0] += a.c * b.c;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#define SN_BARRIER()                                                     \
    do {                                                                 \
        if (!(RING_ABLATE & 2)) __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)

    // Schedule.  Stage s lives in slot s % 3; the producer runs THREE stages ahead of the MFMAs.  An MFMA occupies the
    // matrix pipe for 64 cycles but a wave issues in order, so anything placed AFTER a group of MFMAs is only covered by
    // the last one: every non-MFMA instruction of the loop is therefore placed right after ONE MFMA (sched_barrier pins
    // the order): the two ds_read_b128 of the next fragment pair after the first MFMA of each group, the four DMAs of
    // stage k+3 after the last three MFMAs of the step.  Entering step k the kk = 0 fragments of stage k are in flight
    // to (a0, b0).  The step's barrier sits inside its LAST MFMA group: by then every fragment of stage k is in
    // registers, so the barrier (a) publishes stage k+1, (b) frees slot k % 3 for stage k+3.  At a tile's last step the
    // freed slot first serves as the epilogue's transposition scratch; stage k+3 is issued after it, behind one more barrier.
    f32x4 a0, b0, a1, b1;
    f32x4 sc0, sh0, sc1, sh1;                              // PRO only
    f32x4 ph0, pm0, pl0, ph1, pm1, pl1;                    // PK only: 8 bf16 each
    f32x4 a2, a3;                                          // PK only: A fragments kk = 2, 3
    f32x4 sc2, sh2, sc3, sh3;                              // PK + PRO only
    SnPlanes AP, CP;                                       // PK only: split A planes of k group 0 / 1
    f32x4 qh0, qm0, ql0, qh1, qm1, ql1;                    // PS only: the planes as read back from the exchange buffer
    const unsigned pl_base = lds_base + (unsigned)(3 * STAGE * 4);                                     // PS: [wm][k group][plane][lane]
    const unsigned pl_rd0 = pl_base + (unsigned)((wm * 2 + 0) * 3072 + lane * 16), pl_rd1 = pl_base + (unsigned)((wm * 2 + 1) * 3072 + lane * 16);
    const unsigned pl_wr = pl_base + (unsigned)((wm * 2 + wn) * 3072 + lane * 16);                      // this wave splits k group wn
    unsigned a_own0 = 0u, a_own1 = 0u, s_own0 = 0u, s_own1 = 0u;
    const unsigned bimg_frag = lds_base + (unsigned)(4 * BM * BK + wn * 6144 + lane * 16);
    unsigned s_frag[4] = {0u, 0u, 0u, 0u};
    if constexpr (PRO) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s_frag[kk] = sc_base + (unsigned)(wave * 256 + (8 * kk + 4 * (lane >> 5)) * 4);
    }
    if constexpr (PS) {
        const int ra = wm * 32 + (lane & 31), h = lane >> 5;
        a_own0 = lds_base + 4 * (ra * BK + (((2 * (2 * wn) + h) ^ ((ra >> 1) & 7)) << 2));
        a_own1 = lds_base + 4 * (ra * BK + (((2 * (2 * wn + 1) + h) ^ ((ra >> 1) & 7)) << 2));
        if constexpr (PRO) {
            s_own0 = sc_base + (unsigned)(wave * 256 + (8 * (2 * wn) + 4 * h) * 4);
            s_own1 = sc_base + (unsigned)(wave * 256 + (8 * (2 * wn + 1) + 4 * h) * 4);
        }
    }
    if (RING_ABLATE & 4) { a0 = b0 = a1 = b1 = f32x4{1.f, 2.f, 3.f, 4.f}; }
    issue(0);
    issue(1);
    // (a wave has 4 DMAs per stage in flight, 5 with the scales of the PRO form)
    if (remaining > 1) { if constexpr (PRO && PK) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if constexpr (PRO || PK) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SN_BARRIER();
    SN_STAMP(1);
    issue(2);
    if constexpr (PS) {
        SN_READOWN(0, 0); SN_READB0(0);
        SN_WAIT_A01(3);
        if constexpr (PRO) { SN_PRO(a0, sc0, sh0); SN_PRO(a1, sc1, sh1); }
        const SnPlanes NP0 = sn_split3_pair(a0, a1);
        SN_SB();
        SN_WRITEQ(NP0);
        SN_SB();
    } else if constexpr (PK) {
        SN_READS8(0); SN_READA4(0); SN_READB0(0);
        SN_WAIT_A01(5);
        if constexpr (PRO) { SN_PRO(a0, sc0, sh0); SN_PRO(a1, sc1, sh1); }
        AP = sn_split3_pair(a0, a1);
        SN_SB();
    } else if constexpr (PRO) SN_READ0P(0, 0, 0); else SN_READ0(0, 0);

    // One K-step on ring slot SLOT; returns false after the workgroup's last step.
    auto step = [&](auto slot_c) -> bool {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int OFF = SLOT * STAGE * 4;
        constexpr int OFF_NEXT = ((SLOT + 1) % 3) * STAGE * 4;
        bool tile_done, more;
        if constexpr (PS) {
            // shared split: at entry the planes of this stage (both k groups, written by this wave and its partner during the
            // previous step) sit in the exchange buffer and the B planes of k group 0 were read behind the previous barrier
            constexpr int SOFF_NEXT = ((SLOT + 1) % 3) * 1024;
            SN_PS_GROUP0(OFF);
            --remaining;
            tile_done = (--c_left == 0);
            more = remaining > 0;
            const bool feed = more && !tile_done;
            SN_WAIT_Q1B1();                                    // every fragment of this stage is in registers
            if (more) {
                if (remaining > 1) { if constexpr (PRO) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SN_BARRIER();
            }
            SN_READOWN(OFF_NEXT, SOFF_NEXT); SN_READB0(OFF_NEXT);            // unconditional (see the packed step below)
            SN_PMFMA(ql1, ph1);
            if (feed) { issue_part(SLOT, 0); issue_part(SLOT, 1); SN_SB(); }
            SN_PMFMA(qh1, pl1);
            if (feed) { issue_part(SLOT, 2); issue_part(SLOT, 3); SN_SB(); }
            SN_PS_GROUP1_TAIL();
        } else if constexpr (PK) {
            // packed split, software-pipelined: at entry AP holds the split planes of k group 0 of this stage (made during the previous
            // step's second MFMA group); a2, a3 (k group 1) and the B planes of k group 0 were read behind the previous barrier.
            // Every split level sits between two MFMAs (sched_barrier pins the order): a 32-cycle MFMA hides about six plain VALU.
            SN_PK_GROUP0(OFF);
            --remaining;
            tile_done = (--c_left == 0);
            more = remaining > 0;
            const bool feed = more && !tile_done;
            SN_WAIT_B1();                                      // every fragment of this stage is in registers
            if (more) {
                if (remaining > 1) { if constexpr (PRO) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SN_BARRIER();
            }
            // UNCONDITIONAL (after the workgroup's last step they fetch stale LDS that nobody uses): a read on only some control-flow
            // paths makes the compiler merge the pinned registers with their old contents through copies -- copies of registers
            // whose read is still in flight (seen in the two-group prologue form; tools/check_pinned_regs.py looks for them)
            SN_READS8(((SLOT + 1) % 3) * 1024); SN_READA4(OFF_NEXT); SN_READB0(OFF_NEXT);
            SN_PMFMA(CP.l, ph1);
            if (feed) { issue_part(SLOT, 0); issue_part(SLOT, 1); SN_SB(); }
            SN_PMFMA(CP.h, pl1);
            if (feed) { issue_part(SLOT, 2); issue_part(SLOT, 3); SN_SB(); }
            SN_PK_GROUP1_TAIL();
        } else if constexpr (BF16 >= 2) {
            // split mode (conv_kernel.h, sn_split3): every fragment is decomposed into three bf16 terms when it is read and the stage
            // is six (nine) v_mfma_f32_32x32x16_bf16 per 16-deep k group instead of eight v_mfma_f32_32x32x2_f32.  Same k
            // permutation as the bf16-operand branch below.
            constexpr int NPROD = (BF16 == 3) ? 9 : 6;
            constexpr int SOFF = SLOT * 1024, SOFF_NEXT = ((SLOT + 1) % 3) * 1024;
            if constexpr (PRO) { SN_WAIT0P(); SN_PRO(a0, sc0, sh0); } else SN_WAIT0();
            const SnSplit3 A0 = sn_split3(a0), B0 = sn_split3(b0);
            if constexpr (PRO) SN_READ1P(1, OFF, SOFF); else SN_READ1(1, OFF);
            if constexpr (PRO) { SN_WAIT1P(); SN_PRO(a1, sc1, sh1); } else SN_WAIT1();
            const SnSplit3 A1 = sn_split3(a1), B1 = sn_split3(b1);
            if constexpr (PRO) { SN_READ0P(2, OFF, SOFF); SN_READ1P(3, OFF, SOFF); } else { SN_READ0(2, OFF); SN_READ1(3, OFF); }
            sn_mfma_split3<NPROD>(acc[0][0], A0, A1, B0, B1);
            if constexpr (PRO) { SN_WAIT0P(); SN_WAIT1P(); SN_PRO(a0, sc0, sh0); SN_PRO(a1, sc1, sh1); }
            else { SN_WAIT0(); SN_WAIT1(); }                   // every fragment of this stage is in registers
            const SnSplit3 C0 = sn_split3(a0), D0 = sn_split3(b0);
            const SnSplit3 C1 = sn_split3(a1), D1 = sn_split3(b1);
            __builtin_amdgcn_sched_barrier(0);
            --remaining;
            tile_done = (--c_left == 0);
            more = remaining > 0;
            const bool feed = more && !tile_done;
            if (more) {
                if (remaining > 1) { if constexpr (PRO) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SN_BARRIER();
                if constexpr (PRO) SN_READ0P(0, OFF_NEXT, SOFF_NEXT); else SN_READ0(0, OFF_NEXT);
            }
            if (feed) { issue_part(SLOT, 0); issue_part(SLOT, 1); issue_part(SLOT, 2); issue_part(SLOT, 3); __builtin_amdgcn_sched_barrier(0); }
            sn_mfma_split3<NPROD>(acc[0][0], C0, C1, D0, D1);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (BF16) {
            // bf16-operand mode: the four 8-deep fragment pairs of the stage feed TWO v_mfma_f32_32x32x16_bf16.  A lane half
            // holds k = {4h..4h+3} of fragment kk and of fragment kk+1: a permutation of the 16 k of the instruction that is
            // the same for A and B, so the product sum is the same set of terms.  The matrix pipe is no longer the limit
            // here (64 instead of 1024 cycles per stage): the LDS-DMA stream is.
            SN_WAIT0();
            const sn_bf16x4 al = __builtin_convertvector(a0, sn_bf16x4), bl = __builtin_convertvector(b0, sn_bf16x4);
            SN_READ1(1, OFF);
            SN_WAIT1();
            const sn_bf16x4 ah = __builtin_convertvector(a1, sn_bf16x4), bh = __builtin_convertvector(b1, sn_bf16x4);
            SN_READ0(2, OFF);
            SN_READ1(3, OFF);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(al, ah, 0, 1, 2, 3, 4, 5, 6, 7),
                                                                __builtin_shufflevector(bl, bh, 0, 1, 2, 3, 4, 5, 6, 7), acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            SN_WAIT0();
            SN_WAIT1();                                        // every fragment of this stage is in registers
            const sn_bf16x4 cl = __builtin_convertvector(a0, sn_bf16x4), dl = __builtin_convertvector(b0, sn_bf16x4);
            const sn_bf16x4 ch = __builtin_convertvector(a1, sn_bf16x4), dh = __builtin_convertvector(b1, sn_bf16x4);
            __builtin_amdgcn_sched_barrier(0);
            --remaining;
            tile_done = (--c_left == 0);
            more = remaining > 0;
            const bool feed = more && !tile_done;
            if (more) {
                if (remaining > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SN_BARRIER();
                SN_READ0(0, OFF_NEXT);
            }
            if (feed) { issue_part(SLOT, 0); issue_part(SLOT, 1); issue_part(SLOT, 2); issue_part(SLOT, 3); __builtin_amdgcn_sched_barrier(0); }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(cl, ch, 0, 1, 2, 3, 4, 5, 6, 7),
                                                                __builtin_shufflevector(dl, dh, 0, 1, 2, 3, 4, 5, 6, 7), acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (PRO) {
        // the fp32 schedule of the branch below; the prologue of fragment kk + 1 (8 VALU) sits in front of the LAST MFMA of group
        // kk -- in the shadow of the one before it -- so only the step's first fragment pays for its prologue in the open
        constexpr int SOFF = SLOT * 1024, SOFF_NEXT = ((SLOT + 1) % 3) * 1024;
        SN_WAIT0P();
        SN_PRO(a0, sc0, sh0);
        SN_MFMA1(a0, b0, x);
        SN_READ1P(1, OFF, SOFF);
        SN_MFMA1(a0, b0, y);
        SN_MFMA1(a0, b0, z);
        SN_WAIT1P();
        SN_PRO(a1, sc1, sh1);
        SN_MFMA1(a0, b0, w);
        SN_MFMA1(a1, b1, x);
        SN_READ0P(2, OFF, SOFF);
        SN_MFMA1(a1, b1, y);
        SN_MFMA1(a1, b1, z);
        SN_WAIT0P();
        SN_PRO(a0, sc0, sh0);
        SN_MFMA1(a1, b1, w);
        SN_MFMA1(a0, b0, x);
        SN_READ1P(3, OFF, SOFF);
        SN_MFMA1(a0, b0, y);
        SN_MFMA1(a0, b0, z);
        SN_WAIT1P();                                       // every fragment of this stage is in registers
        SN_PRO(a1, sc1, sh1);
        SN_MFMA1(a0, b0, w);
        SN_MFMA1(a1, b1, x);
        --remaining;
        tile_done = (--c_left == 0);
        more = remaining > 0;
        const bool feed = more && !tile_done;
        if (more) {
            if (remaining > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SN_BARRIER();
            SN_READ0P(0, OFF_NEXT, SOFF_NEXT);
        }
        if (feed) { issue_part(SLOT, 0); __builtin_amdgcn_sched_barrier(0); }
        SN_MFMA1(a1, b1, y);
        if (feed) { issue_part(SLOT, 1); __builtin_amdgcn_sched_barrier(0); }
        SN_MFMA1(a1, b1, z);
        if (feed) { issue_part(SLOT, 2); __builtin_amdgcn_sched_barrier(0); }
        SN_MFMA1(a1, b1, w);
        if (feed) { issue_part(SLOT, 3); __builtin_amdgcn_sched_barrier(0); }
        } else {
        SN_WAIT0();
        SN_MFMA1(a0, b0, x);
        SN_READ1(1, OFF);
        SN_MFMA1(a0, b0, y);
        SN_MFMA1(a0, b0, z);
        SN_MFMA1(a0, b0, w);
        SN_WAIT1();
        SN_MFMA1(a1, b1, x);
        SN_READ0(2, OFF);
        SN_MFMA1(a1, b1, y);
        SN_MFMA1(a1, b1, z);
        SN_MFMA1(a1, b1, w);
        SN_WAIT0();
        SN_MFMA1(a0, b0, x);
        SN_READ1(3, OFF);
        SN_MFMA1(a0, b0, y);
        SN_MFMA1(a0, b0, z);
        SN_MFMA1(a0, b0, w);
        SN_WAIT1();                                        // every fragment of this stage is in registers
        SN_MFMA1(a1, b1, x);
        --remaining;
        tile_done = (--c_left == 0);
        more = remaining > 0;
        const bool feed = more && !tile_done;
        if (more) {
            // stage k+1 has landed for THIS wave when at most stage k+2's 4 DMAs are outstanding.  (Epilogue stores of
            // an earlier tile may be in flight too: the counter retires in order, so the wait is then stricter, never weaker.)
            if (!(RING_ABLATE & 16)) {
                if (remaining > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SN_BARRIER();
            SN_READ0(0, OFF_NEXT);
        }
        if (feed) { issue_part(SLOT, 0); __builtin_amdgcn_sched_barrier(0); }   // stage k+3 into the slot just retired
        SN_MFMA1(a1, b1, y);
        if (feed) { issue_part(SLOT, 1); __builtin_amdgcn_sched_barrier(0); }
        SN_MFMA1(a1, b1, z);
        if (feed) { issue_part(SLOT, 2); __builtin_amdgcn_sched_barrier(0); }
        SN_MFMA1(a1, b1, w);
        if (feed) { issue_part(SLOT, 3); __builtin_amdgcn_sched_barrier(0); }
        }
        if (tile_done) {
            if constexpr (PS) { asm volatile("s_waitcnt lgkmcnt(0)" : "+" SN_PH0(ph0), "+" SN_PM0(pm0), "+" SN_PL0(pl0)); if (!more) SN_BARRIER(); }
            else if constexpr (PK) { SN_WAIT_A23B0(0); if (!more) SN_BARRIER(); }
            else if (more) { if constexpr (PRO) SN_WAIT0P(); else SN_WAIT0(); }   // the next tile's first fragments: landed before the epilogue code
            else SN_BARRIER();                             // (otherwise the barrier above already retired slot SLOT)
            SN_STAMP(2);
            if constexpr (KG > 1) {
                // Slot SLOT of EVERY group is free here (the workgroup-wide barrier above retired it).  Groups 1.. park their
                // accumulators in their own slot, [wave][q][lane] in 16-B units (conflict-free both ways); one barrier; group 0
                // adds them in group order -- a fixed order: the result does not depend on timing -- and runs the full epilogue.
                // Inline asm for the same reason as the fragment reads: a compiler-visible LDS access would drain the DMA queue.
                const unsigned red = lds_base + (unsigned)(OFF + wave * 4096 + lane * 16);
                if (grp > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(red), "v"(v), "n"(q * 1024) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                SN_BARRIER();
                if (grp == 0) {
#pragma unroll
                    for (int g = 1; g < KG; ++g) {
                        f32x4 t0, t1, t2, t3;
                        const unsigned src = red + (unsigned)(g * GROUP_FLOATS * 4);
                        asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:1024\n\t"
                                     "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(src) : "memory");
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[0][0][e] += t0[e]; acc[0][0][4 + e] += t1[e]; acc[0][0][8 + e] += t2[e]; acc[0][0][12 + e] += t3[e];
                        }
                    }
                    conv_epilogue<1, 1, true>(acc, p, c_m0 + wm * 32, c_n0 + wn * 32, lane, 0,
                                              lds_base + (unsigned)(OFF + wave * SN_EPI_WAVE_BYTES));
                }
            } else if constexpr (PRO)
            conv_epilogue<1, 1, false, true>(acc, p, c_m0 + wm * 32, c_n0 + wn * 32, lane, c_z,
                                             lds_base + (unsigned)(OFF + wave * SN_EPI_WAVE_BYTES), c_yoff, c_roff);
            else
            conv_epilogue<1, 1>(acc, p, c_m0 + wm * 32, c_n0 + wn * 32, lane, c_z,
                                lds_base + (unsigned)(OFF + wave * SN_EPI_WAVE_BYTES));
#if RING_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SN_STAMP(3); SN_STAMP_RT(5);
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
            ct += t_stride;
            if (ct < t_end) consumer_setup();
            if (more) {
                SN_BARRIER();                              // every wave is done with the scratch
                issue(SLOT);
            }
        }
        return more;
    };
    // PK: a step in the middle of a tile for both sides of the ring (the consumer stays inside its tile, the producer inside its own,
    // at least two more steps follow): the same instruction stream as `step` without its tile bookkeeping -- with the matrix work of
    // a step down to 12 x 32 cycles, the ~70 scalar instructions and ~10 branches of the general step were what a wave spent its time on
    auto fast_step = [&](auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int OFF = SLOT * STAGE * 4;
        constexpr int OFF_NEXT = ((SLOT + 1) % 3) * STAGE * 4;
        if constexpr (PK) {
            if constexpr (PS) {
                SN_PS_GROUP0(OFF);
                --remaining; --c_left; --p_left;
                SN_WAIT_Q1B1();
            } else {
                SN_PK_GROUP0(OFF);
                --remaining; --c_left; --p_left;
                SN_WAIT_B1();
            }
            if constexpr (PRO) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            SN_BARRIER();
            if constexpr (PS) { SN_READOWN(OFF_NEXT, ((SLOT + 1) % 3) * 1024); SN_READB0(OFF_NEXT); }
            else { SN_READS8(((SLOT + 1) % 3) * 1024); SN_READA4(OFF_NEXT); SN_READB0(OFF_NEXT); }
            if constexpr (PS) SN_PMFMA(ql1, ph1); else SN_PMFMA(CP.l, ph1);
            if (!(RING_ABLATE & 1)) {
                const unsigned lds_a = dma_base + (unsigned)(SLOT * STAGE * 4);
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float* src = ((a_mask[t] >> l_tap) & 1ull) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + a_voff[t]) : zero_page;
                        __builtin_amdgcn_global_load_lds((sn_gbl_ptr_t)src, (sn_lds_ptr_t)(size_t)(lds_a + (unsigned)(t * 4096)), 16, 0, 0);
                    }
                } else {
                    dma16(xb, a_voff[0], lds_a);
                    dma16(xb, a_voff[1], lds_a + 4096u);
                }
                dma16(wb, wimg_voff, lds_a + (unsigned)(BM * BK * 4));
            }
            SN_SB();
            if constexpr (PS) SN_PMFMA(qh1, pl1); else SN_PMFMA(CP.h, pl1);
            if (!(RING_ABLATE & 1)) {
                const unsigned lds_b = dma_base + (unsigned)(SLOT * STAGE * 4 + BM * BK * 4);
                dma16(wb, wimg_voff + 4096u, lds_b + 4096u);
                dma16(wb, wimg_voff + 8192u, lds_b + 8192u);
            }
            if constexpr (PRO) { issue_s(SLOT); scb += BK; }
            xb += BK;
            wb += B_STAGE;
            if constexpr (MODE == 2) {
                if (++l_kw == run_steps) { l_kw = 0; ++l_tap; xb += (p.W + 2 * p.pad) * p.Cin - BK * run_steps; }
            } else {
                l_c0 += BK;
                if (l_c0 == p.Cin) {
                    l_c0 = 0;
                    ++l_tap;
                    xb += ld - p.Cin;
                    if (++l_kw == p.KW) { l_kw = 0; xb += row_jump; }
                }
            }
            SN_SB();
            if constexpr (PS) { SN_PS_GROUP1_TAIL(); } else { SN_PK_GROUP1_TAIL(); }
        }
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    using s2 = std::integral_constant<int, 2>;
    while (true) {
        if constexpr (PK) {
            int nf = (pt < t_end) ? min(min(c_left, p_left) - 1, remaining - 2) : 0;
            for (; nf >= 3; nf -= 3) { fast_step(s0{}); fast_step(s1{}); fast_step(s2{}); }
        }
        if (!step(s0{})) break;
        if (!step(s1{})) break;
        if (!step(s2{})) break;
    }
#undef SN_BARRIER
#undef SN_DS_READ
#undef SN_READ0
#undef SN_READ1
#undef SN_WAIT0
#undef SN_WAIT1
#undef SN_MFMA1
#undef SN_READ0P
#undef SN_READ1P
#undef SN_WAIT0P
#undef SN_WAIT1P
#undef SN_PRO
#undef SN_READA
#undef SN_READB0
#undef SN_READB1
#undef SN_READS8
#undef SN_WAIT_A01
#undef SN_WAIT_A23B0
#undef SN_WAIT_B1
#undef SN_READA4
#undef SN_SB
#undef SN_PMFMA
#undef SN_L1
#undef SN_L2
#undef SN_L3
#undef SN_READQ
#undef SN_WAIT_Q0B0
#undef SN_WAIT_Q1B1
#undef SN_READOWN
#undef SN_WRITEQ
#undef SN_PS_GROUP0
#undef SN_PS_GROUP1_TAIL
#undef SN_PK_GROUP0
#undef SN_PK_GROUP1_TAIL
}
