// Body of the register-staged implicit-GEMM convolution kernel (conv_kernel.h), as TEXT: it is included inside the kernel
// function(s) that share it (template parameters BM, BN, BK, WM, WN, MODE, NBUF, BF16, the argument `p` and the macro
// SN_IGEMM_SAFE_BASE -- float offset of the first image whose rows this workgroup may touch -- and SN_IGEMM_VALID_PTR -- any
// readable address -- are in scope there).
// Text rather than a __device__ function on purpose: moving the body into a function changed the register allocation of the
// inference instantiations (ISA diff), and those kernels are on the headline path.
// (Round 3 built this body with the K split INSIDE the workgroup as well -- two groups of 256 threads, LDS reduction, no reduce
//  launch -- for the block-3 conv1 layers of the 720p frame: 30.8 us against 24.3 + 4.9 us for kernel + reduce launch, rocprofv3
//  in the graph replay; removed again.  The ring kernel keeps its form of it, conv_ring_kernel.h KG.)
    constexpr int PITCH = BK + 4;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int F4_PER_ROW = BK / 4;
    constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
    constexpr int A_PASSES = BM / ROWS_PER_PASS;
    constexpr int B_PASSES = BN / ROWS_PER_PASS;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small for the loader");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = (BM + BN) * PITCH;

    CONV_STAMP_AT(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const ConvTile tile = conv_tile_of_block(p);
    const int m0 = tile.mt * BM, n0 = tile.nt * BN;

    const int cin_steps = p.Cin / BK;
    const int total_steps = p.KH * p.KW * cin_steps;
    const int ks_begin0 = tile.z * p.steps_per_split;
    const int ks_end0 = min(total_steps, ks_begin0 + p.steps_per_split);
    int ks_b = ks_begin0, ks_e = ks_end0;
    // MODE 2 = dgrad of a stride-2 conv: the input is dY dilated by 2, so a tap row kh only meets real samples on output rows of
    // one parity.  For 3x3 / stride 2 the re-packed weights keep their tap rows in the order (1, 0, 2) (pack_dgrad_weights*,
    // `kperm`): the rows valid for a tile whose 64 pixels lie in ONE output row are then one contiguous run of K steps -- the
    // first third (kh = 1) or the last two thirds (kh = 0, 2) -- and the rest is skipped instead of multiplied by zeros.
    const bool kperm = MODE == 2 && p.KH == 3 && p.KW == 3 && p.up == 2;
    if constexpr (MODE == 2) {
        if (kperm && p.Wo % BM == 0) {
            const int oy = (m0 / p.Wo) % p.Ho;
            const int third = p.KW * cin_steps;
            if ((p.pad - oy) & 1) ks_e = min(ks_e, third);
            else ks_b = max(ks_b, third);
        }
    }
    const int ks_begin = ks_b, ks_end = ks_e;

    // ---- loader mapping: thread -> (row within pass, float4 within the BK slice).
    // The K loop must stay (almost) free of vector ALU work: on gfx950 every VALU instruction costs ~3 cycles of
    // f32-MFMA issue (measured, tools/mfma_probe.hip).  So each global address is  (wave-uniform base advanced with
    // scalar adds) + (loop-invariant 32-bit per-lane byte offset): the `saddr + voffset` form of global_load.
    // All offsets are 32-bit; the host checks every tensor has < 2^30 elements.
    const int lrow = tid / F4_PER_ROW, lc4 = tid % F4_PER_ROW;
    const int pad_off = (p.pad * p.W + p.pad) * p.Cin;           // shifts the per-lane offset to be non-negative
    unsigned a_voff[A_PASSES];
    unsigned long long a_mask[A_PASSES];                          // MODE 1: bit t set <=> tap t of this row is in frame
    int a_iy0[A_PASSES], a_ix0[A_PASSES];                         // MODE 2 only
    const unsigned safe_voff = (unsigned)(pad_off + lc4 * 4) + SN_IGEMM_SAFE_BASE;   // + tap base = inside the first KH rows of image 0 (of this tile's tower)
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
        const int m = m0 + lrow + ps * ROWS_PER_PASS;
        const int mc = min(m, p.M - 1);
        const int img = sn_fastdiv(mc, p.div_hw_mul, p.div_hw_shift);
        const int r = mc - img * (p.Ho * p.Wo);
        const int oy = sn_fastdiv(r, p.div_w_mul, p.div_w_shift), ox = r - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_iy0[ps] = iy0;
        a_ix0[ps] = ix0;
        a_mask[ps] = 0;
        if (MODE == 2) {
            a_voff[ps] = (unsigned)(img * p.H * p.W * p.Cin + lc4 * 4);
            if (m >= p.M) a_iy0[ps] = -(1 << 28);
        } else {
            a_voff[ps] = (unsigned)(((img * p.H + iy0 + p.pad) * p.W + ix0 + p.pad) * p.Cin + lc4 * 4);
            if (MODE == 1 && m < p.M) {
                unsigned long long mk = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W)
                            mk |= 1ull << (kh * p.KW + kw);
                a_mask[ps] = mk;
            }
        }
    }
    unsigned w_voff[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
        const int n = n0 + lrow + ps * ROWS_PER_PASS;
        w_voff[ps] = (unsigned)(min(n, p.Cout - 1) * p.K + lc4 * 4);     // rows >= Cout are never stored
    }
    const bool has_pro = p.in_scale != nullptr;
    const float* const scp = has_pro ? p.in_scale : SN_IGEMM_VALID_PTR;     // any valid address when there is no prologue
    const float* const shp = has_pro ? p.in_shift : SN_IGEMM_VALID_PTR;
    const unsigned s_voff = has_pro ? (unsigned)(lc4 * 4) : 0u;

    // K-iteration state (wave-uniform) of the NEXT tile to load.  For MODE 0/1 the operand pointers are RUNNING pointers:
    // consecutive K-steps are contiguous in memory for the weights (K index = (kh,kw,c)) and, inside one kernel row, for
    // the activations too ((kh*W + kw)*Cin + c), so both advance by BK per step; only a kh wrap needs a jump.
    int l_kh, l_kw, l_c0, l_tap;
    {
        l_tap = ks_begin / cin_steps;
        l_c0 = (ks_begin - l_tap * cin_steps) * BK;
        l_kh = l_tap / p.KW;
        l_kw = l_tap - l_kh * p.KW;
    }
    const float* xb = p.x + ((l_kh * p.W + l_kw) * p.Cin + l_c0 - pad_off);       // uniform; may point before p.x
    const float* wb = p.w + (l_tap * p.Cin + l_c0);
    const float* scb = scp + (has_pro ? l_c0 : 0) + s_voff;                         // per-lane, advances uniformly
    const float* shb = shp + (has_pro ? l_c0 : 0) + s_voff;
    const int row_jump = (p.W - p.KW) * p.Cin;

    // Loads are unconditional (out-of-frame taps read a safe in-buffer address and are zeroed when the tile is written
    // to LDS) and the BN+ReLU prologue runs at LDS-store time, AFTER the MFMAs of the current step: nothing between the
    // global loads and the matrix work consumes the loaded registers, so the loads stay in flight under the MFMAs.
    float4 ra[A_PASSES], rb[B_PASSES], sc, sh;
    bool okv[A_PASSES];                                   // lane masks (SGPR pairs), reused at LDS-store time
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) okv[ps] = true;
    auto load_tiles = [&]() {
        if (MODE == 2) {
            const int kh = kperm ? (l_kh == 0 ? 1 : (l_kh == 1 ? 0 : 2)) : l_kh, kw = l_kw, c0 = l_c0;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                int iy = a_iy0[ps] + kh, ix = a_ix0[ps] + kw;
                bool ok = (iy >= 0) & (ix >= 0) & ((iy % p.up) == 0) & ((ix % p.up) == 0);
                iy /= p.up;
                ix /= p.up;
                ok = ok & (iy < p.H) & (ix < p.W);
                const unsigned off = a_voff[ps] + (unsigned)((iy * p.W + ix) * p.Cin + c0);
                if (!(CONV_ABLATE & 1)) ra[ps] = *reinterpret_cast<const float4*>(p.x + (ok ? off : (unsigned)(lc4 * 4)));
                okv[ps] = ok;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                unsigned voff = a_voff[ps];
                if (MODE == 1) {
                    okv[ps] = (a_mask[ps] >> l_tap) & 1ull;
                    voff = okv[ps] ? voff : safe_voff;
                }
                if (!(CONV_ABLATE & 1)) ra[ps] = *reinterpret_cast<const float4*>(xb + voff);
            }
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps)
            if (!(CONV_ABLATE & 1)) rb[ps] = *reinterpret_cast<const float4*>(wb + w_voff[ps]);
        sc = *reinterpret_cast<const float4*>(scb);
        sh = *reinterpret_cast<const float4*>(shb);
        // advance to the next K-step
        xb += BK;
        wb += BK;
        l_c0 += BK;
        if (has_pro) { scb += BK; shb += BK; }
        if (l_c0 == p.Cin) {
            l_c0 = 0;
            ++l_tap;
            if (has_pro) { scb -= p.Cin; shb -= p.Cin; }
            if (++l_kw == p.KW) { l_kw = 0; ++l_kh; xb += row_jump; }
        }
    };
    auto store_tiles = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + BM * PITCH;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            float4 v = ra[ps];
            if (has_pro) {
                v.x = fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), 0.f);
                v.y = fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), 0.f);
                v.w = fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), 0.f);
            }
            if (MODE != 0 && !okv[ps]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(CONV_ABLATE & 2)) *reinterpret_cast<float4*>(As + (lrow + ps * ROWS_PER_PASS) * PITCH + lc4 * 4) = v;
            else asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const float4 v = rb[ps];
            if (!(CONV_ABLATE & 2)) *reinterpret_cast<float4*>(Bs + (lrow + ps * ROWS_PER_PASS) * PITCH + lc4 * 4) = v;
            else asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ks_begin < ks_end) {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();
    CONV_STAMP_AT(1);

    const int frag_off = (lane & 31) * PITCH + (lane >> 5) * 4;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const int buf = (NBUF == 2) ? ((ks - ks_begin) & 1) : 0;
        if (ks + 1 < ks_end) load_tiles();                // global loads in flight under the MFMAs
        const float* Ab = smem + buf * STAGE + (wm * WM) * PITCH + frag_off;
        const float* Bb = smem + buf * STAGE + (BM + wn * WN) * PITCH + frag_off;
        if constexpr (BF16 >= 2) {
            // split mode (conv_kernel.h, sn_split3): exact f32 products as six (nine) bf16 MFMAs per 16-deep k group
            typedef float fl4 __attribute__((ext_vector_type(4)));
            static_assert(BK % 16 == 0, "split operands need BK % 16 == 0");
#pragma unroll
            for (int kp = 0; kp < BK / 16; ++kp) {
                SnSplit3 al[TM], ah[TM], bl[TN], bh[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    al[i] = sn_split3(*reinterpret_cast<const fl4*>(Ab + i * 32 * PITCH + (2 * kp) * 8));
                    ah[i] = sn_split3(*reinterpret_cast<const fl4*>(Ab + i * 32 * PITCH + (2 * kp + 1) * 8));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bl[j] = sn_split3(*reinterpret_cast<const fl4*>(Bb + j * 32 * PITCH + (2 * kp) * 8));
                    bh[j] = sn_split3(*reinterpret_cast<const fl4*>(Bb + j * 32 * PITCH + (2 * kp + 1) * 8));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) sn_mfma_split3<(BF16 == 3) ? 9 : 6>(acc[i][j], al[i], ah[i], bl[j], bh[j]);
            }
        } else if constexpr (BF16) {
            // bf16-operand mode (conv_launch's bf16_operands): two 8-deep fragment pairs -> one v_mfma_f32_32x32x16_bf16
            typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
            typedef float fl4 __attribute__((ext_vector_type(4)));
            static_assert(BK % 16 == 0 || !BF16, "bf16 operands need BK % 16 == 0");
#pragma unroll
            for (int kp = 0; kp < BK / 16; ++kp) {
                fl4 al[TM], ah[TM], bl[TN], bh[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    al[i] = *reinterpret_cast<const fl4*>(Ab + i * 32 * PITCH + (2 * kp) * 8);
                    ah[i] = *reinterpret_cast<const fl4*>(Ab + i * 32 * PITCH + (2 * kp + 1) * 8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bl[j] = *reinterpret_cast<const fl4*>(Bb + j * 32 * PITCH + (2 * kp) * 8);
                    bh[j] = *reinterpret_cast<const fl4*>(Bb + j * 32 * PITCH + (2 * kp + 1) * 8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_shufflevector(__builtin_convertvector(al[i], bf4), __builtin_convertvector(ah[i], bf4), 0, 1, 2, 3, 4, 5, 6, 7),
                            __builtin_shufflevector(__builtin_convertvector(bl[j], bf4), __builtin_convertvector(bh[j], bf4), 0, 1, 2, 3, 4, 5, 6, 7),
                            acc[i][j], 0, 0, 0);
            }
        } else {
        // fragment reads run one kk ahead of the MFMAs that consume them (LDS latency hidden under the matrix work)
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * PITCH);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * PITCH);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * PITCH + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * PITCH + (kk + 1) * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bf[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bf[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bf[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bf[cur][j].w, acc[i][j], 0, 0, 0);
                }
        }
        // pin the interleave: reads of kk+1 are issued BEFORE the MFMAs of kk (hipcc otherwise regroups them into
        // "read 2 kk, wait, 8 MFMAs", exposing the LDS latency at one wave per SIMD)
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            if (kk + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
        }
        }
        if (NBUF == 1) __syncthreads();                   // single LDS stage: everyone is done reading it
        if (ks + 1 < ks_end) store_tiles(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }

    CONV_STAMP_AT(2);
    __syncthreads();                                       // every wave is done with the operand stages the scratch overlays
    conv_epilogue<TM, TN>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, tile.z,
                          (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + wave * SN_EPI_WAVE_BYTES);
    CONV_STAMP_AT(3);
