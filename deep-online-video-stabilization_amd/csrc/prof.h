// Optional per-launch HIP-event instrumentation used by bench.py's roofline leg (never on by default).
#pragma once
#include <vector>
#include "common.h"

enum {
    PK_KERNEL_PAD = 1, PK_KERNEL_POOL = 2, PK_KERNEL_GAP = 3, PK_KERNEL_FC = 4, PK_KERNEL_MESH = 5,
    PK_KERNEL_WARP = 6, PK_KERNEL_ASSEMBLE = 7, PK_KERNEL_PUSH = 8, PK_KERNEL_SPLITK_REDUCE = 9, PK_KERNEL_WGRAD = 10, PK_KERNEL_HEAD = 20,
    PK_KERNEL_WGRAD_SAME = 11 /* + 2*K3 + PRO, + 4 with the bias sums (K3 = 0): conv_wgrad_same_f32_kernel<K3, PRO, BIAS> */,
    PK_KERNEL_CONV_PAIR = 80 /* + 2*MODE + (BK==32): conv_igemm_f32_pair_kernel<64, 64, BK, 32, 32, MODE> */, PK_KERNEL_CONV_KG = 84 /* + mode: conv_ring_f32_kernel<MODE, 0, 3, 0>; + 2: conv_ring_f32_kernel<0, 0, 1, 1> (fragment prologue); + 3: <0, 0, 2, 1> */, PK_KERNEL_CONV_RING = 90 /* + mode; + 3 for the bf16-operand variants */,
    PK_KERNEL_CONV_PACKED = 70 /* + mode: conv_ring_f32_kernel<MODE, 4, 1, 0> (pre-split weight image); + 3: <0, 4, 1, 1>; + 4 + mode: <MODE, 4, 2, 0>; + 6: <0, 4, 2, 1> */,
    PK_KERNEL_CONV_B2B = 96 /* conv_b2b_f32_kernel<2>; + 1: <4> */,
    PK_KERNEL_CONV_BASE = 100   // + MODE*6 + tile*2 + (BK==32) + 18 if NBUF == 1 + 36 if BF16 (conv_igemm_f32_kernel<..., NBUF, BF16>)
};

struct Prof {
    std::vector<hipEvent_t> ev;     // 2 per record
    std::vector<int> kind;
    std::vector<double> flops, bytes;
    std::vector<int> shape;         // 4 ints per record: M, N, K, split-K (0 when not a GEMM)
    int cap = 0, n = 0;
    bool begin(hipStream_t st) {
        if (n >= cap) return false;
        (void)hipEventRecord(ev[2 * n], st);
        return true;
    }
    void end(hipStream_t st, int k, double f, double b, int sm = 0, int sn = 0, int sk = 0, int ss = 0) {
        (void)hipEventRecord(ev[2 * n + 1], st);
        kind[n] = k; flops[n] = f; bytes[n] = b;
        shape[4 * n] = sm; shape[4 * n + 1] = sn; shape[4 * n + 2] = sk; shape[4 * n + 3] = ss;
        ++n;
    }
};
