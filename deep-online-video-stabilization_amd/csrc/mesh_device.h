// get_4_pts + get_H/pinv + get_Hs for ONE (sample, cell) by one 64-lane wave (s_net_bundle_nobm.py:29-71,
// spatial_transformer3.py:144-198).  Shared by mesh_homography_kernel (warp.hip) and the regressor's fused head
// (head_chain.hip), so both produce the same bits.  Built with -ffp-contract=off: one rounding per TF op.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float sn_wshfl(float v, int src) { return __shfl(v, src, 64); }
// a[q] for a per-lane q without private-memory indexing (a dynamically indexed register array lands in scratch)
__device__ __forceinline__ float sn_sel4(const float (&a)[4], int q) { return q == 0 ? a[0] : (q == 1 ? a[1] : (q == 2 ? a[2] : a[3])); }

// in_n: this sample's theta [(gh+1)(gw+1)*2] (in_is_theta: vertex = regular grid + offset, clipped to +-lim) or its
// pts2 [gh+1, gw+1, 2] used as is; may point into LDS.  Writes pts2_n (optional), Hs_n [gh*gw, 9], pts1_n (optional).
// LU with partial pivoting + two column-oriented triangular solves == Eigen PartialPivLU::inverse() for n=8.
__device__ __forceinline__ void sn_mesh_cell(const float* in_n, int in_is_theta, int gh, int gw, float lim, int cell, int lane,
                                             float* __restrict__ pts2_n, float* __restrict__ Hs_n, float* __restrict__ pts1_n) {
    const int ci = cell / gw, cj = cell % gw;
    const double hh = 2.0 / gh, ww = 2.0 / gw;

    // source corners (regular cell) and target corners (mesh vertices): order TL, TR, BL, BR
    float sx[4], sy[4], tu[4], tv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vi = ci + (q >> 1), vj = cj + (q & 1);
        const double bx = cj * ww - 1.0, by = ci * hh - 1.0;          // spatial_transformer3.py:187-189
        sx[q] = (float)((q & 1) ? bx + ww : bx);
        sy[q] = (float)((q >> 1) ? by + hh : by);
        const int v = vi * (gw + 1) + vj;
        float px = in_n[v * 2 + 0];
        float py = in_n[v * 2 + 1];
        if (in_is_theta) {                                             // s_net_bundle_nobm.py:44-58
            px = (float)(vj * ww - 1.0) + px;
            py = (float)(vi * hh - 1.0) + py;
            px = fminf(fmaxf(px, -lim), lim);
            py = fminf(fmaxf(py, -lim), lim);
        }
        tu[q] = px;
        tv[q] = py;
        if (pts2_n != nullptr && lane == q) {
            // each vertex is written by every cell that owns it with the same value
            pts2_n[v * 2 + 0] = px;
            pts2_n[v * 2 + 1] = py;
        }
    }
    if (pts1_n != nullptr && lane < 8)                   // [x_TL,x_TR,x_BL,x_BR,y_TL,y_TR,y_BL,y_BR], s_net_bundle_nobm.py:65-66
        pts1_n[cell * 8 + lane] = (lane < 4) ? sn_sel4(tu, lane & 3) : sn_sel4(tv, lane & 3);

    const int r = lane >> 3, c = lane & 7;
    const int q = r & 3;
    const float tq = (r < 4) ? sn_sel4(tu, q) : sn_sel4(tv, q);
    const float sxq = sn_sel4(sx, q), syq = sn_sel4(sy, q);
    float a;
    {   // A[r][c], spatial_transformer3.py:160-167
        const int cc = (r < 4) ? c : c - 3;             // u-rows use cols 0..2, v-rows cols 3..5
        float val = 0.0f;
        if (c < 6) {
            if (cc == 0) val = sxq;
            else if (cc == 1) val = syq;
            else if (cc == 2) val = 1.0f;
            else val = 0.0f;
            if ((r < 4 && c >= 3) || (r >= 4 && c < 3)) val = 0.0f;
        } else if (c == 6) {
            val = (-sxq) * tq;
        } else {
            val = (-syq) * tq;
        }
        a = val + ((r == c) ? 1e-4f : 0.0f);            // A + eye(8)*1e-4, :145
    }
    float x = (r == c) ? 1.0f : 0.0f;                   // right-hand side: identity, row-swapped with A (P*I)

    for (int k = 0; k < 8; ++k) {
        float best = -1.0f;
        int piv = k;
        for (int rr = k; rr < 8; ++rr) {                // first max |A[rr][k]|
            const float v = fabsf(sn_wshfl(a, rr * 8 + k));
            if (v > best) { best = v; piv = rr; }
        }
        const int src = (r == k) ? piv * 8 + c : ((r == piv) ? k * 8 + c : lane);
        a = sn_wshfl(a, src);
        x = sn_wshfl(x, src);
        const float pivot = sn_wshfl(a, k * 8 + k);
        if (r > k && c == k) a = a / pivot;
        const float l = sn_wshfl(a, r * 8 + k);
        const float u = sn_wshfl(a, k * 8 + c);
        if (r > k && c > k) a = a - l * u;
    }
    for (int i = 0; i < 8; ++i) {                       // unit-lower solve
        const float b = sn_wshfl(x, i * 8 + c);
        const float l = sn_wshfl(a, r * 8 + i);
        if (r > i) x = x - b * l;
    }
    for (int i = 7; i >= 0; --i) {                      // upper solve, reciprocal-diagonal form
        const float d = sn_wshfl(a, i * 8 + i);
        const float inv = 1.0f / d;
        if (r == i) x = x * inv;
        const float b = sn_wshfl(x, i * 8 + c);
        const float u = sn_wshfl(a, r * 8 + i);
        if (r < i) x = x - b * u;
    }
    float acc = 0.0f;                                   // h = inv(A) @ b, k-sequential
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xk = sn_wshfl(x, r * 8 + k);
        const float bk = (k < 4) ? tu[k & 3] : tv[k & 3];
        acc = acc + xk * bk;
    }
    float* Hout = Hs_n + cell * 9;
    if (c == 0) Hout[r] = acc;
    if (lane == 63) Hout[8] = 1.0f;
}
