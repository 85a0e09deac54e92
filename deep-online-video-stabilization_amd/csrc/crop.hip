// Max-inscribed-rectangle crop search (deploy_bundle.py:291,344-366), gfx950.
// The reference accumulates all_black += round(black) on the host every frame and, once per video, runs a pure-Python
// quadruple loop over an integral image.  Here: the accumulation is one elementwise kernel per frame, the integral image
// is two scan kernels, and the search runs one block per start point (i, j) of the step grid with one thread per bottom
// row hh, each finding its widest black-free extent by binary search in the integral image.  Tie-breaking is the
// reference's: the first rectangle in (i, j, hh, ww) order of strictly larger area wins.
#include "common.h"

__global__ __launch_bounds__(256) void black_accumulate_kernel(const float* __restrict__ black, int* __restrict__ all_black, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) all_black[i] += (int)rintf(black[i]);                       // np.round(black).astype(int64), :291
}

// S[(H+1) x (W+1)], S[y+1][x+1] = sum all_black[0..y][0..x] (int64: integer sums, any order gives the same bits).
// Row pass: one WAVE per row -- 64 consecutive elements per trip (coalesced), inclusive scan across the lanes by shuffles, the
// running total carried in a register.  (Round 4: the first form, one THREAD per row walking W dependent loads, took 456 us at
// 720p; this one is bandwidth-shaped.)
__device__ __forceinline__ long long sn_shfl_up_i64(long long v, int d) {
    const int lo = __shfl_up((int)(v & 0xffffffffll), d, 64), hi = __shfl_up((int)(v >> 32), d, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__global__ __launch_bounds__(256) void integral_rows_kernel(const int* __restrict__ a, int H, int W, long long* __restrict__ S) {
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y > H) return;                                                       // wave-uniform
    long long* row = S + (size_t)y * (W + 1);
    if (y == 0) { for (int x = lane; x <= W; x += 64) row[x] = 0; return; }
    if (lane == 0) row[0] = 0;
    const int* src = a + (size_t)(y - 1) * W;
    long long carry = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        long long v = x < W ? (long long)src[x] : 0ll;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long u = sn_shfl_up_i64(v, d);
            if (lane >= d) v += u;
        }
        v += carry;
        if (x < W) row[x + 1] = v;
        carry = ((long long)__shfl((int)(v >> 32), 63, 64) << 32) | (unsigned int)__shfl((int)(v & 0xffffffffll), 63, 64);   // lane 63's total
    }
}
// Column pass: a workgroup owns 16 columns, its 16 row groups sum their rows independently (loads in flight, no chain across the
// whole column), the group offsets go through LDS, then every group rewrites its rows as running totals.  (First form: one
// thread per column, H dependent load -> add -> store steps: 187 us at 720p.)
__global__ __launch_bounds__(256) void integral_cols_kernel(int H, int W, long long* __restrict__ S) {
    __shared__ long long part[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + c;
    const int rows = (H + 15) / 16;                                          // rows per group (rows 1 .. H of S)
    const int y0 = 1 + g * rows, y1 = min(H + 1, y0 + rows);
    long long s = 0;
    if (x <= W)
        for (int y = y0; y < y1; ++y) s += S[(size_t)y * (W + 1) + x];
    part[g][c] = s;
    __syncthreads();
    long long off = 0;
    for (int k = 0; k < g; ++k) off += part[k][c];
    if (x <= W) {
        long long run = off;
        for (int y = y0; y < y1; ++y) { run += S[(size_t)y * (W + 1) + x]; S[(size_t)y * (W + 1) + x] = run; }
    }
}

// key = area << 32 | (0xFFFFFFFF - order): larger area first, then the EARLIER candidate in the reference's loop order.
__global__ __launch_bounds__(256) void crop_search_kernel(const int* __restrict__ all_black, const long long* __restrict__ S, int H,
                                                          int W, int step, int nj, unsigned long long* __restrict__ best_key,
                                                          int* __restrict__ best_rect) {
    const int i = (blockIdx.x / nj) * step, j = (blockIdx.x % nj) * step;
    if (all_black[(size_t)i * W + j] > 0) return;                            // :356-357
    __shared__ unsigned long long s_key[256];
    __shared__ int s_ww[256], s_hh[256];
    unsigned long long key = 0;
    int bww = -1, bhh = -1;
    const long long* Si = S + (size_t)i * (W + 1);
    for (int hh = i + threadIdx.x; hh < H; hh += 256) {
        const long long* Sh = S + (size_t)(hh + 1) * (W + 1);
        // widest ww >= j with zero black in rows i..hh, cols j..ww: the sum is monotone in ww -> binary search
        int lo = j - 1, hi = W - 1;                                          // invariant: cols j..lo are free
        if (Sh[W] - Sh[j] - Si[W] + Si[j] == 0) lo = W - 1;
        else {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (Sh[mid + 1] - Sh[j] - Si[mid + 1] + Si[j] == 0) lo = mid; else hi = mid;
            }
        }
        if (lo >= j) {
            const unsigned long long area = (unsigned long long)(hh - i + 1) * (unsigned long long)(lo - j + 1);
            // order inside this start point: smaller hh first
            const unsigned long long k = (area << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)hh);
            if (k > key) { key = k; bww = lo; bhh = hh; }
        }
    }
    s_key[threadIdx.x] = key; s_ww[threadIdx.x] = bww; s_hh[threadIdx.x] = bhh;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o && s_key[threadIdx.x + o] > s_key[threadIdx.x]) {
            s_key[threadIdx.x] = s_key[threadIdx.x + o]; s_ww[threadIdx.x] = s_ww[threadIdx.x + o]; s_hh[threadIdx.x] = s_hh[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_key[0] != 0) {
        const unsigned long long area = s_key[0] >> 32;
        const unsigned long long gkey = (area << 32) | (unsigned long long)(0xFFFFFFFFu - blockIdx.x);    // earlier (i,j) wins ties
        atomicMax(best_key, gkey);
        best_rect[(size_t)blockIdx.x * 4 + 0] = i; best_rect[(size_t)blockIdx.x * 4 + 1] = j;
        best_rect[(size_t)blockIdx.x * 4 + 2] = s_hh[0]; best_rect[(size_t)blockIdx.x * 4 + 3] = s_ww[0];
    }
}

__global__ void crop_pick_kernel(const unsigned long long* __restrict__ best_key, const int* __restrict__ best_rect,
                                 int* __restrict__ ans5) {
    const unsigned long long k = *best_key;
    if (k == 0) { ans5[0] = ans5[1] = ans5[2] = ans5[3] = -1; ans5[4] = 0; return; }
    const unsigned b = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
    for (int q = 0; q < 4; ++q) ans5[q] = best_rect[(size_t)b * 4 + q];
    ans5[4] = (int)(k >> 32);
}

extern "C" {

/* all_black += round(black)   (deploy_bundle.py:291).  black float [n], all_black int32 [n]. */
int stabnet_black_accumulate(const float* black, int* all_black, long n, void* stream) {
    SN_REQUIRE(black && all_black && n > 0, "black_accumulate: bad arguments");
    black_accumulate_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(black, all_black, n);
    SN_LAUNCH_CHECK("black_accumulate_kernel");
    return STABNET_OK;
}

size_t stabnet_crop_search_workspace_bytes(int H, int W, int step) {
    const size_t nstart = (size_t)((H / 2 + step - 1) / step) * ((W / 2 + step - 1) / step);
    return (size_t)(H + 1) * (W + 1) * 8 + 16 + nstart * 16 + 64;
}

/* Largest black-free rectangle as deploy_bundle.py:344-366.  ans5 (device int[5]) = {i, j, hh, ww, area}; all -1 / 0 when
 * no start pixel is free. */
int stabnet_crop_search(const int* all_black, int H, int W, int step, int* ans5, void* workspace, size_t workspace_bytes,
                        void* stream) {
    SN_REQUIRE(all_black && ans5 && workspace && H >= 2 && W >= 2 && step >= 1, "crop_search: bad arguments");
    SN_REQUIRE((long)H * W < (1L << 31), "crop_search: frame too large");
    SN_REQUIRE(workspace_bytes >= stabnet_crop_search_workspace_bytes(H, W, step) && ((uintptr_t)workspace & 7) == 0,
               "crop_search: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    long long* S = static_cast<long long*>(workspace);
    unsigned long long* key = reinterpret_cast<unsigned long long*>(S + (size_t)(H + 1) * (W + 1));
    int* rects = reinterpret_cast<int*>(key + 2);
    const int ni = (H / 2 + step - 1) / step, nj = (W / 2 + step - 1) / step;       // range(0, floor(H*0.5), step)
    if (hipMemsetAsync(key, 0, 16, st) != hipSuccess) {
        stabnet_set_error("crop_search: memset failed");
        return STABNET_ERR_LAUNCH;
    }
    integral_rows_kernel<<<cdiv(H + 1, 4), 256, 0, st>>>(all_black, H, W, S);          // one wave per row of S
    SN_LAUNCH_CHECK("integral_rows_kernel");
    integral_cols_kernel<<<cdiv(W + 1, 16), 256, 0, st>>>(H, W, S);         // 16 columns x 16 row groups per workgroup
    SN_LAUNCH_CHECK("integral_cols_kernel");
    crop_search_kernel<<<ni * nj, 256, 0, st>>>(all_black, S, H, W, step, nj, key, rects);
    SN_LAUNCH_CHECK("crop_search_kernel");
    crop_pick_kernel<<<1, 1, 0, st>>>(key, rects, ans5);
    SN_LAUNCH_CHECK("crop_pick_kernel");
    return STABNET_OK;
}

}  // extern "C"
