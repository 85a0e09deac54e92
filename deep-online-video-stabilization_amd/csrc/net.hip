// The regressor as one host-side plan: slim resnet_v2_50 (global_pool=False, output_stride=32) -> reduce_mean ->
// FC 2048/1024/512 (ReLU) -> output_layer, as built by get_resnet (s_net_bundle_nobm.py:250-264, resnet.py:44-56;
// architecture: SURVEY.md Appendix A).  The plan owns NO device memory: parameters, folded BN, workspace and I/O
// are caller buffers; a forward is one C call that only enqueues kernels (hipGraph-capturable).
//
// Parameter buffer layout (floats; every entry 16-B aligned), queried through stabnet_net_param_info():
//   [conv weights OHWI (Cin padded to 16) / conv biases / FC weights [out][in] / FC biases, in network order]
//   [all BN gammas][all BN betas]                      <- end of the trainable region
//   [all BN moving means][all BN moving variances]     <- state region
// The four BN sections share one channel ordering, so folding BN is a single elementwise kernel and the folded
// (scale, shift) buffer is [G scales][G shifts].
#include <string>
#include <vector>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "conv.h"
#include "layers.h"
#include "prof.h"
#include "ring.h"
#include "train_layers.h"
#include "warp.h"

extern "C" int stabnet_black_accumulate(const float* black, int* all_black, long n, void* stream);

enum { PK_CONV_W = 0, PK_BIAS = 1, PK_GAMMA = 2, PK_BETA = 3, PK_MEAN = 4, PK_VAR = 5, PK_FC_W = 6, PK_FC_B = 7 };
enum { S_PAD = 0, S_CONV = 1, S_POOL = 2, S_GAP = 3, S_FC = 4, S_CONV_B2B = 5 /* conv2 -> conv3 of a unit as one launch (inference) */ };
static int fc_launches(int M) { int n = 0, m0 = 0; for (; M - m0 > 8; m0 += 16) ++n; return n + (m0 < M ? 1 : 0); }   // launch_fc's passes
static const long EXT_IN = -2, EXT_OUT = -3, NONE = -1;

struct ParamEntry {
    std::string name;
    int kind;
    size_t off;
    int dims[4];
    int aux;   // conv: un-padded Cin
};

struct Step {
    int kind;
    ConvArgs conv;
    ConvArgs conv_pair;                          // training plans: the same conv over BOTH towers' batches (N doubled), own split-K plan
    int pair_ok;                                 // 1: conv_launch_pair() can run it (64x64 register-staged plan, tower rows % 64 == 0)
    long in_off, out_off, res_off, w_off, b_off, bn_off;
    long obn_off;                                // inference only: folded BN + ReLU of the CONSUMER fused behind this conv
    int inf_preactivated;                        // inference only: the input tensor already holds relu(bn(.)) -> no prologue
    long wimg_off;                               // inference only: this conv's pre-split weight image inside the image region of `fold`
    long fvec_off;                               // inference only: per-channel (bias, scale, shift, floor) vectors of a merged
                                                 // shortcut|conv1 launch, offset inside the merge region of `fold` (-1: none)
    // S_CONV_B2B (inference plans): `conv` is the unit's 3x3 conv2 (in_off, w_off; its consumer BN bn2 = mid_obn_off), `conv_b` its
    // 1x1 conv3 (w3_off, b_off, res_off, out_off, obn_off = the NEXT unit's pre-activation, if fused); mid_out_off = conv2's own
    // output tensor, written only when the step runs as two launches (bf16-operand mode)
    ConvArgs conv_b;
    long mid_obn_off, w3_off, mid_out_off;
    int N, H, W, C, Ho, Wo, k, stride, pt, pl;   // pool / pad / gap
    int M, K, Nout, relu;                        // fc
    size_t splitk_bytes;
};

struct Arena {                                    // first-fit free list over the activation workspace (floats)
    struct Blk { size_t off, size; };
    std::vector<Blk> free_;
    size_t end = 0, peak = 0;
    static size_t rnd(size_t n) { return (n + 63) & ~(size_t)63; }
    size_t alloc(size_t n) {
        n = rnd(n);
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].size >= n) {
                const size_t o = free_[i].off;
                free_[i].off += n;
                free_[i].size -= n;
                if (free_[i].size == 0) free_.erase(free_.begin() + i);
                return o;
            }
        const size_t o = end;
        end += n;
        peak = std::max(peak, end);
        return o;
    }
    void release(size_t off, size_t n) {
        n = rnd(n);
        free_.push_back({off, n});
        std::sort(free_.begin(), free_.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_.size();)
            if (free_[i].off + free_[i].size == free_[i + 1].off) {
                free_[i].size += free_[i + 1].size;
                free_.erase(free_.begin() + i + 1);
            } else
                ++i;
        if (!free_.empty() && free_.back().off + free_.back().size == end) {
            end = free_.back().off;
            free_.pop_back();
        }
    }
};

struct TensorRef { long off; size_t size; int N, H, W, C; };

struct UnitInfo {                                 // one bottleneck unit, for the explicit backward schedule
    TensorRef x, sc, r1, r2, out;
    bool proj;
    int stride, cin, dbn, depth;
    long bn_pre, bn1, bn2;                        // channel offsets inside the BN sections
    long w_sc, b_sc, w1, w2, w3, b3;
};
struct BnInfo { long chan_off; int C; long tensor_off; long M; int H, W; };

struct Net {
    int N, H, W, in_ch, in_ch_pad, n_theta, keep_all;
    int stem_rowrun = 0, in_ch_act = 0;            // the stem reads the tight 13-channel stack directly (ring kernel MODE 2)
    int bf16_operands = 0;                        // secondary fast mode of the inference forward (stabnet_net_set_bf16_operands)
    size_t stem_w_floats = 0;
    struct MergeInfo { long off; int depth, dbn; long b_sc, bn1; };
    std::vector<MergeInfo> merges;                // merged shortcut|conv1 launches of the inference plan
    size_t merge_floats = 0;
    size_t wimg_floats = 0;                       // inference plans: pre-split weight images of every convolution (conv.h), behind the merge vectors
    PackTable packs{};                            // dgrad weight re-pack of every unit conv (training)
    int train_split = 0;                          // training plans: dgrad through the packed split kernels (stabnet_net_set_bf16_operands(net, 4))
    WeightImageTable dimg{};                      // ... pre-split images of the re-packed dgrad weights, one launch per step
    WeightImageTable fimg{};                      // ... and of the forward weights (Step::wimg_off inside this region)
    size_t fimg_floats = 0;
    long dimg_of_pack[56];                        // image offset of pack entry i (-1: K not a multiple of 32)
    size_t dimg_floats = 0;
    long pack_w3[16] = {0}, pack_w2[16] = {0}, pack_w1[16] = {0}, pack_sc[16] = {0};   // wt offsets per unit                     // re-laid-out stem weights [64][7][roundup(7*in_ch, 32)] behind the folded BN
    std::vector<UnitInfo> units;
    std::vector<BnInfo> bns;
    TensorRef t_xin{}, t_c1{}, t_pool{}, t_last{}, t_gap{}, t_fc[3]{};
    long w_stem = 0, b_stem = 0, bn_post = 0, fc_w[4] = {0, 0, 0, 0}, fc_b[4] = {0, 0, 0, 0};
    int pool_pt = 0, pool_pl = 0, fc_dims[5] = {0, 0, 0, 0, 0};
    size_t max_net = 0, max_r = 0, max_w = 0, max_c = 0, reduce_floats = 0, dgrad_splitk_bytes = 0;
    std::vector<ParamEntry> params;
    std::vector<ParamEntry> bn_entries[4];        // gamma, beta, mean, var (appended after the weights)
    size_t n_floats = 0, n_trainable = 0, G = 0;
    size_t off_gamma = 0, off_beta = 0, off_mean = 0, off_var = 0;
    std::vector<Step> steps;
    size_t act_floats = 0, splitk_bytes = 0;
    double flops = 0;
    int feat_H = 0, feat_W = 0;
    struct Tap { std::string name; long off; int N, H, W, C; };
    std::vector<Tap> taps;
    // Which training forward filled a workspace last (host-side, in enqueue order): the lockstep forward keeps the FC-head
    // activations of BOTH towers in the pair's first workspace, the single-tower forward in its own -- a backward of the other
    // kind would read rows that were never written.  value: {kind, partner}; kind 1 = stabnet_tower_fwd_train, 2 / 3 = first /
    // second workspace of stabnet_towers_fwd_train.
    struct FwdStamp { int kind; const void* partner; };
    mutable std::mutex stamp_mu;
    mutable std::unordered_map<const void*, FwdStamp> fwd_stamp;
    void stamp(const void* ws, int kind, const void* partner) const {
        std::lock_guard<std::mutex> g(stamp_mu);
        fwd_stamp[ws] = FwdStamp{kind, partner};
    }
    FwdStamp stamp_of(const void* ws) const {
        std::lock_guard<std::mutex> g(stamp_mu);
        auto it = fwd_stamp.find(ws);
        return it == fwd_stamp.end() ? FwdStamp{0, nullptr} : it->second;
    }

    size_t add_param(const std::string& name, int kind, int d0, int d1, int d2, int d3, int aux) {
        ParamEntry e;
        e.name = name; e.kind = kind; e.off = n_floats;
        e.dims[0] = d0; e.dims[1] = d1; e.dims[2] = d2; e.dims[3] = d3; e.aux = aux;
        size_t n = (size_t)d0 * std::max(d1, 1) * std::max(d2, 1) * std::max(d3, 1);
        n_floats += (n + 3) & ~(size_t)3;
        params.push_back(e);
        return e.off;
    }
    long add_bn(const std::string& prefix, int C) {   // returns the channel offset inside the BN sections
        const long off = (long)G;
        const char* suffix[4] = {"/gamma", "/beta", "/moving_mean", "/moving_variance"};
        for (int s = 0; s < 4; ++s) {
            ParamEntry e;
            e.name = prefix + suffix[s]; e.kind = PK_GAMMA + s; e.off = (size_t)off;   // section base added later
            e.dims[0] = C; e.dims[1] = e.dims[2] = e.dims[3] = 0; e.aux = 0;
            bn_entries[s].push_back(e);
        }
        G += (size_t)C;
        return off;
    }
};

static TensorRef new_tensor(Arena& ar, int N, int H, int W, int C) {
    TensorRef t;
    t.N = N; t.H = H; t.W = W; t.C = C;
    t.size = (size_t)N * H * W * C;
    t.off = (long)ar.alloc(t.size);
    return t;
}

static void add_tap(Net& net, const std::string& name, const TensorRef& t) {
    net.taps.push_back({name, t.off, t.N, t.H, t.W, t.C});
}

static void same_pads(int n, int k, int s, int& before, int& out) {
    out = (n + s - 1) / s;
    const int tot = std::max((out - 1) * s + k - n, 0);
    before = tot / 2;
}

static Step conv_step(Net& net, const TensorRef& in, const TensorRef& out, int KH, int stride, int pad, long w_off,
                      long b_off, long bn_off, const TensorRef* res, int res_stride, int real_cin = 0, int rowrun = 0) {
    Step s{};
    s.kind = S_CONV;
    ConvArgs& a = s.conv;
    a.N = in.N; a.H = in.H; a.W = in.W; a.Cin = in.C; a.Cout = out.C; a.KH = KH; a.KW = KH; a.stride = stride; a.pad = pad;
    a.up = 1; a.Ho = out.H; a.Wo = out.W;
    a.res_H = res ? res->H : out.H; a.res_W = res ? res->W : out.W; a.res_stride = res ? res_stride : 1;
    a.relu_out = 0;
    a.cin_real = real_cin ? real_cin : in.C;
    a.in_scale_expected = bn_off >= 0 ? 1 : 0;
    a.rowrun = rowrun;
    s.splitk_bytes = conv_plan(a);
    net.splitk_bytes = std::max(net.splitk_bytes, s.splitk_bytes);
    s.in_off = in.off; s.out_off = out.off; s.res_off = res ? res->off : NONE;
    s.w_off = w_off; s.b_off = b_off; s.bn_off = bn_off;
    s.obn_off = NONE; s.inf_preactivated = 0; s.fvec_off = NONE; s.wimg_off = NONE;
    net.flops += 2.0 * a.M * (double)(a.KH * a.KW * (real_cin ? real_cin : a.Cin)) * a.Cout;   // algorithmic (un-padded)
    return s;
}

static Net* build_net(int N, int H, int W, int in_ch, int n_theta, int keep_all) {
    Net* net = new Net();
    net->N = N; net->H = H; net->W = W; net->in_ch = in_ch; net->n_theta = n_theta; net->keep_all = keep_all;
    net->in_ch_pad = (in_ch + 15) / 16 * 16;
    static const int want_rowrun = []() { const char* v = getenv("STABNET_STEM_ROWRUN"); return v ? atoi(v) : 1; }();
    static const int want_ring = []() { const char* v = getenv("STABNET_CONV_RING"); return v ? atoi(v) : 1; }();   // (debug switch)
    net->stem_rowrun = (want_rowrun && want_ring && 7 * in_ch <= 128) ? 1 : 0;
    net->in_ch_act = net->stem_rowrun ? in_ch : net->in_ch_pad;
    net->stem_w_floats = net->stem_rowrun ? (size_t)64 * 7 * 32 * ((7 * in_ch + 31) / 32) : 0;
    Arena ar;
    auto done = [&](const TensorRef& t) { if (!keep_all) ar.release((size_t)t.off, t.size); };
    const std::string R = "resnet_v2_50/";

    // stem: conv2d_same(64, 7, stride 2) with bias, no BN/ReLU; then max_pool2d 3x3/2 SAME
    // row-run stem: the stack lives as a tight in_ch-channel image with a 3-pixel zero border (+ one slack row: the last
    // run of the last row reads up to 31 floats past its taps); STABNET_STEM_ROWRUN=0: channels padded to 16, no border
    const int border = net->stem_rowrun ? 3 : 0;
    TensorRef xin = net->stem_rowrun ? new_tensor(ar, N, H + 2 * border + 1, W + 2 * border, in_ch)
                                     : new_tensor(ar, N, H, W, net->in_ch_act);
    {
        Step s{};
        s.kind = S_PAD; s.in_off = EXT_IN; s.out_off = xin.off; s.N = N; s.H = H; s.W = W; s.C = in_ch;
        net->steps.push_back(s);
    }
    const int H1 = (H + 6 - 7) / 2 + 1, W1 = (W + 6 - 7) / 2 + 1;
    TensorRef c1 = new_tensor(ar, N, H1, W1, 64);
    {
        const long w = (long)net->add_param(R + "conv1/weights", PK_CONV_W, 64, 7, 7, net->in_ch_pad, in_ch);
        const long b = (long)net->add_param(R + "conv1/biases", PK_BIAS, 64, 0, 0, 0, 0);
        TensorRef xlog = xin;                   // logical (unbordered) geometry of the same buffer
        xlog.H = H; xlog.W = W;
        net->steps.push_back(conv_step(*net, xlog, c1, 7, 2, 3, w, b, NONE, nullptr, 1, in_ch, net->stem_rowrun));
        net->w_stem = w; net->b_stem = b;
    }
    net->t_xin = xin; net->t_c1 = c1;
    done(xin);
    add_tap(*net, "conv1", c1);
    int pt, pl, H2, W2;
    same_pads(H1, 3, 2, pt, H2);
    same_pads(W1, 3, 2, pl, W2);
    TensorRef cur = new_tensor(ar, N, H2, W2, 64);
    {
        Step s{};
        s.kind = S_POOL; s.in_off = c1.off; s.out_off = cur.off; s.obn_off = NONE;
        s.N = N; s.H = H1; s.W = W1; s.C = 64; s.Ho = H2; s.Wo = W2; s.k = 3; s.stride = 2; s.pt = pt; s.pl = pl;
        net->steps.push_back(s);
    }
    done(c1);
    add_tap(*net, "pool1", cur);
    net->t_pool = cur; net->pool_pt = pt; net->pool_pl = pl;

    struct Blk { const char* name; int depth, dbn, units, stride; };
    const Blk blocks[4] = {{"block1", 256, 64, 3, 2}, {"block2", 512, 128, 4, 2}, {"block3", 1024, 256, 6, 2},
                           {"block4", 2048, 512, 3, 1}};
    for (const Blk& b : blocks) {
        for (int u = 1; u <= b.units; ++u) {
            const int stride = (u == b.units) ? b.stride : 1;
            const std::string S = R + b.name + "/unit_" + std::to_string(u) + "/bottleneck_v2/";
            const int cin = cur.C;
            const long bn_pre = net->add_bn(S + "preact", cin);
            net->bns.push_back({bn_pre, cin, cur.off, (long)N * cur.H * cur.W, cur.H, cur.W});
            UnitInfo ui{};
            ui.x = cur; ui.stride = stride; ui.cin = cin; ui.dbn = b.dbn; ui.depth = b.depth; ui.bn_pre = bn_pre;
            ui.w_sc = ui.b_sc = NONE;
            const int Ho = (cur.H + 2 - 3) / stride + 1, Wo = (cur.W + 2 - 3) / stride + 1;
            TensorRef sc = cur;
            bool own_sc = false;
            int merged_ld = 0;
            TensorRef merged_buf{};
            // Inference plan, projection units (the first unit of every block): nothing reads the RAW unit input (the shortcut
            // is a conv of the pre-activation), so its producer -- the max-pool or the previous block's last conv3 -- applies
            // this unit's preact BN + ReLU itself and both convs of the unit run prologue-free on the ring kernel.
            const bool pre_act = !keep_all && want_ring && cin != b.depth && !net->steps.empty() &&
                                 (net->steps.back().kind == S_POOL || net->steps.back().kind == S_CONV || net->steps.back().kind == S_CONV_B2B) &&
                                 net->steps.back().out_off == cur.off && net->steps.back().obn_off < 0;
            if (pre_act) net->steps.back().obn_off = bn_pre;
            auto mark_preactivated = [&](Step& st_) {
                st_.inf_preactivated = 1;
                st_.conv.in_scale_expected = 0;
                st_.splitk_bytes = conv_plan(st_.conv);
                net->splitk_bytes = std::max(net->splitk_bytes, st_.splitk_bytes);
            };
            // Inference plan of a projection unit: shortcut (depth channels, bias) and conv1 (dbn channels, then bn1 + ReLU)
            // are two 1x1 convolutions of the SAME activated input -> ONE launch with Cout = depth + dbn writing one buffer
            // [M][depth + dbn] (weights adjacent in the parameter buffer; bias / BN scale / shift / activation floor per
            // channel come from the fold buffer).  conv2 then reads its 64..512 channels with a pixel stride of depth + dbn,
            // conv3 reads the residual with that row stride.
            const bool merged = pre_act;
            TensorRef r1;
            if (cin != b.depth) {       // projection shortcut: conv1x1(preact) + bias, stride 1 here
                own_sc = true;
                const long w = (long)net->add_param(S + "shortcut/weights", PK_CONV_W, b.depth, 1, 1, cin, cin);
                const long w1 = (long)net->add_param(S + "conv1/weights", PK_CONV_W, b.dbn, 1, 1, cin, cin);   // adjacent to w
                const long bb = (long)net->add_param(S + "shortcut/biases", PK_BIAS, b.depth, 0, 0, 0, 0);
                ui.w_sc = w; ui.b_sc = bb; ui.w1 = w1;
                if (merged) {
                    const int Ct = b.depth + b.dbn;
                    TensorRef xs = new_tensor(ar, N, cur.H, cur.W, Ct);
                    net->steps.push_back(conv_step(*net, cur, xs, 1, 1, 0, w, NONE, bn_pre, nullptr, 1));
                    mark_preactivated(net->steps.back());
                    net->steps.back().fvec_off = (long)net->merge_floats;             // relative to the merge region of `fold`
                    net->merges.push_back({(long)net->merge_floats, b.depth, b.dbn, bb, -1});
                    net->merge_floats += (size_t)4 * Ct;
                    sc = xs; sc.C = b.depth;                                   // views of the shared buffer (row stride Ct)
                    r1 = xs; r1.off = xs.off + b.depth; r1.C = b.dbn;
                    merged_ld = Ct; merged_buf = xs;
                } else {
                    sc = new_tensor(ar, N, cur.H, cur.W, b.depth);
                    net->steps.push_back(conv_step(*net, cur, sc, 1, 1, 0, w, bb, bn_pre, nullptr, 1));
                    r1 = new_tensor(ar, N, cur.H, cur.W, b.dbn);
                    net->steps.push_back(conv_step(*net, cur, r1, 1, 1, 0, w1, NONE, bn_pre, nullptr, 1));
                }
            } else {
                r1 = new_tensor(ar, N, cur.H, cur.W, b.dbn);
                const long w = (long)net->add_param(S + "conv1/weights", PK_CONV_W, b.dbn, 1, 1, cin, cin);
                net->steps.push_back(conv_step(*net, cur, r1, 1, 1, 0, w, NONE, bn_pre, nullptr, 1));
                ui.w1 = w;
            }
            const size_t i_conv1 = net->steps.size() - 1;
            const long bn1 = net->add_bn(S + "conv1/BatchNorm", b.dbn);
            net->bns.push_back({bn1, b.dbn, r1.off, (long)N * r1.H * r1.W, r1.H, r1.W});
            TensorRef r2 = new_tensor(ar, N, Ho, Wo, b.dbn);
            {
                const long w = (long)net->add_param(S + "conv2/weights", PK_CONV_W, b.dbn, 3, 3, b.dbn, b.dbn);
                net->steps.push_back(conv_step(*net, r1, r2, 3, stride, 1, w, NONE, bn1, nullptr, 1));
                if (merged_ld) net->steps.back().conv.x_ld = merged_ld;
                ui.w2 = w;
            }
            if (!merged_ld) done(r1);
            const size_t i_conv2 = net->steps.size() - 1;
            const long bn2 = net->add_bn(S + "conv2/BatchNorm", b.dbn);
            // Inference plan: conv1 and conv2 apply their consumer's folded BN + ReLU in the epilogue, so conv2 and conv3
            // read activated tensors and run prologue-free (LDS-DMA ring kernel).  Same arithmetic per element
            // (fma + max), applied once per activation instead of once per use.  Training keeps the raw conv outputs
            // (the batch statistics are taken over them).
            if (!merged_ld) net->steps[i_conv1].obn_off = bn1;
            else net->merges.back().bn1 = bn1;
            net->steps[i_conv2].obn_off = bn2;
            net->steps[i_conv2].inf_preactivated = 1;
            if (!keep_all) {            // inference plan: conv2 runs prologue-free (ring kernel) -> its own split-K choice
                Step& c2 = net->steps[i_conv2];
                c2.conv.in_scale_expected = 0;
                c2.splitk_bytes = conv_plan(c2.conv);
                net->splitk_bytes = std::max(net->splitk_bytes, c2.splitk_bytes);
            }
            net->bns.push_back({bn2, b.dbn, r2.off, (long)N * r2.H * r2.W, r2.H, r2.W});
            TensorRef nxt = new_tensor(ar, N, Ho, Wo, b.depth);
            {
                const long w = (long)net->add_param(S + "conv3/weights", PK_CONV_W, b.depth, 1, 1, b.dbn, b.dbn);
                const long bb = (long)net->add_param(S + "conv3/biases", PK_BIAS, b.depth, 0, 0, 0, 0);
                // identity shortcut of a strided unit = subsample(x, stride): read the residual at (oy*s, ox*s)
                net->steps.push_back(conv_step(*net, r2, nxt, 1, 1, 0, w, bb, bn2, &sc, own_sc ? 1 : stride));
                if (merged_ld) net->steps.back().conv.res_ld = merged_ld;
                net->steps.back().inf_preactivated = 1;
                if (!keep_all) {
                    Step& c3 = net->steps.back();
                    c3.conv.in_scale_expected = 0;
                    c3.splitk_bytes = conv_plan(c3.conv);
                    net->splitk_bytes = std::max(net->splitk_bytes, c3.splitk_bytes);
                }
                ui.w3 = w; ui.b3 = bb;
            }
            // Inference plan, block 1 / block 2 units with enough 64-pixel tiles to fill the chip: conv2 and conv3 as ONE launch
            // (conv_b2b_kernel.h): the activated conv2 tile stays in LDS, the low-K conv3 launch and the round trip of its input go.
            // OFF by default (STABNET_CONV_B2B_PLAN=1 turns it on): measured in the 720p frame (same box, same build) every choice
            // of units loses to the two launches -- all six eligible units 549.0 against 556.4 frames/s, the three 225-tile
            // units of block 1 / block 2 550.5, block 1's strided unit alone 555.3 (the fused
            // launch has no more matrix throughput per step than the two kernels, 80 / 136 KB of LDS leave two / one workgroup per
            // CU, and the 1x1 phase costs its 8 K-steps plus four epilogues that nothing overlaps).  The window [min_tiles,
            // max_tiles] and the d_b mask (bit 0: 64, bit 1: 128) select units when the plan switch is on.
            if (!keep_all && i_conv2 + 2 == net->steps.size()) {
                auto env = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
                static const int plan_on = env("STABNET_CONV_B2B_PLAN", 0);
                static const int min_tiles = env("STABNET_CONV_B2B_MIN_TILES", 200), max_tiles = env("STABNET_CONV_B2B_MAX_TILES", 1 << 30);
                static const int cmask = env("STABNET_CONV_B2B_CMASK", 3);
                const Step c2 = net->steps[i_conv2], c3 = net->steps[i_conv2 + 1];
                const int tiles = (c3.conv.M + 63) / 64;
                if (plan_on && conv_b2b_supported(c2.conv, c3.conv) && tiles >= min_tiles && tiles <= max_tiles &&
                    ((c2.conv.Cout == 64 && (cmask & 1)) || (c2.conv.Cout == 128 && (cmask & 2)))) {
                    Step f = c2;
                    f.kind = S_CONV_B2B;
                    f.conv_b = c3.conv;
                    f.mid_obn_off = c2.obn_off; f.mid_out_off = c2.out_off;
                    f.w3_off = c3.w_off; f.b_off = c3.b_off; f.res_off = c3.res_off; f.out_off = c3.out_off;
                    f.obn_off = NONE;                        // conv3's consumer BN: set by the next unit if it is a projection unit
                    f.splitk_bytes = std::max(c2.splitk_bytes, c3.splitk_bytes);
                    net->steps.pop_back();
                    net->steps.pop_back();
                    net->steps.push_back(f);
                }
            }
            ui.sc = sc; ui.r1 = r1; ui.r2 = r2; ui.out = nxt; ui.proj = own_sc; ui.bn1 = bn1; ui.bn2 = bn2;
            net->units.push_back(ui);
            net->max_net = std::max({net->max_net, cur.size, nxt.size, sc.size});
            net->max_r = std::max({net->max_r, r1.size, r2.size});
            net->max_w = std::max({net->max_w, (size_t)b.dbn * 9 * b.dbn, (size_t)b.depth * cin, (size_t)b.depth * b.dbn});
            net->max_c = std::max({net->max_c, (size_t)b.depth, (size_t)cin});
            done(r2);
            if (merged_ld) done(merged_buf);
            else if (own_sc) done(sc);
            done(cur);
            cur = nxt;
            add_tap(*net, std::string(b.name) + "/unit_" + std::to_string(u), cur);
        }
    }
    net->feat_H = cur.H; net->feat_W = cur.W;
    const long bn_post = net->add_bn(R + "postnorm", cur.C);
    net->bns.push_back({bn_post, cur.C, cur.off, (long)N * cur.H * cur.W, cur.H, cur.W});
    net->bn_post = bn_post; net->t_last = cur;
    TensorRef g = new_tensor(ar, N, 1, 1, cur.C);
    net->t_gap = g;
    {
        Step s{};
        s.kind = S_GAP; s.in_off = cur.off; s.out_off = g.off; s.bn_off = bn_post; s.N = N; s.H = cur.H; s.W = cur.W; s.C = cur.C;
        net->steps.push_back(s);
    }
    done(cur);
    add_tap(*net, "global_pool", g);
    const int dims[4] = {cur.C, 2048, 1024, 512};
    TensorRef f = g;
    // (the head's few KB of vectors stay allocated together; nothing is gained by recycling them)
    for (int k = 1; k <= 3; ++k) {
        TensorRef o = new_tensor(ar, N, 1, 1, dims[k]);
        Step s{};
        s.kind = S_FC; s.in_off = f.off; s.out_off = o.off; s.M = N; s.K = dims[k - 1]; s.Nout = dims[k]; s.relu = 1;
        s.w_off = (long)net->add_param("fc/fc/fc_" + std::to_string(k) + "/weights", PK_FC_W, dims[k], dims[k - 1], 0, 0, 0);
        s.b_off = (long)net->add_param("fc/fc/fc_" + std::to_string(k) + "/biases", PK_FC_B, dims[k], 0, 0, 0, 0);
        net->steps.push_back(s);
        net->flops += 2.0 * N * dims[k - 1] * dims[k];
        net->t_fc[k - 1] = o; net->fc_w[k - 1] = s.w_off; net->fc_b[k - 1] = s.b_off;
        f = o;
    }
    {
        Step s{};
        s.kind = S_FC; s.in_off = f.off; s.out_off = EXT_OUT; s.M = N; s.K = 512; s.Nout = n_theta; s.relu = 0;
        s.w_off = (long)net->add_param("fc/fc_weights", PK_FC_W, n_theta, 512, 0, 0, 0);
        s.b_off = (long)net->add_param("fc/fc_bias", PK_FC_B, n_theta, 0, 0, 0, 0);
        net->steps.push_back(s);
        net->flops += 2.0 * N * 512 * n_theta;
        net->fc_w[3] = s.w_off; net->fc_b[3] = s.b_off;
    }
    for (int k = 0; k < 4; ++k) net->fc_dims[k] = dims[k];
    net->fc_dims[4] = n_theta;
    // BN sections
    net->off_gamma = net->n_floats;
    net->off_beta = net->off_gamma + net->G;
    net->n_trainable = net->off_beta + net->G;
    net->off_mean = net->n_trainable;
    net->off_var = net->off_mean + net->G;
    net->n_floats = net->off_var + net->G;
    const size_t base[4] = {net->off_gamma, net->off_beta, net->off_mean, net->off_var};
    for (int s = 0; s < 4; ++s)
        for (ParamEntry e : net->bn_entries[s]) {
            e.off += base[s];
            net->params.push_back(e);
        }
    {   // dgrad weight re-pack table, in the order run_backward consumes it
        PackTable& t = net->packs;
        t.n = 0; t.prefix[0] = 0;
        auto add = [&](long w_off, int Cout, int K, int Cin, int stride = 1) -> long {
            const long off = t.prefix[t.n];
            t.d[t.n] = {w_off, Cout, K, Cin, dgrad_kperm(K, K, stride)};
            t.prefix[t.n + 1] = off + (long)Cout * K * K * Cin;
            ++t.n;
            return off;
        };
        for (size_t ui = 0; ui < net->units.size() && ui < 16; ++ui) {
            const UnitInfo& u = net->units[ui];
            net->pack_w3[ui] = add(u.w3, u.depth, 1, u.dbn);
            net->pack_w2[ui] = add(u.w2, u.dbn, 3, u.dbn, u.stride);
            net->pack_w1[ui] = add(u.w1, u.dbn, 1, u.cin);
            net->pack_sc[ui] = u.proj ? add(u.w_sc, u.depth, 1, u.cin) : -1;
        }
        // images of the re-packed dgrad weights: entry i is wt[prefix[i] ...] = [Cin][K*K*Cout] (the dgrad's output channels x its K)
        net->dimg.n = 0;
        for (int i = 0; i < t.n; ++i) {
            const int Kd = t.d[i].K * t.d[i].K * t.d[i].Cout;
            net->dimg_of_pack[i] = -1;
            if (Kd % 32 != 0 || net->dimg.n >= 64) continue;
            net->dimg_of_pack[i] = (long)net->dimg_floats;
            weight_image_table_add(net->dimg, t.prefix[i], (long)net->dimg_floats, t.d[i].Cin, Kd);
            net->dimg_floats += conv_weight_image_floats(t.d[i].Cin, Kd);
        }
    }
    if (net->keep_all) {   // the lockstep forward launches a conv ONCE for both towers where the kernel allows it
        for (Step& st_ : net->steps) {
            if (st_.kind != S_CONV) continue;
            st_.conv_pair = st_.conv;
            st_.conv_pair.N = 2 * st_.conv.N;
            const size_t bytes = conv_plan(st_.conv_pair);
            st_.pair_ok = conv_pair_supported(st_.conv_pair) && st_.conv.M % 64 == 0 &&
                          (long)st_.conv_pair.N * (st_.conv.H + 2 * st_.conv.pad) * (st_.conv.W + 2 * st_.conv.pad) * st_.conv.Cin < (1L << 30) &&
                          (long)st_.conv_pair.M * st_.conv.Cout < (1L << 30);
            if (st_.pair_ok) net->splitk_bytes = std::max(net->splitk_bytes, bytes);
        }
    }
    if (net->keep_all) {   // training plans: images of the forward weights, re-split once per step when the plan runs in split mode
        net->fimg.n = 0;
        for (Step& st_ : net->steps) {
            if (st_.kind != S_CONV || st_.conv.K % 32 != 0 || st_.conv.rowrun || net->fimg.n >= 64) continue;
            st_.wimg_off = (long)net->fimg_floats;
            weight_image_table_add(net->fimg, st_.w_off, st_.wimg_off, st_.conv.Cout, st_.conv.K);
            net->fimg_floats += conv_weight_image_floats(st_.conv.Cout, st_.conv.K);
        }
    }
    if (!net->keep_all) {  // pre-split weight images for the packed split kernel (stabnet_net_set_bf16_operands(net, 4))
        for (Step& st_ : net->steps) {
            if (st_.kind != S_CONV || st_.conv.K % 32 != 0) continue;
            st_.wimg_off = (long)net->wimg_floats;
            net->wimg_floats += conv_weight_image_floats(st_.conv.Cout, st_.conv.K);
        }
    }
    net->act_floats = ar.peak;
    net->splitk_bytes = std::max(net->splitk_bytes, sizeof(float) * (size_t)N * std::max(8, gap_chunks(net->t_last.H * net->t_last.W)) * net->t_last.C);
    net->max_net = std::max({net->max_net, net->t_c1.size, net->t_pool.size});
    return net;
}

// What the online loop hangs behind the regressor's head: get_4_pts + get_Hs (+ the ring-head advance).  With the shortened
// head (head.hip) they ride in the output layer's launch; `done` tells the caller whether run_forward did them.
struct MeshTail { int gh, gw; float lim; float* Hs; int* head_adv; int depth; bool done; const float* prefetch_src = nullptr; /* the frame the sampler behind the mesh gathers from */ };

static int run_forward(const Net* net, const float* params, const float* fold, const float* x, float* theta,
                       float* ws, hipStream_t st, Prof* prof = nullptr, bool skip_pad = false, MeshTail* mesh = nullptr) {
    float* splitk = ws + net->act_floats;
    const float* scale = fold;
    const float* shift = fold + net->G;
    // inference head, batch <= 8: reduce_mean finalize folded into fc_1, output layer + mesh as one launch (head.hip)
    const bool fused_head = head_fused_supported(net->N, net->t_last.C, net->fc_dims);
    if (mesh) mesh->done = false;
    for (const Step& s : net->steps) {
        int rc = STABNET_OK;
        if (s.kind == S_PAD && skip_pad) continue;
        if (s.kind == S_FC && fused_head && (s.in_off == net->t_gap.off || s.out_off == EXT_OUT)) continue;   // fc_1 rides with the GAP, the output layer with the mesh
        const bool rec = (s.kind != S_CONV && s.kind != S_CONV_B2B) && prof != nullptr && prof->begin(st);
        switch (s.kind) {
            case S_PAD:
                if (net->stem_rowrun) rc = launch_embed_border(x, s.N, s.H, s.W, s.C, 3, ws + s.out_off, st);   // (+ the slack row)
                else rc = launch_pad_channels(x, ws + s.out_off, (long)s.N * s.H * s.W, s.C, net->in_ch_act, st);
                break;
            case S_CONV: {
                ConvArgs a = s.conv;
                a.x = ws + s.in_off;
                a.y = ws + s.out_off;
                a.w = a.rowrun ? fold + 2 * net->G : params + s.w_off;
                a.bias = s.b_off >= 0 ? params + s.b_off : nullptr;
                a.residual = s.res_off >= 0 ? ws + s.res_off : nullptr;
                const bool prologue = s.bn_off >= 0 && !s.inf_preactivated;
                a.in_scale = prologue ? scale + s.bn_off : nullptr;
                a.in_shift = prologue ? shift + s.bn_off : nullptr;
                if (s.obn_off >= 0) {
                    a.out_scale = scale + s.obn_off;
                    a.out_shift = shift + s.obn_off;
                    a.relu_out = 1;
                }
                if (s.fvec_off >= 0) {                     // merged shortcut | conv1: everything per channel
                    const float* v = fold + 2 * net->G + net->stem_w_floats + s.fvec_off;
                    a.bias = v; a.out_scale = v + a.Cout; a.out_shift = v + 2 * a.Cout; a.out_floor = v + 3 * a.Cout;
                    a.relu_out = 0;
                }
                a.partial = splitk;
                rc = conv_launch(a, st, prof, net->bf16_operands,
                                 (net->bf16_operands == 4 && s.wimg_off >= 0 && !net->keep_all) ? fold + 2 * net->G + net->stem_w_floats + net->merge_floats + s.wimg_off : nullptr);
                break;
            }
            case S_CONV_B2B: {
                ConvArgs a = s.conv, b = s.conv_b;
                a.x = ws + s.in_off; a.w = params + s.w_off; a.y = ws + s.mid_out_off;
                a.out_scale = scale + s.mid_obn_off; a.out_shift = shift + s.mid_obn_off; a.relu_out = 1;
                b.x = ws + s.mid_out_off; b.w = params + s.w3_off; b.y = ws + s.out_off;
                b.bias = s.b_off >= 0 ? params + s.b_off : nullptr;
                b.residual = s.res_off >= 0 ? ws + s.res_off : nullptr;
                if (s.obn_off >= 0) { b.out_scale = scale + s.obn_off; b.out_shift = shift + s.obn_off; b.relu_out = 1; }
                a.partial = b.partial = splitk;
                if (net->bf16_operands) {                  // the fused kernel is fp32 only: the secondary mode runs the two launches
                    rc = conv_launch(a, st, prof, 1);
                    if (!rc) rc = conv_launch(b, st, prof, 1);
                } else {
                    rc = conv_b2b_launch(a, b, st, prof);
                }
                break;
            }
            case S_POOL:
                rc = launch_max_pool(ws + s.in_off, ws + s.out_off, s.N, s.H, s.W, s.C, s.Ho, s.Wo, s.k, s.stride, s.pt,
                                     s.pl, s.obn_off >= 0 ? scale + s.obn_off : nullptr, s.obn_off >= 0 ? shift + s.obn_off : nullptr, st);
                break;
            case S_GAP:
                if (fused_head) {                                            // partial sums + fc_1 (partials in the split-K scratch)
                    rc = launch_gap_fc1(ws + s.in_off, scale + s.bn_off, shift + s.bn_off, s.N, s.H * s.W, s.C, splitk, ws + s.out_off,
                                        params + net->fc_w[0], params + net->fc_b[0], ws + net->t_fc[0].off, net->fc_dims[1], st);
                    break;
                }
                rc = launch_gap_bn_relu(ws + s.in_off, scale + s.bn_off, shift + s.bn_off, s.N, s.H * s.W, s.C,
                                        ws + s.out_off, splitk, st);          // partials live in the split-K scratch
                break;
            case S_FC:
                rc = launch_fc(ws + s.in_off, params + s.w_off, params + s.b_off,
                               s.out_off == EXT_OUT ? theta : ws + s.out_off, s.M, s.K, s.Nout, s.relu, st);
                break;
        }
        if (rec) {
            static const int kmap[5] = {PK_KERNEL_PAD, 0, PK_KERNEL_POOL, PK_KERNEL_GAP, PK_KERNEL_FC};
            double bytes = 0;
            if (s.kind == S_PAD) bytes = 4.0 * s.N * s.H * s.W * (s.C + net->in_ch_pad);
            if (s.kind == S_POOL) bytes = 4.0 * s.N * s.C * ((double)s.H * s.W + (double)s.Ho * s.Wo);
            if (s.kind == S_GAP) bytes = 4.0 * s.N * s.C * ((double)s.H * s.W + 1);
            if (s.kind == S_FC) bytes = 4.0 * ((double)s.K * s.Nout + (double)s.M * (s.K + s.Nout));
            if (s.kind == S_GAP && fused_head) bytes += 4.0 * net->fc_dims[0] * net->fc_dims[1];      // + fc_1's weights
            prof->end(st, kmap[s.kind], s.kind == S_FC ? 2.0 * s.M * s.K * s.Nout : 0.0, bytes);
        }
        if (rc) return rc;
    }
    if (fused_head) {                                         // output_layer (+ mesh + ring-head advance) as one launch
        const bool rec = prof != nullptr && prof->begin(st);
        const int rc = launch_theta_mesh(ws + net->t_fc[2].off, params + net->fc_w[3], params + net->fc_b[3], net->N, net->n_theta, theta,
                                         mesh ? mesh->gh : 1, mesh ? mesh->gw : 1, mesh ? mesh->lim : 0.f, mesh ? mesh->Hs : nullptr,
                                         mesh ? mesh->head_adv : nullptr, mesh ? mesh->depth : 1, st,
                                         mesh ? mesh->prefetch_src : nullptr, net->H, net->W);
        if (rec) prof->end(st, PK_KERNEL_HEAD, 2.0 * net->N * 512 * net->n_theta, 4.0 * 512 * net->n_theta);
        if (rc) return rc;
        if (mesh) mesh->done = true;
    }
    return STABNET_OK;
}

extern "C" {

int stabnet_prof_create(void** out, int max_records) {
    SN_REQUIRE(out && max_records > 0 && max_records <= (1 << 20), "prof_create: bad arguments");
    Prof* p = new Prof();
    p->cap = max_records;
    p->ev.resize(2 * (size_t)max_records);
    p->kind.resize(max_records); p->flops.resize(max_records); p->bytes.resize(max_records);
    p->shape.resize(4 * (size_t)max_records);
    for (auto& e : p->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            stabnet_set_error("prof_create: hipEventCreate failed");
            return STABNET_ERR_LAUNCH;
        }
    *out = p;
    return STABNET_OK;
}
void stabnet_prof_destroy(void* pp) {
    Prof* p = static_cast<Prof*>(pp);
    if (!p) return;
    for (auto& e : p->ev) (void)hipEventDestroy(e);
    delete p;
}
int stabnet_prof_reset(void* pp) {
    SN_REQUIRE(pp, "prof_reset: null");
    static_cast<Prof*>(pp)->n = 0;
    return STABNET_OK;
}
/* Records one event pair with nothing between them on `stream` (kind 0): the per-pair overhead to subtract. */
int stabnet_prof_record_empty(void* pp, void* stream) {
    Prof* p = static_cast<Prof*>(pp);
    SN_REQUIRE(p, "prof_record_empty: null");
    if (p->begin((hipStream_t)stream)) p->end((hipStream_t)stream, 0, 0.0, 0.0);
    return STABNET_OK;
}
int stabnet_prof_num_records(const void* pp) { return pp ? static_cast<const Prof*>(pp)->n : -1; }
/* The stream the records were taken on must have been synchronised by the caller. */
int stabnet_prof_record(const void* pp, int idx, int* kind, float* ms, double* flops, double* bytes) {
    const Prof* p = static_cast<const Prof*>(pp);
    SN_REQUIRE(p && idx >= 0 && idx < p->n && kind && ms && flops && bytes, "prof_record: bad arguments");
    hipError_t e = hipEventElapsedTime(ms, p->ev[2 * idx], p->ev[2 * idx + 1]);
    if (e != hipSuccess) {
        stabnet_set_error("prof_record: %s", hipGetErrorString(e));
        return STABNET_ERR_LAUNCH;
    }
    *kind = p->kind[idx]; *flops = p->flops[idx]; *bytes = p->bytes[idx];
    return STABNET_OK;
}
/* GEMM view of a record: shape4 = {M, N, K, split-K}; zeros for non-GEMM kernels. */
int stabnet_prof_record_shape(const void* pp, int idx, int* shape4) {
    const Prof* p = static_cast<const Prof*>(pp);
    SN_REQUIRE(p && idx >= 0 && idx < p->n && shape4, "prof_record_shape: bad arguments");
    for (int i = 0; i < 4; ++i) shape4[i] = p->shape[4 * idx + i];
    return STABNET_OK;
}
const char* stabnet_prof_kind_name(int kind) {
    switch (kind) {
        case PK_KERNEL_PAD: return "pad_channels_kernel";
        case PK_KERNEL_POOL: return "max_pool_kernel";
        case PK_KERNEL_GAP: return "gap_bn_relu_partial_kernel";
        case PK_KERNEL_FC: return "fc_kernel";
        case PK_KERNEL_MESH: return "mesh_homography_kernel";
        case PK_KERNEL_HEAD: return "theta_mesh_kernel";
        case PK_KERNEL_WARP: return "warp_sample_kernel";
        case PK_KERNEL_ASSEMBLE: return "stack_assemble_bordered_kernel";
        case PK_KERNEL_PUSH: return "ring_push_kernel";
        case PK_KERNEL_SPLITK_REDUCE: return "conv_splitk_reduce_kernel";
        case PK_KERNEL_WGRAD: return "conv_wgrad_f32_kernel";
        default: break;
    }
    if (kind >= PK_KERNEL_WGRAD_SAME && kind < PK_KERNEL_WGRAD_SAME + 6) {      // names as rocprofv3 prints them: <K3, PRO, BIAS>
        static const char* const names[6] = {"conv_wgrad_same_f32_kernel<0, 0, 0>", "conv_wgrad_same_f32_kernel<0, 1, 0>",
                                             "conv_wgrad_same_f32_kernel<1, 0, 0>", "conv_wgrad_same_f32_kernel<1, 1, 0>",
                                             "conv_wgrad_same_f32_kernel<0, 0, 1>", "conv_wgrad_same_f32_kernel<0, 1, 1>"};
        return names[kind - PK_KERNEL_WGRAD_SAME];
    }
    if (kind >= PK_KERNEL_CONV_PAIR && kind < PK_KERNEL_CONV_PAIR + 4) {
        static const char* const names[4] = {"conv_igemm_f32_pair_kernel<64, 64, 16, 32, 32, 0>", "conv_igemm_f32_pair_kernel<64, 64, 32, 32, 32, 0>",
                                             "conv_igemm_f32_pair_kernel<64, 64, 16, 32, 32, 1>", "conv_igemm_f32_pair_kernel<64, 64, 32, 32, 32, 1>"};
        return names[kind - PK_KERNEL_CONV_PAIR];
    }
    if (kind == PK_KERNEL_CONV_KG) return "conv_ring_f32_kernel<0, 0, 3, 0>";
    if (kind == PK_KERNEL_CONV_KG + 1) return "conv_ring_f32_kernel<1, 0, 3, 0>";
    if (kind == PK_KERNEL_CONV_KG + 2) return "conv_ring_f32_kernel<0, 0, 1, 1>";
    if (kind == PK_KERNEL_CONV_KG + 3) return "conv_ring_f32_kernel<0, 0, 2, 1>";
    // names as rocprofv3 prints the template instantiation <MODE, BF16>
    if (kind == PK_KERNEL_CONV_RING) return "conv_ring_f32_kernel<0, 0, 1, 0>";
    if (kind == PK_KERNEL_CONV_RING + 1) return "conv_ring_f32_kernel<1, 0, 1, 0>";
    if (kind == PK_KERNEL_CONV_RING + 2) return "conv_ring_f32_kernel<2, 0, 1, 0>";
    if (kind == PK_KERNEL_CONV_RING + 3) return "conv_ring_f32_kernel<0, 1, 1, 0>";
    if (kind == PK_KERNEL_CONV_RING + 4) return "conv_ring_f32_kernel<1, 1, 1, 0>";
    if (kind == PK_KERNEL_CONV_RING + 5) return "conv_ring_f32_kernel<2, 1, 1, 0>";
    if (kind >= PK_KERNEL_CONV_PACKED && kind <= PK_KERNEL_CONV_PACKED + 6) {      // conv_ring_f32_kernel<MODE, 4 | 5, KG, PRO>
        static const int shape[7][3] = {{0, 1, 0}, {1, 1, 0}, {2, 1, 0}, {0, 1, 1}, {0, 2, 0}, {1, 2, 0}, {0, 2, 1}};
        static thread_local char pbuf[7][48];
        const int i = kind - PK_KERNEL_CONV_PACKED;
        snprintf(pbuf[i], sizeof(pbuf[i]), "conv_ring_f32_kernel<%d, %d, %d, %d>", shape[i][0], conv_packed_variant(), shape[i][1], shape[i][2]);
        return pbuf[i];
    }
    if (kind == PK_KERNEL_CONV_B2B) return "conv_b2b_f32_kernel<2>";
    if (kind == PK_KERNEL_CONV_B2B + 1) return "conv_b2b_f32_kernel<4>";
    if (kind >= PK_KERNEL_CONV_BASE && kind < PK_KERNEL_CONV_BASE + 72) {
        // names as rocprofv3 prints the template instantiation: <BM, BN, BK, WM, WN, MODE, NBUF, BF16>
        // (kind = base + MODE*6 + tile*2 + (BK==32) + 18 if NBUF == 1 + 36 if BF16, conv.hip)
        static thread_local char buf[96];
        int k = kind - PK_KERNEL_CONV_BASE;
        const int bf16 = k >= 36 ? 1 : 0;
        k -= 36 * bf16;
        const int nbuf = k >= 18 ? 1 : 2;
        k -= (nbuf == 1) ? 18 : 0;
        const int mode = k / 6, t = (k % 6) / 2, bk = (k & 1) ? 32 : 16;
        const int bm = (t == 2) ? 64 : 128, bn = (t == 0) ? 128 : 64, wm = (t == 2) ? 32 : 64, wn = (t == 0) ? 64 : 32;
        snprintf(buf, sizeof(buf), "conv_igemm_f32_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", bm, bn, bk, wm, wn, mode, nbuf, bf16);
        return buf;
    }
    return "?";
}

int stabnet_net_create(void** out, int N, int H, int W, int in_ch, int n_theta, int keep_activations) {
    SN_REQUIRE(out != nullptr, "net_create: null out");
    SN_REQUIRE(N > 0 && N <= 4096 && H >= 32 && W >= 32 && in_ch > 0 && in_ch <= 64 && n_theta > 0,
               "net_create: bad shape N=%d H=%d W=%d in_ch=%d n_theta=%d (H,W >= 32)", N, H, W, in_ch, n_theta);
    *out = build_net(N, H, W, in_ch, n_theta, keep_activations);
    return STABNET_OK;
}

void stabnet_net_destroy(void* net) { delete static_cast<Net*>(net); }

/* SECONDARY fast mode of the inference forward, off by default: the conv operands are rounded to
 * bf16 when the fragments are read (fp32 tensors in memory, fp32 accumulate, v_mfma_f32_32x32x16_bf16).  The reference is
 * fp32 end to end, so this mode has its own, looser parity bar (tests/test_bf16_mode_gpu.py) and is never the headline. */
int stabnet_net_set_bf16_operands(void* netp, int on) {
    Net* net = static_cast<Net*>(netp);
    SN_REQUIRE(net != nullptr, "set_bf16_operands: null net");
    if (net->keep_all) {
        // training plans: mode 4 = the dgrad launches read a pre-split image of the re-packed weights (written once per step); the
        // reduced-precision and read-time split modes do not exist for training.  Call it before stabnet_net_train_workspace_bytes().
        SN_REQUIRE(on == 0 || on == 4, "set_bf16_operands: inference plans only (training stays fp32; mode 4 = packed split dgrad)");
        net->train_split = on == 4 ? 1 : 0;
        return STABNET_OK;
    }
    net->bf16_operands = (on >= 1 && on <= 4) ? on : 0;     // 1: bf16 operands; 2 / 3: split operands (conv_kernel.h sn_split3)
    return STABNET_OK;
}

int stabnet_net_num_params(const void* net) { return net ? (int)static_cast<const Net*>(net)->params.size() : -1; }

/* kind: 0 conv weight (dims Cout,KH,KW,CinPadded; aux = real Cin), 1 conv bias, 2 gamma, 3 beta, 4 moving_mean,
 * 5 moving_variance, 6 FC weight (dims out,in), 7 FC bias. */
int stabnet_net_param_info(const void* netp, int idx, char* name, int name_cap, long* offset, int* kind, int* dims4,
                           int* aux) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && idx >= 0 && idx < (int)net->params.size(), "param_info: bad index %d", idx);
    const ParamEntry& e = net->params[idx];
    if (name && name_cap > 0) {
        std::strncpy(name, e.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (offset) *offset = (long)e.off;
    if (kind) *kind = e.kind;
    if (dims4) for (int i = 0; i < 4; ++i) dims4[i] = e.dims[i];
    if (aux) *aux = e.aux;
    return STABNET_OK;
}

size_t stabnet_net_param_floats(const void* net) { return net ? static_cast<const Net*>(net)->n_floats : 0; }
size_t stabnet_net_trainable_floats(const void* net) { return net ? static_cast<const Net*>(net)->n_trainable : 0; }
size_t stabnet_net_bn_channels(const void* net) { return net ? static_cast<const Net*>(net)->G : 0; }
size_t stabnet_net_workspace_bytes(const void* netp) {
    const Net* net = static_cast<const Net*>(netp);
    return net ? net->act_floats * sizeof(float) + net->splitk_bytes + 256 : 0;
}
double stabnet_net_flops(const void* net) { return net ? static_cast<const Net*>(net)->flops : 0.0; }
int stabnet_net_num_launches(const void* netp) {
    const Net* net = static_cast<const Net*>(netp);
    if (!net) return -1;
    int n = 0;
    // (shortened head: GAP partials + fc_1 = 2 launches, fc_2, fc_3, output layer [+ mesh] = 3; else 2 + 4 x fc_launches)
    for (const Step& s : net->steps) {
        if (s.kind == S_CONV_B2B && net->bf16_operands) n += 2 + conv_reduce_launches(s.conv) + conv_reduce_launches(s.conv_b);   // runs as two launches there
        else n += (s.kind == S_CONV) ? 1 + conv_reduce_launches(s.conv, (net->bf16_operands == 4 && s.wimg_off >= 0) ? 4 : 0)
                                     : (s.kind == S_FC ? fc_launches(s.M) : (s.kind == S_GAP ? 2 : 1));
    }
    if (head_fused_supported(net->N, net->t_last.C, net->fc_dims)) n -= 1 /* gap_finalize */ + (fc_launches(net->N) - 1) * 2;
    return n;
}

int stabnet_deploy_frame_launches(const void* netp, int grid_h, int grid_w) {
    const Net* net = static_cast<const Net*>(netp);
    if (!net) return -1;
    (void)grid_h; (void)grid_w;
    const bool fused = head_fused_supported(net->N, net->t_last.C, net->fc_dims) != 0;     // the mesh rides with the output layer
    return stabnet_net_num_launches(netp) - 1 /* the stack assembly replaces the pad step */ + 1 + (fused ? 0 : 1) + 1;
}

/* Debug taps (valid after a forward only when the net was created with keep_activations=1):
 * names conv1, pool1, block{1-4}/unit_{k}, global_pool -> float offset into the workspace and NHWC dims. */
int stabnet_net_activation_info(const void* netp, const char* name, long* offset, int* dims4) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && name && offset && dims4, "activation_info: null pointer");
    for (const Net::Tap& t : net->taps)
        if (t.name == name) {
            *offset = t.off;
            dims4[0] = t.N; dims4[1] = t.H; dims4[2] = t.W; dims4[3] = t.C;
            return STABNET_OK;
        }
    stabnet_set_error("activation_info: no tap named %s", name);
    return STABNET_ERR_BAD_ARG;
}

/* Moving-average BN -> folded (scale, shift): fold = [G scales][G shifts].  Run once after loading weights. */
int stabnet_net_fold_bn(const void* netp, const float* params, float* fold, float eps, void* stream) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && params && fold, "fold_bn: null pointer");
    int rc = launch_bn_fold(params + net->off_gamma, params + net->off_beta, params + net->off_mean,
                            params + net->off_var, eps, (int)net->G, fold, fold + net->G, (hipStream_t)stream);
    if (rc) return rc;
    for (const Net::MergeInfo& m : net->merges) {          // per-channel epilogue vectors of the merged shortcut | conv1 launches
        rc = launch_merge_vectors(params + m.b_sc, fold + m.bn1, fold + net->G + m.bn1, m.depth, m.dbn,
                                  fold + 2 * net->G + net->stem_w_floats + m.off, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (net->stem_rowrun) {
        // stem weights OHWI [64][7][7][in_ch_pad] -> row-run layout [64][7][roundup(7*in_ch, 32)] (zeros in the run padding)
        rc = launch_stem_repack(params + net->w_stem, fold + 2 * net->G, 64, 7, 7, net->in_ch_pad, net->in_ch, (hipStream_t)stream);
        if (rc) return rc;
    }
    // pre-split weight images (inference plans ONLY: a training plan's Step::wimg_off points into its own workspace region, and its
    // `fold` has no image region): from the weights each launch reads
    if (net->keep_all) return STABNET_OK;
    float* img = fold + 2 * net->G + net->stem_w_floats + net->merge_floats;
    for (const Step& s : net->steps) {
        if (s.kind != S_CONV || s.wimg_off < 0) continue;
        SN_REQUIRE((size_t)s.wimg_off + conv_weight_image_floats(s.conv.Cout, s.conv.K) <= net->wimg_floats, "fold_bn: weight image outside the fold buffer");
        const float* w = s.conv.rowrun ? fold + 2 * net->G : params + s.w_off;
        rc = launch_weight_split_image(w, s.conv.Cout, s.conv.K, img + s.wimg_off, (hipStream_t)stream);
        if (rc) return rc;
    }
    return STABNET_OK;
}

/* Floats of the `fold` buffer: [G scales][G shifts][re-laid-out stem weights of an inference plan]. */
size_t stabnet_net_fold_floats(const void* netp) {
    const Net* net = static_cast<const Net*>(netp);
    return net ? 2 * net->G + net->stem_w_floats + net->merge_floats + net->wimg_floats : 0;
}

/* get_resnet(x_tensor, is_training=False): x_tensor NHWC [N,H,W,in_ch] -> theta [N,n_theta]. */
int stabnet_backbone_fwd_infer(const void* netp, const float* params, const float* fold, const float* x_tensor,
                               float* theta, void* workspace, size_t workspace_bytes, void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && params && fold && x_tensor && theta && workspace, "backbone_fwd_infer: null pointer");
    if (workspace_bytes < stabnet_net_workspace_bytes(netp)) {
        stabnet_set_error("backbone_fwd_infer: workspace %zu B < %zu B needed", workspace_bytes,
                          stabnet_net_workspace_bytes(netp));
        return STABNET_ERR_WORKSPACE;
    }
    int rc = sn_check_device(params, "backbone_fwd_infer: params", (hipStream_t)stream);
    if (rc == 0) rc = sn_check_device(x_tensor, "backbone_fwd_infer: x_tensor", (hipStream_t)stream);
    if (rc) return rc;
    return run_forward(net, params, fold, x_tensor, theta, static_cast<float*>(workspace), (hipStream_t)stream,
                       static_cast<Prof*>(prof));
}

/* History ring initialisation: `depth` copies of the first frame, zero masks (deploy_bundle.py:216-224). */
int stabnet_ring_init(float* frames_ring, float* masks_ring, const float* first_frame, int S, int depth, int H, int W,
                      void* stream) {
    SN_REQUIRE(frames_ring && masks_ring && first_frame && S > 0 && S <= 65535 && depth > 0 && H > 0 && W > 0,
               "ring_init: bad arguments");
    return launch_ring_init(frames_ring, masks_ring, first_frame, S, depth, (long)H * W, (hipStream_t)stream);
}

/* One iteration of the online loop for S = net.N independent streams (deploy_bundle.py:259-296,319-332):
 * stack assembly from the ring -> regressor -> get_4_pts + transformer -> frame = img - black -> push.
 * `head` (DEVICE int[2]) = {ring slot this frame's push writes, reserved}; the call advances head[0] to
 * (head+1) % depth on the device (one thread of the mesh kernel for refine = 1, a one-thread kernel otherwise), so the
 * whole frame has fixed arguments and can be captured once into a hipGraph and replayed.
 * `lags` is a HOST array (read at enqueue time).  all_black (optional, int32 [S][H*W]) += round(black) once per refine
 * pass (deploy_bundle.py:291 sits inside the refine loop). */
int stabnet_deploy_frame(const void* netp, const float* params, const float* fold, float* frames_ring,
                         float* masks_ring, int depth, int* head, const int* lags, int n_lags, const float* cur_frame,
                         int refine, int grid_h, int grid_w, float do_crop_rate, float* theta, float* out_img,
                         float* black, float* x_map, float* y_map, float* Hs, float* frame_fb, int* all_black,
                         void* workspace, size_t workspace_bytes, void* stream, void* profp) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && params && fold && frames_ring && masks_ring && lags && cur_frame && theta && out_img && black &&
                   x_map && y_map && Hs && frame_fb && workspace, "deploy_frame: null pointer");
    SN_REQUIRE(depth > 0 && head != nullptr && refine >= 1, "deploy_frame: bad ring arguments");
    SN_REQUIRE(2 * n_lags + 1 == net->in_ch && n_lags <= 7, "deploy_frame: %d lags do not make %d channels", n_lags,
               net->in_ch);
    SN_REQUIRE((grid_h + 1) * (grid_w + 1) * 2 == net->n_theta, "deploy_frame: grid does not match n_theta");
    if (workspace_bytes < stabnet_net_workspace_bytes(netp)) {
        stabnet_set_error("deploy_frame: workspace %zu B < %zu B needed", workspace_bytes, stabnet_net_workspace_bytes(netp));
        return STABNET_ERR_WORKSPACE;
    }
    int rc = check_warp_args(net->N, net->H, net->W, 1, grid_h, grid_w);
    if (rc) return rc;
    if ((rc = sn_check_device(params, "deploy_frame: params", (hipStream_t)stream)) != 0) return rc;
    if ((rc = sn_check_device(workspace, "deploy_frame: workspace", (hipStream_t)stream)) != 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    Prof* prof = static_cast<Prof*>(profp);
    float* ws = static_cast<float*>(workspace);
    RingLags rl{};
    rl.n = n_lags;
    for (int i = 0; i < n_lags; ++i) {
        SN_REQUIRE(lags[i] > 0 && lags[i] <= depth, "deploy_frame: lag %d outside the ring", lags[i]);
        rl.lag[i] = lags[i];
    }
    const long hw = (long)net->H * net->W;
    const float* cur = cur_frame;
    float* x16 = ws + net->steps[0].out_off;
    const bool fused_push = (refine == 1);
    for (int j = 0; j < refine; ++j) {
        bool rec = prof && prof->begin(st);
        if (net->stem_rowrun)
            rc = launch_stack_assemble_bordered(frames_ring, masks_ring, cur, net->N, depth, head, rl, net->H, net->W, 3, x16, st);
        else
            rc = launch_stack_assemble(frames_ring, masks_ring, cur, net->N, depth, head, rl, hw, net->in_ch_act, x16, st);
        if (rec) prof->end(st, PK_KERNEL_ASSEMBLE, 0, 4.0 * net->N * hw * (net->in_ch + net->in_ch_act));
        if (rc) return rc;
        // fused push: whoever computes the mesh also advances the ring head (nothing between the stack assembly and the
        // sampler reads it): the fused head's last phase, or the mesh kernel
        MeshTail mt{grid_h, grid_w, 1.0f / do_crop_rate, Hs, fused_push ? head : nullptr, depth, false};
        mt.prefetch_src = cur;                               // warmed into the sampler's L2s while the head's latency chain runs
        rc = run_forward(net, params, fold, nullptr, theta, ws, st, prof, true, &mt);
        if (rc) return rc;
        if (!mt.done) {
            rec = prof && prof->begin(st);
            rc = launch_mesh(theta, 1, net->N, grid_h, grid_w, 1.0f / do_crop_rate, nullptr, Hs, st, nullptr, mt.head_adv, depth);
            if (rec) prof->end(st, PK_KERNEL_MESH, 0, 4.0 * net->N * (net->n_theta + grid_h * grid_w * 9));
            if (rc) return rc;
        }
        // the frame being warped is the (possibly refined) current frame: channel 2*n_lags of the stack
        if (fused_push) {
            // sampler + frame = img - black + push into the ring + all_black: ONE launch (slot = advanced head - 1)
            WarpPush wp{frames_ring, masks_ring, frame_fb, all_black, head, depth};
            rec = prof && prof->begin(st);
            rc = launch_sample_push(Hs, cur, net->N, net->H, net->W, grid_h, grid_w, out_img, black, x_map, y_map, wp, st);
            if (rec) prof->end(st, PK_KERNEL_WARP, 0, net->N * (32.0 * hw + 776.0) + (all_black ? 8.0 * net->N * hw : 0.0));
            return rc;
        }
        rec = prof && prof->begin(st);
        rc = launch_sample(Hs, cur, net->N, net->H, net->W, 1, grid_h, grid_w, out_img, black, x_map, y_map, st);
        if (rec) prof->end(st, PK_KERNEL_WARP, 0, net->N * (20.0 * hw + 776.0));
        if (rc) return rc;
        if (all_black != nullptr && (rc = stabnet_black_accumulate(black, all_black, (long)net->N * hw, stream)) != 0) return rc;
        // frame = img - black; with refine > 1 it replaces the current frame of the next pass (:293-295)
        const bool last = (j == refine - 1);
        rec = prof && prof->begin(st);
        if (last)
            rc = launch_ring_push(frames_ring, masks_ring, net->N, depth, head, out_img, black, hw, frame_fb, st);
        else
            rc = launch_ring_push(frame_fb, nullptr, net->N, 1, nullptr, out_img, black, hw, nullptr, st);
        if (rec) prof->end(st, PK_KERNEL_PUSH, 0, 4.0 * net->N * hw * 4);
        if (rc) return rc;
        if (!last) cur = frame_fb;
    }
    return launch_ring_advance(head, depth, st);
}

}  // extern "C"

// =========================================================================================================
// Training: one tower's forward with batch-statistics BN and the explicit backward schedule
// (what `opt.minimize(total_loss)` differentiates for get_resnet(is_training=True), s_net_bundle_nobm.py:301,
// train_bundle_nobm.py:155-160).  The plan must have been created with keep_activations = 1.
// =========================================================================================================
struct TrainLayout {
    size_t bn_scale, bn_shift, bn_mean, bn_invstd, GA, GB, T1, T2, T3, fcg0, fcg1, partial, fcpart, coef, wt, wt_img, fw_img, argmax, splitk, slabs, total;
    size_t fcx[4];                  // FC head inputs of the PAIR ([2N, dims[k]], tower 1's rows behind tower 0's), in tower 0's workspace
    size_t fcpart_floats;
    size_t splitk_bytes, slab_floats;
    size_t stem_w, stem_dw;         // row-run stem: the weights re-laid-out for this step, and their gradient in that layout
};

static size_t rnd64(size_t n) { return (n + 63) & ~(size_t)63; }

static ConvArgs dgrad_args(int N, int H, int W, int Cin, int Cout, int KH, int stride, int pad) {
    ConvArgs a{};
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KH) / stride + 1;
    a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin; a.KH = KH; a.KW = KH; a.stride = 1; a.pad = KH - 1 - pad;
    a.up = stride; a.Ho = H; a.Wo = W; a.res_H = H; a.res_W = W; a.res_stride = 1;
    return a;
}

// slab floats the wgrad launches of backward stages [s0, s1) need for T towers (stage order: see run_backward_stage)
static const int kNumStagesFwd = 4;
static size_t unit_slab_floats(const UnitInfo& u, int T) {
    size_t n = wgrad_slab_floats(u.depth, u.dbn, u.out.N * u.out.H * u.out.W, T, 1) + wgrad_slab_floats(u.dbn, 9 * u.dbn, u.r2.N * u.r2.H * u.r2.W, T) +
               wgrad_slab_floats(u.dbn, u.cin, u.x.N * u.x.H * u.x.W, T);
    if (u.proj) n += wgrad_slab_floats(u.depth, u.cin, u.x.N * u.x.H * u.x.W, T);
    return n;
}
static size_t net_slab_floats(const Net* net, int T, int s0, int s1) {
    const int first[5] = {0, 3, 7, 13, 16};               // units of block 1..4 (resnet_v2_50: 3, 4, 6, 3)
    size_t n = 0;
    for (int sg = s0; sg < s1; ++sg) {
        const int blk = 3 - sg;
        for (int ui = first[blk + 1] - 1; ui >= first[blk] && ui < (int)net->units.size(); --ui) n += unit_slab_floats(net->units[ui], T);
        if (sg == kNumStagesFwd - 1)
            n += wgrad_slab_floats(64, net->stem_rowrun ? (int)(net->stem_w_floats / 64) : 49 * net->in_ch_pad,
                                   net->t_c1.N * net->t_c1.H * net->t_c1.W, T);
    }
    return n;
}

static TrainLayout train_layout(const Net* net) {
    TrainLayout L{};
    size_t o = rnd64(net->act_floats);
    auto take = [&](size_t n) { const size_t r = o; o += rnd64(n); return r; };
    L.bn_scale = take(net->G); L.bn_shift = take(net->G); L.bn_mean = take(net->G); L.bn_invstd = take(net->G);
    // gradient scratch of the backward: sized for BOTH towers -- in the lockstep backward tower 1's copy of a tensor of `sz` floats
    // sits right behind tower 0's (buf + sz), so the pair is one [2N,...] tensor and every dgrad is ONE launch over 2N samples
    L.GA = take(2 * net->max_net); L.GB = take(2 * net->max_net); L.T1 = take(2 * net->max_r); L.T2 = take(2 * net->max_r);
    L.T3 = take(2 * net->max_net);
    L.fcg0 = take((size_t)2 * net->N * 2048); L.fcg1 = take((size_t)2 * net->N * 2048);
    for (int k = 0; k < 4; ++k) L.fcx[k] = take((size_t)2 * net->N * net->fc_dims[k]);
    size_t red = 0;
    for (const BnInfo& b : net->bns) red = std::max(red, col_reduce_workspace_floats(b.M, b.C, 2));       // (both towers' partials)
    for (const UnitInfo& u : net->units)
        red = std::max(red, col_reduce_workspace_floats((long)u.out.N * u.out.H * u.out.W, u.depth, 2));
    red = std::max(red, col_reduce_workspace_floats((long)net->t_c1.N * net->t_c1.H * net->t_c1.W, 64, 2));
    L.partial = take(red);
    L.fcpart_floats = (size_t)64 * 2 * net->N * 2048;        // fc_bwd_x partials: (<= 64 splits) x 2N x K (train_layers.hip picks ~256 workgroups: 8..16 splits here)
    L.fcpart = take(L.fcpart_floats);
    L.coef = take(3 * std::max<size_t>(net->max_c, 2048));
    L.wt = take((size_t)net->packs.prefix[net->packs.n]);
    L.wt_img = take(net->train_split ? net->dimg_floats : 0);
    L.fw_img = take(net->train_split ? net->fimg_floats : 0);
    L.argmax = take((net->t_pool.size + 3) / 4);              // one byte per pooled element
    size_t sk = net->splitk_bytes;
    for (const UnitInfo& u : net->units) {
        for (int T = 1; T <= 2; ++T) {                        // one tower, or both towers as one [2N,...] tensor
            ConvArgs a3 = dgrad_args(T * u.r2.N, u.r2.H, u.r2.W, u.dbn, u.depth, 1, 1, 0);
            ConvArgs a2 = dgrad_args(T * u.r1.N, u.r1.H, u.r1.W, u.dbn, u.dbn, 3, u.stride, 1);
            ConvArgs a1 = dgrad_args(T * u.x.N, u.x.H, u.x.W, u.cin, u.dbn, 1, 1, 0);
            ConvArgs as = dgrad_args(T * u.x.N, u.x.H, u.x.W, u.cin, u.depth, 1, 1, 0);
            sk = std::max({sk, conv_plan(a3), conv_plan(a2), conv_plan(a1), u.proj ? conv_plan(as) : (size_t)0});
        }
    }
    L.splitk_bytes = sk;
    L.splitk = take(sk / sizeof(float) + 64);
    // wgrad partial slabs of every conv of one backward (consumed stage by stage by wgrad_reduce_flush); sized for the lockstep
    // backward of both towers (their slabs of a layer are contiguous, in tower 0's workspace)
    L.slab_floats = net_slab_floats(net, 2, 0, kNumStagesFwd);
    L.slabs = take(L.slab_floats + 64);
    L.stem_w = take(net->stem_w_floats); L.stem_dw = take(net->stem_w_floats);
    L.total = o;
    return L;
}

static const BnInfo* find_bn(const Net* net, long chan_off) {
    for (const BnInfo& b : net->bns)
        if (b.chan_off == chan_off) return &b;
    return nullptr;
}

// T = 1: one tower.  T = 2: the two siamese towers (train_bundle_nobm.py:107-108, same weights) layer by layer in LOCKSTEP:
// every convolution / pool is launched for tower 0 and then for tower 1 (the weights are L2 / Infinity-Cache hot for the
// second launch), and every batch-statistics reduction covers both towers in ONE launch (launch_*_g: these kernels are
// launch-latency sized).  Scratch that lives for one launch only (reduction partials, split-K slabs, the re-packed dgrad
// weights) is taken from tower 0's workspace; everything a tower keeps for its backward is in its own.
static int run_forward_train(const Net* net, float* params, int T, const float* const* x, float* const* theta, float* const* ws,
                             float eps, float decay, hipStream_t st, Prof* prof) {
    const TrainLayout L = train_layout(net);
    float* splitk = ws[0] + L.splitk;
    float* partial = ws[0] + L.partial;
    static const bool pair_convs = getenv("STABNET_TRAIN_PAIR_FWD") == nullptr || atoi(getenv("STABNET_TRAIN_PAIR_FWD")) != 0;   // debug switch
    if (net->train_split) {                                // images of this step's forward weights: one launch, read by both towers
        const int rc0 = launch_weight_split_images(params, ws[0] + L.fw_img, net->fimg, st);
        if (rc0) return rc0;
    }
    std::vector<char> have(net->bns.size(), 0);
    auto need_bn = [&](long chan_off) -> int {
        const BnInfo* b = find_bn(net, chan_off);
        SN_REQUIRE(b != nullptr, "forward_train: unknown BN at channel offset %ld", chan_off);
        const size_t idx = (size_t)(b - net->bns.data());
        if (have[idx]) return STABNET_OK;
        have[idx] = 1;
        const float* xs[2] = {nullptr, nullptr};
        float *sc[2], *sh[2], *mu[2], *is[2];
        for (int t = 0; t < T; ++t) {
            xs[t] = ws[t] + b->tensor_off;
            sc[t] = ws[t] + L.bn_scale + chan_off; sh[t] = ws[t] + L.bn_shift + chan_off;
            mu[t] = ws[t] + L.bn_mean + chan_off; is[t] = ws[t] + L.bn_invstd + chan_off;
        }
        return launch_bn_stats_g(T, xs, b->M, b->C, params + net->off_gamma + chan_off, params + net->off_beta + chan_off, eps,
                                 decay, sc, sh, mu, is, params + net->off_mean + chan_off, params + net->off_var + chan_off,
                                 partial, st);
    };
    for (const Step& s : net->steps) {
        int rc = STABNET_OK;
        if ((s.kind == S_CONV || s.kind == S_GAP) && s.bn_off >= 0 && (rc = need_bn(s.bn_off)) != 0) return rc;
        for (int t = 0; t < T && rc == STABNET_OK; ++t) {
            float* w = ws[t];
            const float* scale = w + L.bn_scale;
            const float* shift = w + L.bn_shift;
            switch (s.kind) {
                case S_PAD:
                    if (net->stem_rowrun) {
                        rc = launch_embed_border(x[t], s.N, s.H, s.W, s.C, 3, w + s.out_off, st);
                        if (rc == STABNET_OK && t == 0)      // this step's stem weights in the row-run layout (both towers read them)
                            rc = launch_stem_repack(params + net->w_stem, ws[0] + L.stem_w, 64, 7, 7, net->in_ch_pad, net->in_ch, st);
                    } else {
                        rc = launch_pad_channels(x[t], w + s.out_off, (long)s.N * s.H * s.W, s.C, net->in_ch_pad, st);
                    }
                    break;
                case S_CONV: {
                    const bool pair = T == 2 && s.pair_ok && pair_convs;
                    if (pair && t == 1) break;                         // launched with tower 0
                    ConvArgs a = pair ? s.conv_pair : s.conv;
                    a.x = w + s.in_off; a.y = w + s.out_off; a.w = a.rowrun ? ws[0] + L.stem_w : params + s.w_off;
                    a.bias = s.b_off >= 0 ? params + s.b_off : nullptr;
                    a.residual = s.res_off >= 0 ? w + s.res_off : nullptr;
                    a.in_scale = s.bn_off >= 0 ? scale + s.bn_off : nullptr;
                    a.in_shift = s.bn_off >= 0 ? shift + s.bn_off : nullptr;
                    a.partial = splitk;
                    // split mode: the prologue-carrying 1x1 layers read the image of this step's weights (both towers, pair or not)
                    const float* wimg = (net->train_split && s.wimg_off >= 0 && a.KH == 1 && a.in_scale != nullptr) ? ws[0] + L.fw_img + s.wimg_off : nullptr;
                    if (pair) {
                        // tower 1's tensors sit at the same offsets of ITS workspace: one element offset for everything, minus the
                        // N images (or tower rows) the pair's row index has already advanced by
                        const long delta = (long)(ws[1] - ws[0]);
                        const ConvArgs& c = s.conv;
                        ConvPair pr;
                        pr.m_tower = c.M;
                        pr.x_tower_floats = c.N * c.H * c.W * c.Cin;
                        pr.dx = delta - (long)c.N * c.H * c.W * c.Cin;
                        pr.dy = delta - (long)c.M * c.Cout;
                        pr.dres = delta - (long)c.N * c.res_H * c.res_W * c.res_ld;
                        pr.dscale = delta;
                        rc = conv_launch_pair(a, pr, st, prof, wimg);
                    } else {
                        rc = conv_launch(a, st, prof, wimg != nullptr ? 4 : 0, wimg);
                    }
                    break;
                }
                case S_POOL:
                    rc = launch_max_pool_argmax(w + s.in_off, w + s.out_off, reinterpret_cast<unsigned char*>(w + L.argmax), s.N,
                                                s.H, s.W, s.C, s.Ho, s.Wo, s.k, s.stride, s.pt, s.pl, st);
                    break;
                case S_GAP:         // the FC head works on the pair: tower t's pooled rows go behind tower 0's
                    rc = launch_gap_bn_relu(w + s.in_off, scale + s.bn_off, shift + s.bn_off, s.N, s.H * s.W, s.C,
                                            ws[0] + L.fcx[0] + (size_t)t * s.N * s.C, splitk, st);
                    break;
                case S_FC: {
                    int k = 0;
                    while (k < 3 && net->fc_w[k] != s.w_off) ++k;
                    if (s.out_off == EXT_OUT)       // output layer: theta of each tower is the caller's own buffer
                        rc = launch_fc(ws[0] + L.fcx[3] + (size_t)t * s.M * s.K, params + s.w_off, params + s.b_off, theta[t], s.M, s.K,
                                       s.Nout, s.relu, st);
                    else if (t == 0)                // fc_1..3: ONE launch over the [T*N, K] pair (the weights are streamed once)
                        rc = launch_fc(ws[0] + L.fcx[k], params + s.w_off, params + s.b_off, ws[0] + L.fcx[k + 1], T * s.M, s.K,
                                       s.Nout, s.relu, st);
                    break;
                }
            }
        }
        if (rc) return rc;
    }
    return STABNET_OK;
}

// Backward in STAGES so that a data-parallel host can hand each parameter bucket to the collective while the earlier
// layers are still in backward (reverse layer order).  stage 0 = FC head + postnorm + block4, 1 = block3,
// 2 = block2, 3 = block1 + stem.  A stage leaves d(unit input) in GA/GB (which one: parity of the units processed) and
// ends with the ordered reduction of its wgrad slabs, so its gradients are final when the stage's kernels are done.
static const int kNumStages = 4;
static void stage_units(const Net* net, int stage, int& u_hi, int& u_lo) {      // units [u_lo, u_hi] processed descending
    // blocks of resnet_v2_50: 3, 4, 6, 3 units
    const int first[5] = {0, 3, 7, 13, 16};
    const int blk = 3 - stage;
    u_lo = first[blk]; u_hi = first[blk + 1] - 1;
    (void)net;
}

static int run_backward_stage(const Net* net, const float* params, int T, const float* const* d_theta, float* grads,
                              float* const* ws, int stage, hipStream_t st, Prof* prof) {
    const TrainLayout L = train_layout(net);
    float* partial = ws[0] + L.partial;                  // (one launch at a time: shared scratch from tower 0's workspace)
    float* wt = ws[0] + L.wt;                            // dgrad weights re-packed ONCE per step for both towers
    float* splitk = ws[0] + L.splitk;
    const int N = net->N;
    int rc;
    SN_REQUIRE(net->units.size() == 16, "tower_bwd: unexpected unit count %zu", net->units.size());
    WgradReduceTable table{};
    // slab cursor at the start of this stage = slab floats of the stages before it (layers are visited in a fixed order)
    size_t cursor = net_slab_floats(net, T, 0, stage);
    // per-tower views.  The gradient scratch (GA, GB, T1, T2, T3) lives in tower 0's workspace for both towers: tower t's copy
    // of a tensor of `sz` floats is at buf + t * sz, i.e. the pair is ONE [T*N, ...] tensor and a dgrad is one launch over it.
    struct TW { float* ws; const float *scale, *shift, *bmean, *binv; float *coef, *slabs; } tw[2];
    int u_hi, u_lo;
    stage_units(net, stage, u_hi, u_lo);
    for (int t = 0; t < T; ++t) {
        float* w = ws[t];
        tw[t] = {w, w + L.bn_scale, w + L.bn_shift, w + L.bn_mean, w + L.bn_invstd, w + L.coef, w + L.slabs};
    }
    float *bGA = ws[0] + L.GA, *bGB = ws[0] + L.GB, *bT1 = ws[0] + L.T1, *bT2 = ws[0] + L.T2, *bT3 = ws[0] + L.T3;
    if ((15 - u_hi) & 1) std::swap(bGA, bGB);                          // one swap per unit already processed
    auto V = [](float* buf, size_t sz) { return [buf, sz](int t) -> float* { return buf + (size_t)t * sz; }; };
    auto CV = [](const float* buf, size_t sz) { return [buf, sz](int t) -> const float* { return buf + (size_t)t * sz; }; };
    // ONE wgrad launch per layer for both towers (grid.z = tower x split), slabs [tower][split] in tower 0's workspace
    auto wgrad = [&](auto xin, auto dyin, long w_off, long bn, int H, int W, int Cin, int Cout, int K, int stride, int pad,
                     int rowrun = 0, long b_off = -1, long b_off2 = -1) -> int {
        const float *xs[2], *dys[2], *sc[2], *sh[2];
        for (int t = 0; t < T; ++t) {
            xs[t] = xin(t); dys[t] = dyin(t);
            sc[t] = bn >= 0 ? tw[t].scale + bn : nullptr; sh[t] = bn >= 0 ? tw[t].shift + bn : nullptr;
        }
        return wgrad_launch_g(T, xs, dys, grads, w_off, bn >= 0 ? sc : nullptr, bn >= 0 ? sh : nullptr, N, H, W, Cin, Cout, K, K, stride,
                              pad, tw[0].slabs, &cursor, L.slab_floats, &table, st, prof, rowrun, b_off, b_off2);
    };
    // BN + ReLU backward of both towers: one reduction launch, one finalize, one apply
    auto bn_bwd = [&](long bn, const TensorRef& xt, auto gin, auto addin, bool has_add, int add_stride, auto dxout) -> int {
        const float *xs[2], *gs[2], *sc[2], *sh[2], *mu[2], *is[2], *ad[2];
        float *dx[2], *cf[2];
        for (int t = 0; t < T; ++t) {
            xs[t] = tw[t].ws + xt.off; gs[t] = gin(t); sc[t] = tw[t].scale + bn; sh[t] = tw[t].shift + bn;
            mu[t] = tw[t].bmean + bn; is[t] = tw[t].binv + bn; ad[t] = has_add ? addin(t) : nullptr; dx[t] = dxout(t); cf[t] = tw[t].coef;
        }
        return launch_bn_relu_bwd_g(T, xs, gs, sc, sh, mu, is, params + net->off_gamma + bn, (long)xt.N * xt.H * xt.W, xt.C,
                                    has_add ? ad : nullptr, add_stride, xt.H, xt.W, grads + net->off_gamma + bn,
                                    grads + net->off_beta + bn, dx, partial, cf, st);
    };
    auto bias_grad = [&](auto gin, long M, int C, long b_off, long b_off2 = -1) -> int {
        const float* gs[2];
        for (int t = 0; t < T; ++t) gs[t] = gin(t);
        return launch_bias_grad_g(T, gs, M, C, grads + b_off, partial, st, b_off2 >= 0 ? grads + b_off2 : nullptr);
    };
    // dgrad of both towers: ONE launch over the [T*N, ...] pair (dy, dx, residual are pair bases)
    auto dgrad = [&](const float* dy, long pack_off, float* dx, const float* res, int H, int W, int Cin, int Cout, int K, int stride,
                     int pad) -> int {
        const float* img = nullptr;
        if (net->train_split)
            for (int i = 0; i < net->packs.n; ++i)
                if (net->packs.prefix[i] == pack_off && net->dimg_of_pack[i] >= 0) img = ws[0] + L.wt_img + net->dimg_of_pack[i];
        return dgrad_launch(dy, wt + pack_off, dx, res, T * N, H, W, Cin, Cout, K, K, stride, pad, splitk, L.splitk_bytes, st, prof, img);
    };
    auto none = [&](int) -> const float* { return nullptr; };

    if (stage == 0) {
        if ((rc = pack_dgrad_weights_all(params, wt, net->packs, st)) != 0) return rc;
        if (net->train_split && (rc = launch_weight_split_images(wt, ws[0] + L.wt_img, net->dimg, st)) != 0) return rc;
        // ---- FC head (resnet.py:44-56, s_net_bundle_nobm.py:256-259) on the pair: the output layer per tower (d_theta are the
        // caller's buffers; dW += in tower order), fc_3..1 and the reduce_mean backward as ONE launch over [T*N, ...]
        const TensorRef& last = net->t_last;
        float* w0 = tw[0].ws;
        float* fg[2] = {w0 + L.fcg0, w0 + L.fcg1};
        for (int t = 0; t < T; ++t) {
            rc = launch_fc_bwd(w0 + L.fcx[3] + (size_t)t * N * net->fc_dims[3], params + net->fc_w[3], nullptr, d_theta[t], N,
                               net->fc_dims[3], net->fc_dims[4], 0, grads + net->fc_w[3], grads + net->fc_b[3],
                               fg[1] + (size_t)t * N * net->fc_dims[3], w0 + L.fcpart, L.fcpart_floats, st);
            if (rc) return rc;
        }
        const float* dy = fg[1];
        for (int k = 2; k >= 0; --k) {
            float* dx = fg[k & 1];
            rc = launch_fc_bwd(w0 + L.fcx[k], params + net->fc_w[k], w0 + L.fcx[k + 1], dy, T * N, net->fc_dims[k], net->fc_dims[k + 1], 1,
                               grads + net->fc_w[k], grads + net->fc_b[k], dx, w0 + L.fcpart, L.fcpart_floats, st);
            if (rc) return rc;
            dy = dx;
        }
        // ---- reduce_mean backward
        if ((rc = launch_gap_bwd(dy, T * N, last.H * last.W, last.C, bT3, st)) != 0) return rc;
        // ---- postnorm BN + ReLU
        if ((rc = bn_bwd(net->bn_post, last, CV(bT3, last.size), none, false, 1, V(bGA, last.size))) != 0) return rc;
    }
    // ---- bottleneck units of this stage, last to first.  G = d(unit output) = GA
    for (int ui = u_hi; ui >= u_lo; --ui) {
        const UnitInfo& u = net->units[ui];
        const long Mo = (long)u.out.N * u.out.H * u.out.W;
        const size_t so = u.out.size, sr2 = u.r2.size, sr1 = u.r1.size, sx = u.x.size;
        auto X = [&](long off) { return [&, off](int t) -> const float* { return tw[t].ws + off; }; };
        // conv3 (1x1, bias) : input relu(bn2(r2))
        // (the projection shortcut's bias receives the same column sums of G: one reduction for both)
        // -- and where conv3's wgrad runs on the stride-1 kernel (every unit of the regressor) they come out of ITS pass over G
        const bool bias_in_wgrad = wgrad_bias_fusable(N, u.r2.H, u.r2.W, u.dbn, u.depth, 1, 1, 1, 0) != 0;
        if (!bias_in_wgrad && (rc = bias_grad(CV(bGA, so), Mo, u.depth, u.b3, u.proj ? u.b_sc : -1)) != 0) return rc;
        if ((rc = wgrad(X(u.r2.off), CV(bGA, so), u.w3, u.bn2, u.r2.H, u.r2.W, u.dbn, u.depth, 1, 1, 0, 0, bias_in_wgrad ? u.b3 : -1,
                        bias_in_wgrad && u.proj ? u.b_sc : -1)) != 0) return rc;
        if ((rc = dgrad(bGA, net->pack_w3[ui], bT1, nullptr, u.r2.H, u.r2.W, u.dbn, u.depth, 1, 1, 0)) != 0) return rc;
        if ((rc = bn_bwd(u.bn2, u.r2, CV(bT1, sr2), none, false, 1, V(bT1, sr2))) != 0) return rc;             // T1 = d r2
        // conv2 (3x3, stride) : input relu(bn1(r1))
        if ((rc = wgrad(X(u.r1.off), CV(bT1, sr2), u.w2, u.bn1, u.r1.H, u.r1.W, u.dbn, u.dbn, 3, u.stride, 1)) != 0) return rc;
        if ((rc = dgrad(bT1, net->pack_w2[ui], bT2, nullptr, u.r1.H, u.r1.W, u.dbn, u.dbn, 3, u.stride, 1)) != 0) return rc;
        if ((rc = bn_bwd(u.bn1, u.r1, CV(bT2, sr1), none, false, 1, V(bT2, sr1))) != 0) return rc;             // T2 = d r1
        // conv1 (1x1) : input relu(bn_pre(x))
        if ((rc = wgrad(X(u.x.off), CV(bT2, sr1), u.w1, u.bn_pre, u.x.H, u.x.W, u.cin, u.dbn, 1, 1, 0)) != 0) return rc;
        if ((rc = dgrad(bT2, net->pack_w1[ui], bT3, nullptr, u.x.H, u.x.W, u.cin, u.dbn, 1, 1, 0)) != 0) return rc;
        if (u.proj) {   // projection shortcut conv1x1(preact) + bias: d preact += dgrad(G)
            if ((rc = wgrad(X(u.x.off), CV(bGA, so), u.w_sc, u.bn_pre, u.x.H, u.x.W, u.cin, u.depth, 1, 1, 0)) != 0) return rc;
            if ((rc = dgrad(bGA, net->pack_sc[ui], bT3, bT3, u.x.H, u.x.W, u.cin, u.depth, 1, 1, 0)) != 0) return rc;
            if ((rc = bn_bwd(u.bn_pre, u.x, CV(bT3, sx), none, false, 1, V(bGB, sx))) != 0) return rc;
        } else {        // identity shortcut (subsample by the unit's stride): d x += upsample(G)
            if ((rc = bn_bwd(u.bn_pre, u.x, CV(bT3, sx), CV(bGA, so), true, u.stride, V(bGB, sx))) != 0) return rc;
        }
        std::swap(bGA, bGB);
    }
    if (stage == kNumStages - 1) {
        // ---- stem: max-pool backward, conv1 weight/bias gradient (the input needs no gradient)
        const TensorRef& c1 = net->t_c1;
        const TensorRef& pl = net->t_pool;
        for (int t = 0; t < T; ++t)
            if ((rc = launch_max_pool_bwd(reinterpret_cast<const unsigned char*>(tw[t].ws + L.argmax), bGA + (size_t)t * pl.size,
                                          bGB + (size_t)t * c1.size, N, c1.H, c1.W, c1.C, pl.H, pl.W, 3, 2, net->pool_pt, net->pool_pl,
                                          st)) != 0) return rc;
        auto Xin = [&](int t) -> const float* { return tw[t].ws + net->t_xin.off; };
        if ((rc = bias_grad(CV(bGB, c1.size), (long)N * c1.H * c1.W, 64, net->b_stem)) != 0) return rc;
        if (net->stem_rowrun) {
            // the gradient comes out in the forward's row-run layout [64][7][roundup(7 * in_ch, 32)] (a scratch buffer the slab
            // reduction accumulates into: addressed relative to `grads`, as every entry of the table is), then moves to OHWI
            float* dw_rr = ws[0] + L.stem_dw;
            if (hipMemsetAsync(dw_rr, 0, net->stem_w_floats * sizeof(float), st) != hipSuccess) return STABNET_ERR_LAUNCH;
            if ((rc = wgrad(Xin, CV(bGB, c1.size), sn_float_distance(grads, dw_rr), -1, net->H, net->W, net->in_ch, 64, 7, 2, 3, 1)) != 0) return rc;
            if ((rc = wgrad_reduce_flush(grads, table, st)) != 0) return rc;
            return launch_wgrad_rowrun_scatter(dw_rr, grads + net->w_stem, 64, 7, 7, net->in_ch, net->in_ch_pad, st);
        }
        if ((rc = wgrad(Xin, CV(bGB, c1.size), net->w_stem, -1, net->H, net->W, net->in_ch_pad, 64, 7, 2, 3)) != 0) return rc;
    }
    return wgrad_reduce_flush(grads, table, st);
}

extern "C" {

/* Bytes of one tower's training workspace (activations kept for backward + batch BN buffers + gradient scratch). */
size_t stabnet_net_train_workspace_bytes(const void* netp) {
    const Net* net = static_cast<const Net*>(netp);
    if (!net || !net->keep_all) return 0;
    return train_layout(net).total * sizeof(float) + 256;
}

/* get_resnet(x_tensor, is_training=True): batch-statistics BN; moving averages inside `params` are updated
 * (slim UPDATE_OPS, s_net_bundle_nobm.py:355-356).  Activations stay in `workspace` for stabnet_tower_bwd. */
int stabnet_tower_fwd_train(const void* netp, float* params, const float* x_tensor, float* theta, void* workspace,
                            size_t workspace_bytes, float bn_eps, float bn_decay, void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && params && x_tensor && theta && workspace, "tower_fwd_train: null pointer");
    SN_REQUIRE(net->keep_all, "tower_fwd_train: the plan must be created with keep_activations = 1");
    SN_REQUIRE(workspace_bytes >= stabnet_net_train_workspace_bytes(netp), "tower_fwd_train: workspace too small");
    if (int rc = sn_check_device(params, "tower_fwd_train: params", (hipStream_t)stream)) return rc;
    if (int rc = sn_check_device(x_tensor, "tower_fwd_train: x_tensor", (hipStream_t)stream)) return rc;
    float* ws[1] = {static_cast<float*>(workspace)};
    net->stamp(workspace, 1, nullptr);
    return run_forward_train(net, params, 1, &x_tensor, &theta, ws, bn_eps, bn_decay, (hipStream_t)stream, static_cast<Prof*>(prof));
}

/* Both siamese towers of a training step (train_bundle_nobm.py:107-108: two towers over the SAME weights) in lockstep, layer
 * by layer: per layer the convolution of tower 1 and of tower 2 (weights cache-hot for the second), and ONE batch-statistics
 * reduction launch covering both.  Results are those of stabnet_tower_fwd_train(x1) followed by stabnet_tower_fwd_train(x2)
 * (the moving averages receive tower 1's update, then tower 2's).  One workspace per tower, as for the single-tower call. */
int stabnet_towers_fwd_train(const void* netp, float* params, const float* x1, const float* x2, float* theta1, float* theta2,
                             void* workspace1, void* workspace2, size_t workspace_bytes, float bn_eps, float bn_decay,
                             void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && params && x1 && x2 && theta1 && theta2 && workspace1 && workspace2 && workspace1 != workspace2,
               "towers_fwd_train: null pointer (or one workspace for both towers)");
    SN_REQUIRE(net->keep_all, "towers_fwd_train: the plan must be created with keep_activations = 1");
    SN_REQUIRE(workspace_bytes >= stabnet_net_train_workspace_bytes(netp), "towers_fwd_train: workspace too small");
    if (int rc = sn_check_device(params, "towers_fwd_train: params", (hipStream_t)stream)) return rc;
    const float* xs[2] = {x1, x2};
    float* th[2] = {theta1, theta2};
    float* ws[2] = {static_cast<float*>(workspace1), static_cast<float*>(workspace2)};
    net->stamp(workspace1, 2, workspace2);
    net->stamp(workspace2, 3, workspace1);
    return run_forward_train(net, params, 2, xs, th, ws, bn_eps, bn_decay, (hipStream_t)stream, static_cast<Prof*>(prof));
}

/* Backward of the tower given d_theta [N,n_theta]; gradients are ACCUMULATED into `grads` (same layout as the
 * trainable prefix of `params`; zero it once per step -- both siamese towers add into it, in stream order). */
static int tower_bwd_checks(const Net* net, const float* params, const float* d_theta, float* grads, void* workspace,
                            size_t workspace_bytes, const void* netp, void* stream) {
    SN_REQUIRE(net && params && d_theta && grads && workspace, "tower_bwd: null pointer");
    SN_REQUIRE(net->keep_all, "tower_bwd: the plan must be created with keep_activations = 1");
    SN_REQUIRE(workspace_bytes >= stabnet_net_train_workspace_bytes(netp), "tower_bwd: workspace too small");
    return sn_check_device(grads, "tower_bwd: grads", (hipStream_t)stream);
}
// The backward must match the forward that filled the workspace (see Net::fwd_stamp).
static int tower_bwd_pairing(const Net* net, const void* ws1, const void* ws2) {
    const Net::FwdStamp a = net->stamp_of(ws1);
    if (ws2 == nullptr) {
        SN_REQUIRE(a.kind == 1, "tower_bwd: this workspace was last filled by %s; the single-tower backward needs "
                   "stabnet_tower_fwd_train (the lockstep forward keeps both towers' FC activations in the pair's first workspace: "
                   "use stabnet_towers_bwd_stage)", a.kind == 0 ? "no training forward of this plan" : "stabnet_towers_fwd_train");
        return STABNET_OK;
    }
    const Net::FwdStamp b = net->stamp_of(ws2);
    SN_REQUIRE(a.kind == 2 && a.partner == ws2 && b.kind == 3 && b.partner == ws1,
               "towers_bwd_stage: the two workspaces were not filled together by stabnet_towers_fwd_train(workspace1, workspace2) "
               "(after two stabnet_tower_fwd_train calls use stabnet_tower_bwd per tower)");
    return STABNET_OK;
}
int stabnet_tower_bwd(const void* netp, const float* params, const float* d_theta, float* grads, void* workspace,
                      size_t workspace_bytes, void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    int rc = tower_bwd_checks(net, params, d_theta, grads, workspace, workspace_bytes, netp, stream);
    if (rc == 0) rc = tower_bwd_pairing(net, workspace, nullptr);
    float* ws[1] = {static_cast<float*>(workspace)};
    for (int stage = 0; rc == 0 && stage < kNumStages; ++stage)
        rc = run_backward_stage(net, params, 1, &d_theta, grads, ws, stage, (hipStream_t)stream, static_cast<Prof*>(prof));
    return rc;
}
/* One stage of the same backward (0: FC head + block4, 1: block3, 2: block2, 3: block1 + stem; call them in this order).
 * When a stage's kernels are done, the gradients of stabnet_net_grad_bucket(stage) are final for this tower. */
int stabnet_tower_bwd_stage(const void* netp, const float* params, const float* d_theta, float* grads, void* workspace,
                            size_t workspace_bytes, int stage, void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    int rc = tower_bwd_checks(net, params, d_theta, grads, workspace, workspace_bytes, netp, stream);
    if (rc == 0) rc = tower_bwd_pairing(net, workspace, nullptr);
    if (rc) return rc;
    SN_REQUIRE(stage >= 0 && stage < kNumStages, "tower_bwd_stage: stage %d outside [0, %d)", stage, kNumStages);
    float* ws[1] = {static_cast<float*>(workspace)};
    return run_backward_stage(net, params, 1, &d_theta, grads, ws, stage, (hipStream_t)stream, static_cast<Prof*>(prof));
}
/* One backward stage of BOTH towers in lockstep (see stabnet_towers_fwd_train): per layer ONE wgrad and ONE dgrad launch for the pair,
 * one BN-backward reduction / finalize / apply and one bias-gradient reduction for both, the dgrad weights re-packed once,
 * one slab reduction per stage.  After stage k the bucket stabnet_net_grad_bucket(k) holds the sum of both towers. */
int stabnet_towers_bwd_stage(const void* netp, const float* params, const float* d_theta1, const float* d_theta2, float* grads,
                             void* workspace1, void* workspace2, size_t workspace_bytes, int stage, void* stream, void* prof) {
    const Net* net = static_cast<const Net*>(netp);
    int rc = tower_bwd_checks(net, params, d_theta1, grads, workspace1, workspace_bytes, netp, stream);
    if (rc) return rc;
    SN_REQUIRE(d_theta2 && workspace2 && workspace2 != workspace1, "towers_bwd_stage: null pointer (or one workspace for both towers)");
    SN_REQUIRE(stage >= 0 && stage < kNumStages, "towers_bwd_stage: stage %d outside [0, %d)", stage, kNumStages);
    if ((rc = tower_bwd_pairing(net, workspace1, workspace2)) != 0) return rc;
    const float* dt[2] = {d_theta1, d_theta2};
    float* ws[2] = {static_cast<float*>(workspace1), static_cast<float*>(workspace2)};
    return run_backward_stage(net, params, 2, dt, grads, ws, stage, (hipStream_t)stream, static_cast<Prof*>(prof));
}
int stabnet_net_num_grad_stages(void) { return kNumStages; }
/* Float range [lo, hi) of the gradient buffer that backward stage `stage` completes (weights and biases in network order;
 * the BN gamma / beta sections [stabnet_net_bn_grad_range] are touched by every stage and are final after the last one). */
int stabnet_net_grad_bucket(const void* netp, int stage, long* lo, long* hi) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && lo && hi && stage >= 0 && stage < kNumStages && net->units.size() == 16, "grad_bucket: bad arguments");
    int u_hi, u_lo;
    stage_units(net, stage, u_hi, u_lo);
    const UnitInfo& first = net->units[u_lo];
    *lo = (stage == kNumStages - 1) ? 0 : (first.proj ? std::min(first.w_sc, first.w1) : first.w1);
    if (stage == 0) *hi = (long)net->off_gamma;
    else {
        int ph, pl;
        stage_units(net, stage - 1, ph, pl);
        const UnitInfo& nx = net->units[pl];
        *hi = nx.proj ? std::min(nx.w_sc, nx.w1) : nx.w1;
    }
    return STABNET_OK;
}
int stabnet_net_bn_grad_range(const void* netp, long* lo, long* hi) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && lo && hi, "bn_grad_range: null pointer");
    *lo = (long)net->off_gamma; *hi = (long)net->n_trainable;
    return STABNET_OK;
}

/* Batch-statistics view of a tower's last forward: folded (scale, shift) [2][G] copied out for inspection/tests. */
int stabnet_net_train_bn_offsets(const void* netp, long* scale_off, long* shift_off, long* mean_off, long* invstd_off) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && scale_off && shift_off && mean_off && invstd_off, "train_bn_offsets: null pointer");
    const TrainLayout L = train_layout(net);
    *scale_off = (long)L.bn_scale; *shift_off = (long)L.bn_shift; *mean_off = (long)L.bn_mean; *invstd_off = (long)L.bn_invstd;
    return STABNET_OK;
}

/* Where the training forward left what its discrete decisions can be read from (tests force them onto the float64 autograd
 * oracle): `what` = "bn:<channel offset>" -> the tensor that BN normalises (float offset, element count M*C);
 * "fcx<k>", k = 0..3 -> input of FC layer k for the PAIR, [2N][dims[k]] in tower 0's workspace; "argmax" -> the max-pool argmax
 * BYTES (float offset of their start, byte count); "pool" -> the pooled tensor. */
int stabnet_net_train_debug_offset(const void* netp, const char* what, long* off, long* count) {
    const Net* net = static_cast<const Net*>(netp);
    SN_REQUIRE(net && what && off && count && net->keep_all, "train_debug_offset: bad arguments (needs a keep_activations plan)");
    const TrainLayout L = train_layout(net);
    const std::string w(what);
    if (w.rfind("bn:", 0) == 0) {
        const BnInfo* b = find_bn(net, atol(w.c_str() + 3));
        SN_REQUIRE(b != nullptr, "train_debug_offset: no BN at channel offset %s", w.c_str() + 3);
        *off = b->tensor_off; *count = b->M * b->C;
        return STABNET_OK;
    }
    if (w.rfind("fcx", 0) == 0 && w.size() == 4 && w[3] >= '0' && w[3] <= '3') {
        const int k = w[3] - '0';
        *off = (long)L.fcx[k]; *count = 2L * net->N * net->fc_dims[k];
        return STABNET_OK;
    }
    if (w == "argmax") { *off = (long)L.argmax; *count = (long)net->t_pool.size; return STABNET_OK; }
    if (w == "pool") { *off = net->t_pool.off; *count = (long)net->t_pool.size; return STABNET_OK; }
    stabnet_set_error("train_debug_offset: unknown item %s", what);
    return STABNET_ERR_BAD_ARG;
}

/* slim L2 regularisers (s_net_bundle_nobm.py:324-325; resnet.py:35-37): value = sum_seg coef*0.5*sum w^2 added to
 * *loss_out (zero it first; may be NULL); grads[seg] += gscale*coef*w (grads may be NULL).  seg_* are DEVICE arrays. */
int stabnet_weight_decay(const float* params, float* grads, const long* seg_off, const long* seg_len,
                         const float* seg_coef, int nseg, float gscale, float* loss_out, float* workspace, void* stream) {
    SN_REQUIRE(params && seg_off && seg_len && seg_coef && nseg > 0 && nseg <= 65535, "weight_decay: bad arguments");
    SN_REQUIRE(loss_out == nullptr || workspace != nullptr, "weight_decay: the loss value needs a workspace of 64*nseg floats");
    return launch_weight_decay(params, grads, seg_off, seg_len, seg_coef, nseg, gscale, loss_out, workspace, (hipStream_t)stream);
}

/* tf.train.AdamOptimizer.apply_gradients (train_bundle_nobm.py:159-160), TF 1.x ApplyAdam op for op in float32:
 *   alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)   (the powers are float32 running products, as TF's
 *   beta1_power / beta2_power variables are: initialised to beta, multiplied by beta after every step)
 *   m += (g - m)(1 - b1); v += (g*g - v)(1 - b2); w -= (m * alpha) / (sqrt(v) + eps).   `step` is 1-based.
 * g is scaled by gscale first (1/world for data-parallel averaging); g = grads + grads2 when grads2 is given. */
int stabnet_adam_step(float* params, const float* grads, const float* grads2, float* m, float* v, long n, float lr,
                      float beta1, float beta2, float eps, int step, float gscale, void* stream) {
    SN_REQUIRE(params && grads && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    // beta^step as TF holds it: a float32 product built one multiplication per step (cached so that consecutive steps cost one)
    static thread_local struct { float b1, b2, p1, p2; int step; } c = {0.f, 0.f, 0.f, 0.f, 0};
    if (c.step == 0 || c.b1 != beta1 || c.b2 != beta2 || step < c.step) { c.b1 = beta1; c.b2 = beta2; c.p1 = beta1; c.p2 = beta2; c.step = 1; }
    for (; c.step < step; ++c.step) { c.p1 = c.p1 * beta1; c.p2 = c.p2 * beta2; }
    const float alpha = lr * sqrtf(1.0f - c.p2) / (1.0f - c.p1);
    return launch_adam(params, grads, grads2, m, v, n, alpha, beta1, beta2, eps, gscale, (hipStream_t)stream);
}

}  // extern "C"
