"""Feasibility probe (round 4): what does a stem-sized convolution on a SECOND stream cost the 720p frame chain?
If the frame with a co-running 180 us conv is slower by less than what taking the stem's history part off the chain saves
(~115 us), pipelining the stem across frames pays.  Prints ms/frame for: frame alone, frame + side conv (graph fork/join),
eager two-stream with priorities."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stabnet_amd import synthetic, ops
from stabnet_amd.config import Config
from stabnet_amd.deploy import StabNetStream

H, W = 720, 1280
dev = torch.device("cuda:0")
cfg = Config(height=H, width=W)
P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
clip = torch.from_numpy(synthetic.make_clip(H, W, 8, seed=1234)).to(dev)
st = StabNetStream(P, H, W, cfg, streams=1, device=dev, use_graph=False)
st.start(clip[0:1])
st.cur.copy_(clip[1:2])

# side load: 3x3 conv, M = 230400, N = 64, K = 576 on the ring kernel (the stem is M = 230400, N = 64, K = 672)
xs = torch.randn(1, 360, 640, 64, device=dev) * 0.1
wsd = torch.randn(64, 3, 3, 64, device=dev) * 0.05
ys = ops.conv2d(xs, wsd, pad=1)
torch.cuda.synchronize()
from stabnet_amd import _lib
from stabnet_amd._tensor import ptr, stream_ptr
L = _lib.lib()
ws_bytes = L.stabnet_conv2d_workspace_bytes(1, 360, 640, 64, 64, 3, 3, 1, 1)
wsb = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)


def side_conv():
    _lib.call("stabnet_conv2d_fwd_ex", ptr(xs), ptr(wsd), 0, 0, 0, 0, 0, 0, 1, 0, 0, ptr(ys), 1, 360, 640, 64, 64, 3, 3, 1, 1,
              0, ptr(wsb), ws_bytes, stream_ptr(dev), device=dev)


def timeit(fn, n=300, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def capture(body):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g


res = {}
g0 = capture(lambda: st._enqueue())
res["frame graph"] = timeit(g0.replay)
gs = capture(side_conv)
res["side conv alone (graph)"] = timeit(gs.replay)


def both_seq():
    st._enqueue(); side_conv()
g1 = capture(both_seq)
res["frame + conv, same stream"] = timeit(g1.replay)

for prio in (0, -1):
    side = torch.cuda.Stream(device=dev, priority=0)
    def fork_join(at_start=True):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            side_conv()
        st._enqueue()
        cur.wait_stream(side)
    g2 = capture(fork_join)
    res["frame || conv (graph fork/join)"] = timeit(g2.replay)
    break

# eager, two streams with priorities
for pm, ps in ((0, 0), (-1, 0)):
    main = torch.cuda.Stream(device=dev, priority=pm)
    side = torch.cuda.Stream(device=dev, priority=ps)
    def eager_two():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            side_conv()
        with torch.cuda.stream(main):
            st._enqueue()
            main.wait_stream(side)
    res["eager frame || conv, prio main %d side %d" % (pm, ps)] = timeit(eager_two)
    def eager_one():
        with torch.cuda.stream(main):
            st._enqueue()
    res["eager frame alone, prio %d" % pm] = timeit(eager_one)

# two side convs half the size each? (M halves): smaller-lived workgroups -> skip
for k, v in res.items():
    print("%-50s %.4f ms" % (k, v))
