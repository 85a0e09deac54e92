#!/usr/bin/env python3
"""Online stabilisation driver on MI355X -- drop-in for the reference's deploy_bundle.py call surface.

Same flags as the reference CLI (deploy_bundle.py:12-31).  What runs on the GPU is the part the reference times
(deploy_bundle.py:285-287): 13-channel stack -> regressor -> multi-grid warp, plus the history ring and the feedback,
which the reference does in NumPy on the host.  Differences forced by the offline image, stated rather than hidden:
  * --model-dir/--model-name take either a TF checkpoint prefix (`model-80000.index` + `.data-*`, read by
    stabnet_amd/tf_checkpoint.py without TensorFlow; the `.meta` graph is not needed) or a `.npz` written by
    train_bundle_nobm.py (TF variable names); without one, seeded synthetic weights are used and said so.
  * OpenCV is absent: clips are `.npy` arrays ([T,H,W] grey in [0,255] or [T,H,W,3] BGR) under
    <prefix>/unstable/<name>; results are written as `.npy` (stabilised grey frames, x/y maps, black masks).  Video
    decode / MJPG encode (cv2.VideoCapture / VideoWriter in the reference) are outside the path and not implemented.
  * --before-ch is parsed and ignored exactly as in the reference (deploy_bundle.py:15,41): the ring depth is
    max(indices[1:]) = 32 and six frames are sampled at lags 1,2,4,8,16,32.
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--model-dir')
    p.add_argument('--model-name')
    p.add_argument('--before-ch', type=int)
    p.add_argument('--output-dir', default='data_video_local')
    p.add_argument('--infer-with-stable', action='store_true')
    p.add_argument('--infer-with-last', action='store_true')
    p.add_argument('--test-list', nargs='+', default=['data_video/test_list', 'data_video/train_list_deploy'])
    p.add_argument('--prefix', default='data_video')
    p.add_argument('--max-span', type=int, default=1)
    p.add_argument('--random-black', type=int, default=None)
    p.add_argument('--start-with-stable', action='store_true')
    p.add_argument('--refine', type=int, default=1)
    p.add_argument('--no_bm', type=int, default=1)
    p.add_argument('--gpu_memory_fraction', type=float, default=0.1)
    p.add_argument('--deploy-vis', action='store_true')
    # extensions of this build
    p.add_argument('--height', type=int, default=None, help='network/warp height (reference: fixed 288)')
    p.add_argument('--width', type=int, default=None, help='network/warp width (reference: fixed 512)')
    p.add_argument('--synthetic', type=int, default=0, help='stabilise a synthetic shaky clip of this many frames')
    p.add_argument('--device', default='cuda:0')
    p.add_argument('--pipeline', action='store_true',
                   help='overlap upload / frame / download of neighbouring frames on three HIP streams (stabnet_amd.deploy.ClipPipeline); '
                        'same output bytes, fps is then the host-to-host rate of the whole loop')
    p.add_argument('--operand-mode', type=int, default=4, choices=[0, 1, 2, 3, 4],
                   help='conv operand mode of the regressor (include/stabnet_hip.h): 4 = packed split kernels -- float32 operands as exact '
                        'sums of three bf16 terms on the bf16 matrix pipe, float32 accumulation, float32-level results (default); '
                        '0 = exact f32 MFMA; 1 = bf16 operands (reduced precision)')
    return p


def grey_train(frame, H, W):
    """config.cvt_img2train (config.py:6-21) without cv2/PIL: BGR->grey (OpenCV weights), nearest-size bilinear
    resize when needed, scale to [-0.5, 0.5]."""
    f = np.asarray(frame, np.float32)
    if f.ndim == 3:
        f = 0.114 * f[..., 0] + 0.587 * f[..., 1] + 0.299 * f[..., 2]
    if f.shape != (H, W):
        ys = (np.arange(H) + 0.5) * f.shape[0] / H - 0.5
        xs = (np.arange(W) + 0.5) * f.shape[1] / W - 0.5
        y0 = np.clip(np.floor(ys).astype(int), 0, f.shape[0] - 1); y1 = np.clip(y0 + 1, 0, f.shape[0] - 1)
        x0 = np.clip(np.floor(xs).astype(int), 0, f.shape[1] - 1); x1 = np.clip(x0 + 1, 0, f.shape[1] - 1)
        wy = np.clip(ys - y0, 0, 1)[:, None]; wx = np.clip(xs - x0, 0, 1)[None, :]
        f = (f[y0][:, x0] * (1 - wy) * (1 - wx) + f[y0][:, x1] * (1 - wy) * wx
             + f[y1][:, x0] * wy * (1 - wx) + f[y1][:, x1] * wy * wx)
    return (f * (1.0 / 255) - 0.5).astype(np.float32)


def load_weights(args, cfg):
    from stabnet_amd import synthetic
    if args.model_dir and args.model_name:
        path = os.path.join(args.model_dir, args.model_name)
        for cand in (path, path + '.npz'):
            if os.path.exists(cand) and cand.endswith('.npz'):
                z = np.load(cand)
                print('restored weights from', cand)
                return {k: z[k] for k in z.files if not k.startswith('__')}
        if os.path.exists(path + '.index'):
            # the reference's own checkpoint format (new_saver.restore, deploy_bundle.py:45-47): read without TensorFlow
            from stabnet_amd import tf_checkpoint
            params, _ = tf_checkpoint.load_stabnet_variables(path)
            print('restored weights from TF checkpoint', path)
            return params
        print('WARNING: neither %s.npz nor %s.index found' % (path, path))
    print('using seeded synthetic weights (no trained model is available offline)')
    return synthetic.make_params(cfg, seed=0, theta_scale=0.2)


def is_colour(frame, H, W):
    f = np.asarray(frame)
    return f.ndim == 3 and f.shape[:2] == (H, W)


def run_serial(stream, clip, H, W, dev, frames_out, colour_out, xmaps, ymaps, blacks):
    """The loop as the reference writes it (deploy_bundle.py:244-342): one frame at a time, the host waiting for each step;
    fps = frames / time inside the step, as the reference prints it (:285-289)."""
    import torch
    from stabnet_amd import warp
    tot_time, length = 0.0, 0
    first = grey_train(clip[0], H, W)
    stream.start(torch.from_numpy(first[None]).to(dev))                       # ring = 32 x first frame, zero masks
    for t in range(1, len(clip)):
        cur = torch.from_numpy(grey_train(clip[t], H, W)[None]).to(dev)
        torch.cuda.synchronize()
        start = time.time()
        r = stream.step(cur)                                                  # one sess.run-equivalent
        torch.cuda.synchronize()
        tot_time += time.time() - start
        if is_colour(clip[t], H, W):
            # warpRevBundle2 (deploy_bundle.py:136-146,303) on the device: colour frame remapped by the smoothed maps
            bgr = torch.from_numpy(np.ascontiguousarray(clip[t], dtype=np.uint8)).to(dev)
            colour_out.append(warp.warpRevBundle2(bgr, r['x_map'], r['y_map']).cpu().numpy())
        net_output = ((r['output'][0, :, :, 0].cpu().numpy() + 0.5) * 255).clip(0, 255).astype(np.uint8)
        frames_out.append(net_output)
        xmaps.append(r['x_map'][0, :, :, 0].cpu().numpy()); ymaps.append(r['y_map'][0, :, :, 0].cpu().numpy())
        blacks.append(r['black_pix'][0].cpu().numpy().astype(np.uint8))
        length += 1
        if length % 10 == 0:
            print('length: ' + str(length))
            print('fps={}'.format(length / tot_time))
    return length, tot_time


def run_pipelined(stream, clip, H, W, frames_out, colour_out, xmaps, ymaps, blacks):
    """--pipeline: the same frames through stabnet_amd.deploy.ClipPipeline (upload / frame / download of neighbouring frames on
    three HIP streams).  Same output bytes; fps = frames / wall time of the whole loop, host conversion and copies included."""
    from stabnet_amd.deploy import ClipPipeline
    colour = is_colour(clip[0], H, W)

    class Grey:                                                               # frames converted as the pipeline asks for them
        def __len__(self):
            return len(clip)

        def __getitem__(self, t):
            return grey_train(clip[t], H, W)

    def sink(r):                                                              # views of pinned staging memory: copy out
        frames_out.append(r['output'].copy())
        if colour:
            colour_out.append(r['bgr'].copy())
        xmaps.append(r['x_map'].copy()); ymaps.append(r['y_map'].copy()); blacks.append(r['black'].copy())
        if len(frames_out) % 10 == 0:
            print('length: ' + str(len(frames_out)))

    start = time.time()
    ClipPipeline(stream, colour=colour).run(Grey(), clip if colour else None, sink=sink, maps=True)
    tot_time = time.time() - start
    if frames_out:
        print('fps={}'.format(len(frames_out) / tot_time))
    return len(frames_out), tot_time


def main():
    args = build_parser().parse_args()
    import torch
    from stabnet_amd import synthetic, warp
    from stabnet_amd.config import Config
    from stabnet_amd.deploy import StabNetStream

    base = Config()
    H, W = args.height or base.height, args.width or base.width
    cfg = Config(height=H, width=W)
    lags = cfg.indices[1:]
    print('inference with {}'.format(list(lags)))
    if args.before_ch is not None and args.before_ch != max(lags):
        print('note: --before-ch %d is ignored (as in the reference); ring depth = %d' % (args.before_ch, max(lags)))
    ignored = [flag for flag, on in (('--max-span', args.max_span != 1), ('--random-black', args.random_black is not None),
                                     ('--no_bm=0', args.no_bm == 0), ('--infer-with-last', args.infer_with_last),
                                     ('--infer-with-stable', args.infer_with_stable),
                                     ('--start-with-stable', args.start_with_stable), ('--deploy-vis', args.deploy_vis)) if on]
    if ignored:
        # --infer-with-stable / --start-with-stable / --deploy-vis read the ground-truth STABLE video (deploy_bundle.py:73,
        # 88-89,237-246,319-320): debugging aids that need the paired clip; with --infer-with-stable the reference also stops
        # appending to before_masks while still popping it (:319-328), i.e. it only runs for 32 frames.
        print('note: %s: debugging paths of the reference that are not on the timed path; accepted and ignored'
              % '/'.join(ignored))
    params = load_weights(args, cfg)
    dev = torch.device(args.device)
    torch.cuda.set_device(dev)
    stream = StabNetStream(params, H, W, cfg, streams=1, device=dev, refine=args.refine, before_ch=args.before_ch,
                           use_graph=True, operand_mode=args.operand_mode)   # one frame = fixed-argument launches: captured once, replayed per frame
    stream.track_black()            # all_black += round(black) inside every refine pass (deploy_bundle.py:234,291), on the device

    clips = []
    if args.synthetic > 0:
        clips.append(('synthetic', (synthetic.make_clip(H, W, args.synthetic, seed=1234) + 0.5) * 255.0))
    else:
        for lst in args.test_list:
            if not os.path.exists(lst):
                continue
            for name in open(lst).read().splitlines():
                if not name:
                    continue
                path = os.path.join(args.prefix, 'unstable', name)
                if os.path.exists(path) and path.endswith('.npy'):
                    clips.append((name, np.load(path, mmap_mode='r')))
                elif os.path.exists(path + '.npy'):
                    clips.append((name, np.load(path + '.npy', mmap_mode='r')))
                else:
                    print('skipping %s: only .npy clips can be read without OpenCV' % path)
    out_dir = os.path.join(args.output_dir, 'output')
    os.makedirs(out_dir, exist_ok=True)

    for name, clip in clips:
        print(name)
        tot_time, length = 0.0, 0
        frames_out, xmaps, ymaps, blacks, colour_out = [], [], [], [], []
        try:
            if args.pipeline:
                length, tot_time = run_pipelined(stream, clip, H, W, frames_out, colour_out, xmaps, ymaps, blacks)
            else:
                length, tot_time = run_serial(stream, clip, H, W, dev, frames_out, colour_out, xmaps, ymaps, blacks)
        except Exception:
            traceback.print_exc()                    # the reference swallows per-video errors and still finalises
        finally:
            print('total length={}'.format(length + 2))
            if frames_out:
                stem = os.path.join(out_dir, os.path.splitext(os.path.basename(name))[0])
                np.save(stem + '_stable.npy', np.stack(frames_out))
                if colour_out:
                    np.save(stem + '_stable_bgr.npy', np.stack(colour_out))
                np.savez_compressed(stem + '_maps.npz', x_map=np.stack(xmaps), y_map=np.stack(ymaps), black=np.stack(blacks))
                print('wrote', stem + '_stable.npy')
                # max-inscribed black-free rectangle over the whole clip (deploy_bundle.py:344-371), searched on the device
                ans, area = warp.max_inscribed_rect(stream.all_black[0])
                if ans:
                    src = np.stack(colour_out) if colour_out else np.stack(frames_out)
                    np.save(stem + '_cut.npy', src[:, ans[0]:ans[2] + 1, ans[1]:ans[3] + 1])
                    print('crop', ans, 'area', area)


if __name__ == '__main__':
    main()
