#!/usr/bin/env python3
"""Siamese StabNet training driver on MI355X -- drop-in for the reference's train_bundle_nobm.py call surface.

Flags of the reference (train_bundle_nobm.py:34-37) are kept; the loop mirrors train_bundle_nobm.py:216-348:
loss-schedule gates from the step index, display every disp_freq, checkpoint every save_freq, 10 held-out batches
every test_freq, Adam with the staircase learning rate.  Differences forced by the offline image:
  * no TFRecord dataset / ImageNet resnet_v2_50.ckpt exists here: batches come from the seeded synthetic generator
    (SURVEY.md 8d); the ImageNet checkpoint is read (stabnet_amd/tf_checkpoint.py, no TensorFlow needed) when present,
    otherwise weights start from the seeded initialiser; checkpoints are `.npz` of TF-named variables.
  * data parallel (new): launch with torchrun, one process per GPU; each rank generates its own shard of the global
    batch, BN statistics stay local, gradients are summed over RCCL.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--gpu_memory_fraction', type=float, default=0.95)
    p.add_argument('--restore', action='store_true')
    # extensions of this build
    p.add_argument('--iters', type=int, default=None, help='stop after this many steps (reference: training_iter)')
    p.add_argument('--batch-size', type=int, default=None, help='pairs per GPU (reference: 10)')
    p.add_argument('--height', type=int, default=None)
    p.add_argument('--width', type=int, default=None)
    p.add_argument('--model-dir', default=None)
    p.add_argument('--disp-freq', type=int, default=None)
    p.add_argument('--imagenet-ckpt', default='data_video/resnet_v2_50.ckpt',
                   help='TF checkpoint (V1 single file as slim ships it, or V2) that initialises the backbone except conv1 and '
                        'the fc head (reference: train_bundle_nobm.py:184-191,208).  Like the reference, training does not start '
                        'without it -- unless --no-imagenet-init is given')
    p.add_argument('--no-imagenet-init', action='store_true',
                   help='start from the seeded initialiser instead of the ImageNet backbone (synthetic runs, benchmarks)')
    p.add_argument('--augment', action='store_true',
                   help='assemble every batch on the device from un-augmented pair material with the reference\'s random '
                        'crop / flip / contrast / brightness / homography masks (get_data_mini_after.py)')
    return p


def latest_checkpoint(model_dir):
    if not os.path.isdir(model_dir):
        return None
    c = [f for f in os.listdir(model_dir) if f.startswith('model-') and f.endswith('.npz')]
    return os.path.join(model_dir, max(c, key=lambda f: int(f[6:-4]))) if c else None


def main():
    args = build_parser().parse_args()
    import torch
    from stabnet_amd import parallel, synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.train import Trainer, learning_rate, loss_gates

    base = Config()
    cfg = Config(height=args.height or base.height, width=args.width or base.width,
                 batch_size=args.batch_size or base.batch_size)
    if args.disp_freq:
        cfg.disp_freq = args.disp_freq
    model_dir = args.model_dir or cfg.model_dir
    rank, local_rank, world = parallel.env_world()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    pg = parallel.init_process_group(device=dev)
    N, H, W = cfg.batch_size, cfg.height, cfg.width
    init = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    from stabnet_amd import tf_checkpoint
    # train_bundle_nobm.py:204-208: `saver.restore(latest_checkpoint)` when --restore, `restorer.restore(resnet_v2_50.ckpt)` ONLY
    # otherwise -- resuming needs no ImageNet file.  (--restore with an empty model_dir fails in the reference; here it falls
    # back to the warm start so that the first run of a job can carry the flag.)
    resume = latest_checkpoint(model_dir) if args.restore else None
    if resume is not None:
        note = 'resuming from %s: ImageNet warm start skipped' % resume
    elif args.no_imagenet_init:
        note = 'seeded initialiser (--no-imagenet-init)'
    else:
        pre, note = tf_checkpoint.try_load_imagenet_resnet(args.imagenet_ckpt)       # train_bundle_nobm.py:184-191,208
        if pre is None:
            # the reference's restorer.restore() raises here; silently training from scratch would be a different experiment
            raise SystemExit('train_bundle_nobm.py: cannot warm-start the backbone: %s.  Provide --imagenet-ckpt <resnet_v2_50.ckpt> '
                             '(V1 or V2 TensorFlow checkpoint) or pass --no-imagenet-init to train from the seeded initialiser.' % note)
        try:
            n_hit = tf_checkpoint.apply_imagenet_init(init, pre)    # every expected variable or an error, like Saver.restore
        except ValueError as e:
            raise SystemExit('train_bundle_nobm.py: %s' % e)
        note = 'initialised %d backbone variables from %s' % (n_hit, args.imagenet_ckpt)
    if rank == 0:
        print('note: ' + note)
    tr = Trainer(init, N, H, W, cfg, device=dev, process_group=pg, world_size=world)
    if resume is not None:
        ck = resume
        if ck:
            z = np.load(ck)
            tr.load_state_dict({'params': tr.plan.pack({k: z[k] for k in z.files if not k.startswith('__')}),
                                'adam_m': z['__adam_m'], 'adam_v': z['__adam_v'], 'global_step': int(z['__global_step'])})
            print('restoring {}'.format(ck))
    st_step = tr.global_step
    training_iter = args.iters if args.iters is not None else cfg.training_iter

    def batch_for(step, split):
        seed = (1234 if split == 'train' else 987654) + step * world + rank       # every rank its own shard
        if args.augment:
            from stabnet_amd import data
            raw = synthetic.make_raw_pairs(cfg, N, H, W, seed)
            para, jitter, Hs = data.draw(np.random.default_rng(seed), cfg, N, H, W)
            t = lambda k: torch.from_numpy(raw[k]).to(dev)
            x1, y1, x2, y2, flow, fm1, mk1, fm2, mk2 = data.augment_pairs(
                t('stable'), t('unstable'), t('flow'), t('matches1'), raw['n1'], t('matches2'), raw['n2'], para, jitter, Hs, cfg)
            return {'x1': x1, 'y1': y1, 'x2': x2, 'y2': y2, 'flow': flow, 'matches1': fm1, 'mask1': mk1, 'matches2': fm2,
                    'mask2': mk2}
        b = synthetic.make_train_batch(cfg, N, H, W, seed)
        return {k: torch.from_numpy(v).to(dev) for k, v in b.items()}

    tot_time = tot_train_time = 0.0
    for i in range(st_step, training_iter):
        t0 = time.time()
        batch = batch_for(i, 'train')
        gates = loss_gates(i, cfg)
        tot_time += time.time() - t0
        if (i % cfg.disp_freq == 0 or i == training_iter - 1) and rank == 0 and tr.last is not None:
            print('==========================')
            print('read data time:' + str(tot_time / cfg.disp_freq) + 's')
            print('train time:' + str(tot_train_time / cfg.disp_freq) + 's')
            tot_train_time = tot_time = 0.0
            lo = tr.losses()
            print('Iteration: ' + str(i) + ' Loss: ' + str(lo['total_loss']))
            print({k: round(float(v), 6) for k, v in lo.items() if not isinstance(v, dict)})
            print(learning_rate(i, cfg))
        if (i % cfg.save_freq == 0 or i == training_iter - 1) and i > st_step:
            tr.sync_moving_stats()        # (collective: every rank) BN moving statistics are per-rank (local BN); checkpoint their mean
        if (i % cfg.save_freq == 0 or i == training_iter - 1) and rank == 0 and i > st_step:
            os.makedirs(model_dir, exist_ok=True)
            sd = tr.state_dict()
            arrays = tr.plan.unpack(sd['params'])
            np.savez(os.path.join(model_dir, 'model-%d.npz' % i), __adam_m=sd['adam_m'], __adam_v=sd['adam_v'],
                     __global_step=np.int64(sd['global_step']), **arrays)
        if (i % cfg.test_freq == 0 or i == training_iter - 1) and i > st_step:
            s = 0.0
            for j in range(10):                                       # test_batches
                tr.forward_backward(batch_for(i * 10 + j, 'test'), gates, apply_update=False)
                s += tr.losses()['total_loss']
                tr.global_step -= 1
            if rank == 0:
                print('Test Loss: ' + str(s / 10))
        t1 = time.time()
        tr.forward_backward(batch, gates)
        torch.cuda.synchronize()
        tot_train_time += time.time() - t1
    if rank == 0 and tr.last is not None:
        print('final loss', tr.losses()['total_loss'])
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
