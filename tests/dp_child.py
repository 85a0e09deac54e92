"""Child rank of tests/test_parallel_gpu.py (not a test module): one data-parallel training step on this rank's shard
of a seeded global batch (process group given by the torchrun environment), plus the single-process gradient of the same
shard.  Writes <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    from stabnet_amd import parallel, synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.train import Trainer
    rank, _, world = parallel.env_world()
    dev = torch.device("cuda", 0)                                # both ranks share the box's one GPU
    torch.cuda.set_device(dev)
    pg = parallel.init_process_group(os.environ.get("STABNET_TEST_BACKEND", "gloo"), device=dev)
    n_local, H, W = 2, 64, 96
    cfg = Config(height=H, width=W, batch_size=n_local, max_matches=48)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    glob = synthetic.make_train_batch(cfg, n_local * world, H, W, 5)
    mine = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in parallel.shard_batch(glob, rank, world).items()}
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    single = Trainer(P, n_local, H, W, cfg, device=dev)          # this shard alone, no collective, no update
    single.forward_backward(mine, gates, apply_update=False)
    g_single = single.grad_flat().cpu().numpy()
    tr = Trainer(P, n_local, H, W, cfg, device=dev, process_group=pg, world_size=world)
    p0 = tr.params[:tr.nt].cpu().numpy()
    tr.comm_timing = []
    tr.forward_backward(mine, gates, apply_update=True)
    torch.cuda.synchronize()
    mov_before = tr.params[tr.nt:].cpu().numpy()
    tr.sync_moving_stats()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), g_single=g_single, g_dp=tr.grad_flat().cpu().numpy(), p0=p0,
             p1=tr.params[:tr.nt].cpu().numpy(), adam_m=tr.adam_m.cpu().numpy(), mov_before=mov_before,
             mov_after=tr.params[tr.nt:].cpu().numpy(), n_buckets=len(tr.comm_timing),
             bucket_bytes=np.array([b for _, _, b in tr.comm_timing], np.int64))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
