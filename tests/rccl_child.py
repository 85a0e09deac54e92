"""Child of tests/test_rccl_gpu.py (not a test module): ONE fresh rank with backend "nccl" (= RCCL).  Runs `steps` optimiser
steps of Trainer with the communication path forced on (every bucket all-reduced on the communication stream, wait_stream
joins) next to the same steps of a Trainer without a process group, and writes both end states to <out>/rccl.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir, steps=3):
    from stabnet_amd import parallel, synthetic
    from stabnet_amd.config import Config
    from stabnet_amd.train import Trainer
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pg = parallel.init_process_group("nccl", device=dev, single_rank_group=True)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    N, H, W = 2, 64, 96
    cfg = Config(height=H, width=W, batch_size=N, max_matches=48)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.3)
    gates = {"use_theta_loss": 1, "use_temp_loss": 1, "use_black_loss": 1, "use_theta_only": 0}
    batches = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
                for k, v in synthetic.make_train_batch(cfg, N, H, W, 11 + i).items()} for i in range(steps)]
    end = {}
    for tag, kw in (("plain", {}), ("rccl", {"process_group": pg, "world_size": 1, "force_comm": True})):
        tr = Trainer(P, N, H, W, cfg, device=dev, **kw)
        if tag == "rccl":
            assert tr.comm and tr.comm_stream is not None
            tr.comm_timing = []
        for b in batches:
            tr.forward_backward(b, gates, apply_update=True)
        torch.cuda.synchronize()
        end[tag + "_params"] = tr.params.cpu().numpy()
        end[tag + "_grads"] = tr.grad_flat().cpu().numpy()
        end[tag + "_m"] = tr.adam_m.cpu().numpy()
        if tag == "rccl":
            end["n_collectives"] = len(tr.comm_timing)
            end["bucket_bytes"] = np.array([b for _, _, b in tr.comm_timing], np.int64)
            end["allreduce_ms"] = np.array([a.elapsed_time(b) for a, b, _ in tr.comm_timing])
            end["nt"] = tr.nt
    # a collective that really moves data through RCCL: 4 MiB all-reduce + broadcast on the default stream
    t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    dist.all_reduce(t, group=pg)
    dist.broadcast(t, 0, group=pg)
    torch.cuda.synchronize()
    end["probe_ok"] = bool(torch.equal(t, torch.arange(1 << 20, dtype=torch.float32, device=dev)))
    end["rccl_mapped"] = any("librccl" in ln for ln in open("/proc/self/maps"))
    np.savez(os.path.join(out_dir, "rccl.npz"), **end)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
