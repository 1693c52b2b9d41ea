"""One data-parallel rank of tests/test_dp_gpu.py (not a test module): joins a gloo group of `world` processes that
share cuda:0, runs ONE PPNTrainer.train_step on its own minibatch shard and saves what the parent compares.

    python tests/dp_worker.py RANK WORLD PORT OUT.pt SECOND_ORDER DTYPE [SIZE]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard_inputs(rank, size=96, batch=2):
    """The shard of rank `rank`: frames seed 500+rank, targets seed 600+rank (host arrays)."""
    from oracle import forward_ref as Fr, targets_ref as T
    from pytorch_pose_proposal_network_amd import prng
    x = Fr.normalize_u8(prng.u8_frames(500 + rank, batch, (size, size)))
    tg = T.synthetic_batch(600 + rank, batch, insize=(size, size), outsize=(size // 16, size // 16))
    return x, tg


def make_trainer(second_order, dtype_name, size=96):
    import torch
    from pytorch_pose_proposal_network_amd import lib as L, synth
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 0),
                    compute_dtype=L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16, insize=(size, size), lr=7e-4,
                    lr_weights=0.01, alpha=0.12, second_order=second_order)
    tr.task.w.copy_(torch.tensor([1.3, 0.8, 1.1, 0.7, 1.1]))
    tr.base = torch.tensor([2.0, 1.5, 0.6, 0.4, 3.0], device="cuda")
    return tr


def main():
    rank, world, port, out, second_order, dtype_name = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                        sys.argv[5] == "1", sys.argv[6])
    size = int(sys.argv[7]) if len(sys.argv) > 7 else 96
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_trainer(second_order, dtype_name, size)
    x, tg = shard_inputs(rank, size)
    xd = torch.as_tensor(x).cuda()
    tgd = {k: torch.from_numpy(v).cuda() for k, v in tg.items()}
    losses, w = tr.train_step(xd, tgd)
    torch.cuda.synchronize()
    torch.save({"losses": losses.cpu(), "w": w.cpu(), "grad": tr.grad.cpu(), "flat": tr.flat.cpu(),
                "world": dist.get_world_size()}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
