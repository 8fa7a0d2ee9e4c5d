"""Worker of tests/test_sp_utils.py: one gloo rank exercising the collectives of sink_attention.sp_utils on CPU."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
from sink_attention.sp_utils import _AllGatherSeq, prepare_sink_kv_for_sp, reduce_sink_kv_grads


def main():
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    grp = dist.group.WORLD
    B, H, n, D, ns = 1, 2, 6, 4, 3
    g = torch.Generator().manual_seed(5)
    k_full, v_full = torch.randn(B, H, P * n, D, generator=g), torch.randn(B, H, P * n, D, generator=g)
    k, v = k_full[:, :, rank * n:(rank + 1) * n].clone(), v_full[:, :, rank * n:(rank + 1) * n].clone()

    ks, vs = prepare_sink_kv_for_sp(k, v, ns, grp)
    if rank == 0:
        assert ks is k and vs is v
    else:
        assert ks.shape[2] == n + ns
        assert torch.equal(ks[:, :, :ns], k_full[:, :, :ns]) and torch.equal(vs[:, :, :ns], v_full[:, :, :ns])
        assert torch.equal(ks[:, :, ns:], k)

    # every rank contributes (rank + 1) to the sink rows; rank 0 must end up with the sum, the others stripped
    dk = torch.full((B, H, ks.shape[2], D), float(rank + 1))
    dv = 2 * dk
    rk, rv = reduce_sink_kv_grads(dk, dv, ns, grp)
    tot = float(sum(range(1, P + 1)))
    if rank == 0:
        assert rk.shape[2] == n and torch.all(rk[:, :, :ns] == tot) and torch.all(rk[:, :, ns:] == 1.0)
        assert torch.all(rv[:, :, :ns] == 2 * tot)
    else:
        assert rk.shape[2] == n and torch.all(rk == float(rank + 1))

    # autograd-aware all-gather: forward = the full sequence, backward = sum over ranks of their gradient slice
    kl = k.clone().requires_grad_(True)
    full = _AllGatherSeq.apply(kl, grp)
    assert torch.equal(full.detach(), k_full)
    w = torch.arange(P * n, dtype=torch.float32).view(1, 1, -1, 1) * (rank + 1)
    (full * w).sum().backward()
    exp = torch.arange(rank * n, (rank + 1) * n, dtype=torch.float32).view(1, 1, -1, 1) * tot
    assert torch.allclose(kl.grad, exp.expand_as(kl.grad))
    dist.barrier()
    if rank == 0:
        print("SP_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
