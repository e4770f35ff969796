"""GPU: data-parallel semantics of the HIP engine -- 2 ranks sharing cuda:0 (gloo transports the CUDA tensors;
RCCL needs one device per rank, which the 8-GPU bench provides).  SyncBatchNorm-converted model under
DistributedDataParallel must reproduce the single-process result on the concatenated batch: global BN
statistics in forward, cross-rank BN sums in backward, gradients averaged by DDP."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
MP = dict(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
          value_fc_size=32, score_fc_size=16, obs_channels=50)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    g = torch.Generator().manual_seed(7)
    return (torch.randn(8, 50, 9, 9, generator=g), torch.randn(8, 9, 9, 139, generator=g),
            torch.randn(8, 3, generator=g), torch.randn(8, 1, generator=g))


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from keisei_amd.training.model_registry import build_model
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    torch.manual_seed(3)
    model = build_model("se_resnet", MP).to("cuda:0")
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
    obs, cp, cv, cs = (t.to("cuda:0") for t in _inputs())
    sl = slice(rank * 4, rank * 4 + 4)
    out = ddp(obs[sl])
    loss = (out.policy_logits * cp[sl]).sum() + (out.value_logits * cv[sl]).sum() + (out.score_lead * cs[sl]).sum()
    loss.backward()
    torch.save({"policy": out.policy_logits.detach().cpu(), "grads": {n: p.grad.cpu() for n, p in model.named_parameters()},
                "rm": model.input_bn.running_mean.cpu()}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_syncbn_ddp_matches_single_process_full_batch():
    from keisei_amd.training.model_registry import build_model
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
    torch.manual_seed(3)
    model = build_model("se_resnet", MP).to("cuda:0")
    obs, cp, cv, cs = (t.to("cuda:0") for t in _inputs())
    out = model(obs)
    ((out.policy_logits * cp).sum() + (out.value_logits * cv).sum() + (out.score_lead * cs).sum()).backward()
    full = out.policy_logits.detach().cpu()
    assert torch.allclose(torch.cat([r0["policy"], r1["policy"]]), full, rtol=1e-4, atol=2e-5)
    assert torch.allclose(r0["rm"], model.input_bn.running_mean.cpu(), rtol=1e-5, atol=1e-6)
    for n, p in model.named_parameters():
        ref = p.grad.cpu()
        got = 2.0 * r0["grads"][n]                       # DDP averaged the two per-rank gradients
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
        assert err < 2e-3, (n, err)
