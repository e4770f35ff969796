"""2-rank data-parallel update on CPU (gloo), mirroring the reference's tests/integration/test_ddp_training.py:
each rank owns its own rollout data, DDP averages gradients, learned weights end up identical on every rank
(BatchNorm running statistics stay per-rank without SyncBatchNorm).  Also covers the distributed helpers."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keisei_amd.training import distributed as kd

MP = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
          value_fc_size=32, score_fc_size=16, obs_channels=50)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, outdir: str) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2 if world <= 2 else 1)
    from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer
    from keisei_amd.training.model_registry import build_model
    from keisei_amd.training.value_adapter import MultiHeadValueAdapter

    ctx = kd.get_distributed_context()
    assert ctx.is_distributed and ctx.rank == rank and ctx.world_size == world and ctx.is_main == (rank == 0)
    kd.setup_distributed(ctx, backend="gloo")
    kd.seed_all_ranks(42 + rank)                         # per-rank seeds: weights differ until DDP's broadcast
    model = build_model("se_resnet", MP)
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=8, epochs_per_batch=2), model, forward_model=ddp)
    T, N = 4, 4
    buf = KataGoRolloutBuffer(N, (50, 9, 9), 11259)
    g = torch.Generator().manual_seed(100 + rank)
    for t in range(T):
        legal = torch.rand(N, 11259, generator=g) < 0.02
        acts = torch.randint(0, 11259, (N,), generator=g)
        legal[torch.arange(N), acts] = True
        last = t == T - 1
        buf.add(torch.randn(N, 50, 9, 9, generator=g), acts, -5 + 0.1 * torch.randn(N, generator=g),
                0.2 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g), torch.full((N,), last),
                torch.full((N,), last), legal, torch.randint(0, 3, (N,), generator=g) if last else torch.full((N,), -1),
                torch.randn(N, generator=g).clamp(-1.5, 1.5))
    metrics = algo.update(buf, torch.zeros(N), value_adapter=MultiHeadValueAdapter())
    torch.save({"sd": model.state_dict(), "metrics": metrics}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    kd.cleanup_distributed(ctx)


@pytest.mark.timeout(300)
def test_two_rank_ddp_update_keeps_ranks_in_sync():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        a = torch.load(os.path.join(d, "rank0.pt"))
        b = torch.load(os.path.join(d, "rank1.pt"))
    assert a["metrics"]["policy_loss"] != b["metrics"]["policy_loss"]      # different data per rank
    for k, v in a["sd"].items():
        if "running_" in k or "num_batches" in k:
            continue
        assert torch.allclose(v, b["sd"][k], rtol=0, atol=1e-7), k


@pytest.mark.timeout(600)
def test_eight_rank_ddp_update_keeps_ranks_in_sync():
    """The rank count of BASELINE configs[3] (keisei-ddp.toml: 8 GPUs), rehearsed on CPU over gloo: eight ranks with their own
    rollouts, one update() each, identical learned weights everywhere afterwards."""
    world = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world)]
    assert len({o["metrics"]["policy_loss"] for o in outs}) == world          # different data on every rank
    for o in outs[1:]:
        for k, v in outs[0]["sd"].items():
            if "running_" in k or "num_batches" in k:
                continue
            assert torch.allclose(v, o["sd"][k], rtol=0, atol=1e-7), k


def test_distributed_context_without_torchrun(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    ctx = kd.get_distributed_context()
    assert not ctx.is_distributed and ctx.world_size == 1 and ctx.is_main and ctx.device.type in ("cpu", "cuda")
    kd.setup_distributed(ctx)          # no-op
    kd.cleanup_distributed(ctx)
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(RuntimeError, match="LOCAL_RANK"):
        kd.get_distributed_context()
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    ctx = kd.get_distributed_context()
    assert ctx.rank == 1 and not ctx.is_main
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="requires CUDA"):
            kd.setup_distributed(ctx, backend="nccl")


def _reducer_worker(rank: int, world: int, port: int, outdir: str) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo")
    from keisei_amd.hip.grad_reducer import OverlappedGradReducer

    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)

    dist.all_reduce = counting
    red = OverlappedGradReducer(bucket_bytes=4 * 100)
    timeline = []
    buckets = []
    for b in range(3):                                   # three "blocks": each fills one bucket, exchanged at once
        flat = torch.full((100,), float(rank + 1 + b))
        timeline.append(("backward", b))
        red.launch(flat, f"conv[{b}]")
        timeline.append(("launched", b))
        buckets.append(flat)
    smalls = [torch.full((3,), float(rank)), torch.full((2, 2), 10.0 * rank)]
    for t in smalls:
        red.add_small(t)
    smalls = red.finish()                                # views of the reduced packed buffer, in add_small order
    dist.all_reduce = real
    torch.save({"buckets": buckets, "smalls": [t.clone() for t in smalls], "calls": calls, "log": red.log, "collectives": red.collectives},
               os.path.join(outdir, f"red{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_overlapped_reducer_two_ranks():
    """The engine-driven gradient exchange (hip/grad_reducer.py) on 2 gloo ranks: every bucket's collective is issued
    while the 'backward' is still running (before the next block), the small tensors travel in ONE coalesced collective
    at the end, the result is the rank average (DDP semantics), and the number of collectives is buckets + 1."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_reducer_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        a, b = torch.load(os.path.join(d, "red0.pt")), torch.load(os.path.join(d, "red1.pt"))
    for r in (a, b):
        assert r["collectives"] == 4 and r["calls"] == [100, 100, 100, 7]
        assert [e[0] for e in r["log"]] == ["launch", "launch", "launch", "launch", "finish"]
        assert [e[1] for e in r["log"][:4]] == ["conv[0]", "conv[1]", "conv[2]", "small"]
        for k, flat in enumerate(r["buckets"]):
            assert torch.equal(flat, torch.full((100,), 1.5 + k))          # mean of (1 + k) and (2 + k)
        assert torch.equal(r["smalls"][0], torch.full((3,), 0.5)) and torch.equal(r["smalls"][1], torch.full((2, 2), 5.0))
