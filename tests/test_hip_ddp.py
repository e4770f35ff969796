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


def _fused_worker(rank, world, port, outdir, overlap):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      KA_DDP_OVERLAP="1" if overlap else "0")
    from keisei_amd.hip.grad_reducer import OverlappedGradReducer
    from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams
    from keisei_amd.training.model_registry import build_model
    from keisei_amd.training.value_adapter import MultiHeadValueAdapter
    import bench

    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    torch.manual_seed(3)
    mp_ = dict(MP, num_blocks=3)
    model = build_model("se_resnet", mp_).to("cuda:0")
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
    algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=8, learning_rate=1e-3), model, forward_model=ddp)
    adapter = MultiHeadValueAdapter()
    data = bench.synth_dataset(16, 50 + rank, torch.device("cuda:0"))
    fs = algo._fused_begin(data, torch.device("cuda:0"), adapter)
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)

    dist.all_reduce = counting
    if fs["reducer"] is not None:
        fs["reducer"].bucket_bytes = 4 * 2 * 9 * 32 * 32          # one block per bucket -> 3 conv buckets
    ddp.train()
    # DistributedDataParallel arms its own reducer in its FORWARD: record what it saw there
    sync_flags = []
    hook = ddp.module.register_forward_pre_hook(lambda mod, inp: sync_flags.append(bool(ddp.require_backward_grad_sync)))
    algo._fused_step(fs, torch.arange(8, device="cuda:0"), torch.device("cuda:0"))
    hook.remove()
    dist.all_reduce = real
    algo._fused_end(fs)
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "calls": calls, "ddp_sync_in_forward": sync_flags,
                "log": fs["reducer"].log if fs["reducer"] is not None else None}, os.path.join(outdir, f"f{int(overlap)}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_fused_step_under_ddp_overlapped_exchange_equals_ddp_reducer():
    """The fused PPO step on a SyncBatchNorm + DDP model, 2 ranks (gloo, one GPU): with the engine-driven bucketed
    exchange (default) and with DDP's own post-backward reducer (KA_DDP_OVERLAP=0) the weights after the step agree, both
    ranks stay identical, and the collectives are exactly: one per BatchNorm layer and direction (8 + 8 for 3 blocks) +
    3 conv buckets + FC flat + small tensors -- with every conv bucket issued before the pass ends."""
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for overlap in (True, False):
            mp.spawn(_fused_worker, args=(2, _free_port(), d, overlap), nprocs=2, join=True)
            res[overlap] = [torch.load(os.path.join(d, f"f{int(overlap)}{r}.pt")) for r in (0, 1)]
    for overlap in (True, False):
        a, b = res[overlap]
        for k, v in a["sd"].items():
            assert torch.equal(v, b["sd"][k]), (overlap, k)
    on, off = res[True][0], res[False][0]
    for k, v in on["sd"].items():
        if v.dtype.is_floating_point:
            assert torch.allclose(v, off["sd"][k], rtol=1e-5, atol=1e-6), k
    n_bn = 2 + 2 * 3
    sync = [n for n in on["calls"] if n in (2 * 32 + 1, 2 * 8 + 1)]          # [sums | count] vectors (C = 32, policy 8)
    assert len(sync) == 2 * n_bn
    tags = [e[1] for e in on["log"] if e[0] == "launch"]
    assert tags == ["conv[2:3]", "conv[1:2]", "conv[0:1]", "fc", "small"], tags
    assert len(on["calls"]) == 2 * n_bn + 5
    # with the engine-driven exchange DDP's own reducer must stay disarmed (its forward ran under no_sync): otherwise every
    # gradient is copied into DDP's buckets and all-reduced a second time; without it, DDP's reducer is the exchange
    assert on["ddp_sync_in_forward"] == [False] and off["ddp_sync_in_forward"] == [True]


@pytest.mark.timeout(300)
def test_rccl_initialises_and_carries_the_step_collectives():
    """RCCL itself (backend "nccl") on the one GPU of the box: a world of ONE rank launched exactly as bench.py's N > 1
    branch does (init_process_group("nccl", device_id=...), SyncBatchNorm conversion, DDP wrap), with the SyncBatchNorm
    and gradient-bucket collectives FORCED on (KA_FORCE_COLLECTIVES=1) so that every all-reduce of a step really goes
    through RCCL.  (Two ranks cannot share a device under RCCL; the multi-GPU curve is the driver's.)"""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               KA_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--workload", "2x32", "--dist-dry-run"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    import json
    out = json.loads(line)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["collectives_per_step"]["syncbn"] == 2 * (2 + 2 * 2) and out["collectives_per_step"]["gradient"] >= 3
    assert all(map(lambda v: v == v, out["train_metrics"].values()))
    # every statistics all-reduce is issued asynchronously and waited for once, right before its coefficient kernel
    exp = out["syncbn_allreduce_exposure"]
    assert exp["waits_per_step"] == out["collectives_per_step"]["syncbn"] and exp["exposed_us_per_step"] >= 0.0
