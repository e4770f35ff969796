"""GPU: the fused supervised-learning step (ka_policy_ce + ka_value_loss + hand-written backward + fused clip/Adam)
against the reference's SLTrainer run (golden g8_sl) and against fp32 torch autograd for the loss kernel."""
import math

import pytest
import torch
import torch.nn.functional as F

from keisei_amd import _lib
from keisei_amd.sl.dataset import SLDataset, write_shard
from keisei_amd.sl.trainer import SLConfig, SLTrainer
from keisei_amd.training.model_registry import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda"
A = 11259
MP = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
          value_fc_size=32, score_fc_size=16, obs_channels=50)


def test_policy_ce_kernel_matches_autograd():
    B, S = 6, 11
    g = torch.Generator().manual_seed(4)
    logits = (3 * torch.randn(B, A, generator=g)).requires_grad_(True)
    targets = torch.randint(0, A, (S,), generator=g)
    targets[0], targets[1] = 0, A - 1
    idx = torch.tensor([1, 0, 10, 4, 4, 7])
    w = 0.7 / B
    per_row = F.cross_entropy(logits, targets[idx], reduction="none")
    (w * per_row.sum()).backward()
    dl = torch.empty(B, A, device=DEV)
    rl = torch.empty(B, device=DEV)
    flags = torch.zeros(2, dtype=torch.int32, device=DEV)
    gs = torch.tensor([4.0], device=DEV)
    _lib.call("ka_policy_ce", logits.detach().to(DEV), targets.to(DEV), idx.to(DEV), dl, rl, flags, gs, w, B, A, _lib.stream_ptr())
    assert flags.cpu().tolist() == [0, 0]
    assert torch.allclose(rl.cpu(), per_row.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(dl.cpu(), 4.0 * logits.grad, rtol=1e-4, atol=1e-9)
    bad = targets.clone(); bad[7] = A
    _lib.call("ka_policy_ce", logits.detach().to(DEV), bad.to(DEV), idx.to(DEV), None, rl, flags, None, w, B, A, _lib.stream_ptr())
    assert flags.cpu().tolist() == [0, 1]
    flags.zero_()
    nan = logits.detach().clone(); nan[2, 5] = float("nan")
    _lib.call("ka_policy_ce", nan.to(DEV), targets.to(DEV), idx.to(DEV), None, rl, flags, None, w, B, A, _lib.stream_ptr())
    assert flags.cpu().tolist() == [1, 0]


def make_dir(g, tmp_path):
    for shard in (0, 1, 2):
        write_shard(tmp_path / f"shard_{shard}.bin", g.np(f"shard{shard}.obs"), g.np(f"shard{shard}.policy"),
                    g.np(f"shard{shard}.value"), g.np(f"shard{shard}.score"))
    return tmp_path


def test_fused_trainer_reproduces_the_reference_run(golden, tmp_path, monkeypatch):
    g = golden("g8_sl")
    make_dir(g, tmp_path)
    model = build_model("se_resnet", MP)
    model.load_state_dict(g.sub("sd0."))
    model.to(DEV)
    trainer = SLTrainer(model, SLConfig(data_dir=str(tmp_path), batch_size=8, learning_rate=1e-3, total_epochs=5, lambda_score=0.05))
    assert trainer._fused_path_available()
    visited = []
    real = SLDataset.read_batch
    monkeypatch.setattr(SLDataset, "read_batch", lambda self, idx, pin=False: visited.extend(int(i) for i in idx) or real(self, idx, pin))
    monkeypatch.setattr(SLDataset, "__getitem__", lambda self, i: pytest.fail("the fused path decodes whole batches"))
    torch.manual_seed(82)
    for ep in range(2):
        visited.clear()
        met = trainer.train_epoch()
        assert visited == g.np(f"order{ep}").tolist()
        for k in ("policy_loss", "value_loss", "score_loss"):
            ref = float(g.np(f"metric{ep}.{k}"))
            assert abs(met[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (ep, k, met[k], ref)
        assert abs(trainer.optimizer.param_groups[0]["lr"] - float(g.np(f"lr{ep}"))) < 1e-12
        sd, ref_sd = model.state_dict(), g.sub(f"sd{ep + 1}.")
        steps = 3 * (ep + 1)
        for k, v in ref_sd.items():
            if v.dtype.is_floating_point:      # Adam moves numerically-zero-gradient elements by +-lr on rounding noise alone
                diff = (sd[k].cpu() - v).abs()
                assert float(diff.max()) <= 0.05 * steps * 1e-3, (ep, k, float(diff.max()))
                assert float((diff > 3e-5).float().mean()) <= 5e-3, (ep, k)
            else:
                assert int(sd[k]) == int(v)
    assert float(trainer.optimizer.state_dict()["state"][0]["step"]) == 6.0


def test_fused_trainer_bf16_amp(golden, tmp_path):
    g = golden("g8_sl")
    make_dir(g, tmp_path)
    model = build_model("se_resnet", MP)
    model.load_state_dict(g.sub("sd0."))
    model.to(DEV)
    trainer = SLTrainer(model, SLConfig(data_dir=str(tmp_path), batch_size=8, total_epochs=5, lambda_score=0.05, use_amp=True))
    assert trainer._fused_path_available() and trainer.scaler.is_enabled()
    torch.manual_seed(82)
    first = trainer.train_epoch()
    for _ in range(3):
        last = trainer.train_epoch()
    assert all(math.isfinite(v) for v in last.values())
    assert abs(first["policy_loss"] - float(g.np("metric0.policy_loss"))) < 0.05
    assert last["policy_loss"] < first["policy_loss"] and last["value_loss"] < first["value_loss"]     # 22 positions: it memorises
    assert float(trainer.scaler.get_scale()) == 65536.0
