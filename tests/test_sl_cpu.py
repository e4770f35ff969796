"""CPU: the supervised-learning path (keisei_amd.sl) against the reference's own shard writer / SLDataset / SLTrainer
run (golden g8_sl, made by oracle/make_golden.py from keisei/sl/*)."""
import json

import numpy as np
import pytest
import torch
from torch.utils.data import default_collate

from keisei_amd.sl import dataset as ds_mod
from keisei_amd.sl.dataset import SLDataset, write_shard
from keisei_amd.sl.trainer import SLConfig, SLTrainer
from keisei_amd.training.model_registry import build_model

MP = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
          value_fc_size=32, score_fc_size=16, obs_channels=50)


def make_dir(g, tmp_path):
    for shard in (0, 1, 2):
        write_shard(tmp_path / f"shard_{shard}.bin", g.np(f"shard{shard}.obs"), g.np(f"shard{shard}.policy"),
                    g.np(f"shard{shard}.value"), g.np(f"shard{shard}.score"))
    return tmp_path


def test_constants_and_shard_bytes(golden, tmp_path):
    g = golden("g8_sl")
    assert (ds_mod.OBS_SIZE, ds_mod.OBS_BYTES, ds_mod.RECORD_SIZE, ds_mod.SCORE_NORMALIZATION) == (4050, 16200, 16220, 76.0)
    make_dir(g, tmp_path)
    for shard in (0, 1, 2):
        mine = np.frombuffer((tmp_path / f"shard_{shard}.bin").read_bytes(), dtype=np.uint8)
        assert np.array_equal(mine, g.np(f"shard{shard}.bytes"))            # the reference writer's file, byte for byte


def test_dataset_items_batches_and_guards(golden, tmp_path):
    g = golden("g8_sl")
    ds = SLDataset(make_dir(g, tmp_path))
    assert len(ds) == 22 and [n for _, n in ds.shards] == [9, 6, 7]
    item = ds[10]                                                          # second shard, local 1
    assert torch.equal(item["observation"], g["shard1.obs"][1].reshape(50, 9, 9))
    assert int(item["policy_target"]) == int(g.np("shard1.policy")[1]) and item["policy_target"].dtype == torch.long
    assert int(item["value_target"]) == int(g.np("shard1.value")[1])
    assert float(item["score_target"]) == float(g.np("shard1.score")[1]) and item["score_target"].dtype == torch.float32
    idx = [21, 0, 9, 8, 15, 14, 3]
    got, ref = ds.read_batch(idx), default_collate([ds[i] for i in idx])
    assert set(got) == set(ref) and all(torch.equal(got[k], ref[k]) and got[k].dtype == ref[k].dtype for k in ref)
    for bad in (-1, 22):
        with pytest.raises(IndexError, match="out of range for dataset with 22 positions"):
            ds[bad]
        with pytest.raises(IndexError, match="out of range"):
            ds.read_batch([0, bad])
    with pytest.raises(ValueError, match="max_cache_size must be >= 1"):
        SLDataset(tmp_path, max_cache_size=0)
    small = SLDataset(tmp_path, max_cache_size=1)                           # LRU of one map: still correct across shards
    assert torch.equal(small.read_batch(idx)["observation"], ref["observation"]) and len(small._mmap_cache) == 1
    small.clear_cache()
    assert not small._mmap_cache
    # numeric shard order, empty files skipped, trailing bytes tolerated
    (tmp_path / "shard_10.bin").write_bytes((tmp_path / "shard_0.bin").read_bytes()[:ds_mod.RECORD_SIZE + 5])
    (tmp_path / "shard_3.bin").write_bytes(b"")
    ds2 = SLDataset(tmp_path)
    assert [p.name for p, _ in ds2.shards] == ["shard_0.bin", "shard_1.bin", "shard_2.bin", "shard_10.bin"] and len(ds2) == 23
    assert torch.equal(ds2[22]["observation"], ds2[0]["observation"])


def test_dataset_rejects_bad_records_and_placeholders(tmp_path):
    obs = np.zeros((3, ds_mod.OBS_SIZE), dtype=np.float32)
    write_shard(tmp_path / "shard_0.bin", obs, np.array([5, 11259, 7]), np.array([0, 1, 3]), np.zeros(3, dtype=np.float32))
    ds = SLDataset(tmp_path)
    assert int(ds[0]["policy_target"]) == 5
    with pytest.raises(ValueError, match=r"Invalid policy_target=11259 at index 1 \(shard=shard_0.bin, local=1\)"):
        ds[1]
    with pytest.raises(ValueError, match=r"Invalid value_target=3 at index 2"):
        ds[2]
    with pytest.raises(ValueError, match="Invalid value_target=3 at index 2"):
        ds.read_batch([0, 2, 1])                 # the first offending position in batch order, as item-wise collation would
    (tmp_path / "shard_meta.json").write_text(json.dumps({"placeholder": True}))
    with pytest.raises(ValueError, match="contains placeholder data"):
        SLDataset(tmp_path)
    assert len(SLDataset(tmp_path, allow_placeholder=True)) == 3
    (tmp_path / "shard_meta.json").write_text("{not json")
    with pytest.raises(ValueError, match="Corrupt shard_meta.json"):
        SLDataset(tmp_path)


def test_config_validation():
    for kw, msg in ((dict(grad_clip=0), "grad_clip must be > 0"), (dict(total_epochs=-1), "total_epochs must be >= 0"),
                    (dict(batch_size=0), "batch_size must be > 0"), (dict(learning_rate=0.0), "learning_rate must be > 0"),
                    (dict(num_workers=-2), "num_workers must be >= 0"), (dict(lambda_value=-0.1), "lambda_value must be >= 0"),
                    (dict(lambda_score=float("nan")), "lambda_score must be finite"),
                    (dict(lambda_policy=float("inf")), "lambda_policy must be finite")):
        with pytest.raises(ValueError, match=msg):
            SLConfig(data_dir="x", **kw)
    c = SLConfig(data_dir="x")
    assert (c.batch_size, c.learning_rate, c.total_epochs, c.lambda_policy, c.lambda_value, c.lambda_score, c.grad_clip,
            c.use_amp) == (4096, 1e-3, 30, 1.0, 1.5, 0.02, 0.5, False)


def test_trainer_reproduces_the_reference_run(golden, tmp_path, monkeypatch):
    g = golden("g8_sl")
    make_dir(g, tmp_path)
    model = build_model("se_resnet", MP)
    model.load_state_dict(g.sub("sd0."))
    trainer = SLTrainer(model, SLConfig(data_dir=str(tmp_path), batch_size=8, learning_rate=1e-3, total_epochs=5, lambda_score=0.05))
    visited = []
    real = SLDataset.__getitem__
    monkeypatch.setattr(SLDataset, "__getitem__", lambda self, i: visited.append(int(i)) or real(self, i))
    torch.manual_seed(82)
    for ep in range(2):
        visited.clear()
        met = trainer.train_epoch()
        assert visited == g.np(f"order{ep}").tolist()                      # same shuffling as the reference's DataLoader
        for k in ("policy_loss", "value_loss", "score_loss"):
            ref = float(g.np(f"metric{ep}.{k}"))
            assert abs(met[k] - ref) <= 1e-5 * max(1.0, abs(ref)), (ep, k, met[k], ref)
        assert abs(trainer.optimizer.param_groups[0]["lr"] - float(g.np(f"lr{ep}"))) < 1e-12
        sd, ref_sd = model.state_dict(), g.sub(f"sd{ep + 1}.")
        for k, v in ref_sd.items():
            if v.dtype.is_floating_point:
                assert float((sd[k] - v).abs().max()) <= 2e-5, (ep, k)
            else:
                assert int(sd[k]) == int(v)


def test_empty_directory_does_not_tick_the_scheduler(tmp_path):
    model = build_model("se_resnet", MP)
    trainer = SLTrainer(model, SLConfig(data_dir=str(tmp_path), batch_size=4))
    assert trainer.train_epoch() == {"policy_loss": 0.0, "value_loss": 0.0, "score_loss": 0.0}
    assert trainer.optimizer.param_groups[0]["lr"] == 1e-3 and trainer.scheduler.last_epoch == 0
