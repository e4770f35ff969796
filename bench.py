#!/usr/bin/env python3
"""PPO samples/sec of the Keisei SE-ResNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: started plainly like that, the process spawns its own N ranks (`python -m torch.distributed.run --nproc-per-node N`
on 127.0.0.1 with a free port, as the reference's run.sh:309-310 does with torchrun) before anything touches the GPU,
forwards rank 0's JSON line and exits with the ranks' status; started BY torch.distributed.run (WORLD_SIZE set) it is a rank.

One "step" = one KataGo-PPO minibatch through the product path
(keisei_amd.training.katago_ppo.KataGoPPOAlgorithm._fused_step): fused row-gather + SE-ResNet
forward (train-mode BN), fused loss+gradient kernel, hand-written backward, fused
GradScaler/clip/Adam -- on a device-resident synthetic epoch dataset (inputs already in HBM).
Default workload = BASELINE.json configs[2]: se_resnet 40x256, minibatch 4096, bf16 activations/MFMA
(the reference's production AMP mode), one rank per GPU, DDP (+SyncBatchNorm, as keisei-ddp.toml)
gradient all-reduce over RCCL when N > 1; per-rank batch fixed (weak scaling).

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline     -- dominant kernel (conv3x3 implicit-GEMM, fwd+dgrad): achieved MFMA TFLOP/s from the
                  algorithmic FLOPs per launch / average launch duration (HIP events on the launch stream)
  cpu_baseline -- the oracle (CPU restatement of the reference) timed on this host, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (num_blocks, channels, se_reduction, global_pool_channels, policy_channels, value_fc, score_fc, T, N, B)
    "40x256": (40, 256, 16, 128, 32, 256, 128, 128, 128, 4096),
    "6x128": (6, 128, 16, 128, 32, 256, 128, 128, 64, 2048),
    "2x32": (2, 32, 8, 16, 8, 32, 16, 8, 8, 32),
}
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def kernel_source_id(names=("common.h", "conv3x3.hip")) -> str:
    """sha256[:16] over the HIP sources the roofline kernel (conv3x3_kernel) is built from: ties a counter profile under
    profiles/ to the build of that kernel."""
    import hashlib
    h = hashlib.sha256()
    for name in names:
        f = ROOT / "keisei_amd" / "csrc" / name
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_sample(nb, C, P, n_params, B, act_bytes):
    """SURVEY 8d contract figure, scaled to the activation storage actually used (4 -> act_bytes)."""
    act = act_bytes * 81 * C * (23 * nb + 10) + 4 * 81 * (100 + 556 + 4 * P)
    return act + 40.0 * n_params / B


def train_flops_per_sample(nb, C, G, R, P, V, S):
    fwd = 2 * 81 * 9 * 50 * C + nb * (2 * (2 * 81 * 9 * C * C) + 2 * (3 * C * G + G * C + C * R + R * 2 * C)) \
        + 2 * 81 * (C * P + P * 139) + 2 * (3 * C * V + 3 * V + 3 * C * S + S)
    return 3.0 * fwd


def synth_dataset(total, seed, device):
    """Mirror of the reference's scripts/profile_hotpath.py:436-454 synthetic update inputs."""
    g = torch.Generator().manual_seed(seed)
    A = 11259
    obs = torch.randn(total, 50, 9, 9, generator=g)
    masks = torch.zeros(total, A, dtype=torch.bool)
    masks[:, : A // 3] = True
    actions = torch.randint(0, A // 3, (total,), generator=g)
    cats = torch.randint(-1, 3, (total,), generator=g)
    return {
        "obs": obs.to(device), "masks": masks.to(device), "mask_words": 0, "n_actions": A, "actions": actions.to(device),
        "old_lp": (-8.2 + 0.05 * torch.randn(total, generator=g)).to(device),
        "adv": torch.randn(total, generator=g).to(device), "cats": cats.to(device),
        "score_t": torch.randn(total, generator=g).clamp(-1.5, 1.5).to(device),
    }


def whole_update_rate(algo, adapter, T, N, device, stores=("host", "device")):
    """SURVEY 8(d) secondary metric and the 8(f1) row: T add() calls of N transitions with the step tensors on the
    device (as katago_loop.py:1523 passes them), then one update() through the public API -- batched GAE, advantage
    normalisation, the epochs_per_batch x ceil(TN/B) minibatch steps, the metric read-back.  Measured for both
    stores: "host" = the reference's CPU buffer (ten .cpu() per add, pinned H2D of the epoch inside update()),
    "device" = the device-resident store (one append launch per add, nothing uploaded)."""
    from keisei_amd.training.katago_ppo import KataGoRolloutBuffer
    A = 11259
    out = {"transitions": T * N, "epochs_per_batch": algo.params.epochs_per_batch,
           "includes": "GAE, advantage normalisation, epoch dataset hand-over, minibatch gathers, metric read-back"}
    for store in stores:
        g = torch.Generator().manual_seed(99)
        os.environ["KA_ROLLOUT_BUFFER"] = store
        buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
        masks = torch.zeros(N, A, dtype=torch.bool)
        masks[:, : A // 3] = True
        masks = masks.to(device)
        fill = 0.0
        for t in range(T):
            last = t == T - 1
            done = torch.full((N,), last, dtype=torch.bool)
            cats = torch.randint(0, 3, (N,), generator=g) if last else torch.full((N,), -1, dtype=torch.long)
            step = [torch.randn(N, 50, 9, 9, generator=g), torch.randint(0, A // 3, (N,), generator=g),
                    -8.2 + 0.05 * torch.randn(N, generator=g), torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g),
                    done, done, None, cats, torch.randn(N, generator=g).clamp(-1.5, 1.5)]
            step = [masks if v is None else v.to(device) for v in step]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            buf.add(*step)
            torch.cuda.synchronize()
            fill += time.perf_counter() - t0
        nv = torch.randn(N, generator=g).to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        algo.update(buf, nv, value_adapter=adapter)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[store] = {"samples_per_s": round(algo.params.epochs_per_batch * T * N / dt, 1), "update_seconds": round(dt, 3),
                      "add_ms_per_step": round(1e3 * fill / T, 3)}
    os.environ.pop("KA_ROLLOUT_BUFFER", None)
    out["samples_per_s"] = out["device"]["samples_per_s"]
    return out


def sl_epoch_rate(make_model, positions, batch, use_amp):
    """one SLTrainer.train_epoch() over a synthetic shard directory (write_shard format, 16 220 B per
    position, under /tmp) through the public API -- shard gather on a helper thread, pinned H2D, fused step."""
    import tempfile
    from pathlib import Path

    import numpy as np
    from keisei_amd.sl.dataset import OBS_SIZE, write_shard
    from keisei_amd.sl.trainer import SLConfig, SLTrainer
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as tmp:
        per = 4096
        for i in range(positions // per):
            write_shard(Path(tmp) / f"shard_{i}.bin", rng.standard_normal((per, OBS_SIZE), dtype=np.float32),
                        rng.integers(0, 11259, per), rng.integers(0, 3, per), rng.standard_normal(per).astype(np.float32))
        trainer = SLTrainer(make_model(), SLConfig(data_dir=tmp, batch_size=batch, use_amp=use_amp))
        assert trainer._fused_path_available()
        trainer.train_epoch()                       # warm-up: page cache, allocator, graph-free first launches
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = trainer.train_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"positions_per_s": round(positions / dt, 1), "seconds": round(dt, 3), "positions": positions, "batch": batch,
            "includes": "shuffled shard gather (mmap), pinned H2D, forward, CE/CE/MSE loss, backward, clip, Adam",
            "policy_loss": round(m["policy_loss"], 4)}


def _cpu_share() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota when there is one."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return avail


def cpu_baseline(shape, seconds_budget=25.0):
    """Times the oracle's PPO minibatch step (fp32 CPU PyTorch restatement of the reference) on this host, two ways, and
    reports the faster as `value` with both stated: (a) at most 16 threads (a one-GPU box's CPU share) with the minibatch sized
    from a probe step; (b) SURVEY 8d's specification -- every core the process may use, minibatch 256 for 40x256.  Bounded: each
    leg stops after ~12 s of timed CPU work (at least one step)."""
    from oracle import keisei_oracle as orc

    nb, C, Rr, G, P, V, S = shape
    ns = orc.NetShape(nb, C, Rr, G, P, V, S)
    avail = _cpu_share()
    sd = orc.synth_state_dict(ns)
    w = orc.LossWeights(1.0, 1.5, 0.1, 0.01, 0.2)

    def leg(threads, Bc, label, min_steps):
        torch.set_num_threads(threads)
        mb = orc.synth_minibatch(Bc, seed=1234)
        state, times = None, []
        t_all = time.perf_counter()
        while len(times) < min_steps or ((time.perf_counter() - t_all) < 12.0 and len(times) < 40):
            t0 = time.perf_counter()
            _, state, _ = orc.ppo_minibatch_step(sd, nb, mb, w, state)
            times.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline {label} step {len(times)}: {times[-1]:.1f} s (minibatch {Bc}, {threads} threads)",
                  file=sys.stderr, flush=True)
            if time.perf_counter() - t_all > 30.0:
                break
        med = sorted(times)[len(times) // 2]
        return {"samples_per_s": round(Bc / med, 2), "threads": threads, "minibatch": Bc, "steps": len(times),
                "cpu_seconds": round(sum(times), 1)}

    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    orc.ppo_minibatch_step(sd, nb, orc.synth_minibatch(4, seed=1), w, None)          # probe (also warms the allocator)
    probe = time.perf_counter() - t0
    print(f"[bench] cpu_baseline probe: minibatch 4 took {probe:.1f} s on {threads} threads", file=sys.stderr, flush=True)
    Bc = 4
    while Bc < 256 and probe * (2 * Bc / 4) * 3 < seconds_budget:                 # ~linear in the minibatch
        Bc *= 2
    legs = {"share16": leg(threads, Bc, "(a)", 3)}
    if avail > threads or Bc != 256:
        legs["survey_8d"] = leg(avail, 256 if nb * C >= 40 * 256 else Bc, "(b)", 1)
    best = max(legs.values(), key=lambda d: d["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "host_cores_total": os.cpu_count(),
            "host_cores_available_to_this_process": avail, "kind": "port", "legs": legs,
            "sample": f"oracle ppo_minibatch_step (fp32 CPU PyTorch restatement of the reference), se_resnet {nb}x{C}: the faster of "
                      f"(a) <= 16 threads, probe-sized minibatch and (b) every usable core at minibatch 256 (SURVEY 8d); here minibatch "
                      f"{best['minibatch']} on {best['threads']} threads, median of {best['steps']} step(s) (~{best['cpu_seconds']:.0f} s of CPU work)"}


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of a parent that has not touched
    the GPU (torch.distributed.run on 127.0.0.1 with a free port; the reference's launcher does the same with torchrun,
    run.sh:161-162,309-310), let rank 0's JSON line through on stdout, return the launcher's exit status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] spawning {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    # stdout carries exactly rank 0's JSON line: whatever else the ranks' libraries print there (gloo's "[Gloo] Rank 0 is
    # connected to ..." banners) is passed on to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        dest = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        dest.write(line); dest.flush()
    return proc.wait()


def run_transformer(args, device):
    """BASELINE configs[4]: transformer (d_model 256, nhead 8, 6 layers -- SURVEY 8d fixes the sizing, no reference config
    names one), minibatch 4096, one GPU: forward (bf16 autocast, train mode with the encoder's dropout 0.1) + clipped-
    surrogate policy loss over the legal actions + MSE value loss against returns (ScalarValueAdapter, value_adapter.py:43-
    59) + backward + fused clip / Adam.  The reference cannot train this model, so there is no training
    loop to mirror; the step is the same statement sequence as katago_ppo.py:849-933 with the scalar adapter."""
    from keisei_amd import _lib
    from keisei_amd.training.fused_optim import FusedAdamMixin
    from keisei_amd.training.model_registry import build_model

    d, H, L, B = 256, 8, 6, args.batch or 4096
    torch.manual_seed(1234)
    model = build_model("transformer", {"d_model": d, "nhead": H, "num_layers": L}).to(device)
    model.train()
    n_params = sum(p.numel() for p in model.parameters())
    total = 4 * B
    data = synth_dataset(total, 1234, device)
    returns = torch.randn(total, generator=torch.Generator().manual_seed(5)).clamp(-1, 1).to(device)

    class Step(FusedAdamMixin):
        def __init__(self):
            self.optimizer = torch.optim.Adam(model.parameters(), lr=2e-4)
            self._hip_state = {}
            assert self._fused_optimizer_ok()
            self.st = self._adam_tables(device)
            self.flags = torch.zeros(2, dtype=torch.int32, device=device)
            self.acc = torch.zeros(5, device=device)

        def __call__(self, idx):
            sp = _lib.stream_ptr(device)
            obs = data["obs"][idx]
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
                logits, value = model(obs)
            Bn, A = logits.shape
            dlogits = torch.empty_like(logits)
            new_lp = torch.empty(Bn, device=device); rowloss = torch.empty(Bn, device=device); rowent = torch.empty(Bn, device=device)
            _lib.call("ka_policy_loss", logits, data["masks"], data["actions"], data["old_lp"], data["adv"], idx, dlogits, new_lp,
                      rowloss, rowent, self.flags, None, 0.2, 1.0 / Bn, 0.01 / Bn, Bn, A, 0, sp)
            dv = (2.0 * 1.5 / Bn) * (value.detach() - returns[idx].unsqueeze(1))          # d(1.5 * MSE)/dv: (B, 1) scalars
            self.optimizer.zero_grad(set_to_none=True)
            torch.autograd.backward([logits, value], [dlogits, dv])
            st = self.st
            tab = self._upload_table(st, device)
            _lib.call("ka_clip_adam_step", tab, st["blk_t"], st["blk_o"], st["nblocks"], st["partial"], st["ctl"], st["step_dev"],
                      None, self.flags, self.acc[4:5], 1.0, 2e-4, 0.9, 0.999, 1e-8, sp)
            model._hip_engine.notify_weights_updated()

    step = Step()
    perm = torch.randperm(total, device=device)
    nmb = total // B
    print(f"[bench] transformer d={d} h={H} L={L}: {n_params / 1e6:.1f} M parameters, {args.warmup} warm-up + {args.steps} timed steps",
          file=sys.stderr, flush=True)
    for i in range(args.warmup):
        step(perm[(i % nmb) * B:(i % nmb + 1) * B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        j = (args.warmup + i) % nmb
        step(perm[j * B:(j + 1) * B])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    flags = step.flags.cpu().tolist()
    M = B * 81
    lin = 2.0 * M * (64 * d + L * (3 * d * d + d * d + 8 * d * d)) + 2.0 * B * 81 * d * 11259
    attn = L * 4.0 * B * H * 81 * 81 * (d // H)
    flop_step = 3.0 * (lin + attn)
    # the step's HBM bytes from the committed rocprofv3 PMC passes of this same command (tools/profile_round.sh), only when that
    # profile was taken on this build of transformer.hip
    hbm_step = None
    tfp = sorted((ROOT / "profiles").glob("r*_transformer_hbm_traffic.json"))
    if tfp and B == 4096 and args.dtype == "bf16":
        prof = json.loads(tfp[-1].read_text())
        if prof.get("transformer_source_sha16") == kernel_source_id(("common.h", "transformer.hip")) and prof.get("step_summary"):
            byt = prof["step_summary"]["hbm_bytes_per_step"]
            tbps = byt / (elapsed / args.steps) / 1e12
            hbm_step = {"bytes_per_step": byt, "achieved_TBps": round(tbps, 3), "frac_of_8TBps": round(tbps / 8.0, 4), "source": tfp[-1].name}
    out = {"metric": "PPO samples/sec, transformer d256 h8 L6 on 50x9x9", "value": round(B * args.steps / elapsed, 1), "unit": "samples/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"transformer d_model {d}, nhead {H}, {L} layers ({n_params / 1e6:.0f} M parameters, 233 M of them "
                                  f"the 81*d -> 11259 policy layer), minibatch {B}, clip loss + value MSE + backward + clip + Adam, "
                                  "train mode (dropout 0.1)", "per_gpu_batch": B, "global_batch": B, "parallelism": "dp1"},
           "roofline": {"bound": "mfma", "kernel": "whole step (linear layers + attention on the matrix cores)",
                        "achieved": round(flop_step / (elapsed / args.steps) / 1e12, 1),
                        "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flop_step / (elapsed / args.steps) / 1e12 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 4),
                        "traffic": None, "flop_per_step": flop_step, "hbm_step": hbm_step},
           "guard_flags": flags, "cpu_baseline": None}
    return out


def run_workload(args, dtype, steps, warmup, device, rank, world, dev_index, events_steps):
    """Build the model + synthetic epoch, run `warmup` untimed and `steps` timed minibatch steps of the product path
    (barrier + synchronize on both sides, MAX over ranks), then -- OUTSIDE the timed region -- `events_steps` more steps
    with a HIP event pair around every conv3x3 / wgrad launch on its launch stream (per-launch durations for the
    roofline; VERDICT r1: the timed region itself carries no event records)."""
    from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams
    from keisei_amd.training.model_registry import build_model
    from keisei_amd.training.value_adapter import MultiHeadValueAdapter

    nb, C, Rr, G, P, V, S, T, N, B = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(1234 + rank)
    model = build_model("se_resnet", dict(num_blocks=nb, channels=C, se_reduction=Rr, global_pool_channels=G,
                                          policy_channels=P, value_fc_size=V, score_fc_size=S, obs_channels=50))
    model.to(device)
    n_params = sum(p.numel() for p in model.parameters())
    fwd_model = model
    if world > 1 or dist.is_initialized():
        fwd_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        fwd_model = torch.nn.parallel.DistributedDataParallel(fwd_model, device_ids=[dev_index], gradient_as_bucket_view=True)
        model = fwd_model.module
    pp = KataGoPPOParams(batch_size=B, use_amp=(dtype == "bf16"), lambda_score=0.1, score_blend_alpha=0.1,
                         compile_mode="default")        # keisei-katago.toml:33-49 (compile_mode accepted, unused)
    algo = KataGoPPOAlgorithm(pp, model, forward_model=fwd_model)
    adapter = MultiHeadValueAdapter(pp.lambda_value, pp.lambda_score, pp.score_blend_alpha)
    assert algo._fused_path_available(device, adapter), "fused HIP path unavailable"
    total = T * N
    data = synth_dataset(total, 1234 + rank, device)
    fs = algo._fused_begin(data, device, adapter)
    fwd_model.train()
    perm = torch.randperm(total, device=device)
    nmb = max(1, total // B)

    def one_step(i):
        lo = (i % nmb) * B
        algo._fused_step(fs, perm[lo:lo + B], device)

    print(f"[bench] rank {rank}: {dtype} model + dataset ready, {warmup} warm-up + {steps} timed steps", file=sys.stderr, flush=True)
    for i in range(warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if os.environ.get("KA_HOST_TIMING"):
        fs["host_ms"] = {}
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(warmup + i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if fs.get("host_ms") is not None:
        print(f"[bench] host enqueue ms/step: total {1e3 * t_enq / steps:.1f} " +
              " ".join(f"{k} {v / steps:.1f}" for k, v in fs["host_ms"].items()), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # ---- per-launch durations: a separate pass, after the timed region
    engine = model._hip_engine
    events, ev_elapsed = None, None
    if events_steps > 0:
        engine.kernel_events = {"conv3x3": [], "wgrad": [], "conv3x3_fwd": []}
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(events_steps):
            one_step(warmup + steps + i)
        torch.cuda.synchronize()
        ev_elapsed = time.perf_counter() - t1
        events, engine.kernel_events = engine.kernel_events, None
    metrics = algo._fused_end(fs)
    return {"elapsed": elapsed, "metrics": metrics, "events": events, "events_elapsed": ev_elapsed, "events_steps": events_steps,
            "B": B, "n_params": n_params, "algo": algo, "adapter": adapter, "model": model, "shape": (nb, C, Rr, G, P, V, S, T, N),
            "fs": fs}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="40x256", choices=sorted(WORKLOADS) + ["transformer"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="override the minibatch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whole-update", action="store_true",
                    help="also time one KataGoPPOAlgorithm.update() from a host rollout buffer (GAE, H2D, gather included)")
    ap.add_argument("--sl-epoch", action="store_true", help="also time one SLTrainer.train_epoch() over synthetic shards")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary fp32-mode (parity numerics) run of the same workload")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block of the default line (whole update(), the 6x128 workload, the transformer workload)")
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="initialise the process group (RCCL when the backend is nccl) even for one rank, wrap the model as the "
                         "N > 1 bench does, run ONE step with its collectives, print the collective counts and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))          # the parent never initialises the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: pass the same N to torch.distributed.run and to bench.py")
    backend = os.environ.get("KA_BENCH_BACKEND", "nccl")     # "gloo": rehearse the N > 1 wiring with several ranks on ONE GPU
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if args.dist_dry_run:
        os.environ.setdefault("KA_FORCE_COLLECTIVES", "1")     # world size 1: issue every collective of the N > 1 step anyway
    if world > 1 or args.dist_dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))      # (only reached without a launcher: --dist-dry-run at world 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload == "transformer":
        if world > 1:
            raise SystemExit("the transformer workload is a single-GPU bench line (BASELINE configs[4])")
        print(json.dumps(run_transformer(args, device)), flush=True)
        return
    if args.dist_dry_run:
        counts = {"syncbn": 0, "gradient": 0, "other": 0}
        real = dist.all_reduce

        def counting(t, *a, **k):
            kind = "syncbn" if t.dtype == torch.float64 else ("gradient" if t.dtype == torch.float32 and t.numel() > 1 else "other")
            counts[kind] += 1
            return real(t, *a, **k)

        res = run_workload(args, args.dtype, 0, 1, device, rank, world, dev_index, 0)       # allocator / first-launch warm-up
        dist.all_reduce = counting
        res["algo"]._fused_step(res["fs"], torch.arange(res["B"], device=device), device)
        dist.all_reduce = real
        torch.cuda.synchronize()
        # what the collectives of the N > 1 step cost on ONE rank (RCCL launch + stream hand-offs; no wire time): a few
        # timed steps with them, to compare with the plain single-GPU line
        def timed5():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5):
                res["algo"]._fused_step(res["fs"], torch.arange(res["B"], device=device), device)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 5 * 1e3
        ms_with = timed5()
        # the same with the SyncBN all-reduces turned into no-ops (their packing / coefficient kernels still run): what is
        # left of the difference is the collective calls themselves
        dist.all_reduce = lambda t, *a, **k: None if t.dtype == torch.float64 else real(t, *a, **k)
        ms_nosync = timed5()
        dist.all_reduce = real
        # how long the compute stream STALLS on each SyncBN statistics all-reduce: the collectives are issued asynchronously
        # (the process group's stream) and waited for right before the coefficient kernel that reads them, with the independent
        # global-pool chain launched in between where there is one; an event pair around every wait, two steps
        engine = res["model"]._hip_engine
        engine.sync_events = []
        for i in range(2):
            res["algo"]._fused_step(res["fs"], torch.arange(res["B"], device=device), device)
        torch.cuda.synchronize()
        pairs, engine.sync_events = engine.sync_events, None
        stalls = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
        exposed = {"waits_per_step": len(stalls) // 2, "exposed_us_per_step": round(sum(stalls) / 2, 1),
                   "median_us": round(stalls[len(stalls) // 2], 2) if stalls else None,
                   "max_us": round(stalls[-1], 2) if stalls else None,
                   "note": "stall of the compute stream per wait (event pair around work.wait()), world size 1: launch + stream hand-off, no wire time"}
        m = res["algo"]._fused_end(res["fs"])
        if rank == 0:
            print(json.dumps({"dist_dry_run": True, "backend": backend, "world": world, "collectives_per_step": counts,
                              "overlapped_gradient_exchange": res["fs"]["reducer"] is not None,
                              "ms_per_step_with_collectives_on_one_rank": round(ms_with, 2),
                              "ms_per_step_with_syncbn_allreduce_as_noop": round(ms_nosync, 2),
                              "syncbn_allreduce_exposure": exposed,
                              "train_metrics": {k: round(v, 5) for k, v in m.items()}}), flush=True)
        dist.destroy_process_group()
        return

    ev_steps = 0 if args.no_kernel_events else min(args.steps, 4)
    res = run_workload(args, args.dtype, args.steps, args.warmup, device, rank, world, dev_index, ev_steps)
    elapsed, metrics, B, n_params = res["elapsed"], res["metrics"], res["B"], res["n_params"]
    nb, C, Rr, G, P, V, S, T, N = res["shape"]
    total = T * N
    whole = None
    if args.whole_update and rank == 0:
        whole = whole_update_rate(res["algo"], res["adapter"], T, N, device)
    want_secondary = world == 1 and rank == 0 and args.workload == "40x256" and args.dtype == "bf16" and not args.no_secondary
    if want_secondary and whole is None:
        whole = whole_update_rate(res["algo"], res["adapter"], T, N, device, stores=("device",))
    sl = None
    if args.sl_epoch and rank == 0:
        from keisei_amd.training.model_registry import build_model
        sl = sl_epoch_rate(lambda: build_model("se_resnet", dict(num_blocks=nb, channels=C, se_reduction=Rr, global_pool_channels=G,
                                                                   policy_channels=P, value_fc_size=V, score_fc_size=S,
                                                                   obs_channels=50)).to(device), 4 * B, B, args.dtype == "bf16")
    # secondary: the same workload in the fp32 (parity numerics) mode, SURVEY 8d config #3 "use_amp=true AND an fp32 run"
    fp32 = None
    if world == 1 and args.dtype == "bf16" and not args.no_fp32:
        del res["algo"], res["fs"], res["model"]
        torch.cuda.empty_cache()
        r32 = run_workload(args, "f32", 3, 1, device, rank, world, dev_index, 0 if args.no_kernel_events else 1)
        conv_flop = 2.0 * B * 81 * 9 * C * C
        fp32 = {"samples_per_s": round(B * 3 / r32["elapsed"], 1), "ms_per_step": round(1e3 * r32["elapsed"] / 3, 2), "steps": 3,
                "warmup": 1, "numerics": "fp32 activations, exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the mode the fp32 parity tests run"}
        if r32["events"] and r32["events"]["conv3x3"]:
            ms = [a.elapsed_time(b) for a, b in r32["events"]["conv3x3"]]
            avg = sum(ms) / len(ms)
            fp32["conv3x3_avg_launch_ms"] = round(avg, 3)
            fp32["conv_frac_of_157TF"] = round(conv_flop / (avg * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)
        del r32

    # SURVEY 8d "report both" + BASELINE configs[1] / configs[4], where the driver's default run sees them: a bounded leg each
    # (<= 3 timed steps), after the timed region, one GPU, rank 0 -- the full-length versions are the --whole-update /
    # --workload 6x128 / --workload transformer lines
    secondary = None
    if want_secondary:
        secondary = {}
        if whole is not None:
            secondary["whole_update"] = {"samples_per_s": whole["samples_per_s"], "update_seconds": whole["device"]["update_seconds"],
                                         "transitions": whole["transitions"], "epochs_per_batch": whole["epochs_per_batch"],
                                         "store": "device", "includes": whole["includes"]}
        for k in ("algo", "fs", "model", "adapter"):
            res.pop(k, None)
        torch.cuda.empty_cache()
        sub = argparse.Namespace(**vars(args))
        sub.workload, sub.batch = "6x128", 0
        n6, w6 = 20, 5      # (a 4.5 ms step: three of them are a noisy number -- 357 k to 451 k samples/s on one build; twenty are 0.1 s)
        r6 = run_workload(sub, "bf16", n6, w6, device, rank, world, dev_index, 1)
        B6 = r6["B"]
        line6 = {"samples_per_s": round(B6 * n6 / r6["elapsed"], 1), "ms_per_step": round(1e3 * r6["elapsed"] / n6, 3), "steps": n6, "warmup": w6,
                 "workload": f"se_resnet 6x128 KataGo-PPO minibatch step, minibatch {B6} (BASELINE configs[1])", "dtype": "bf16"}
        if r6["events"] and r6["events"]["conv3x3"]:
            ms6 = [a.elapsed_time(b) for a, b in r6["events"]["conv3x3"]]
            f6 = 2.0 * B6 * 81 * 9 * 128 * 128
            line6["roofline"] = {"bound": "mfma", "kernel": "tower conv3x3 launches (forward + data gradient)", "avg_launch_ms": round(sum(ms6) / len(ms6), 4),
                                 "achieved": round(f6 / (sum(ms6) / len(ms6) * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(f6 / (sum(ms6) / len(ms6) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
        sh6 = r6["shape"]
        fl6 = train_flops_per_sample(sh6[0], sh6[1], sh6[3], sh6[1] // sh6[2], sh6[4], sh6[5], sh6[6])
        line6["model_tflops_frac"] = round(line6["samples_per_s"] * fl6 / 1e12 / PEAK_BF16_TFLOPS, 4)
        secondary["workload_6x128"] = line6
        r6["algo"]._fused_end(r6["fs"])
        del r6
        torch.cuda.empty_cache()
        sub = argparse.Namespace(**vars(args))
        sub.steps, sub.warmup, sub.batch = 3, 2, 0
        tl = run_transformer(sub, device)
        secondary["transformer"] = {"samples_per_s": tl["value"], "ms_per_step": tl["ms_per_step"], "steps": 3, "warmup": 2,
                                    "workload": tl["config"]["workload"], "dtype": tl["dtype"],
                                    "roofline": {k: tl["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}}
        torch.cuda.empty_cache()

    n_ranks_seen = 1
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, torch.cuda.current_device()))
        n_ranks_seen = len({r for r, _ in seen})
        dist.barrier()
        dist.destroy_process_group()          # the other ranks are done; rank 0 goes on to the CPU baseline alone
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        act_bytes = 2 if args.dtype == "bf16" else 4
        bps = algorithmic_bytes_per_sample(nb, C, P, n_params, B, act_bytes)
        bps32 = algorithmic_bytes_per_sample(nb, C, P, n_params, B, 4)
        flops = train_flops_per_sample(nb, C, G, C // Rr, P, V, S)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        roof = None
        extra = {}
        ev = res["events"]
        if ev and ev["conv3x3"]:
            conv_ms = [a.elapsed_time(b) for a, b in ev["conv3x3"]]
            conv_flop = 2.0 * B * 81 * 9 * C * C
            avg = sum(conv_ms) / len(conv_ms)
            ach = conv_flop / (avg * 1e-3) / 1e12
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload -- only when that profile
            # was taken on THIS build of the kernels (source hash recorded by tools/pmc_traffic.py); otherwise null
            traffic, traffic_note = None, None
            pmcs = sorted((ROOT / "profiles").glob("r*_pmc_hbm_traffic.json"))
            pmc = pmcs[-1] if pmcs else None
            if pmc is not None and args.workload == "40x256" and args.dtype == "bf16" and B == 4096:
                prof = json.loads(pmc.read_text())
                if prof.get("kernel_source_sha16") == kernel_source_id():
                    # the tower's launches (conv3x3_pc2_kernel: forward and data gradients; older forms when a switch selects them)
                    # -- launch-weighted mean, like the timed launches
                    recs = [v for k, v in prof["kernels"].items() if k.startswith("conv3x3_kernel<bf16_t") and v["launches"] >= 10]
                    recs += [v for k, v in prof["kernels"].items() if k.startswith(("conv3x3_pc_kernel", "conv3x3_pc2_kernel"))]
                    n = sum(v["launches"] for v in recs)
                    traffic = round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in recs) / n) if n else None
                    corner = prof["kernels"].get("conv3x3_corner_kernel")      # (square 80 as a launch of its own: KA_CONV_CORNER_IN=0, older profiles)
                    if traffic is not None and corner:
                        traffic += corner["hbm_bytes_per_launch"]
                    traffic_note = f"{pmc.name} (same kernel sources)"
                else:
                    traffic_note = (f"{pmc.name} was collected on kernel sources {prof.get('kernel_source_sha16')}, this build is "
                                    f"{kernel_source_id()}: not reported")
            ev_ms_step = 1e3 * res["events_elapsed"] / res["events_steps"]
            two_streams = os.environ.get("KA_WGRAD_OVERLAP", "0") != "0"
            roof = {"bound": "mfma", "kernel": "conv3x3_pc2_kernel (implicit-GEMM 3x3 conv, two boards per weight fragment: squares 0..79 of both as ten row "
                                               "tiles, square 80 of the workgroup's 16 boards as one more tile inside the kernel -- by conv3x3_corner_kernel behind "
                                               "the masked form; forward and data-gradient launches, the "
                                               "latter with the fused BatchNorm-backward input and -- behind conv2 -- the masked epilogue" + (", concurrent with wgrad on a 2nd stream)" if two_streams else ")"),
                    "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_note, "launches_timed": len(conv_ms), "avg_launch_ms": round(avg, 4),
                    "flop_per_launch": conv_flop,
                    "launch_timing": f"HIP event pair per launch on its launch stream, {res['events_steps']} step(s) run right after "
                                     f"the timed region ({ev_ms_step:.2f} ms/step with the events vs {ms_step:.2f} without)"}
            fwd_ms = [a.elapsed_time(b) for a, b in ev.get("conv3x3_fwd", [])]
            if fwd_ms:
                favg = sum(fwd_ms) / len(fwd_ms)
                extra["conv3x3_forward_launches_only"] = {"avg_launch_ms": round(favg, 4), "launches_timed": len(fwd_ms),
                                                          "achieved_tflops": round(conv_flop / (favg * 1e-3) / 1e12, 1)}
            w_ms = [a.elapsed_time(b) for a, b in ev["wgrad"]]
            if w_ms:
                wavg = sum(w_ms) / len(w_ms)
                extra["wgrad_kernel"] = {"avg_launch_ms": round(wavg, 4), "achieved_tflops": round(conv_flop / (wavg * 1e-3) / 1e12, 1),
                                         "launches_timed": len(w_ms)}
            # all MFMA launches (conv + wgrad, which overlap on two streams) over the whole timed region, every phase included
            n_mfma = (len(conv_ms) + len(w_ms)) // res["events_steps"]
            extra["mfma_step_average"] = {"launches_per_step": n_mfma, "achieved_tflops": round(n_mfma * conv_flop / (elapsed / args.steps) / 1e12, 1),
                                          "frac": round(n_mfma * conv_flop / (elapsed / args.steps) / 1e12 / peak, 4)}
        out = {
            "metric": "PPO samples/sec, se_resnet 40x256 on 50x9x9" if args.workload == "40x256" else f"PPO samples/sec, se_resnet {args.workload}",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "n_ranks_seen": n_ranks_seen, "backend": backend if world > 1 else None,
            "config": {"workload": f"se_resnet {nb}x{C} KataGo-PPO minibatch step (gather+fwd+loss+bwd+clip+Adam), "
                                   f"minibatch {B}/GPU from a {total}-sample device-resident epoch dataset, obs 50x9x9",
                       "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world}" + ("+syncbn" if world > 1 else "")},
            "roofline": roof,
            "hbm_roofline": {"algorithmic_bytes_per_sample": round(bps), "activation_storage_bytes": act_bytes,
                             "achieved_GBps": round(value / world * bps / 1e9, 1), "peak_GBps": PEAK_HBM_GBS,
                             "frac": round(value / world * bps / 1e9 / PEAK_HBM_GBS, 4),
                             "frac_on_fp32_contract_bytes": round(value / world * bps32 / 1e9 / PEAK_HBM_GBS, 4),
                             "fp32_contract_bytes_per_sample": round(bps32)},
            "model_tflops": {"train_flop_per_sample": flops, "achieved": round(value / world * flops / 1e12, 1), "peak": peak,
                             "frac": round(value / world * flops / 1e12 / peak, 4)},
            "train_metrics": {k: round(v, 5) for k, v in metrics.items()},
            "kernel_source_sha16": kernel_source_id(),
            # every run-time switch of the library / engine set in this process's environment (none in a default run)
            "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("KA_", "KEISEI_AMD")) and k != "KA_CHECK_ARGS"},
        }
        out.update(extra)
        if fp32 is not None:
            out["fp32_mode"] = fp32
        if whole is not None and args.whole_update:
            out["whole_update"] = whole
        if secondary is not None:
            out["secondary"] = secondary
        if sl is not None:
            out["sl_epoch"] = sl
        # (N > 1: taken after the timed region and after the process group is gone -- the other ranks have exited)
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline((nb, C, Rr, G, P, V, S))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
