"""GPU parity of the whole SE-ResNet (HIP engine, through the nn.Module API) against golden
vectors produced by the reference implementation, in the fp32 (exact-MFMA) mode with
rtol/atol 1e-4/5e-5-class bounds and in the bf16 mode with the stated bf16 bound."""
import pytest
import torch

from keisei_amd.training.models.se_resnet import GlobalPoolBiasBlock, SEResNetModel, SEResNetParams
from oracle import keisei_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def tiny_model(sd):
    p = SEResNetParams(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16,
                       policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)
    m = SEResNetModel(p)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def freeze_bn(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0


@pytest.mark.parametrize("tag", ["randn.", "board."])
def test_tiny_model_fp32_forward_backward(golden, tag):
    g = golden("g2_model_tiny")
    m = tiny_model(g.sub("sd."))
    obs = g[tag + "obs"].to(DEV)
    m.eval()
    with torch.no_grad():
        o = m(obs)
    assert o.policy_logits.shape == (4, 9, 9, 139) and o.value_logits.shape == (4, 3) and o.score_lead.shape == (4, 1)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "eval.policy"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(o.value_logits.cpu(), g[tag + "eval.value"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(o.score_lead.cpu(), g[tag + "eval.score"], rtol=1e-4, atol=2e-5)
    m.train(); freeze_bn(m)
    o = m(obs)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "train.policy"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.value_logits.cpu(), g[tag + "train.value"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.score_lead.cpu(), g[tag + "train.score"], rtol=1e-4, atol=5e-5)
    loss = ((o.policy_logits * g[tag + "cot.policy"].to(DEV)).sum() / 4 + (o.value_logits * g[tag + "cot.value"].to(DEV)).sum()
            + (o.score_lead * g[tag + "cot.score"].to(DEV)).sum())
    loss.backward()
    worst = 0.0
    for n, p in m.named_parameters():
        ref = g[f"{tag}grad.{n}"]
        assert p.grad is not None, n
        err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-6)
        worst = max(worst, err)
        assert err < 2e-3, (n, err)
    print("worst relative grad error", worst)


def test_running_stats_update_like_reference(golden):
    g = golden("g2_model_tiny")
    sd = g.sub("sd.")
    m = tiny_model(sd)
    obs = g["randn.obs"]
    m.train()
    with torch.no_grad():
        m(obs.to(DEV))
    ref = dict(sd)
    orc.seresnet_forward(ref, obs, 2, train=True, momentum=0.1, update_running=True)
    got = m.state_dict()
    for k in ref:
        if "running_" in k:
            assert torch.allclose(got[k].cpu(), ref[k], rtol=1e-4, atol=1e-5), k
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(ref[k]) == 1, k


def _cots(B):
    return tuple(t.to(DEV) for t in orc.closed_form_cotangents(B))


def _backward(o, B):
    cp, cv, cs = _cots(B)
    ((o.policy_logits * cp).sum() / B + (o.value_logits * cv).sum() + (o.score_lead * cs).sum()).backward()


def _grad_errors(g, tag, m):
    """(worst relative norm error, worst relative L2 over the stored tensors / slices, their median) against the
    reference's fp64 gradients."""
    names = list(g.np(tag + "grad_names"))
    norms = dict(zip(names, g.np(tag + "grad_norms64")))
    grads = dict((n, p.grad) for n, p in m.named_parameters())
    worst_n, l2 = 0.0, []
    for n in names:
        assert grads[n] is not None, n
        worst_n = max(worst_n, abs(float(grads[n].double().norm()) - norms[n]) / (norms[n] + 1e-30))
        for key, got in ((f"{tag}grad64.{n}", grads[n]), (f"{tag}grad64.{n}[:2]", grads[n][:2])):
            if key in g:
                ref = g[key].double()
                l2.append(float((got.double().cpu() - ref).norm() / (ref.norm() + 1e-30)))
    l2.sort()
    return worst_n, l2[-1], l2[len(l2) // 2]


def _build(shape, amp):
    m = SEResNetModel(SEResNetParams(**shape.__dict__))
    m.load_state_dict(orc.init_like_state_dict(shape), strict=True)
    m.to(DEV)
    if amp:
        m.configure_amp(True, torch.bfloat16, "cuda")
    freeze_bn(m)
    return m


# fp32 mode against the reference (g2_model_mid16: 16 boards = 8 randn + 8 board-like, init-like closed-form weights,
# batch picked among 200 seeds so that no ReLU input lies within 2.4e-6 of zero -- the reference's own fp32 gradients
# are then 1.2e-6 from its fp64 ones, i.e. no mask element sits on a knife edge).  Gradients are compared with the fp64
# run of the reference: every tensor's norm, every tensor of <= 20 000 elements in full, two output channels of every
# 3x3 convolution.  Measured on MI355X: norms 9.4e-7 / 6.0e-7, relative L2 worst 2.3e-6 / 2.9e-6 (6x128 / 3x256).
MID_GRAD_TOL = 5e-5


@pytest.mark.parametrize("tag,shape", [("s6x128.", orc.NetShape(6, 128)), ("s3x256.", orc.NetShape(3, 256))])
def test_mid_models_fp32(golden, tag, shape):
    g = golden("g2_model_mid16")
    m = _build(shape, False)
    obs = g[tag + "obs"].to(DEV)
    B = obs.shape[0]
    m.eval()
    with torch.no_grad():
        o = m(obs)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "eval.policy"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.value_logits.cpu(), g[tag + "eval.value"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.score_lead.cpu(), g[tag + "eval.score"], rtol=1e-4, atol=5e-5)
    m.train()
    o = m(obs)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "train.policy"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.value_logits.cpu(), g[tag + "train.value"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.score_lead.cpu(), g[tag + "train.score"], rtol=1e-4, atol=5e-5)
    _backward(o, B)
    worst_n, worst_l2, med_l2 = _grad_errors(g, tag, m)
    print(f"{tag} fp32 gradients vs reference fp64: norm {worst_n:.2e}, rel L2 worst {worst_l2:.2e} median {med_l2:.2e}")
    assert worst_n <= MID_GRAD_TOL and worst_l2 <= MID_GRAD_TOL, (worst_n, worst_l2)


def test_headline_model_fp32(golden):
    """se_resnet 40x256 (BASELINE configs[2], keisei-katago.toml:15-23) in the fp32 mode against the reference on 16
    boards: outputs at the stated rtol 1e-4 / atol 5e-5 (the reference's own fp32 forward is 7.2e-6 from its fp64
    forward), gradients of all 576 tensors against the reference's fp64 run.  27 M ReLU inputs: a few lie within fp32
    rounding of zero, two fp32 implementations take different sides there, and the reference's own fp32 gradients are
    up to 8.2e-3 (median 2.3e-3) from its fp64 gradients (recorded per tensor in the fixture).  The HIP path is held
    to that yardstick: worst <= 2.5 x the reference's worst, median <= 2.5 x its median, norms within 1 %."""
    g = golden("g2_model_full")
    shape = orc.NetShape()
    m = _build(shape, False)
    obs = g["obs"].to(DEV)
    B = obs.shape[0]
    m.eval()
    with torch.no_grad():
        o = m(obs)
    for got, key in ((o.policy_logits, "eval.policy"), (o.value_logits, "eval.value"), (o.score_lead, "eval.score")):
        assert torch.allclose(got.cpu(), g[key], rtol=1e-4, atol=5e-5), key
    m.train()
    o = m(obs)
    for got, key in ((o.policy_logits, "train.policy"), (o.value_logits, "train.value"), (o.score_lead, "train.score")):
        d = float((got.detach().cpu() - g[key]).abs().max())
        print(f"40x256 fp32 {key}: max abs diff {d:.2e} (|ref|max {float(g[key].abs().max()):.3f})")
        assert torch.allclose(got.detach().cpu(), g[key], rtol=1e-4, atol=5e-5), key
    _backward(o, B)
    worst_n, worst_l2, med_l2 = _grad_errors(g, "", m)
    ref = g.np("grad_rel32v64")
    print(f"40x256 fp32 gradients vs reference fp64: norm {worst_n:.2e}, rel L2 worst {worst_l2:.2e} median {med_l2:.2e} "
          f"(reference fp32 vs fp64: worst {ref.max():.2e} median {float(sorted(ref)[len(ref) // 2]):.2e})")
    assert worst_n <= 1e-2
    assert worst_l2 <= 2.5 * float(ref.max()) and med_l2 <= 2.5 * float(sorted(ref)[len(ref) // 2])


# bf16 mode (the throughput mode the bench runs): bf16 activations and conv operands, fp32 accumulation / statistics /
# FC layers.  The comparison is against the reference's fp32 outputs and fp64 gradients, so it includes the inherent
# distance of bf16 storage.  Yardstick: the reference's OWN bf16 mode (CPU autocast, katago_ppo.py:23-24), recorded in
# the fixtures -- distance from its fp64 run (policy eval / train as a fraction of |logit|max; gradient relative L2
# worst / median; worst norm ratio):
#     6x128   0.002 / 0.021;  0.295 / 0.152;  0.057        this build, measured on MI355X:  0.0056 / 0.0169;  0.369 / 0.146;  0.169
#     3x256   0.001 / 0.011;  0.166 / 0.121;  0.049                                        0.0036 / 0.0099;  0.170 / 0.104;  0.050
#     40x256  0.005 / 0.037;  0.682 / 0.300;  0.195                                        0.0123 / 0.0387;  0.468 / 0.282;  0.208
# i.e. the same class in train mode and for gradients; in eval mode 2.5-3x further (the residual stream x is STORED in
# bf16 here, the reference keeps it in fp32 and only casts conv inputs).  The bounds below are those measurements,
# frozen with ~1.4x headroom (the kernels are deterministic; the headroom is for future kernel changes).
BF16_BOUNDS = {
    "s6x128.": dict(eval=0.009, train=0.030, ratio=0.25, l2=0.50, med=0.20),
    "s3x256.": dict(eval=0.006, train=0.020, ratio=0.10, l2=0.25, med=0.15),
    "": dict(eval=0.020, train=0.060, ratio=0.30, l2=0.65, med=0.38),
}


@pytest.mark.parametrize("tag,shape,fixture", [("s6x128.", orc.NetShape(6, 128), "g2_model_mid16"),
                                               ("s3x256.", orc.NetShape(3, 256), "g2_model_mid16"),
                                               ("", orc.NetShape(), "g2_model_full")])
def test_models_bf16_bound(golden, tag, shape, fixture):
    g = golden(fixture)
    bound = BF16_BOUNDS[tag]
    m = _build(shape, True)
    obs = g[tag + "obs"].to(DEV)
    B = obs.shape[0]
    for train in (False, True):
        m.train(train)
        with torch.no_grad():
            o = m(obs)
        ref = g[tag + ("train.policy" if train else "eval.policy")]
        e = float((o.policy_logits.float().cpu() - ref).abs().max()) / float(ref.abs().max())
        refv = g[tag + ("train.value" if train else "eval.value")]
        ev = float((o.value_logits.float().cpu() - refv).abs().max())
        print(f"{tag or '40x256.'} bf16 train={train}: policy max diff / |logit|max {e:.4f}, value logits max diff {ev:.4f}")
        assert e <= bound["train" if train else "eval"]
    m.train()
    o = m(obs)
    _backward(o, B)
    worst_n, worst_l2, med_l2 = _grad_errors(g, tag, m)
    print(f"{tag or '40x256.'} bf16 gradients vs reference fp64: norm ratio off by {worst_n:.3f}, rel L2 worst {worst_l2:.3f} "
          f"median {med_l2:.3f}")
    assert worst_n <= bound["ratio"] and worst_l2 <= bound["l2"] and med_l2 <= bound["med"]


def test_standalone_block_matches_reference(golden):
    g = golden("g1_block")
    blk = GlobalPoolBiasBlock(32, 8, 16)
    blk.load_state_dict(g.sub("sd."))
    blk.to(DEV)
    x = g["x"].to(DEV)
    blk.eval()
    with torch.no_grad():
        out = blk(x)
    assert torch.allclose(out.cpu(), g["out_eval"], rtol=1e-4, atol=2e-5)
    blk.train()
    for bn in (blk.bn1, blk.bn2):
        bn.momentum = 0.0
    with torch.no_grad():
        out = blk(x)
    assert torch.allclose(out.cpu(), g["out_train"], rtol=1e-4, atol=5e-5)


def test_bad_obs_shape_raises():
    m = SEResNetModel(SEResNetParams(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16,
                                     policy_channels=8, value_fc_size=32, score_fc_size=16)).to(DEV)
    with pytest.raises(ValueError, match=r"Expected obs shape \(batch, 50, 9, 9\)"):
        m(torch.zeros(2, 46, 9, 9, device=DEV))


def test_bf16_backward_tracks_fp32_gradients():
    """bf16 mode (fused BN-backward convs, side-stream wgrad) vs the fp32 oracle gradients on default-initialised
    6x128 weights, B=64: every gradient tensor keeps cosine > 0.95 and a norm ratio within [0.8, 1.25].  This is the
    class of the reference's own bf16 autocast: its CPU bf16-autocast backward measures cosine >= 0.968 against its fp32
    backward on the same setup (worst tensors: global_fc.0 / se_fc1, which sit behind max/std pooling)."""
    shape = orc.NetShape(6, 128)
    torch.manual_seed(1)
    m = SEResNetModel(SEResNetParams(**shape.__dict__))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    B = 64
    g = torch.Generator().manual_seed(5)
    obs = torch.randn(B, 50, 9, 9, generator=g)
    cp, cv, cs = torch.randn(B, 9, 9, 139, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, 1, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    live = dict(sd); live.update(leaves)
    p, v, s = orc.seresnet_forward(live, obs, shape.num_blocks, train=True, momentum=0.0)
    ((p * cp).sum() / B + (v * cv).sum() + (s * cs).sum()).backward()
    m.to(DEV).train()
    m.configure_amp(True, torch.bfloat16, "cuda")
    freeze_bn(m)
    o = m(obs.to(DEV))
    ((o.policy_logits * cp.to(DEV)).sum() / B + (o.value_logits * cv.to(DEV)).sum() + (o.score_lead * cs.to(DEV)).sum()).backward()
    worst = 1.0
    for n, prm in m.named_parameters():
        ref, got = leaves[n].grad.flatten().double(), prm.grad.flatten().double().cpu()
        if float(ref.norm()) == 0:
            continue
        cos = float((ref * got).sum() / (ref.norm() * got.norm() + 1e-30))
        ratio = float(got.norm() / ref.norm())
        worst = min(worst, cos)
        assert cos > 0.95 and 0.8 < ratio < 1.25, (n, cos, ratio)
    print("worst cosine", worst)


def test_chain_without_dz_is_no_further_from_fp32_than_the_dz_form(monkeypatch):
    """At training batch sizes the block-boundary launches leave dz = du * gate + add unformed (ka_block_dx_tail_bwd_du_gate) and
    the conv2 data gradient builds it inside its input transform (ka_conv3x3_dgrad_fused_gated): one bf16 rounding less on the
    way into BatchNorm's backward, so the gradients are not bit-identical to the dz form (KA_TAIL_GATE=0).  Both are measured
    against the fp32 oracle's gradients on a 3x128 tower at 512 boards (the smallest batch the two-board kernel takes): the
    default must be in the same class -- every tensor's relative L2 error at most 1.15x the dz form's (+1e-3), cosine > 0.95."""
    shape = orc.NetShape(3, 128)
    torch.manual_seed(2)
    ref_m = SEResNetModel(SEResNetParams(**shape.__dict__))
    sd = {k: v.clone() for k, v in ref_m.state_dict().items()}
    B = 512
    g = torch.Generator().manual_seed(9)
    obs = (torch.rand(B, 50, 9, 9, generator=g) < 0.15).float()
    cp, cv, cs = torch.randn(B, 9, 9, 139, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, 1, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    live = dict(sd); live.update(leaves)
    p, v, s = orc.seresnet_forward(live, obs, shape.num_blocks, train=True, momentum=0.0)
    ((p * cp).sum() / B + (v * cv).sum() + (s * cs).sum()).backward()
    errs = {}
    for gate in ("0", "1"):
        monkeypatch.setenv("KA_TAIL_GATE", gate)
        m = SEResNetModel(SEResNetParams(**shape.__dict__))
        m.load_state_dict(sd)
        m.to(DEV).train()
        m.configure_amp(True, torch.bfloat16, "cuda")
        freeze_bn(m)
        o = m(obs.to(DEV))
        ((o.policy_logits * cp.to(DEV)).sum() / B + (o.value_logits * cv.to(DEV)).sum() + (o.score_lead * cs.to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        e = {}
        for n, prm in m.named_parameters():
            ref, got = leaves[n].grad.flatten().double(), prm.grad.flatten().double().cpu()
            if float(ref.norm()) == 0:
                continue
            e[n] = (float((got - ref).norm() / ref.norm()), float((ref * got).sum() / (ref.norm() * got.norm() + 1e-30)), prm.grad.clone())
        errs[gate] = e
    monkeypatch.delenv("KA_TAIL_GATE")
    assert any(not torch.equal(errs["0"][n][2], errs["1"][n][2]) for n in errs["0"]), "the switch changed nothing: gate form not taken?"
    worst = 0.0
    for n in errs["0"]:
        (e0, c0, _), (e1, c1, _) = errs["0"][n], errs["1"][n]
        assert c1 > 0.95, (n, c1)
        assert e1 <= 1.15 * e0 + 1e-3, (n, e0, e1)
        worst = max(worst, e1 / max(e0, 1e-9))
    print("worst ratio gate/dz", worst, "median errors", sorted(x[0] for x in errs["0"].values())[len(errs["0"]) // 2],
          sorted(x[0] for x in errs["1"].values())[len(errs["1"]) // 2])


@pytest.mark.parametrize("amp", [False, True])
def test_backward_schedules_give_identical_gradients(monkeypatch, amp):
    """The backward's launch schedule is a choice, not arithmetic: weight gradients on the main stream (default) or on a second
    stream (KA_WGRAD_OVERLAP=1), block boundaries as one launch (default) or two (KA_DX_TAIL=0), the global-pool FC chains on the
    main stream (default) or forked beside the statistics kernels (KA_FC_SIDE=2, KA_FC_BWD_SIDE=1), BatchNorm statistics as two
    launches (default) or one (KA_BN_ONE_LAUNCH=1) -- every parameter gradient bit for bit the same."""
    shape = orc.NetShape(3, 64, 8, 32, 16, 64, 32)
    sd = orc.init_like_state_dict(shape)
    g = torch.Generator().manual_seed(11)
    B = 37
    obs = torch.randn(B, 50, 9, 9, generator=g).to(DEV)
    cp, cv, cs = torch.randn(B, 9, 9, 139, generator=g).to(DEV), torch.randn(B, 3, generator=g).to(DEV), torch.randn(B, 1, generator=g).to(DEV)
    grads = {}
    for name, env in (("default", {}), ("two streams", {"KA_WGRAD_OVERLAP": "1"}), ("two launches", {"KA_DX_TAIL": "0"}),
                      ("forked chains", {"KA_FC_SIDE": "2", "KA_FC_BWD_SIDE": "1"}), ("one-launch statistics", {"KA_BN_ONE_LAUNCH": "1"})):
        for k in ("KA_WGRAD_OVERLAP", "KA_DX_TAIL", "KA_FC_SIDE", "KA_FC_BWD_SIDE", "KA_BN_ONE_LAUNCH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = SEResNetModel(SEResNetParams(**shape.__dict__))              # (the engine reads its switches when it is built)
        m.load_state_dict(sd)
        m.to(DEV).train()
        if amp:
            m.configure_amp(True, torch.bfloat16, "cuda")
        o = m(obs)
        ((o.policy_logits * cp).sum() / B + (o.value_logits * cv).sum() + (o.score_lead * cs).sum()).backward()
        torch.cuda.synchronize()
        grads[name] = {n: prm.grad.clone() for n, prm in m.named_parameters()}
    for name in ("two streams", "two launches", "forked chains", "one-launch statistics"):
        for n, ref in grads["default"].items():
            assert torch.equal(grads[name][n], ref), (name, n)


@pytest.mark.parametrize("amp", [False, True])
def test_se_chain_inside_the_tail_launch(monkeypatch, amp):
    """KA_SE_IN_TAIL=1: the squeeze-excite FC chain of a block runs inside its forward-tail launch (ka_block_tail_fwd_se) instead of
    as a ka_fc_chain launch in front of it.  Same operands, plain fp32 FMAs in another order than the chain kernel's MFMAs:
    outputs and every parameter gradient agree to fp32 rounding (fp32 mode) / to the bf16 mode's own noise floor."""
    shape = orc.NetShape(3, 64, 8, 32, 16, 64, 32)
    sd = orc.init_like_state_dict(shape)
    g = torch.Generator().manual_seed(12)
    B = 37
    obs = torch.randn(B, 50, 9, 9, generator=g).to(DEV)
    cp, cv, cs = torch.randn(B, 9, 9, 139, generator=g).to(DEV), torch.randn(B, 3, generator=g).to(DEV), torch.randn(B, 1, generator=g).to(DEV)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("KA_SE_IN_TAIL", flag)
        m = SEResNetModel(SEResNetParams(**shape.__dict__))
        m.load_state_dict(sd)
        m.to(DEV).train()
        if amp:
            m.configure_amp(True, torch.bfloat16, "cuda")
        o = m(obs)
        ((o.policy_logits * cp).sum() / B + (o.value_logits * cv).sum() + (o.score_lead * cs).sum()).backward()
        torch.cuda.synchronize()
        runs[flag] = (o.policy_logits.detach().clone(), o.value_logits.detach().clone(), {n: prm.grad.clone() for n, prm in m.named_parameters()})
    monkeypatch.delenv("KA_SE_IN_TAIL")
    tol = 2e-2 if amp else 1e-4
    for a, b in ((runs["0"][0], runs["1"][0]), (runs["0"][1], runs["1"][1])):
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + 1e-6
    for n, ref in runs["0"][2].items():
        got = runs["1"][2][n]
        den = float(ref.norm())
        if den > 0:
            assert float((got - ref).norm()) / den <= (5e-2 if amp else 1e-4), n


@pytest.mark.parametrize("nb,B", [(3, 5), (40, 130)])
def test_eval_tower_kernel_matches_the_per_layer_path(monkeypatch, nb, B):
    """bf16 eval forward of a 256-channel model: the one-launch tower (csrc/tower.hip, the default in eval mode) against the
    per-layer launch sequence (KA_TOWER=0).  Same rounding points (bf16 conv outputs, bf16 block outputs), so the two agree
    to a few bf16 steps even after 40 blocks; both are held to the reference by test_models_bf16_bound."""
    shape = orc.NetShape(nb, 256)
    m = _build(shape, True).eval()
    g = torch.Generator().manual_seed(nb * 100 + B)
    obs = torch.randn(B, 50, 9, 9, generator=g).to(DEV)
    monkeypatch.setenv("KA_EVAL_GRAPH", "0")
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("KA_TOWER", flag)
        with torch.no_grad():
            o = m(obs)
        outs[flag] = (o.policy_logits.float().cpu(), o.value_logits.float().cpu(), o.score_lead.float().cpu())
    for a, b, name in zip(outs["0"], outs["1"], ("policy", "value", "score")):
        assert torch.isfinite(b).all(), name
        scale = float(a.abs().max())
        err = float((a - b).abs().max())
        print(f"tower vs per-layer, {nb} blocks, {name}: max diff {err:.3e} of |max| {scale:.3e}")
        assert err <= 0.02 * scale + 1e-3, (name, err, scale)


def test_eval_graph_matches_eager_and_tracks_weights(monkeypatch):
    """The graph-captured eval forward (rollout inference) equals the eager launch sequence, is deterministic on
    replay, and sees in-place weight / running-statistics updates made after the capture."""
    torch.manual_seed(3)
    m = SEResNetModel(SEResNetParams(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16,
                                     policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)).to(DEV).eval()
    obs = torch.randn(6, 50, 9, 9, device=DEV)

    def run(graph):
        monkeypatch.setenv("KA_EVAL_GRAPH", "1" if graph else "0")
        with torch.no_grad():
            o = m(obs)
        return o.policy_logits.clone(), o.value_logits.clone(), o.score_lead.clone()

    eager = run(False)
    first = run(True)        # capture
    replay = run(True)       # replay
    for a, b, c in zip(eager, first, replay):
        assert torch.equal(a, b) and torch.equal(a, c)
    with torch.no_grad():
        m.blocks[0].conv1.weight.mul_(1.25)
        m.input_bn.running_mean.add_(0.05)
        m.value_fc2.bias.add_(0.5)
    eager2 = run(False)
    replay2 = run(True)
    assert not torch.equal(eager2[0], eager[0])
    for a, b in zip(eager2, replay2):
        assert torch.equal(a, b)
    # a different batch size gets its own graph
    obs = torch.randn(3, 50, 9, 9, device=DEV)
    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


def test_eval_graphs_survive_alternating_dtypes(monkeypatch):
    """a captured eval graph reads the weight-pack / BatchNorm-coefficient buffers it was captured with.  A
    bf16 eval, then an fp32 eval (which builds its own pack set), then bf16 again must replay the first graph against
    LIVE buffers: packs are kept per (dtype, parameter storage), never freed while the engine lives."""
    torch.manual_seed(4)
    m = SEResNetModel(SEResNetParams(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16,
                                     policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)).to(DEV).eval()
    obs = torch.randn(5, 50, 9, 9, device=DEV)

    def run(bf16, graph):
        monkeypatch.setenv("KA_EVAL_GRAPH", "1" if graph else "0")
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            o = m(obs)
        return o.policy_logits.float().clone()

    eager16, eager32 = run(True, False), run(False, False)
    a = run(True, True)             # capture bf16
    b = run(False, True)            # capture fp32: a second pack set
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]   # recycle whatever the allocator had freed
    c = run(True, True)             # replay bf16
    d = run(False, True)            # replay fp32
    del junk
    assert torch.equal(a, eager16) and torch.equal(c, eager16)
    assert torch.equal(b, eager32) and torch.equal(d, eager32)
    with torch.no_grad():
        m.blocks[1].conv2.weight.mul_(0.5)
    assert torch.equal(run(True, True), run(True, False)) and torch.equal(run(False, True), run(False, False))


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("channels,se_red,gpc,pol", [(96, 8, 24, 8), (48, 4, 20, 12)])
def test_unusual_channel_counts(channels, se_red, gpc, pol, amp):
    """Channel counts that are not powers of two (the 16-byte board kernels, the single-pass tail backward and the wide
    conv slabs do not apply; their fallbacks do): forward and gradients against the same module on the CPU.  fp32:
    rel-L2 2e-2 per gradient tensor; bf16 (channels % 32 == 0 only): cosine > 0.95 as in the 6x128 test above."""
    if amp and channels % 32:
        pytest.skip("bf16 convolutions need channels % 32 == 0 (DESIGN.md section 3)")
    torch.manual_seed(11)
    params = SEResNetParams(num_blocks=2, channels=channels, se_reduction=se_red, global_pool_channels=gpc,
                            policy_channels=pol, value_fc_size=40, score_fc_size=24, obs_channels=50)
    ref = SEResNetModel(params)
    m = SEResNetModel(params)
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    if amp:
        m.configure_amp(True, torch.bfloat16, "cuda")
    B = 37
    obs = torch.randn(B, 50, 9, 9)
    cot = [torch.randn(B, 9, 9, 139) / 50, torch.randn(B, 3), torch.randn(B, 1)]
    for mod in (ref, m):
        mod.train()
        freeze_bn(mod)
    o_ref = ref(obs)
    (o_ref.policy_logits * cot[0]).sum().add((o_ref.value_logits * cot[1]).sum()).add((o_ref.score_lead * cot[2]).sum()).backward()
    o = m(obs.to(DEV))
    ((o.policy_logits * cot[0].to(DEV)).sum() + (o.value_logits * cot[1].to(DEV)).sum() + (o.score_lead * cot[2].to(DEV)).sum()).backward()
    if not amp:
        assert torch.allclose(o.policy_logits.cpu(), o_ref.policy_logits, rtol=2e-4, atol=1e-4)
        assert torch.allclose(o.value_logits.cpu(), o_ref.value_logits, rtol=2e-4, atol=1e-4)
    else:
        d = float((o.policy_logits.detach().float().cpu() - o_ref.policy_logits.detach()).abs().max())
        assert d <= 0.06 * float(o_ref.policy_logits.detach().abs().max())
    gr = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        r, got = gr[n].grad.double().flatten(), p.grad.double().cpu().flatten()
        if float(r.norm()) == 0:
            continue
        if amp:
            cos = float((r * got).sum() / (r.norm() * got.norm() + 1e-30))
            assert cos > 0.95 and 0.8 < float(got.norm() / r.norm()) < 1.25, (n, cos)
        else:
            err = float((got - r).norm() / r.norm())
            assert err < 2e-2, (n, err)
