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


# Gradient tolerances of the mid-size fp32 models.  The fixture batch is 2 boards with batch-statistics BatchNorm, and
# among its ~290k ReLU inputs a few lie within 1e-5 of zero -- inside the fp32 rounding noise of a K=2304 convolution
# (measured: changing only the summation order of the conv, 4 K-chunks instead of 2, moves y by 1e-5 relative, flips
# ONE output-ReLU mask element of block 2 and with it moves gradient norms by up to 1.1 % and single elements by up to
# 5.5 %; tools/debug_kc.py reproduces it).  The reference's own CPU fp32 result sits on the same knife edge, so these
# bounds admit a couple of such flips; formula-level parity is held to 5e-3 by the tiny-model tests above and to
# 2e-5 by the per-kernel tests in test_hip_kernels.py.
GRAD_NORM_TOL = 2e-2
GRAD_ELEM_TOL = 8e-2


@pytest.mark.parametrize("tag,shape", [("s6x128.", orc.NetShape(6, 128)), ("s3x256.", orc.NetShape(3, 256))])
def test_mid_models_fp32(golden, tag, shape):
    g = golden("g2_model_mid")
    m = SEResNetModel(SEResNetParams(**shape.__dict__))
    m.load_state_dict(orc.synth_state_dict(shape), strict=True)
    m.to(DEV)
    obs = g[tag + "obs"].to(DEV)
    m.eval()
    with torch.no_grad():
        o = m(obs)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "eval.policy"], rtol=1e-4, atol=5e-5)
    assert torch.allclose(o.value_logits.cpu(), g[tag + "eval.value"], rtol=1e-4, atol=5e-5)
    m.train(); freeze_bn(m)
    o = m(obs)
    assert torch.allclose(o.policy_logits.cpu(), g[tag + "train.policy"], rtol=2e-4, atol=1e-4)
    assert torch.allclose(o.score_lead.cpu(), g[tag + "train.score"], rtol=2e-4, atol=1e-4)
    B = obs.shape[0]
    loss = ((o.policy_logits * g[tag + "cot.policy"].to(DEV)).sum() / B + (o.value_logits * g[tag + "cot.value"].to(DEV)).sum()
            + (o.score_lead * g[tag + "cot.score"].to(DEV)).sum())
    loss.backward()
    names = list(g.np(tag + "grad_names"))
    norms = dict(zip(names, g.np(tag + "grad_norms")))
    grads = dict((n, p.grad) for n, p in m.named_parameters())
    for n in names:
        got = float(grads[n].double().norm())
        assert abs(got - norms[n]) <= GRAD_NORM_TOL * norms[n] + 1e-6, (n, got, norms[n])
    for n in ("input_bn.weight", "blocks.0.bn1.bias", "blocks.1.se_fc1.weight", "policy_conv1.weight",
              "value_fc2.weight", "score_fc2.bias"):
        ref = g[f"{tag}grad.{n}"]
        err = float((grads[n].cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
        assert err < GRAD_ELEM_TOL, (n, err)
    for n in ("blocks.0.conv1.weight", "input_conv.weight"):
        ref = g[f"{tag}grad.{n}[:4]"]
        err = float((grads[n][:4].cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
        assert err < GRAD_ELEM_TOL, (n, err)


@pytest.mark.parametrize("tag,shape", [("s6x128.", orc.NetShape(6, 128)), ("s3x256.", orc.NetShape(3, 256))])
def test_mid_models_bf16_bound(golden, tag, shape):
    """bf16 mode (bf16 activations + bf16 MFMA, fp32 accumulate).  Stated tolerance: within 5 % of
    |logit|max of the CPU emulation that rounds to bf16 at the same storage points
    (oracle.seresnet_policy_bf16_storage), and no further from the fp32 reference than that
    emulation is (x1.25 + 1 %).  The distance of bf16 storage from fp32 is inherent and
    input-dependent (SURVEY 8d measured 1-5 % for the reference's own CPU bf16 autocast)."""
    g = golden("g2_model_mid")
    sd = orc.synth_state_dict(shape)
    m = SEResNetModel(SEResNetParams(**shape.__dict__))
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.configure_amp(True, torch.bfloat16, "cuda")
    obs = g[tag + "obs"]
    freeze_bn(m)
    for train in (False, True):
        m.train(train)
        with torch.no_grad():
            got = m(obs.to(DEV)).policy_logits.float().cpu()
        ref = g[tag + ("train.policy" if train else "eval.policy")]
        emu = orc.seresnet_policy_bf16_storage(sd, obs, shape.num_blocks, train)
        mx = float(ref.abs().max())
        e_hip, e_emu, e_he = (float((a - b).abs().max()) / mx for a, b in ((got, ref), (emu, ref), (got, emu)))
        print(f"{tag} train={train}: hip-vs-fp32 {e_hip:.4f}  emulation-vs-fp32 {e_emu:.4f}  hip-vs-emulation {e_he:.4f}")
        assert e_he < 0.05
        assert e_hip < 1.25 * e_emu + 0.01
    m.train()
    o = m(obs.to(DEV))
    (o.policy_logits.sum() + o.value_logits.sum() + o.score_lead.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


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
