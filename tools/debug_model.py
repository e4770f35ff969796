import sys; sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from keisei_amd.training.models.se_resnet import *
from keisei_amd.hip.seresnet import SEResNetEngine
from oracle import keisei_oracle as orc
z = np.load('tests/golden/g2_model_tiny.npz')
sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd.')}
p = SEResNetParams(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)
m = SEResNetModel(p); m.load_state_dict(sd); m.cuda().eval()
obs = torch.from_numpy(z['randn.obs'])
eng = SEResNetEngine(m)
logits, v, s, sv = eng.forward(obs.cuda(), False, True, torch.float32)
def nchw(t): return t.float().cpu().reshape(t.shape[0], 9, 9, t.shape[2]).permute(0, 3, 1, 2)
def cmp(name, a, b): print(f"{name:20s} maxabs {float((a-b).abs().max()):.3e}  refmax {float(b.abs().max()):.3e}")
y0 = F.conv2d(obs, sd['input_conv.weight'], padding=1)
cmp('y0', nchw(sv.stem[0]), y0)
x = torch.relu(F.batch_norm(y0, sd['input_bn.running_mean'], sd['input_bn.running_var'], sd['input_bn.weight'], sd['input_bn.bias'], False, 0.1, 1e-5))
for i in range(2):
    (bx, bpool, y1, sc1, sh1, mu1, is1, g1, g, y2, sc2, sh2, mu2, is2, sqz, se1, se, out) = sv.blocks[i]
    pre = f'blocks.{i}.'
    cmp(f'b{i}.x', nchw(bx), x)
    cmp(f'b{i}.pool', bpool.cpu(), orc.global_pool(x))
    ry1 = F.conv2d(x, sd[pre+'conv1.weight'], padding=1)
    cmp(f'b{i}.y1', nchw(y1), ry1)
    h = torch.relu(F.batch_norm(ry1, sd[pre+'bn1.running_mean'], sd[pre+'bn1.running_var'], sd[pre+'bn1.weight'], sd[pre+'bn1.bias'], False, 0.1, 1e-5))
    rg1 = torch.relu(F.linear(orc.global_pool(x), sd[pre+'global_fc.0.weight'], sd[pre+'global_fc.0.bias']))
    cmp(f'b{i}.g1', g1.cpu(), rg1)
    rg = F.linear(rg1, sd[pre+'global_fc.2.weight'], sd[pre+'global_fc.2.bias'])
    cmp(f'b{i}.g', g.cpu(), rg)
    h = h + rg[:, :, None, None]
    ry2 = F.conv2d(h, sd[pre+'conv2.weight'], padding=1)
    cmp(f'b{i}.y2', nchw(y2), ry2)
    zz = F.batch_norm(ry2, sd[pre+'bn2.running_mean'], sd[pre+'bn2.running_var'], sd[pre+'bn2.weight'], sd[pre+'bn2.bias'], False, 0.1, 1e-5)
    cmp(f'b{i}.sqz', sqz.cpu(), zz.mean(dim=(2, 3)))
    rse = F.linear(torch.relu(F.linear(zz.mean(dim=(2, 3)), sd[pre+'se_fc1.weight'], sd[pre+'se_fc1.bias'])), sd[pre+'se_fc2.weight'], sd[pre+'se_fc2.bias'])
    cmp(f'b{i}.se', se.cpu(), rse)
    x = torch.relu(zz * torch.sigmoid(rse[:, :32])[:, :, None, None] + rse[:, 32:, None, None] + x)
    cmp(f'b{i}.out', nchw(out), x)
pol, val, sco = orc.seresnet_forward(dict(sd), obs, 2, False)
cmp('policy', logits.cpu(), pol); cmp('value', v.cpu(), val); cmp('score', s.cpu(), sco)
