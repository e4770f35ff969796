import sys, torch
sys.path.insert(0, '.')
from keisei_amd import _lib
DEV='cuda'; st=_lib.stream_ptr
def to_nhwc(t, dt): return t.permute(0,2,3,1).reshape(t.shape[0],81,t.shape[1]).contiguous().to(DEV).to(dt)
for dt in (torch.float32, torch.bfloat16):
  for (B,C,H,heads) in [(3,32,4,False),(5,128,8,True),(4,256,16,False),(515,256,16,False)]:
    code=_lib.dtype_code(dt)
    g = torch.Generator().manual_seed(77 + C + B)
    A = lambda: to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    dxc, dout_up, out_up, y = A(), A(), A(), A()
    x = to_nhwc(torch.relu(torch.randn(B, C, 9, 9, generator=g)), dt)
    pool = torch.empty(B, 4 * C, device=DEV)
    _lib.call("ka_pool_fwd", x, pool, B, C, code, st())
    dpool = torch.randn(B, 3 * C, generator=g).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    mu, istd = (0.1 * torch.randn(C, generator=g)).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    se, se1 = torch.randn(B, 2 * C, generator=g).to(DEV), torch.randn(B, H, generator=g).to(DEV)
    W2 = (torch.randn(2 * C, H, generator=g) / H ** 0.5).to(DEV)
    W1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(DEV)
    up = (None, None) if heads else (dout_up, out_up)
    def outs():
        return (torch.empty_like(x), torch.empty_like(x), torch.empty(B, 2 * C, device=DEV), torch.empty(B, H, device=DEV),
                torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV))
    r = outs(); f = outs()
    _lib.call("ka_block_dx", dxc, *up, x, pool, dpool, r[0], B, C, code, st())
    _lib.call("ka_tail_bwd_fused", r[0], x, y, sc, sh, se, se1, W2, W1, mu, istd, r[1], r[2], r[3], r[4], r[5], B, C, H, code, st())
    _lib.call("ka_block_dx_tail_bwd", dxc, *up, x, pool, dpool, f[0], y, sc, sh, se, se1, W2, W1, mu, istd, f[1], f[2], f[3], f[4], f[5], B, C, H, code, st())
    torch.cuda.synchronize()
    for name, a, b in zip(("dx","dz","dse","dh","s1","s2"), f, r):
        d = (a.float()-b.float()).abs()
        print(dt, B, C, H, heads, name, "equal" if torch.equal(a,b) else f"max diff {float(d.max()):.3e} rel {float(d.max()/b.float().abs().max()):.2e} n={int((d>0).sum())}/{d.numel()} firstC={int((d>0).nonzero()[0][-1])}")
