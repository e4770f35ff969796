"""Per-layer error of Winograd F(2x2,3x3) / F(3x3,3x3) with bf16-rounded transformed operands and fp32 accumulation, against the direct
bf16 convolution on the same inputs (CPU, float64 reference).  profiles/NOTES_r04.md section 13."""
import numpy as np, torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float64)
# F(2,3)
Bt2 = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=torch.float64)
G2 = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=torch.float64)
At2 = torch.tensor([[1,1,1,0],[0,1,-1,-1]],dtype=torch.float64)
# F(3,3): points 0,1,-1,2,inf
At3 = torch.tensor([[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,1]],dtype=torch.float64)
G3 = torch.tensor([[1/2,0,0],[-1/2,-1/2,-1/2],[-1/6,1/6,-1/6],[1/6,1/3,2/3],[0,0,1]],dtype=torch.float64)
Bt3 = torch.tensor([[2,-1,-2,1,0],[0,-2,-1,1,0],[0,2,-3,1,0],[0,-1,0,1,0],[0,2,-1,-2,1]],dtype=torch.float64)
def check(At,G,Bt):
    m=At.shape[0]; n=At.shape[1]
    d=torch.randn(n,dtype=torch.float64); g=torch.randn(3,dtype=torch.float64)
    y=At@((G@g)*(Bt@d))
    ref=torch.stack([sum(d[i+k]*g[k] for k in range(3)) for i in range(m)])
    return (y-ref).abs().max().item()
print('F23',check(At2,G2,Bt2),'F33',check(At3,G3,Bt3))
def conv_wino(x,w,At,G,Bt,m,round_v=True):
    # x (B,C,9,9) float64 (bf16 values), w (O,C,3,3)
    B,C,H,W=x.shape; O=w.shape[0]; n=m+2
    nt=(H+m-1)//m
    xp=torch.zeros(B,C,nt*m+2,nt*m+2,dtype=torch.float64); xp[:,:,1:H+1,1:W+1]=x
    U=torch.einsum('ai,ocij,bj->ocab',G,w,G)
    if round_v: U=bf(U)
    out=torch.zeros(B,O,nt*m,nt*m,dtype=torch.float64)
    for ti in range(nt):
        for tj in range(nt):
            d=xp[:,:,ti*m:ti*m+n,tj*m:tj*m+n]
            V=torch.einsum('ai,bcij,dj->bcad',Bt,d,Bt)
            if round_v: V=bf(V)
            M=torch.einsum('bcad,ocad->boad',V,U).to(torch.float32).to(torch.float64)
            Y=torch.einsum('ia,boad,jd->boij',At,M,At)
            out[:,:,ti*m:ti*m+m,tj*m:tj*m+m]=Y
    return out[:,:,:H,:W]
B,C,O=8,256,256
x=bf(torch.relu(torch.randn(B,C,9,9,dtype=torch.float64)))
w=bf(torch.randn(O,C,3,3,dtype=torch.float64)*(2/(9*C))**.5)
ref=torch.nn.functional.conv2d(x,w,padding=1)
xf=torch.relu(torch.randn(B,C,9,9,dtype=torch.float64))
def rel(a,b): return ((a-b).norm()/b.norm()).item()
print('exactness F23', rel(conv_wino(x,w,At2,G2,Bt2,2,False),ref),'F33',rel(conv_wino(x,w,At3,G3,Bt3,3,False),ref))
e2=rel(conv_wino(x,w,At2,G2,Bt2,2),ref); e3=rel(conv_wino(x,w,At3,G3,Bt3,3),ref)
print('bf16-rounded transforms: F23 rel L2',e2,'F33',e3)
# reference noise level: rounding output to bf16 and rounding of inputs
print('bf16 output rounding rel', rel(bf(ref),ref))
x64=torch.relu(torch.randn(B,C,9,9,dtype=torch.float64)); w64=torch.randn(O,C,3,3,dtype=torch.float64)*(2/(9*C))**.5
print('bf16 input+weight rounding rel', rel(torch.nn.functional.conv2d(bf(x64),bf(w64),padding=1), torch.nn.functional.conv2d(x64,w64,padding=1)))
