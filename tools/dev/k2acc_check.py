"""K2 weight gradient, plain and accumulating (beta = 1) instantiations, against float64 on the host at a few layer shapes (dev check)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from uno_amd import _native
dev=torch.device('cuda:0')
for (Ci,Co,m,B) in [(32,48,22,32),(48,96,14,32),(96,192,6,32),(8,8,4,4),(192,192,6,32)]:
    g=torch.Generator().manual_seed(1)
    xt=torch.randn(B,Ci,2*m,m,dtype=torch.cfloat,generator=g); gO=torch.randn(B,Co,2*m,m,dtype=torch.cfloat,generator=g)
    base=[torch.randn(Ci,Co,m,m,dtype=torch.cfloat,generator=g) for _ in range(2)]
    pure=[torch.einsum("bixy,boxy->ioxy", xt[:,:,c*m:(c+1)*m].conj().to(torch.complex128), gO[:,:,c*m:(c+1)*m].to(torch.complex128)) for c in range(2)]
    plain=_native.mode_wgrad(xt.to(dev), gO.to(dev), (Ci,Co,m,m), 2)
    out=[b.clone().to(dev) for b in base]
    _native.profile_begin(16)
    _native.mode_wgrad(xt.to(dev), gO.to(dev), (Ci,Co,m,m), 2, out=out, accumulate=True)
    torch.cuda.synchronize(); names=[n for n,_,_ in _native.profile_end()]
    for c in range(2):
        ep=(plain[c].cpu().to(torch.complex128)-pure[c]).abs().max().item()/pure[c].abs().max().item()
        ea=(out[c].cpu().to(torch.complex128)-(pure[c]+base[c])).abs().max().item()/pure[c].abs().max().item()
        eb=(out[c].cpu().to(torch.complex128)-pure[c]).abs().max().item()/pure[c].abs().max().item()
        print((Ci,Co,m,B),c,'plain err',ep,'acc err',ea,'acc-vs-pure',eb,names)
