import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch
from oracle import sink_oracle as O
from sink_attention import _native
from sink_attention.sink_flash_attention import SinkFlashAttentionFunc
lib = _native.lib()
for (B,Hq,Hkv,N,D,W,aux) in [(1,4,1,128,80,128,True),(1,4,1,128,80,128,False)]:
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B,Hq,N,D,generator=g).bfloat16(); k = torch.randn(B,Hkv,N,D,generator=g).bfloat16(); v = torch.randn(B,Hkv,N,D,generator=g).bfloat16()
    sa = torch.randn(Hq,generator=g)*0.5 if aux else None
    qd,kd,vd = q.cuda(),k.cuda(),v.cuda()
    o = torch.empty_like(qd); lse = torch.empty(B,Hq,N,device="cuda")
    st = lib.sfa_fwd(_native.desc(qd),_native.desc(kd),_native.desc(vd),_native.desc(o),lse.data_ptr(), sa.cuda().data_ptr() if aux else None, 0, W, D**-0.5, 0, _native.stream_ptr(qd.device))
    torch.cuda.synchronize()
    ref,lse_r = O.sink_attention_dense(q,k,v,0,W,sa)
    o = o.float().cpu(); lse = lse.cpu()
    print((B,Hq,Hkv,N,D,W,aux), _native.last_path())
    print(" lse err", (lse.double()-lse_r).abs().max().item(), "nan", torch.isnan(lse).sum().item())
    err = (o.double()-ref).abs()
    print(" o err by 16-col group:", [round(err[..., c:c+16].max().item(), 3) for c in range(0, D, 16)])
    print(" o err by head:", [round(err[:, h].max().item(), 3) for h in range(Hq)])
    print(" o err by 32-row block:", [round(err[:, :, r:r+32].max().item(), 3) for r in range(0, N, 32)])
    print(" sample o[0,0,0,:8]", o[0,0,0,:8].tolist(), "ref", ref[0,0,0,:8].tolist())
