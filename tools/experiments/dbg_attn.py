import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shapegen_amd
from shapegen_amd import ops
torch.set_grad_enabled(False)
for (B,N,C,H) in [(1,64,64,4),(1,128,64,4),(1,256,64,4),(1,2048,64,4),(1,256,128,4)]:
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * N, 3 * C, generator=g) * 0.5
    q, k, v = qkv.half().float().split(C, dim=1)
    d = C // H
    qh, kh, vh = (z.reshape(N, H, d).permute(1, 0, 2).double() for z in (q, k, v))
    w = torch.softmax(qh @ kh.transpose(1, 2) / d ** 0.5, dim=-1)
    want = (w @ vh).permute(1, 0, 2).reshape(N, C)
    got = ops.set_attention_f16(qkv.half().cuda(), B, N, C, H).float().cpu().double()
    err = (got - want).abs()
    print(B,N,C, "rel", float((got-want).norm()/want.norm()), "max", float(err.max()), "argmax row/col", divmod(int(err.argmax()), C), "nan", int(torch.isnan(got).sum()))
    # per-head error
    for h in range(H):
        e = err[:, h*d:(h+1)*d]
        print("   head",h,"max",float(e.max()), "rows>1e-2:", int((e.max(1)[0]>1e-2).sum()))
