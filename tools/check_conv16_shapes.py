import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from multimodal_mvd_seg_amd import ops
DEV = "cuda:0"
BF = torch.bfloat16
cl = torch.channels_last_3d
def ints(g, shape, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).float()
ok = True
for (N, D, H, W) in [(2, 52, 60, 44), (2, 33, 70, 97), (1, 128, 64, 64), (2, 128, 128, 128), (3, 17, 41, 130)]:
    g = torch.Generator().manual_seed(D * 7 + H)
    x = ints(g, (N, 32, D, H, W), -2, 2)
    w = ints(g, (32, 32, 3, 3, 3), -2, 2)
    b = ints(g, (32,), -3, 3)
    xr = x.clone().requires_grad_()
    ref = F.conv3d(xr, w, b, 1, 1)
    gy = ints(g, tuple(ref.shape), -1, 1)
    ref.backward(gy)
    gx = x.to(DEV).to(BF).contiguous(memory_format=cl).requires_grad_()
    gw = w.to(DEV).requires_grad_(); gb = b.to(DEV).requires_grad_()
    y = ops.Conv3dFn.apply(gx, None, gw, gb, (1, 1, 1))
    y.backward(gy.to(DEV).to(BF).contiguous(memory_format=cl))
    e1 = torch.equal(y.detach().cpu(), ref.detach().to(BF))
    e2 = torch.equal(gx.grad.cpu(), xr.grad.to(BF))
    print((N, D, H, W), "y", e1, "dx", e2, flush=True)
    if not e1:
        d = (y.detach().cpu().float() - ref.detach()).abs()
        idx = (d > 0).nonzero()
        print("  mismatches", idx.shape[0], "first", idx[:5].tolist(), "max", float(d.max()))
    ok = ok and e1 and e2
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
