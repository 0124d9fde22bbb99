"""Diagnostic: bf16 forward of the DDP test network on a batch of 2 vs the same two samples one at a time: which layer's
output first differs per sample, and by how much."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ddp_worker as W
DEV = torch.device("cuda:0")
torch.manual_seed(0)
tr = W.build_trainer(2, DEV, sys.argv[1] if len(sys.argv) > 1 else "bf16")
tr.initialize()
one = W.build_trainer(2, DEV)
one.batch_size, one.num_input_channels, one.local_rank = 1, 4, 0
s0, s1 = W.rank_batch(one, 0), W.rank_batch(one, 1)
x2 = torch.cat([s0["data"], s1["data"]]).to(DEV)
acts = {}
def hook(name):
    def f(m, i, o):
        acts.setdefault(name, []).append(o.detach().float().cpu() if torch.is_tensor(o) else None)
    return f
from multimodal_mvd_seg_amd import network
for n, m in tr.network.named_modules():
    if isinstance(m, network.ConvDropoutNormReLU) or isinstance(m, network.HipConvTranspose3d):
        m.register_forward_hook(hook(n))
with torch.no_grad():
    tr.network(x2)
    tr.network(x2[:1].contiguous())
    tr.network(x2[1:].contiguous())
for n, (a, b0, b1) in acts.items():
    b = torch.cat([b0, b1])
    d = (a - b).abs()
    print(f"{n:40s} max|diff| {float(d.max()):.3e}  frac differing {float((d > 0).float().mean()):.3e}  relL2 {float((a-b).norm()/a.norm()):.3e}")
