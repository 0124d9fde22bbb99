import sys, torch, numpy as np
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from multimodal_mvd_seg_amd import ops
N, D, H, W = 2, 128, 128, 128
g = torch.Generator().manual_seed(1)
ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
x, w, b = ints((N, 32, D, H, W), -2, 2), ints((64, 32, 3, 3, 3), -2, 2), ints((64,), -3, 3)
ref = F.conv3d(x, w, b, 2, 1).to(torch.bfloat16)
with torch.no_grad():
    y = ops.Conv3dFn.apply(x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d), None, w.cuda(), b.cuda(), (2, 2, 2)).cpu()
bad = (y != ref)
print('bad frac', bad.float().mean().item())
for name, dim in (('n', 0), ('c', 1), ('z', 2), ('y', 3), ('x', 4)):
    other = tuple(i for i in range(5) if i != dim)
    f = bad.float().mean(other)
    print(name, np.array2string(f.numpy(), precision=2, max_line_width=200))
