"""Diagnostic (variant library built with -DMVD_F16_DBG=64): prints the in-kernel s_memtime stamps one wave of k_fwd16 wrote
into the workspace -- per tap group: [0] start, [1] after the weight store to LDS, [2] after the barrier, [3] after the
loads of group+2 were issued, [4] after the group's MFMAs were issued."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops
from multimodal_mvd_seg_amd._lib import call, i3, query
dev = torch.device("cuda:0")
N, C, K, S = 2, 64, 64, 64
x = ops.empty_cl3d((N, C, S, S, S), dev, torch.bfloat16).normal_()
w = torch.randn(K, C, 3, 3, 3, device=dev) * 0.03
wf, wb = ops.pack_weight_bf16(w, False)
bias = torch.zeros(K, device=dev)
y = ops.empty_cl3d((N, K, S, S, S), dev, torch.bfloat16)
ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(3):
    ws.zero_()
    call("mvd_conv3d_fwd_bf16", P(x), C, None, 0, P(wf), P(bias), P(y), N, S, S, S, K, i3((3, 3, 3)), i3((1, 1, 1)), P(ws), ws.numel(), s)
    torch.cuda.synchronize()
st = ws.view(torch.int64)[:40 * 8].cpu().view(40, 8)
t0 = int(st[0, 0])
print("grp  start   +store  +barrier +loadiss +mfma   (cycles; start relative to the first group)")
for g in range(20):
    r = [int(v) for v in st[g, :5]]
    if r[0] == 0:
        break
    print(f"{g:3d} {r[0]-t0:7d} {r[1]-r[0]:7d} {r[2]-r[1]:8d} {r[3]-r[2]:8d} {r[4]-r[3]:7d}")
