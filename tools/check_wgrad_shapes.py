"""Diagnostic: fp32 weight gradient of 3x3x3 stride-1 convs through the C ABI against torch (MIOpen) on a list of shapes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops
from multimodal_mvd_seg_amd._lib import call, i3, query
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (N, C1, C2, K, D, H, W) in [(2, 32, 0, 32, 64, 64, 64), (2, 64, 0, 64, 32, 32, 32), (2, 128, 0, 128, 16, 16, 16),
                                (2, 256, 0, 256, 8, 8, 8), (2, 64, 64, 64, 32, 32, 32), (2, 128, 128, 128, 16, 16, 16),
                                (2, 256, 256, 256, 8, 8, 8), (1, 32, 0, 32, 16, 24, 40), (2, 32, 0, 64, 10, 18, 26)]:
    C = C1 + C2
    x = torch.randn(N, C, D, H, W, device=dev)
    dy = torch.randn(N, K, D, H, W, device=dev)
    x1 = ops.to_cl3d(x[:, :C1]) if hasattr(ops, "to_cl3d") else x[:, :C1].contiguous(memory_format=torch.channels_last_3d)
    x2 = (x[:, C1:].contiguous(memory_format=torch.channels_last_3d)) if C2 else None
    x1 = x[:, :C1].contiguous(memory_format=torch.channels_last_3d)
    dyc = dy.contiguous(memory_format=torch.channels_last_3d)
    dw = torch.empty(K, C, 3, 3, 3, device=dev)
    db = torch.empty(K, device=dev)
    ws = torch.empty(query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, D, H, W), dtype=torch.uint8, device=dev)
    call("mvd_conv3d_wgrad", P(x1), C1, P(x2), C2, P(dyc), P(dw), P(db), N, D, H, W, K, i3((3, 3, 3)), i3((1, 1, 1)), P(ws), ws.numel(), s)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv3d_weight(x.double(), (K, C, 3, 3, 3), dy.double(), padding=1)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    eb = float((db.double() - dy.double().sum((0, 2, 3, 4))).abs().max() / dy.double().sum((0, 2, 3, 4)).abs().max())
    print(f"N={N} C={C1}+{C2} K={K} {D}x{H}x{W}: dW rel err {err:.2e}  db rel err {eb:.2e}")
