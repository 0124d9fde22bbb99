"""Diagnostic (variant library built with -DMVD_WG16_DBG=64): in-kernel s_memtime stamps of one wave of k_wgrad16 per tile:
[0] loop top, [1] barrier passed (every wave done with the previous tile), [2] tile written to LDS (includes the wait for
the global loads issued a tile earlier), [3] barrier, [4] next tile's loads issued, [5] the tile's MFMA steps issued."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops, _lib
from multimodal_mvd_seg_amd._lib import call, i3, query
dev = torch.device("cuda:0")
N, C, K, S = 2, 32, 32, 128
x = ops.empty_cl3d((N, C, S, S, S), dev, torch.bfloat16).normal_()
dy = ops.empty_cl3d((N, K, S, S, S), dev, torch.bfloat16).normal_()
dw = torch.empty(K, C, 3, 3, 3, device=dev)
db = torch.empty(K, device=dev)
ws = torch.empty(query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, S, S, S), dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(4):
    call("mvd_conv3d_wgrad_bf16", P(x), C, None, 0, P(dy), P(dw), P(db), N, S, S, S, K, i3((3, 3, 3)), i3((1, 1, 1)), P(ws), ws.numel(), s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 512)()
lib.mvd_debug_wg16_stamps.restype = ctypes.c_int
assert lib.mvd_debug_wg16_stamps(buf) == 0
st = [list(buf[i * 8:(i + 1) * 8]) for i in range(60)]
print("tile  +barrier +lds-write +barrier +load-issue +mfma-steps  | tile period   [cycles]")
for t in range(1, 33):
    r = st[t]
    if r[0] == 0:
        break
    print(f"{t:4d} {r[1]-r[0]:8d} {r[2]-r[1]:10d} {r[3]-r[2]:8d} {r[4]-r[3]:11d} {r[5]-r[4]:11d}  | {r[0]-st[t-1][0]:8d}")
