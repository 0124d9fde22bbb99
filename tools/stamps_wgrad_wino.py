"""Diagnostic (variant library built with -DMVD_WG16_DBG=192): in-kernel s_memtime stamps of one wave of the fp32 Winograd
weight-gradient kernel (k_wgrad_wino2w12) per 2x8x8 tile: [0] loop top, [1] barrier passed, [2] tile written to LDS
(includes the wait for the loads issued a tile earlier), [3] barrier, [4] next tile's loads issued, [5] the 16 steps."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops, _lib
from multimodal_mvd_seg_amd._lib import call, i3, query
dev = torch.device("cuda:0")
N, C, K, S = 2, 32, 32, 128
x = ops.empty_cl3d((N, C, S, S, S), dev).normal_()
dy = ops.empty_cl3d((N, K, S, S, S), dev).normal_()
dw = torch.empty(K, C, 3, 3, 3, device=dev)
db = torch.empty(K, device=dev)
ws = torch.empty(query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, S, S, S), dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(3):
    call("mvd_conv3d_wgrad", P(x), C, None, 0, P(dy), P(dw), P(db), N, S, S, S, K, i3((3, 3, 3)), i3((1, 1, 1)), P(ws), ws.numel(), s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 512)()
lib.mvd_debug_wg16_stamps.restype = ctypes.c_int
assert lib.mvd_debug_wg16_stamps(buf) == 0
st = [list(buf[i * 8:(i + 1) * 8]) for i in range(60)]
db = st[1][6] != 0  # two LDS images (default): [3] loop top, [4] next tile's loads issued, [5] steps, [6] next tile written, [7] barrier
if db:
    print("tile  +load-issue  +steps  +lds-write  +barrier  | tile period   [ticks]")
    for t in range(1, 60):
        r = st[t]
        if r[0] == 0:
            break
        if t % 4 == 0:
            print(f"{t:4d} {r[4]-r[3]:11d} {r[5]-r[4]:7d} {r[6]-r[5]:11d} {r[7]-r[6]:9d}  | {r[0]-st[t-1][0]:8d}")
else:
    print("tile  +barrier +lds-write +barrier +load-issue +steps  | tile period   [ticks]")
    for t in range(1, 60):
        r = st[t]
        if r[0] == 0:
            break
        if t % 4 == 0:
            print(f"{t:4d} {r[1]-r[0]:8d} {r[2]-r[1]:10d} {r[3]-r[2]:8d} {r[4]-r[3]:11d} {r[5]-r[4]:7d}  | {r[0]-st[t-1][0]:8d}")
