"""Diagnostic (variant library built with -DMVD_WINO_DBG=64): in-kernel s_memtime stamps of one wave of k_fwd_wino2, per
32-channel chunk: [0] chunk start, [1] first weights requested, [2] barrier (previous chunk's halo free), [3] halo loaded
and written to LDS, [4] barrier, [5] the chunk's 192 MFMAs issued; after the last chunk [6] = output transform + stores."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops, _lib
from multimodal_mvd_seg_amd._lib import call, i3, query
dev = torch.device("cuda:0")
N, C1, C2, K, S = 2, 32, 32, 32, 128
x1 = ops.empty_cl3d((N, C1, S, S, S), dev).normal_()
x2 = ops.empty_cl3d((N, C2, S, S, S), dev).normal_()
w = torch.randn(K, C1 + C2, 3, 3, 3, device=dev) * 0.03
wf, wb = ops.pack_weight(w, False)
uf = torch.empty(query("mvd_wino_weight_elems", C1 + C2, K), device=dev)
ub = torch.empty_like(uf)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
call("mvd_pack_weight_wino", P(w), P(uf), P(ub), K, C1 + C2, s)
bias = torch.zeros(K, device=dev)
y = ops.empty_cl3d((N, K, S, S, S), dev)
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
for it in range(4):
    call("mvd_conv3d_fwd_wino", P(x1), C1, P(x2), C2, P(wf), P(uf), P(bias), P(y), N, S, S, S, K, i3((3, 3, 3)), i3((1, 1, 1)),
         P(ws), ws.numel(), s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 512)()
lib.mvd_debug_wino_stamps.restype = ctypes.c_int
assert lib.mvd_debug_wino_stamps(buf) == 0
st = [list(buf[i * 8:(i + 1) * 8]) for i in range(4)]
print("chunk  +wreq  +barrier  +halo   +barrier  +mfma(192)   [cycles]")
for c in range(2):
    r = st[c]
    print(f"{c:4d} {r[1]-r[0]:6d} {r[2]-r[1]:8d} {r[3]-r[2]:7d} {r[4]-r[3]:8d} {r[5]-r[4]:9d}")
e = st[2]
print("epilogue: combine", e[0] - st[1][5], "barrier", e[1] - e[0], "lds writes", e[2] - e[1], "barrier", e[3] - e[2], "reads+stores", e[6] - e[3])
print("epilogue (output transform + stores):", st[2][6] - st[1][5] if st[2][6] else st[1][6] - st[1][5], "cycles; whole tile:", (st[2][6] or st[1][6]) - st[0][0])
