#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3c; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q > $O/fused_tests.log 2>&1; echo "fused tests rc=$?"; grep -E "^(FAILED|PASSED|ERROR)|passed|failed" $O/fused_tests.log | tail -30
run timeout -k 10 300 python - > $O/det.log 2>&1 <<'PY'
import torch, ctypes, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_fused_block as T
x, w, b, gamma, beta = T._rand_block(2, 128, 128, 128, 1)
for st in (False, True):
    ys = [T._conv_fused(x, w, b, want_stats=st)[0] for _ in range(3)]
    print("stats", st, "run-to-run equal:", torch.equal(ys[0], ys[1]), torch.equal(ys[1], ys[2]))
ya = T._conv_fused(x, w, b, want_stats=False)[0]
yb = T._conv_fused(x, w, b, want_stats=True)[0]
d = (ya.float() - yb.float()).abs()
nz = d.nonzero()
print("differing elements", nz.shape[0], "of", d.numel(), "max", float(d.max()))
if nz.shape[0]:
    print("first few (n,c,z,y,x):", nz[:20].tolist())
    import collections
    print("by channel", collections.Counter(nz[:, 1].tolist()).most_common(8))
    print("by z", collections.Counter(nz[:, 2].tolist()).most_common(8))
    print("by y%8", collections.Counter((nz[:, 3] % 8).tolist()).most_common(8))
    print("by x%32", collections.Counter((nz[:, 4] % 32).tolist()).most_common(8))
PY
cat $O/det.log | tail -12
echo done
