"""Diagnostic: error of the bf16 mixed-precision step (HIP) and of torch-CPU autocast(bf16) relative to the fp64 truth
of the same network.  usage: PYTHONPATH=. python tools/diag_bf16.py [patch] [n_stages] [batch]"""
import copy
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from tests.test_gpu_parity import _cfg2_pair, DEV  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ora, loss_fn, batch, tr = _cfg2_pair(P, batch_size=B, n_stages=S)
from multimodal_mvd_seg_amd.network import set_precision  # noqa: E402

ora64 = copy.deepcopy(ora).double()
out64 = ora64(batch["data"].double())
l64 = loss_fn(out64, [t.double() for t in batch["target"]])
l64.backward()
g64 = {n: p.grad for n, p in ora64.named_parameters()}


def report(tag, outs, loss, grads):
    e = [float((o.double().cpu() - r).abs().max()) / float(r.abs().max()) for o, r in zip(outs, out64)]
    print(f"[{tag}] loss {float(loss):.6f} (fp64 {float(l64):.6f})  logits max-rel-err per level {['%.2e' % v for v in e]}")
    worst = []
    for n, r in g64.items():
        nr = float(r.norm())
        if nr < 1e-12:
            continue
        g = grads[n].double().cpu()
        rel = float((g - r).norm()) / nr
        cos = float((g * r).sum() / (g.norm() * r.norm()))
        worst.append((rel, cos, n))
    worst.sort(reverse=True)
    for rel, cos, n in worst[:6]:
        print(f"   relL2 {rel:.3e} cos {cos:.5f} {n}")
    print(f"   median relL2 {sorted(w[0] for w in worst)[len(worst) // 2]:.3e}  min cos {min(w[1] for w in worst):.5f}")


for prec in ("fp32", "bf16"):
    set_precision(tr.network, prec)
    tr.optimizer.zero_grad()
    outs = tr.network(batch["data"].to(DEV))
    l = tr.loss(outs, [t.to(DEV) for t in batch["target"]])
    l.backward()
    report("hip " + prec, [o.detach() for o in outs], l.detach().cpu(), {n: p.grad for n, p in tr.network.named_parameters()})

try:
    o2 = copy.deepcopy(ora)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        outs = o2(batch["data"])
        l = loss_fn(outs, batch["target"])
    l.backward()
    report("cpu autocast bf16", [o.detach().float() for o in outs], l.detach(), {n: p.grad for n, p in o2.named_parameters()})
except Exception as e:  # noqa: BLE001
    print("cpu autocast bf16 failed:", repr(e))
