"""Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the torch-CPU fp32 oracle, both measured
against the SAME network evaluated in fp64 on the CPU.  Tells fp32 round-off apart from a real defect.

usage: python tools/diag_grad.py [patch]   (default 64)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_mvd_seg_amd import trainer  # noqa: E402
from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    NS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.set_num_threads(os.cpu_count() or 8)
    strides = UO.CONFIGS["cfg2"]["strides"][:NS]
    ora = UO.build_plainconv_unet(4, 5, NS, strides, seed=0)
    batch = SO.synthetic_batch(B, 4, (P, P, P), strides, num_classes=5, seed=1234)
    loss_fn = LO.build_loss(len(batch["target"]))
    out32 = ora(batch["data"])
    l32 = loss_fn(out32, batch["target"])
    l32.backward()
    g32 = {n: p.grad.clone() for n, p in ora.named_parameters()}
    import copy
    ora64 = copy.deepcopy(ora).double()
    for p in ora64.parameters():
        p.grad = None
    out64 = ora64(batch["data"].double())
    l64 = loss_fn(out64, [t.double() for t in batch["target"]])
    l64.backward()
    g64 = {n: p.grad.clone() for n, p in ora64.named_parameters()}

    dev = torch.device("cuda:0")
    if os.environ.get("MVD_ENGINE"):
        from multimodal_mvd_seg_amd import ops
        ops.set_conv_engine(os.environ["MVD_ENGINE"])
    plans = trainer.make_plans((P, P, P), strides, batch_size=B)
    ds = {"channel_names": {str(i): str(i) for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=dev)
    tr.initialize()
    tr.network.load_state_dict(ora.state_dict())
    tr.optimizer.zero_grad()
    out = tr.network(batch["data"].to(dev))
    l = tr.loss(out, [t.to(dev) for t in batch["target"]])
    l.backward()
    print(f"loss: hip {float(l):.8f} cpu32 {float(l32):.8f} cpu64 {float(l64):.8f}")
    for i in range(len(out)):
        e_h = float((out[i].cpu().double() - out64[i]).abs().max())
        e_c = float((out32[i].double() - out64[i]).abs().max())
        print(f"logits{i}: max|hip-f64| {e_h:.2e}  max|cpu32-f64| {e_c:.2e}")
    worst = 0.0
    for n, p in tr.network.named_parameters():
        r = g64[n]
        nr = float(r.norm()) + 1e-30
        e_h = float((p.grad.cpu().double() - r).norm()) / nr
        e_c = float((g32[n].double() - r).norm()) / nr
        flag = " <<<" if e_h > 10 * e_c + 1e-5 else ""
        worst = max(worst, e_h / (e_c + 1e-12))
        print(f"{n:55s} |g|={nr:.3e} relL2 hip {e_h:.2e} cpu32 {e_c:.2e}{flag}")
    print("worst hip/cpu32 error ratio:", worst)


if __name__ == "__main__":
    main()
