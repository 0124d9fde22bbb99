import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from multimodal_mvd_seg_amd import trainer
for prec in ("bf16", "fp32"):
    plans = trainer.make_plans((128, 128, 128), [[1,1,1]] + [[2,2,2]]*5, batch_size=2)
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, bench.dataset_json(), device=torch.device("cuda:0"))
    tr.precision = prec
    torch.manual_seed(0)
    tr.initialize()
    batch = tr.make_dummy_batch(seed=3)
    n = 160 if prec == "bf16" else 60
    losses = []
    for i in range(n):
        losses.append(float(tr.train_step(batch)["loss"]))
    print(prec, "graphed:", tr._step_graph is not None, "loss", [round(losses[i], 4) for i in range(0, n, n // 8)], "last", round(losses[-1], 4),
          "finite", all(l == l and abs(l) < 1e6 for l in losses))
    del tr
    torch.cuda.empty_cache()
