"""120 train steps on the synthetic benchmark batch (BASELINE cfg 2): prints the first / last losses; the loss must stay
finite and fall (random labels: the net memorises the one batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from multimodal_mvd_seg_amd import trainer
dev=torch.device("cuda",0)
PREC = "bf16" if "--precision" in sys.argv and sys.argv[sys.argv.index("--precision") + 1] == "bf16" else "fp32"
plans=trainer.make_plans(bench.PATCH, bench.STRIDES, batch_size=2)
tr=trainer.nnUNetTrainerMI355Benchmark_noDataLoading(plans,"3d_fullres",0,bench.dataset_json(),device=dev)
tr.precision = PREC
torch.manual_seed(0); tr.initialize(); tr.on_train_epoch_start()
b=tr.dummy_batch
ls=[]
for i in range(120):
    ls.append(float(tr.train_step(b)["loss"]))
print("loss[0,1,2]=",ls[:3],"loss[-3:]=",ls[-3:], "finite", all(l==l and abs(l)<1e6 for l in ls), "decreasing", ls[-1]<ls[0])
