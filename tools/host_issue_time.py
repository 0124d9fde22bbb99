"""Host-side cost of issuing one train step (no device sync inside the loop) next to the device time of the step:
tells whether launch capture (hipGraph) could buy anything."""
import time

import torch

import bench
from multimodal_mvd_seg_amd import trainer


def main():
    dev = torch.device("cuda", 0)
    plans = trainer.make_plans(bench.PATCH, bench.STRIDES, batch_size=bench.PER_GPU_BATCH)
    tr = trainer.nnUNetTrainerMI355Benchmark_noDataLoading(plans, "3d_fullres", 0, bench.dataset_json(), device=dev)
    torch.manual_seed(0)
    tr.initialize()
    tr.on_train_epoch_start()
    batch = tr.dummy_batch
    for _ in range(3):
        tr.train_step(batch, return_device_loss=True)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(batch, return_device_loss=True)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host issue {t_issue / n * 1e3:.2f} ms/step, device-complete {t_all / n * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
