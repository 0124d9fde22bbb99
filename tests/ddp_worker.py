"""Child process of tests/test_gpu_ddp.py: one data-parallel rank running nnUNetTrainerMI355.train_step.

usage: python tests/ddp_worker.py BACKEND RANK WORLD PORT OUT_DIR STEPS [PRECISION]
Every rank uses cuda:0 (the test box has one GPU; on a real node LOCAL_RANK picks the device).  The process is started
fresh (no HIP call before torch.distributed is up), joins the group, builds the trainer the way the reference's run_ddp
does (run_training.py:152-183: init_process_group, set_device, trainer.initialize -> DDP wrap) and writes what the
parent compares: parameters after each step, the reduced gradient of the first step, the loss."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_trainer(batch_size, device, precision="fp32"):
    from multimodal_mvd_seg_amd import trainer
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    plans = trainer.make_plans((32, 32, 32), strides, batch_size=batch_size)   # features 32 / 64 / 128: MFMA engines
    ds = {"channel_names": {str(i): str(i) for i in range(4)},
          "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=device)
    tr.precision = precision
    return tr


def rank_batch(tr, rank):
    """the synthetic batch of rank `rank` (nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22, seed 1234 + rank)"""
    keep = tr.local_rank
    tr.local_rank = rank
    try:
        return tr.make_dummy_batch()
    finally:
        tr.local_rank = keep


def main():
    backend, rank, world, port, out_dir, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], \
        sys.argv[5], int(sys.argv[6])
    precision = sys.argv[7] if len(sys.argv) > 7 else "fp32"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        torch.manual_seed(100 + rank)          # different init per rank: the DDP wrap must broadcast rank 0's weights
        tr = build_trainer(2, dev, precision)  # GLOBAL batch 2 -> 1 sample per rank at world 2
        tr.ddp_bucket_bytes = 2 << 20          # the 3-stage test network has 11 MB of gradients: several buckets
        tr.initialize()
        assert tr.is_ddp and tr.batch_size == 2 // world
        assert (tr.reducer is not None) and (tr.reducer.world == world)
        fp = tr.optimizer.fp
        batch = rank_batch(tr, rank)
        tr.on_train_epoch_start()
        out = {"init": fp.flat.detach().cpu().clone()}
        fired = []
        if world > 1:
            fp.listeners.append(lambda i: fired.append(i))
        for s in range(steps):
            res = tr.train_step(batch)
            out[f"loss{s}"] = float(res["loss"])
            out[f"flat{s}"] = fp.flat.detach().cpu().clone()
            if s == 0:
                out["grad0"] = fp.grad.detach().cpu().clone()      # SUM over ranks (the mean is taken in the optimizer)
                out["gradnorm0"] = float(tr.optimizer.grad_norm())
        out["direct_sink_reports"] = len(fired)
        out["n_params"] = len(fp.params)
        out["n_buckets"] = len(tr.reducer.buckets)
        if backend == "nccl":
            # RCCL smoke on the real buffers: SUM all-reduce of the flat gradient (world 1: identity) + barrier
            before = fp.grad.clone()
            dist.all_reduce(fp.grad)
            dist.barrier()
            torch.cuda.synchronize()
            out["nccl_allreduce_identity"] = bool(torch.equal(before, fp.grad)) if world == 1 else None
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
