"""Where the gradient all-reduce of a train step can hide, measured on the hardware at hand (ONE GPU):

1. `python tools/ddp_overlap_probe.py overlap [fp32|bf16]` -- two ranks on cuda:0 over gloo (as tests/test_gpu_ddp.py) run
   the configs[1] train step (31.2 M parameters, 4x128^3, batch 1 per rank, 25 MB buckets = DDP's default).  A HIP event is
   recorded on the compute stream when backward starts, at every bucket's launch (the moment its last gradient is
   complete) and when backward ends: the table shows, per bucket, how much backward compute is still to run when its
   all-reduce is issued -- the window in which the collective can hide on a multi-GPU node.  (gloo stages through the
   host, so its own duration says nothing about RCCL; the ISSUE times are what carries over.)
2. `python tools/ddp_overlap_probe.py hooks [fp32|bf16]` -- one process, world 1: eager step time with and without the
   ~220 post-accumulate-grad hooks + bucket bookkeeping (BucketedGradReducer(always_hook=True)).
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATCH = (128, 128, 128)
STRIDES = [[1, 1, 1]] + [[2, 2, 2]] * 5
DS = {"channel_names": {str(i): f"m{i}" for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}


def build(precision, batch, dev):
    from multimodal_mvd_seg_amd import trainer
    import torch
    plans = trainer.make_plans(PATCH, STRIDES, batch_size=batch)
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, DS, device=dev)
    tr.precision = precision
    tr.use_hip_graph = False
    torch.manual_seed(0)
    tr.initialize()
    tr.on_train_epoch_start()
    return tr


def worker(rank, world, port, precision, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tr = build(precision, world, dev)       # global batch = world -> 1 sample per rank
    batch = tr.make_dummy_batch()
    red = tr.reducer
    marks = []
    launch = red._launch

    def marked(b):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((b, e, time.perf_counter()))
        launch(b)
    red._launch = marked
    rows = None
    for step in range(4):
        marks.clear()
        red.reset()
        tr.optimizer.zero_grad()
        l, _ = tr._forward_loss(batch["data"], batch["target"])
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        l.backward()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        t1 = time.perf_counter()
        red.wait()
        tr.optimizer.step()
        torch.cuda.synchronize()
        bwd = e0.elapsed_time(e1)
        rows = [{"bucket": b, "MB": round((red.buckets[b][1] - red.buckets[b][0]) * 4 / 1e6, 1),
                 "issued_ms_into_backward_device": round(e0.elapsed_time(e), 2),
                 "backward_left_ms_device": round(bwd - e0.elapsed_time(e), 2),
                 "issued_ms_into_backward_host": round((th - t0) * 1e3, 2)} for b, e, th in marks]
        summary = {"step": step, "backward_ms_device": round(bwd, 2), "backward_ms_host_issue": round((t1 - t0) * 1e3, 2)}
    if rank == 0:
        json.dump({"precision": precision, "world": world, "backend": "gloo (two ranks on one MI355X)", **summary,
                   "buckets": rows}, open(out, "w"), indent=1)
    dist.barrier()
    dist.destroy_process_group()


def hooks(precision):
    import torch
    from multimodal_mvd_seg_amd.parallel import BucketedGradReducer
    dev = torch.device("cuda:0")
    tr = build(precision, 2, dev)
    batch = tr.make_dummy_batch()

    def run(n):
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.train_step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    a = run(15)
    tr.reducer = BucketedGradReducer(tr.optimizer.fp, tr.ddp_bucket_bytes, optimizer=tr.optimizer, always_hook=True)
    b = run(15)
    tr.reducer.remove_hooks()
    tr.reducer = None
    c = run(15)
    print(json.dumps({"precision": precision, "eager_step_ms_no_reducer": [round(a, 3), round(c, 3)],
                      "eager_step_ms_with_hooks_and_buckets": round(b, 3), "parameters_hooked": len(tr.optimizer.fp.params)}))


if __name__ == "__main__":
    mode = sys.argv[1]
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    if mode == "hooks":
        hooks(precision)
    elif mode == "overlap":
        out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/ddp_overlap.json"
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, __file__, "_worker", str(r), "2", "29677", precision, out], env=env)
              for r in range(2)]
        rc = [p.wait(timeout=600) for p in ps]
        assert rc == [0, 0], rc
        print(open(out).read())
    elif mode == "_worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6])
