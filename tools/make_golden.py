"""Generates tests/golden/* (run in the BUILD container only: it imports reference modules by file path from
/root/reference and the reference's C++ persistence extension compiled into oracle/_ref).

Each fixture holds inputs + expected outputs (data only).  `source` in every file says what produced the
expected values: "reference" = a module of /root/reference executed here; "oracle" = this repo's CPU
restatement (the reference's own module is not importable -- SURVEY.md 8c).

usage: python tools/make_golden.py
"""
import ast
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as UO, loss_oracle as LO, step_oracle as SO, cc_oracle, build_ref  # noqa: E402

R = "/root/reference/nnUNet/nnunetv2"
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(R, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name)


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ conv3d (torch CPU ops = reference arithmetic)
def conv_fixtures():
    for tag, stride in (("s1", (1, 1, 1)), ("s2", (2, 2, 2)), ("s122", (1, 2, 2))):
        for cin, cout in ((1, 32), (4, 32), (32, 64)):
            g = gen(10 + cin)
            x = torch.randn(2, cin, 12, 10, 14, generator=g, dtype=torch.float64, requires_grad=True)
            w = (torch.randn(cout, cin, 3, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
            b = (torch.randn(cout, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
            y = F.conv3d(x, w, b, stride, 1)
            gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
            y.backward(gy)
            save(f"conv3d_{tag}_c{cin}_{cout}.npz", source="torch.nn.functional.conv3d fp64 (CPU)",
                 stride=stride, x=x.float(), w=w.float(), b=b.float(), y=y.float(), gy=gy.float(),
                 gx=x.grad.float(), gw=w.grad.float(), gb=b.grad.float())
    for tag, st, cin, cout, sp in (("k2s2", (2, 2, 2), 96, 40, (3, 4, 5)), ("k122s122", (1, 2, 2), 64, 32, (3, 4, 5)),
                                   ("k2s2_small", (2, 2, 2), 8, 4, (4, 3, 5))):
        g = gen(20)
        x = torch.randn(2, cin, *sp, generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(cin, cout, *st, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
        b = (torch.randn(cout, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
        y = F.conv_transpose3d(x, w, b, st)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(gy)
        save(f"convT3d_{tag}.npz", source="torch.nn.functional.conv_transpose3d fp64 (CPU)", stride=st,
             x=x.float(), w=w.float(), b=b.float(), y=y.float(), gy=gy.float(), gx=x.grad.float(),
             gw=w.grad.float(), gb=b.grad.float())
    # 1x1x1 seg head
    g = gen(21)
    x = torch.randn(2, 32, 5, 6, 7, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(5, 32, 1, 1, 1, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
    b = (torch.randn(5, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
    y = F.conv3d(x, w, b)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    save("conv1x1.npz", source="torch conv3d k=1 fp64 (CPU)", x=x.float(), w=w.float(), b=b.float(), y=y.float(),
         gy=gy.float(), gx=x.grad.float(), gw=w.grad.float(), gb=b.grad.float())


def instnorm_fixtures():
    g = gen(30)
    x = (torch.randn(2, 5, 7, 9, 11, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_()
    gamma = (torch.rand(5, generator=g, dtype=torch.float64) + 0.5).requires_grad_()
    beta = (torch.randn(5, generator=g, dtype=torch.float64) * 0.2).requires_grad_()
    y = F.leaky_relu(F.instance_norm(x, None, None, gamma, beta, True, 0.1, 1e-5), 0.01)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    save("instnorm_lrelu.npz", source="torch instance_norm(eps=1e-5)+leaky_relu(0.01) fp64 (CPU)", x=x.float(),
         gamma=gamma.float(), beta=beta.float(), y=y.float(), gy=gy.float(), gx=x.grad.float(),
         ggamma=gamma.grad.float(), gbeta=beta.grad.float())


# ------------------------------------------------------------------ reference-pinned small modules
def reference_fixtures():
    rce = ref_module("training/loss/robust_ce_loss.py", "ref_rce")
    g = gen(40)
    logits = torch.randn(2, 5, 4, 4, 4, generator=g, requires_grad=True)
    tgt = torch.round(torch.rand(2, 1, 4, 4, 4, generator=g) * 4)
    l = rce.RobustCrossEntropyLoss()(logits, tgt)
    l.backward()
    assert torch.equal(l.detach(), LO.RobustCrossEntropyLoss()(logits.detach(), tgt))
    save("robust_ce.npz", source="reference robust_ce_loss.py:6-16", logits=logits, target=tgt, loss=l,
         glogits=logits.grad)

    sk = ref_module("training/loss/soft_skeleton.py", "ref_skel")
    for shape, binary, seed in (((1, 1, 16, 16, 16), False, 41), ((2, 3, 12, 10, 14), False, 42),
                                ((1, 1, 16, 16, 16), True, 43), ((2, 1, 9, 11, 8), True, 44)):
        for it in (1, 3, 10):
            g = gen(seed)
            x = torch.rand(shape, generator=g)
            if binary:
                # tie-heavy input (blobby binary mask)
                x = (F.avg_pool3d(x, 3, 1, 1) > 0.5).float()
            x.requires_grad_()
            y = sk.soft_skel(x, it)
            gy = torch.randn(y.shape, generator=g)
            y.backward(gy)
            yo = LO.soft_skel(x.detach(), it)
            assert torch.equal(y.detach(), yo), "oracle soft_skel differs from the reference"
            e = sk.soft_erode(x.detach())
            d = sk.soft_dilate(x.detach())
            save(f"soft_skel_{'bin' if binary else 'u'}_{'x'.join(map(str, shape))}_iter{it}.npz",
                 source="reference soft_skeleton.py:6-37", iter=it, x=x, skel=y, gy=gy, gx=x.grad, erode=e, dilate=d)

    poly = ref_module("training/lr_scheduler/polylr.py", "ref_poly")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], 1e-2)
    # the reference ctor passes a 4th positional arg (`verbose`) that torch 2.10's LRScheduler no longer takes
    # (ordinary TypeError); its step() body (polylr.py:13-20) is what defines the schedule, so run that on a stub
    import types
    sch = types.SimpleNamespace(optimizer=opt, initial_lr=1e-2, max_steps=200, exponent=0.9, ctr=0)
    lrs = []
    for e in range(200):
        poly.PolyLRScheduler.step(sch, e)
        lrs.append(opt.param_groups[0]['lr'])
    assert np.allclose(lrs, [SO.poly_lr(1e-2, e, 200) for e in range(200)], rtol=0, atol=0)
    json.dump({"source": "reference polylr.py:4-20", "initial_lr": 1e-2, "max_steps": 200, "lrs": lrs},
              open(os.path.join(OUT, "polylr.json"), "w"))

    init = ref_module("utilities/network_initialization.py", "ref_init")
    stats = {}
    torch.manual_seed(0)
    for name, mod in (("conv_32_32", torch.nn.Conv3d(32, 32, 3)), ("conv_4_32", torch.nn.Conv3d(4, 32, 3)),
                      ("convT_64_32", torch.nn.ConvTranspose3d(64, 32, 2, 2)), ("conv1_32_5", torch.nn.Conv3d(32, 5, 1))):
        mod.apply(init.InitWeights_He(1e-2))
        fan_in = mod.weight.size(1) * mod.weight[0, 0].numel()
        stats[name] = {"std": float(mod.weight.std()), "expected_std": float(np.sqrt(2 / (1 + 1e-4) / fan_in)),
                       "bias_abs_max": float(mod.bias.abs().max()), "shape": list(mod.weight.shape)}
    json.dump({"source": "reference network_initialization.py:4-12", "stats": stats},
              open(os.path.join(OUT, "he_init_stats.json"), "w"), indent=1)

    topo = ref_module("experiment_planning/experiment_planners/network_topology.py", "ref_topo")
    props = {}
    for name, patch, spacing in (("cfg1", (64, 64, 64), (1, 1, 1)), ("cfg2", (128, 128, 128), (1, 1, 1)),
                                 ("cfg5", (160, 160, 128), (1, 1, 1)), ("author", (64, 128, 256), (1, 1, 1))):
        r = topo.get_pool_and_conv_props(spacing, patch, 4, 999999)
        props[name] = {"patch": list(patch), "num_pool_per_axis": [int(i) for i in r[0]],
                       "pool_op_kernel_sizes": [[int(j) for j in i] for i in r[1]],
                       "conv_kernel_sizes": [[int(j) for j in i] for i in r[2]],
                       "patch_size": [int(i) for i in r[3]], "must_be_divisible_by": [int(i) for i in r[4]]}
    json.dump({"source": "reference network_topology.py:30-105", "props": props},
              open(os.path.join(OUT, "topology_props.json"), "w"), indent=1)
    print("wrote polylr.json he_init_stats.json topology_props.json")


# ------------------------------------------------------------------ reference functions whose MODULE cannot be imported
def ref_function(rel, func_name, class_name=None, **ns):
    """Compiles ONE function definition out of a reference source file and returns it as a live function.  Used for
    pure torch/numpy functions whose module fails at import time on an absent third-party package (other_loss.py ->
    lightly, base_data_loader.py / nnUNetTrainer.py -> batchgenerators, sliding_window_prediction.py -> acvl_utils):
    the function body itself is executed unchanged, so the fixtures below are reference-pinned.  Runs in the build
    container only (needs /root/reference); nothing of the source text is stored."""
    import typing
    path = os.path.join(R, rel)
    tree = ast.parse(open(path).read(), path)
    body = tree.body
    if class_name is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == class_name).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == func_name)
    node.decorator_list = []
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    glob = {"np": np, "torch": torch, "F": F, "Union": typing.Union, "Tuple": typing.Tuple, "List": typing.List}
    glob.update(ns)
    exec(code, glob)
    return glob[func_name], f"{rel}:{node.lineno}-{node.end_lineno}"


def extracted_reference_fixtures():
    import types
    # ---- distill_kl / l2_loss (other_loss.py:51-78); distill_kl carries a stray `self` first parameter
    kl, kl_src = ref_function("training/loss/other_loss.py", "distill_kl")
    l2, l2_src = ref_function("training/loss/other_loss.py", "l2_loss")
    for name, shape, T in (("distill_kl_c5_T1", (2, 5, 6, 6, 6), 1), ("distill_kl_c5_T4", (2, 5, 6, 6, 6), 4),
                           ("distill_kl_c1_T1", (2, 1, 6, 6, 6), 1), ("distill_kl_c1_T4", (2, 1, 6, 6, 6), 4)):
        g = gen(50)
        ys = (torch.randn(shape, generator=g) * 2).requires_grad_()
        yt = (torch.randn(shape, generator=g) * 2).requires_grad_()
        l = kl(None, ys, yt, T)
        l.backward()
        assert torch.equal(l.detach(), LO.distill_kl(ys.detach(), yt.detach(), T)), "oracle distill_kl != reference"
        save(name + ".npz", source="reference " + kl_src, T=T, ys=ys, yt=yt, loss=l, gys=ys.grad, gyt=yt.grad)
    for name, T in (("feat_kl_T1", 1), ("feat_kl_T4", 4)):
        g = gen(51)
        a = torch.randn(2, 32, 5, 6, 4, generator=g, requires_grad=True)
        b = torch.randn(2, 32, 5, 6, 4, generator=g, requires_grad=True)
        l = l2(a, b, True, T)
        l.backward()
        assert torch.equal(l.detach(), LO.l2_loss(a.detach(), b.detach(), True, T)), "oracle l2_loss != reference"
        save(name + ".npz", source="reference " + l2_src, T=T, a=a, b=b, loss=l, ga=a.grad, gb=b.grad)
    g = gen(52)
    a = torch.randn(2, 8, 5, 6, 4, generator=g, requires_grad=True)
    b = torch.randn(2, 8, 5, 6, 4, generator=g, requires_grad=True)
    l = l2(a, b, False)
    l.backward()
    assert torch.equal(l.detach(), LO.l2_loss(a.detach(), b.detach(), False))
    save("l2_loss_plain.npz", source="reference " + l2_src, a=a, b=b, loss=l, ga=a.grad, gb=b.grad)

    # ---- per-rank batch split (nnUNetTrainer.py:304-349): the method on a stub trainer, torch.distributed stubbed
    cases = {}
    for gb in range(2, 17):
        for ws in range(1, 9):
            if gb < ws:
                continue
            bs, ov = [], []
            for rank in range(ws):
                dist_stub = types.SimpleNamespace(get_world_size=lambda ws=ws: ws, get_rank=lambda rank=rank: rank)
                fn, split_src = ref_function("training/nnUNetTrainer/nnUNetTrainer.py", "_set_batch_size_and_oversample",
                                             "nnUNetTrainer", dist=dist_stub, print=lambda *a, **k: None)
                me = types.SimpleNamespace(is_ddp=True, oversample_foreground_percent=0.33,
                                           configuration_manager=types.SimpleNamespace(batch_size=gb))
                fn(me)
                bs.append(int(me.batch_size))
                ov.append(float(me.oversample_foreground_percent))
            o_bs, o_ov = SO.ddp_batch_split(gb, ws)
            assert o_bs == bs and o_ov == ov, (gb, ws, bs, o_bs, ov, o_ov)
            cases[f"{gb}_{ws}"] = {"batch_sizes": bs, "oversample": ov}
    json.dump({"source": "reference " + split_src + " (method executed on a stub trainer per rank)", "cases": cases},
              open(os.path.join(OUT, "ddp_split.json"), "w"))

    # ---- sliding-window step placement and Gaussian importance map (sliding_window_prediction.py:10-56)
    steps, steps_src = ref_function("inference/sliding_window_prediction.py", "compute_steps_for_sliding_window")
    from scipy.ndimage import gaussian_filter
    gauss, gauss_src = ref_function("inference/sliding_window_prediction.py", "compute_gaussian",
                                    gaussian_filter=gaussian_filter)
    sw = []
    for image, tile, step in (((110,), (64,), 0.5), ((128, 128, 128), (128, 128, 128), 0.5),
                              ((192, 256, 256), (128, 128, 128), 0.5), ((133, 160, 201), (64, 96, 128), 0.5),
                              ((70, 70, 70), (64, 64, 64), 1.0), ((100, 300, 77), (48, 160, 64), 0.25),
                              ((65, 64, 200), (64, 64, 64), 0.75)):
        sw.append({"image_size": list(image), "tile_size": list(tile), "tile_step_size": step,
                   "steps": [[int(v) for v in ax] for ax in steps(image, tile, step)]})
    gm = []
    for tile in ((8, 8, 8), (6, 10, 12), (16, 16, 16)):
        m = gauss(tile, 1. / 8, 1000, torch.float32, torch.device("cpu"))
        gm.append({"tile_size": list(tile), "sigma_scale": 0.125, "value_scaling_factor": 1000,
                   "map": [float(v) for v in m.reshape(-1)]})
    json.dump({"source": "reference " + steps_src + " and " + gauss_src, "steps": sw, "gaussian": gm},
              open(os.path.join(OUT, "sw_steps.json"), "w"))

    # ---- patch sampler (base_data_loader.py:64-139): get_bbox on a stub loader, numpy global RNG seeded per case
    bbox, bbox_src = ref_function("training/dataloading/base_data_loader.py", "get_bbox", "nnUNetDataLoaderBase",
                                  print=lambda *a, **k: None)
    rng = np.random.default_rng(11)
    out = []

    def locs(shape, n):
        if n == 0:
            return np.zeros((0, 4), dtype=np.int64)
        return np.stack([np.zeros(n, dtype=np.int64)] + [rng.integers(0, s, n) for s in shape], 1)

    scen = []
    for shape, patch, pad_extra in (((40, 50, 60), (32, 32, 32), (0, 0, 0)), ((20, 70, 33), (32, 48, 32), (0, 0, 0)),
                                    ((30, 30, 30), (32, 32, 32), (0, 0, 0)), ((64, 64, 64), (32, 40, 48), (5, 4, 3)),
                                    ((17, 90, 41), (16, 64, 48), (0, 7, 0))):
        for force_fg in (False, True):
            for variant in ("plain", "empty_class", "all_empty", "overwrite", "region_key", "ignore"):
                scen.append((shape, patch, pad_extra, force_fg, variant))
    for ci, (shape, patch, pad_extra, force_fg, variant) in enumerate(scen):
        all_labels = (1, 2, 3)
        cl = {1: locs(shape, 40), 2: locs(shape, 25), 3: locs(shape, 7)}
        overwrite, has_ignore = None, False
        if variant == "empty_class":
            cl[2] = locs(shape, 0)
        elif variant == "all_empty":
            cl = {k: locs(shape, 0) for k in cl}
        elif variant == "overwrite":
            overwrite = 3
        elif variant == "region_key":
            cl[all_labels] = locs(shape, 30)  # the annotated-classes tuple key (removed when other classes exist)
        elif variant == "ignore":
            has_ignore = True
            cl[all_labels] = locs(shape, 30 if ci % 2 else 0)
        me = types.SimpleNamespace(need_to_pad=np.array(pad_extra, dtype=int), patch_size=patch,
                                   has_ignore=has_ignore, annotated_classes_key=all_labels)
        seed = 1000 + ci
        np.random.seed(seed)
        lbs, ubs = bbox(me, np.array(shape), force_fg, cl, overwrite)
        tail = float(np.random.uniform())  # the RNG state after the call: pins the NUMBER and ORDER of draws
        out.append({"shape": list(shape), "patch_size": list(patch), "need_to_pad": list(pad_extra),
                    "force_fg": force_fg, "has_ignore": has_ignore, "overwrite_class": overwrite, "seed": seed,
                    "annotated_classes_key": list(all_labels),
                    "class_locations": [[(list(k) if isinstance(k, tuple) else k), v.tolist()] for k, v in cl.items()],
                    "bbox_lbs": [int(v) for v in lbs], "bbox_ubs": [int(v) for v in ubs], "rng_tail": tail})
    json.dump({"source": "reference " + bbox_src + " (method executed on a stub loader)", "cases": out},
              open(os.path.join(OUT, "get_bbox.json"), "w"))
    print("wrote distill_kl_* feat_kl_* l2_loss_plain ddp_split.json sw_steps.json get_bbox.json (reference-extracted)")


# ------------------------------------------------------------------ oracle-generated (reference not importable)
def loss_fixtures():
    # DC+CE on one level, batch_dice False/True
    for bd in (False, True):
        g = gen(53)
        logits = (torch.randn(2, 5, 8, 8, 8, generator=g) * 2).requires_grad_()
        tgt = torch.round(torch.rand(2, 1, 8, 8, 8, generator=g) * 4)
        lf = LO.build_loss(1, batch_dice=bd, deep_supervision=False)
        l = lf(logits, tgt)
        l.backward()
        tp, fp, fn = LO.validation_counts(logits.detach(), tgt)
        per, mean = LO.dice_from_counts(tp, fp, fn)
        save(f"dc_ce_batchdice{int(bd)}.npz", source="oracle restatement (App. B; nnUNetTrainer.py:351-375)",
             logits=logits, target=tgt, loss=l, glogits=logits.grad, tp=tp, fp=fp, fn=fn, dice_per_class=per,
             dice_mean=mean)
    # soft clDice
    g = gen(54)
    p = torch.rand(2, 1, 10, 12, 9, generator=g).requires_grad_()
    t = (F.avg_pool3d(torch.rand(2, 1, 10, 12, 9, generator=g), 3, 1, 1) > 0.5).float()
    l = LO.soft_cldice(p, t, 3, 1.0)
    l.backward()
    save("soft_cldice.npz", source="oracle (reference soft_skel + clDice_metric.py:7-36 formula)", iter=3, smooth=1.0,
         pred=p, target=t, loss=l, gpred=p.grad)


class _RefStepHarness:
    """The reference's own step code on a stub trainer (nothing of its text is stored; see ref_function):
    * `nnUNetTrainer.train_step` (nnUNetTrainer.py:888-925) -- executed unchanged with `autocast` / `dummy_context`
      injected (the device is cpu, so the reference takes its dummy_context branch) and its debug prints muted;
    * `nnUNetTrainer.configure_optimizers` (:473-477) -- executed unchanged; `PolyLRScheduler` is a shim whose `step`
      is the reference's own `PolyLRScheduler.step` (polylr.py:13-20; the ctor passes a `verbose` positional that
      torch 2.10's LRScheduler no longer takes, an ordinary TypeError);
    * `nnUNetTrainer._get_deep_supervision_scales` (:296-302);
    * `UNetDecoder_return_last_fea.forward` (UNetDecoder.py:1001-1027) -- the fork's plain decoder forward (no
      attention insert, `attn_skip = skips[-1]`), executed with `self` = the oracle decoder module, which holds the
      `stages / transpconvs / seg_layers / deep_supervision` attributes the body reads.
    The encoder blocks, the loss classes and the MVD composite stay the oracle's restatement (un-vendored
    `dynamic_network_architectures`, three loss files missing from the fork: SURVEY 8c)."""
    TR = "training/nnUNetTrainer/nnUNetTrainer.py"

    def __init__(self):
        import contextlib
        import types
        quiet = lambda *a, **k: None
        self.types = types
        self.train_step, self.src_step = ref_function(self.TR, "train_step", "nnUNetTrainer", autocast=torch.autocast,
                                                      dummy_context=contextlib.nullcontext, print=quiet)
        poly = ref_module("training/lr_scheduler/polylr.py", "ref_poly_step")

        class PolyShim:
            def __init__(sh, optimizer, initial_lr, max_steps, exponent=0.9, current_step=None):
                sh.optimizer, sh.initial_lr, sh.max_steps, sh.exponent, sh.ctr = optimizer, initial_lr, max_steps, exponent, 0
            step = poly.PolyLRScheduler.step

        self.configure_optimizers, self.src_opt = ref_function(self.TR, "configure_optimizers", "nnUNetTrainer",
                                                               PolyLRScheduler=PolyShim)
        self.ds_scales, self.src_ds = ref_function(self.TR, "_get_deep_supervision_scales", "nnUNetTrainer")
        self.dec_forward, self.src_dec = ref_function("training/my_network/UNetDecoder.py", "forward",
                                                      "UNetDecoder_return_last_fea", print=quiet)

    def network(self, net):
        """encoder (oracle) -> the reference's decoder forward.  Returns a callable with the `parameters()` the trainer
        methods use and a `.last` slot holding the outputs of the latest call."""
        h = self

        class Net:
            last = None

            def __call__(s, x):
                skips = net.encoder(x)
                r = h.dec_forward(net.decoder, skips, skips[-1])
                s.last = r
                return r[0] if net.decoder.deep_supervision else r

            def parameters(s):
                return net.parameters()
        return Net()

    def trainer(self, net, loss_fn, num_epochs=200):
        me = self.types.SimpleNamespace(network=self.network(net), loss=loss_fn, device=torch.device("cpu"),
                                        grad_scaler=None, initial_lr=1e-2, weight_decay=3e-5, num_epochs=num_epochs)
        me.optimizer, me.lr_scheduler = self.configure_optimizers(me)
        return me

    def scales(self, strides, enabled=True):
        me = self.types.SimpleNamespace(enable_deep_supervision=enabled,
                                        configuration_manager=self.types.SimpleNamespace(pool_op_kernel_sizes=strides))
        return self.ds_scales(me)


def _assert_same_state(net_a, net_b, what):
    for (n, a), (_, b) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), (what, n)
        if a.grad is not None or b.grad is not None:
            assert torch.equal(a.grad, b.grad), (what, "grad", n)


def ds_scales_fixture(H):
    cases = []
    for strides in ([[1, 1, 1]] + [[2, 2, 2]] * 5, [[1, 1, 1]] + [[2, 2, 2]] * 4, [[1, 1, 1], [2, 2, 2], [2, 2, 2]],
                    [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]], [[1, 1, 1]]):
        sc = H.scales(strides)
        assert [list(map(float, i)) for i in sc] == [list(map(float, i)) for i in SO.ds_scales(strides)]
        cases.append({"pool_op_kernel_sizes": strides, "scales": [[float(v) for v in i] for i in sc]})
    assert H.scales([[1, 1, 1], [2, 2, 2]], enabled=False) is None
    # configure_optimizers (:473-477) on a stub trainer: the SGD hyper-parameters and the schedule it attaches
    me = H.types.SimpleNamespace(network=torch.nn.Linear(2, 2), initial_lr=1e-2, weight_decay=3e-5, num_epochs=50)
    opt, sch = H.configure_optimizers(me)
    g0 = opt.param_groups[0]
    hp = {k: g0[k] for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov")}
    lrs = []
    for e in range(50):
        sch.step(e)
        lrs.append(opt.param_groups[0]['lr'])
    json.dump({"source": "reference " + H.src_ds + " and " + H.src_opt + " (methods executed on a stub trainer)",
               "cases": cases, "disabled": None,
               "optimizer": {"class": type(opt).__name__, "hyper": hp, "num_epochs": 50, "lrs": lrs}},
              open(os.path.join(OUT, "ds_scales.json"), "w"))
    print("wrote ds_scales.json")


def unet_step_fixture():
    """Tiny end-to-end: 3 stages [8,16,32], C_in=2, K=3, 16^3, B=2 (App. E `unet_tiny_step`).  Round 3: the three steps
    are run by the REFERENCE's train_step / configure_optimizers / decoder forward (_RefStepHarness); the oracle
    restatement runs beside it on a copy of the network and must agree bit for bit at every step."""
    import copy
    H = _RefStepHarness()
    ds_scales_fixture(H)
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    net = UO.build_plainconv_unet(2, 3, 3, strides, features_per_stage=[8, 16, 32], seed=0)
    # non-trivial norm affine + biases so that every parameter's gradient is exercised
    g = gen(60)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("norm.weight"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    batch = SO.synthetic_batch(2, 2, (16, 16, 16), strides, num_classes=3, seed=1234)
    loss_fn = LO.build_loss(len(batch['target']))
    twin = copy.deepcopy(net)                       # the oracle restatement, same start
    opt_twin = SO.make_optimizer(twin.parameters())
    me = H.trainer(net, loss_fn)
    g0 = me.optimizer.param_groups[0]
    assert isinstance(me.optimizer, torch.optim.SGD) and (g0['lr'], g0['momentum'], g0['weight_decay'], g0['nesterov']) == \
        (1e-2, 0.99, 3e-5, True)
    out = {"source": f"reference {H.src_step}, {H.src_opt} (methods executed on a stub trainer; cpu branch), network = "
                     f"oracle encoder blocks + reference {H.src_dec}; loss = oracle restatement (App. B)",
           "strides": strides, "features": [8, 16, 32], "num_classes": 3, "data": batch['data']}
    for i, t in enumerate(batch['target']):
        out[f"target{i}"] = t
    for k, v in sd0.items():
        out["sd0/" + k] = v
    for step in range(3):
        r = H.train_step(me, batch)
        l, (logits, feat) = r['loss'], me.network.last
        lo, logits_o, gn = SO.train_step(twin, loss_fn, opt_twin, batch)
        assert l.dtype == np.float32 and np.array_equal(l, lo), (step, l, lo)
        assert all(torch.equal(a, b) for a, b in zip(logits, logits_o))
        _assert_same_state(net, twin, f"step {step}")
        out[f"loss{step}"] = l
        out[f"gradnorm{step}"] = gn   # norm before clipping (the reference drops clip_grad_norm_'s return value; the
        #                               twin's post-clip gradients equal the reference's, asserted above)
        if step == 0:
            for i, lg in enumerate(logits):
                out[f"logits{i}"] = lg.detach()
            out["feat"] = feat.detach()
            for n, p in net.named_parameters():
                out["grad0/" + n] = p.grad.clone()
        for n, p in net.named_parameters():
            if step in (0, 2):
                out[f"sd{step + 1}/" + n] = p.detach().clone()
    tp, fp, fn = LO.validation_counts(net(batch['data'])[0].detach(), batch['target'][0])
    out["val_tp"], out["val_fp"], out["val_fn"] = tp, fp, fn
    save("unet_tiny_step.npz", **out)

    # the decoder forward alone: DS on (list + last feature) and DS off (bare tensor), reference vs restatement
    with torch.no_grad():
        skips = net.encoder(batch['data'])
        r_on = H.dec_forward(net.decoder, skips, skips[-1])
        o_on = net.decoder(skips, True)
        assert all(torch.equal(a, b) for a, b in zip(r_on[0], o_on[0])) and torch.equal(r_on[1], o_on[1])
        net.decoder.deep_supervision = False
        r_off = H.dec_forward(net.decoder, skips, skips[-1])
        o_off = net.decoder(skips)
        net.decoder.deep_supervision = True
        assert torch.is_tensor(r_off) and torch.equal(r_off, o_off) and torch.equal(r_off, r_on[0][0])
    dec = {"source": "reference " + H.src_dec + " (executed with self = the oracle decoder module)",
           "strides": strides, "features": [8, 16, 32], "num_classes": 3, "feat": r_on[1], "logits_ds_off": r_off}
    for i, sk in enumerate(skips):
        dec[f"skip{i}"] = sk
    for i, lg in enumerate(r_on[0]):
        dec[f"logits{i}"] = lg
    for k, v in net.decoder.state_dict().items():
        if not k.startswith("encoder."):
            dec["sd/" + k] = v
    save("decoder_forward.npz", **dec)

    # anisotropic / no-DS variant: strides (1,2,2) then (2,2,2); checks per-axis stride handling
    strides = [[1, 1, 1], [1, 2, 2], [2, 2, 2]]
    net = UO.build_plainconv_unet(1, 2, 3, strides, features_per_stage=[4, 8, 16], seed=1)
    g = gen(61)
    data = torch.rand(1, 1, 8, 16, 12, generator=g)
    outs = H.network(net)(data)
    assert all(torch.equal(a, b) for a, b in zip(outs, net(data)))
    o = {"source": "oracle encoder blocks + reference " + H.src_dec, "strides": strides, "features": [4, 8, 16],
         "num_classes": 2, "data": data}
    for k, v in net.state_dict().items():
        o["sd0/" + k] = v
    for i, lg in enumerate(outs):
        o[f"logits{i}"] = lg.detach()
    save("unet_aniso_fwd.npz", **o)


def mvd_step_fixture():
    """Dual-branch step.  The composite loss is the build's restatement (MVDTrainer.py:879-925 is not executable: undefined
    names); each branch's decoder runs through the REFERENCE's UNetDecoder_return_last_fea.forward, the
    optimizer comes from the reference's configure_optimizers, and a twin on the oracle's own decoder must agree exactly."""
    import copy
    H = _RefStepHarness()
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    n1 = UO.build_plainconv_unet(2, 4, 3, strides, features_per_stage=[8, 16, 32], seed=2)
    n2 = UO.build_plainconv_unet(2, 4, 3, strides, features_per_stage=[8, 16, 32], seed=3)
    net = UO.DualBranchNet(n1, n2)
    twin = copy.deepcopy(net)
    batch = SO.synthetic_batch(2, 2, (16, 16, 16), strides, num_classes=4, seed=77)
    loss_fn = LO.build_loss(len(batch['target']))

    class RefDecoders:
        """(logits_1, logits_2, feat_1, feat_2) with both decoders run by the reference's forward."""
        def __call__(s, x):
            sk1, sk2 = n1.encoder(x), n2.encoder(x)
            r1, r2 = H.dec_forward(n1.decoder, sk1, sk1[-1]), H.dec_forward(n2.decoder, sk2, sk2[-1])
            return r1[0], r2[0], r1[1], r2[1]

        def parameters(s):
            return net.parameters()

    me = H.types.SimpleNamespace(network=net, initial_lr=1e-2, weight_decay=3e-5, num_epochs=200)
    opt, _sch = H.configure_optimizers(me)
    opt_twin = SO.make_optimizer(twin.parameters())
    out = {"source": "oracle restatement of MVDTrainer.py:879-925 (see oracle/step_oracle.py); decoders run by reference "
                     + H.src_dec + ", optimizer from reference " + H.src_opt, "strides": strides,
           "features": [8, 16, 32], "num_classes": 4, "data": batch['data'], "skel_iter": 3}
    for i, t in enumerate(batch['target']):
        out[f"target{i}"] = t
    for k, v in net.state_dict().items():
        out["sd0/" + k] = v.clone()
    l, outs, gn = SO.mvd_train_step(RefDecoders(), loss_fn, opt, batch, use_topo=True, skel_iter=3, feat_kl=True)
    lt, _o, gnt = SO.mvd_train_step(twin, loss_fn, opt_twin, batch, use_topo=True, skel_iter=3, feat_kl=True)
    assert np.array_equal(l, lt) and gn == gnt
    _assert_same_state(net, twin, "mvd step")
    out["loss0"], out["gradnorm0"] = l, gn
    for n, p in net.named_parameters():
        out["grad0/" + n] = p.grad.clone()
        out["sd1/" + n] = p.detach().clone()
    save("mvd_tiny_step.npz", **out)


def persistence_fixtures():
    m = build_ref.load()
    assert m is not None, "run `python oracle/build_ref.py` first"
    srt = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]
    cases = []
    rng = np.random.default_rng(7)

    def run_ref(cells, f, maxdim=0):
        s = m.SimplicialComplex()
        for c in cells:
            s.append(list(c))
        s.initialize()
        s.extendFloat(torch.from_numpy(f.reshape(-1).copy()))
        hom = m.persistenceForwardHom(s, maxdim, 0)[0].detach().numpy().copy()
        s2 = m.SimplicialComplex()
        for c in cells:
            s2.append(list(c))
        s2.initialize()
        s2.extendFloat(torch.from_numpy(f.reshape(-1).copy()))
        coh = m.persistenceForwardCohom(s2, maxdim)[0].detach().numpy().copy()
        assert np.array_equal(srt(hom), srt(coh))
        return srt(hom)

    def grid_cells(D, H, W, conn):
        cells = [[i] for i in range(D * H * W)]
        offs = {6: [(0, 0, 1), (0, 1, 0), (1, 0, 0)],
                14: [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)]}[conn]
        for z in range(D):
            for y in range(H):
                for x in range(W):
                    for dz, dy, dx in offs:
                        zz, yy, xx = z + dz, y + dy, x + dx
                        if zz < D and yy < H and xx < W:
                            cells.append([(z * H + y) * W + x, (zz * H + yy) * W + xx])
        return cells

    for (D, H, W), conn, ties in (((1, 1, 5), 6, False), ((1, 1, 64), 6, False), ((1, 4, 4), 14, False),
                                  ((1, 8, 8), 14, True), ((1, 16, 16), 14, False), ((4, 4, 4), 6, False),
                                  ((5, 6, 7), 6, True), ((4, 5, 6), 14, False), ((6, 6, 6), 6, False)):
        f = rng.random((D, H, W)).astype(np.float32)
        if ties:
            f = np.round(f * 4) / 4
        dgm = run_ref(grid_cells(D, H, W, conn), f)
        b, d, _ = cc_oracle.h0_persistence(f, conn)
        assert np.array_equal(dgm, srt(np.stack([b, d], 1))), "C oracle differs from the reference C++"
        cases.append({"shape": [D, H, W], "conn": conn, "f": f.reshape(-1).tolist(),
                      "dgm0_sorted": [[float(a), (None if np.isinf(c) else float(c))] for a, c in dgm]})
    # the survey's 5-vertex line known answer
    f = np.array([0.0, 3.0, 0.5, 2.0, 1.0], dtype=np.float32).reshape(1, 1, 5)
    dgm = run_ref(grid_cells(1, 1, 5, 6), f)
    cases.append({"shape": [1, 1, 5], "conn": 6, "f": f.reshape(-1).tolist(),
                  "dgm0_sorted": [[float(a), (None if np.isinf(c) else float(c))] for a, c in dgm]})
    json.dump({"source": "reference C++ persistence (hom.cpp + cohom.cpp) compiled into oracle/_ref", "cases": cases},
              open(os.path.join(OUT, "persistence_grid.json"), "w"))
    # connected components: #infinite H0 bars of the thresholded mask == CC count (reference-pinned count)
    cc_cases = []
    for (D, H, W), conn in (((6, 7, 8), 6), ((5, 5, 5), 14), ((1, 12, 12), 6)):
        mask = (rng.random((D, H, W)) > 0.55)
        labels, n = cc_oracle.cc_label(mask, conn)
        # reference: sub-complex on mask vertices -> count essential bars
        idx = -np.ones(mask.size, dtype=np.int64)
        idx[mask.reshape(-1)] = np.arange(mask.sum())
        cells = [c for c in grid_cells(D, H, W, conn) if all(mask.reshape(-1)[v] for v in c)]
        cells = [[int(idx[v]) for v in c] for c in cells]
        dgm = run_ref(cells, np.zeros(int(mask.sum()), dtype=np.float32))
        assert int(np.isinf(dgm[:, 1]).sum()) == n
        cc_cases.append({"shape": [D, H, W], "conn": conn, "mask": mask.reshape(-1).astype(int).tolist(),
                         "labels": labels.reshape(-1).tolist(), "count": n})
    json.dump({"source": "labels: oracle/cc_oracle.c (canonical = 1 + min linear index); count cross-checked "
                         "against the reference C++ (essential H0 bars)", "cases": cc_cases},
              open(os.path.join(OUT, "cc_label.json"), "w"))
    print("wrote persistence_grid.json cc_label.json")


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "extracted":  # only the fixtures compiled out of reference function bodies
        extracted_reference_fixtures()
        sys.exit(0)
    conv_fixtures()
    instnorm_fixtures()
    reference_fixtures()
    extracted_reference_fixtures()
    loss_fixtures()
    unet_step_fixture()
    mvd_step_fixture()
    persistence_fixtures()
