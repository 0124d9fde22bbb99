"""CPU tests of the host-side mirror of the reference's plugin interface (no kernel is launched here)."""
import inspect
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

from conftest import GOLDEN, ROOT
from multimodal_mvd_seg_amd import losses, network, optim, parallel, trainer
from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO

DATASET_JSON = {"channel_names": {"0": "T1", "1": "T2", "2": "TOF", "3": "FLAIR"},
                "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}


def build(cfg="cfg2"):
    c = UO.CONFIGS[cfg]
    plans = trainer.make_plans(c["patch"], c["strides"])
    pm = trainer.PlansManager(plans)
    cm = pm.get_configuration("3d_fullres")
    ds = dict(DATASET_JSON)
    ds["channel_names"] = {str(i): f"m{i}" for i in range(c["input_channels"])}
    return trainer.nnUNetTrainerMI355.build_network_architecture(pm, ds, cm, c["input_channels"], True), c


def test_state_dict_keys_and_shapes_equal_the_reference_naming():
    net, c = build("cfg2")
    ora = UO.build_plainconv_unet(c["input_channels"], 5, c["n_stages"], c["strides"])
    sd, so = net.state_dict(), ora.state_dict()
    assert list(sd.keys()) == list(so.keys())
    for k in sd:
        assert sd[k].shape == so[k].shape, k
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in ora.parameters())
    # checkpoints interchange both ways
    net.load_state_dict(so)
    ora.load_state_dict(net.state_dict())
    # '.seg_layers.' is special-cased by load_pretrained_weights.py:21-23
    assert any(".seg_layers." in k for k in sd)
    assert net.decoder.deep_supervision is True and net.decoder.encoder is net.encoder


def test_constructor_signature_matches_plainconvunet_call_site():
    # get_network_from_plans.py:70-83 passes exactly these keyword arguments
    names = set(inspect.signature(network.MI355PlainConvUNet.__init__).parameters)
    for kw in ("input_channels", "n_stages", "features_per_stage", "conv_op", "kernel_sizes", "strides", "num_classes",
               "deep_supervision", "n_conv_per_stage", "n_conv_per_stage_decoder", "conv_bias", "norm_op",
               "norm_op_kwargs", "dropout_op", "dropout_op_kwargs", "nonlin", "nonlin_kwargs"):
        assert kw in names, kw
    enc = build("cfg1")[0].encoder
    for attr in ("output_channels", "strides", "kernel_sizes", "conv_op", "conv_bias", "norm_op", "norm_op_kwargs",
                 "dropout_op", "dropout_op_kwargs", "nonlin", "nonlin_kwargs"):  # read at UNetDecoder.py:39-65
        assert hasattr(enc, attr), attr
    assert enc.output_channels == [32, 64, 128, 256, 320]


def test_trainer_surface_signatures():
    T = trainer.nnUNetTrainerMI355
    assert list(inspect.signature(T.__init__).parameters) == \
        ["self", "plans", "configuration", "fold", "dataset_json", "unpack_dataset", "device", "specified_cfg"]
    assert list(inspect.signature(T.build_network_architecture).parameters) == \
        ["plans_manager", "dataset_json", "configuration_manager", "num_input_channels", "enable_deep_supervision"]
    assert isinstance(inspect.getattr_static(T, "build_network_architecture"), staticmethod)
    for m in ("initialize", "_build_loss", "configure_optimizers", "train_step", "validation_step",
              "set_deep_supervision_enabled", "_get_deep_supervision_scales", "_set_batch_size_and_oversample"):
        assert callable(getattr(T, m))
    with pytest.raises(RuntimeError, match="no CPU path"):
        T(trainer.make_plans((16, 16, 16), [[1, 1, 1], [2, 2, 2]]), "3d_fullres", 0, DATASET_JSON,
          device=torch.device("cpu"))


def test_network_refuses_cpu_input():
    net, c = build("cfg1")
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 1, 16, 16, 16))


def test_he_init_applies_to_hip_modules():
    net, _ = build("cfg1")
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
            assert float(m.bias.abs().max()) == 0.0
    w = net.encoder.stages[1][0].convs[1].conv.weight  # 64 -> 64, fan_in = 64*27
    assert abs(float(w.std()) - np.sqrt(2 / (1 + 1e-4) / (64 * 27))) < 2e-3


def test_ds_scales_and_weights_follow_the_reference():
    c = UO.CONFIGS["cfg2"]
    t = trainer.nnUNetTrainerMI355.__new__(trainer.nnUNetTrainerMI355)
    t.enable_deep_supervision = True
    t.configuration_manager = trainer.PlansManager(trainer.make_plans(c["patch"], c["strides"])).get_configuration(
        "3d_fullres")
    assert np.allclose(t._get_deep_supervision_scales(), SO.ds_scales(c["strides"]))
    assert len(t._get_deep_supervision_scales()) == 5
    assert np.allclose(losses.ds_weights(5), LO.ds_weights(5))
    # the reference method itself, executed on a stub trainer (tests/golden/ds_scales.json)
    d = json.load(open(os.path.join(GOLDEN, "ds_scales.json")))
    for c in d["cases"]:
        t.configuration_manager = trainer.PlansManager(trainer.make_plans(
            (64, 64, 64), c["pool_op_kernel_sizes"])).get_configuration("3d_fullres")
        assert [list(map(float, i)) for i in t._get_deep_supervision_scales()] == c["scales"]
    t.enable_deep_supervision = False
    assert t._get_deep_supervision_scales() is d["disabled"]


def test_configure_optimizers_hyperparameters_equal_the_reference_method():
    """nnUNetTrainer.configure_optimizers (:473-477) executed on a stub trainer -> ds_scales.json["optimizer"]: the fused
    optimizer the product's configure_optimizers builds carries the same hyper-parameters and schedule (host side only;
    the arithmetic of its step is a -m gpu test)."""
    o = json.load(open(os.path.join(GOLDEN, "ds_scales.json")))["optimizer"]
    t = trainer.nnUNetTrainerMI355.__new__(trainer.nnUNetTrainerMI355)
    t.network, t.initial_lr, t.weight_decay, t.num_epochs = nn.Linear(3, 2), 1e-2, 3e-5, o["num_epochs"]
    opt, sch = t.configure_optimizers()   # construction only: flat buffers on the parameters' (cpu) device, no kernel
    g0 = opt.param_groups[0]
    for k, v in o["hyper"].items():
        if k != "dampening":              # SGD's dampening is 0 in the reference call; the fused kernel has none
            assert g0[k] == v, k
    assert o["hyper"]["dampening"] == 0 and opt.max_grad_norm == 12   # clip_grad_norm_(..., 12), nnUNetTrainer.py:918/923
    for e, lr in enumerate(o["lrs"]):
        sch.step(e)
        assert opt.param_groups[0]["lr"] == lr


def test_plans_inheritance():
    plans = trainer.make_plans((16, 16, 16), [[1, 1, 1], [2, 2, 2]])
    plans["configurations"]["child"] = {"inherits_from": "3d_fullres", "batch_size": 7}
    cm = trainer.PlansManager(plans).get_configuration("child")
    assert cm.batch_size == 7 and cm.UNet_class_name == "PlainConvUNet"


def test_ddp_batch_split_matches_fixture():
    d = json.load(open(os.path.join(GOLDEN, "ddp_split.json")))["cases"]
    for key, v in d.items():
        gb, ws = map(int, key.split("_"))
        bs, ov = parallel.ddp_batch_split(gb, ws)
        assert bs == v["batch_sizes"] and np.allclose(ov, v["oversample"])


def test_polylr_matches_reference_fixture():
    d = json.load(open(os.path.join(GOLDEN, "polylr.json")))

    class Opt:
        param_groups = [{"lr": 0.0}]
    o = Opt()
    sch = optim.PolyLRScheduler(o, d["initial_lr"], d["max_steps"])
    assert o.param_groups[0]["lr"] == d["initial_lr"]
    for e, lr in enumerate(d["lrs"]):
        sch.step(e)
        assert o.param_groups[0]["lr"] == lr


def test_flat_params_rehome_parameters_without_changing_them():
    net, _ = build("cfg1")
    before = {k: v.clone() for k, v in net.state_dict().items()}
    fp = optim.FlatParams(list(net.parameters()))
    assert fp.numel >= sum(p.numel() for p in net.parameters())
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k
    # views: writing the flat buffer changes the module parameter and vice versa
    p0 = fp.params[0]
    fp.flat[fp.offsets[0]] = 123.0
    assert float(p0.view(-1)[0]) == 123.0
    assert all(o % 4 == 0 for o in fp.offsets)  # 16-byte alignment for the float4 kernels
    assert all(p.grad is not None and p.grad.data_ptr() == fp.grad[o:].data_ptr() for p, o in zip(fp.params, fp.offsets))
    # shared (aliased) parameters appear once
    assert len(fp.params) == len({id(p) for p in net.parameters()})


def test_mvd_dual_branch_contract():
    b1, _ = build("cfg1")
    b2, _ = build("cfg1")
    net = network.MVDDualBranchNet(b1, b2)
    assert net.do_ds is True
    net.do_ds = False
    assert b1.decoder.deep_supervision is False and b2.decoder.deep_supervision is False
    t = trainer.ContrastiveTrainerMI355.__new__(trainer.ContrastiveTrainerMI355)
    trainer.ContrastiveTrainerMI355.__init__(t, trainer.make_plans((16, 16, 16), [[1, 1, 1], [2, 2, 2]]), "3d_fullres", 0,
                                             DATASET_JSON) if torch.cuda.is_available() else None
    assert (trainer.ContrastiveTrainerMI355.__dict__["build_network_architecture"].__func__ is not
            trainer.nnUNetTrainerMI355.__dict__["build_network_architecture"].__func__)


def test_flat_params_direct_gradient_sink_semantics():
    """ops.py writes gradients straight into FlatParams.grad: a parameter's slice can be taken once per step (then
    autograd accumulation takes over), zero_grad() opens a new step, and listeners hear every completed gradient."""
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7))]
    fp = optim.FlatParams(ps)
    heard = []
    fp.listeners.append(heard.append)
    t0 = ps[0]._mvd_take_grad()
    assert t0 is not None and t0.data_ptr() == ps[0].grad.data_ptr() and t0.shape == ps[0].shape
    t0.fill_(2.0)
    ps[0]._mvd_grad_done()
    assert heard == [0]
    assert ps[0]._mvd_take_grad() is None          # second contribution in the same step: not taken
    assert ps[1]._mvd_take_grad() is not None
    assert float(fp.grad[:15].sum()) == 30.0       # the write landed in the flat buffer
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0.0
    assert ps[0]._mvd_take_grad() is not None      # new step
    assert ps[0].grad.data_ptr() == fp.grad.data_ptr()


# ------------------------------------------------------------------------------------------ device feed: host decisions
class _ToyDataset:
    def __init__(self, shapes, seed=0):
        rng = np.random.default_rng(seed)
        self.cases = {}
        for i, shp in enumerate(shapes):
            data = rng.standard_normal((2, *shp)).astype(np.float32)
            seg = (rng.random((1, *shp)) > 0.97).astype(np.int16) * rng.integers(1, 3, (1, *shp)).astype(np.int16)
            locs = {c: np.argwhere(seg == c) for c in (1, 2)}  # rows (0, z, y, x)
            self.cases[f"case{i}"] = (data, seg, {"class_locations": locs})

    def keys(self):
        return self.cases.keys()

    def load_case(self, k):
        return self.cases[k]


class _Labels:
    all_labels = [1, 2]
    has_ignore_label = False


def test_device_loader_bbox_rules_follow_reference():
    from multimodal_mvd_seg_amd.dataloading import DeviceDataLoader3D
    ds = _ToyDataset([(20, 24, 28), (9, 30, 12)])
    patch = (12, 16, 16)
    dl = DeviceDataLoader3D(ds, 4, patch, patch, _Labels(), oversample_foreground_percent=0.33, mirror_axes=(0, 1, 2),
                            device="cpu")
    # oversampling: the last round(B * 0.33)-ish samples are forced foreground (base_data_loader.py:46-50)
    assert [dl.get_do_oversample(j) for j in range(4)] == [False, False, False, True]
    np.random.seed(7)
    for _ in range(50):
        shape = (20, 24, 28)
        lbs, ubs = dl.get_bbox(shape, False, None)
        for i in range(3):
            assert 0 <= lbs[i] <= shape[i] - patch[i] and ubs[i] - lbs[i] == patch[i]
    # a volume smaller than the patch along z is padded on both sides: lb in [-(12-9)//2 .. ] (:66-74)
    seen = set()
    for _ in range(50):
        lbs, _ = dl.get_bbox((9, 30, 12), False, None)
        seen.add(lbs[0])
        assert -2 <= lbs[0] <= -1 and lbs[2] in (-2,) and 0 <= lbs[1] <= 14
    assert seen == {-2, -1}
    # forced foreground: the box contains the selected voxel, clamped at the lower bound only (:131-132)
    data, seg, props = ds.load_case("case0")
    for _ in range(50):
        state = np.random.get_state()
        lbs, ubs = dl.get_bbox(seg.shape[1:], True, props["class_locations"])
        np.random.set_state(state)
        eligible = [c for c in props["class_locations"] if len(props["class_locations"][c]) > 0]
        cls = eligible[np.random.choice(len(eligible))]
        vox = props["class_locations"][cls][np.random.choice(len(props["class_locations"][cls]))]
        assert lbs == [max(0, vox[i + 1] - patch[i] // 2) for i in range(3)]
    # plan: RNG order keys -> (oversample, bbox) per sample -> mirror draws; reproducible from the seed
    np.random.seed(3)
    p1 = dl.plan_batch()
    np.random.seed(3)
    p2 = dl.plan_batch()
    assert p1 == p2 and len(p1[0]) == 4 and all(0 <= f < 8 for f in p1[2])
    with pytest.raises(RuntimeError):
        dl.generate_train_batch(p1)  # no CPU path
    with pytest.raises(NotImplementedError):
        DeviceDataLoader3D(ds, 2, (16, 20, 20), patch, _Labels(), device="cpu")


# ------------------------------------------------------------------------- fixtures generated by reference functions
def test_get_bbox_equals_reference_method_fixture():
    """tests/golden/get_bbox.json: boxes AND the numpy RNG state after each call, produced by the reference's own
    nnUNetDataLoaderBase.get_bbox (tools/make_golden.py compiles the method out of base_data_loader.py:64-139)."""
    from multimodal_mvd_seg_amd.dataloading import DeviceDataLoader3D
    d = json.load(open(os.path.join(GOLDEN, "get_bbox.json")))
    assert d["source"].startswith("reference ")
    seen_fg = seen_pad = 0
    for c in d["cases"]:
        dl = DeviceDataLoader3D.__new__(DeviceDataLoader3D)  # host logic only: no dataset, no device
        dl.patch_size = tuple(c["patch_size"])
        dl.need_to_pad = np.array(c["need_to_pad"], dtype=int)
        dl.has_ignore = c["has_ignore"]
        dl.annotated_classes_key = tuple(c["annotated_classes_key"])
        cl = {(tuple(k) if isinstance(k, list) else k): np.array(v, dtype=np.int64).reshape(-1, 4)
              for k, v in c["class_locations"]}
        np.random.seed(c["seed"])
        lbs, ubs = dl.get_bbox(np.array(c["shape"]), c["force_fg"], cl, c["overwrite_class"])
        tail = float(np.random.uniform())
        assert [int(v) for v in lbs] == c["bbox_lbs"] and [int(v) for v in ubs] == c["bbox_ubs"], c
        assert tail == c["rng_tail"], "different number / order of RNG draws than the reference"
        seen_fg += c["force_fg"]
        seen_pad += any(s < p for s, p in zip(c["shape"], c["patch_size"]))
    assert seen_fg >= 20 and seen_pad >= 10


def _inference_host_namespace():
    src = open(os.path.join(ROOT, "multimodal_mvd_seg_amd", "inference.py")).read()
    ns = {}
    exec(compile(src.replace("from ._lib import call", "call = None"), "inference.py", "exec"), ns)
    return ns


def test_sliding_window_steps_and_gaussian_equal_reference_fixture():
    """tests/golden/sw_steps.json: compute_steps_for_sliding_window / compute_gaussian of the reference
    (sliding_window_prediction.py:10-56) executed in the build container."""
    ns = _inference_host_namespace()
    d = json.load(open(os.path.join(GOLDEN, "sw_steps.json")))
    assert d["source"].startswith("reference ")
    for c in d["steps"]:
        assert ns["compute_steps_for_sliding_window"](c["image_size"], c["tile_size"], c["tile_step_size"]) == c["steps"]
    for c in d["gaussian"]:
        g = ns["compute_gaussian"](c["tile_size"], c["sigma_scale"], c["value_scaling_factor"])
        ref = np.array(c["map"], dtype=np.float64).reshape(c["tile_size"])
        assert np.abs(g - ref).max() <= 2e-6 * ref.max()
    with pytest.raises(ValueError):
        ns["compute_steps_for_sliding_window"]((10, 10), (12, 8), 0.5)


def test_decoder_feature_map_size_known_answer():
    """compute_conv_feature_map_size (the planner's VRAM proxy, UNetDecoder.py:123-150 + the encoder's): values of the
    cfg-2 network, with and without deep supervision (frozen from the formula)."""
    net, _ = build("cfg2")
    assert int(net.compute_conv_feature_map_size((128, 128, 128))) == 458488320
    assert int(net.decoder.compute_conv_feature_map_size((128, 128, 128))) == 279861760
    net.decoder.deep_supervision = False
    assert int(net.decoder.compute_conv_feature_map_size((160, 160, 128))) == 434944000


def test_conv_block_refuses_plans_it_would_silently_mis_execute():
    from multimodal_mvd_seg_amd.network import ConvDropoutNormReLU
    ok = dict(norm_op=nn.InstanceNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True}, nonlin=nn.LeakyReLU,
              nonlin_kwargs={'inplace': True})
    blk = ConvDropoutNormReLU(nn.Conv3d, 4, 8, 3, 1, True, **ok)
    assert blk.nonlin.negative_slope == 0.01
    blk = ConvDropoutNormReLU(nn.Conv3d, 4, 8, 3, 1, True, **{**ok, "nonlin_kwargs": {'negative_slope': 0.2}})
    assert blk.nonlin.negative_slope == 0.2
    for bad in ({"norm_op": None}, {"norm_op": nn.BatchNorm3d}, {"nonlin": nn.ReLU}, {"nonlin": None},
                {"norm_op_kwargs": {'eps': 1e-5, 'affine': False}}):
        with pytest.raises(NotImplementedError):
            ConvDropoutNormReLU(nn.Conv3d, 4, 8, 3, 1, True, **{**ok, **bad})


def test_flat_layout_follows_execution_order_and_the_last_reduced_bucket_is_small():
    """DDP overlap (nnUNetTrainer.py:220-222 wraps the network in DDP, whose buckets fire in gradient-ready order): the flat
    gradient buffer is laid out in forward-execution order, so contiguous buckets cut from its end complete in backward
    order; the bucket reduced last (the first encoder stages, complete only when backward ends) is <= 4 MB."""
    net, c = build("cfg2")
    order = net.parameters_in_execution_order()
    assert len(order) == len(list(net.parameters())) and len({id(p) for p in order}) == len(order)
    names = {id(p): n for n, p in net.named_parameters()}
    seq = [names[id(p)] for p in order]
    first_dec = next(i for i, n in enumerate(seq) if n.startswith("decoder."))
    assert all(n.startswith("encoder.") for n in seq[:first_dec])
    # per decoder level: transposed conv, refining convs, seg layer -- contiguous, level 0 (bottleneck side) first
    lv = [(("transpconvs", "stages", "seg_layers").index(n.split(".")[1]), int(n.split(".")[2])) for n in seq[first_dec:]]
    levels = [l for _k, l in lv]
    assert levels == sorted(levels)
    for L in set(levels):
        kinds = [k for k, l in lv if l == L]
        assert kinds == sorted(kinds)
    fp = optim.FlatParams(order)
    assert [id(p) for p in fp.params] == [id(p) for p in order]
    red = parallel.BucketedGradReducer(fp, 25 * 1024 * 1024)
    sizes = [(e - s) * 4 for s, e, _ in red.buckets]
    assert sum(len(idx) for _s, _e, idx in red.buckets) == len(order)
    assert red.buckets[0][1] == fp.offsets[-1] + ((order[-1].numel() + 3) // 4) * 4   # the first bucket ends the buffer
    assert red.buckets[-1][0] == 0 and sizes[-1] <= 4 * 1024 * 1024                  # the last one starts it and is small
    assert all(s >= 25 * 1024 * 1024 for s in sizes[:-2])
    # buckets are contiguous and ordered from the end of the buffer
    for (s0, e0, _), (s1, e1, _) in zip(red.buckets[:-1], red.buckets[1:]):
        assert e1 == s0
