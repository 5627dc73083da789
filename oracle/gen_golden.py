"""Generates tests/golden/* by RUNNING the reference modules that import in the build container.

Run once, here (the GPU box has no /root/reference):   python oracle/gen_golden.py [post|lmffnet|metrics ...]

  * maskrefiner/modeling/mask_refiner/post_processing.py  (torch only; loaded by file path)
  * explicit_error_estimation/util.py                     (cv2 / segmentation_models_pytorch are imported at
    module top but unused by the two functions called; empty stand-in modules are registered for the import)
  * eval/preprocess_utils.py (normalize_depth), foreground_segmentation/lmffnet.py (torch only),
    eval/evaluation.py + eval/munkres.py (cv2 / eval/utilities.py are imported at module top but untouched with
    compute_boundary_stuff=False; empty stand-ins)

Only inputs and the reference's outputs are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in kw.items()})


def gen_lmffnet():
    """lmffnet_*.npz: logits of the reference LMFFNet(classes=3).eval() with quber_amd.lmff_arch's seeded weights."""
    from oracle import lmffnet_torch as L
    from quber_amd import lmff_arch
    m = load("ref_lmff", "foreground_segmentation/lmffnet.py")
    net = m.LMFFNet(classes=3).eval()
    sd = lmff_arch.init_state_dict(seed=0)
    ref_keys = [k for k in net.state_dict() if "num_batches" not in k]
    assert set(ref_keys) == set(sd) and all(tuple(net.state_dict()[k].shape) == sd[k].shape for k in ref_keys)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for (h, wd, seed) in [(64, 96, 1), (120, 160, 2)]:
        sc = synth.make_scene(seed, h, wd, 3)
        with torch.no_grad():
            ref = net(L.preprocess(sc["rgb"], sc["depth"]))
        save(f"lmffnet_{h}x{wd}", rgb=sc["rgb"], depth=sc["depth"], logits=ref[0].numpy(), seed=np.array(0))


def gen_metrics():
    """metrics_*.npz: dictionaries returned by the reference multilabel_metrics(compute_boundary_stuff=False);
    munkres_cases.npz / munkres_expected.json: assignments of the vendored eval/munkres.py on tie-heavy matrices."""
    import json
    sys.path.insert(0, os.path.join(REF, "eval"))
    for n in ("cv2", "utilities"):
        sys.modules.setdefault(n, types.ModuleType(n))
    m = load("ref_eval", "eval/evaluation.py")
    import munkres

    def labelmap(masks):
        out = np.zeros(masks.shape[1:], np.int32)
        for i, mk in enumerate(masks):
            out[mk != 0] = i + 1
        return out

    run = lambda p, g: m.multilabel_metrics(p, g, 0, 1, compute_boundary_stuff=False)
    cases = {}
    for name, (h, w, n, seed, drop, extra) in {"scene_a": (96, 128, 6, 1, 0, 0), "scene_b": (120, 160, 9, 2, 2, 0),
                                               "scene_c": (96, 128, 5, 3, 0, 3), "scene_big": (480, 640, 20, 4, 1, 1)}.items():
        r = np.random.default_rng(seed)
        gt_m, init_m = synth.make_masks(r, n, h, w)
        gt = labelmap(gt_m)
        pm = list(init_m)[drop:]          # missing predictions
        for _ in range(extra):            # spurious predictions
            z = np.zeros((h, w), bool)
            y0, x0 = int(r.integers(0, h - 12)), int(r.integers(0, w - 12))
            z[y0:y0 + 10, x0:x0 + 10] = True
            pm.append(z)
        pred = labelmap(np.array(pm))
        cases[name] = (pred, gt, run(pred, gt))
    z = np.zeros((32, 48), np.int32)
    one = z.copy()
    one[4:20, 5:30] = 3
    two = z.copy()
    two[0:10, 0:10] = 1
    two[20:30, 30:40] = 7
    cases["none_pred"] = (z, one, run(z, one))
    cases["none_gt"] = (one, z, run(one, z))
    cases["none_both"] = (z, z, run(z, z))
    cases["disjoint"] = (two, one, run(two, one))
    for k, (p, g, res) in cases.items():
        save("metrics_" + k, pred=p, gt=g,
             result=np.array(json.dumps({a: (None if b is None else float(b)) for a, b in res.items()})))
    mats = {f"m{t}": np.round(np.random.default_rng(100 + t).random((np.random.default_rng(t).integers(1, 8),
                                                                    np.random.default_rng(50 + t).integers(1, 8))), 1)
            for t in range(40)}
    save("munkres_cases", **mats)
    json.dump({k: munkres.Munkres().compute(x.max() - x.copy()) for k, x in mats.items()},
              open(os.path.join(OUT, "munkres_expected.json"), "w"))


def main():
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1:] or ["post", "lmffnet", "metrics"]
    if "lmffnet" in what:
        gen_lmffnet()
    if "metrics" in what:
        gen_metrics()
    if "post" in what:
        gen_post()
    if "raster" in what or "post" in what:
        gen_group_raster()
    if "configs" in what:
        gen_configs()


def gen_configs():
    """configs.json: what quber_amd/config.py makes of EVERY refiner yaml the reference ships (configs/**/mask-refiner-*.yaml,
    _BASE_ chains resolved against the reference tree): relative path -> {"arch": keyword arguments of arch.param_specs,
    "post": the post-processing constants, "input": the input switches} or {"unsupported": reason}.  Data about the
    reference's config zoo (no yaml text is stored); tests/test_oracle_golden.py asserts that the reader still derives
    the same and that the built set covers what it claims."""
    import glob
    import json
    import yaml
    from quber_amd import config as qconfig
    out = {}
    root = os.path.join(REF, "configs")
    for path in sorted(glob.glob(os.path.join(root, "**", "mask-refiner-*.yaml"), recursive=True)):
        rel = os.path.relpath(path, root)
        try:
            cfg = qconfig.validate(qconfig.merge_from_file(qconfig.get_cfg(), path))
        except qconfig.UnsupportedConfig as e:
            out[rel] = {"unsupported": str(e)}
            continue
        except FileNotFoundError as e:        # a _BASE_ that does not exist in the reference tree: the reference cannot load it either
            out[rel] = {"broken": "missing _BASE_ " + os.path.relpath(e.filename, root)}
            continue
        except yaml.YAMLError as e:           # not valid yaml (yacs uses the same parser)
            out[rel] = {"broken": "yaml syntax error at line %d" % (e.problem_mark.line + 1)}
            continue
        kw = qconfig.arch_kwargs(cfg)
        kw["hierarchy"] = [list(l) for l in kw["hierarchy"]]
        kw["fusion_target"] = list(kw["fusion_target"])
        pd = cfg.MODEL.PANOPTIC_DEEPLAB
        out[rel] = {"arch": kw,
                    "post": {"center_threshold": pd.CENTER_THRESHOLD, "nms_kernel": pd.NMS_KERNEL, "top_k": pd.TOP_K_INSTANCE,
                             "stuff_area": pd.STUFF_AREA, "res5_dilation": cfg.MODEL.RESNETS.RES5_DILATION},
                    "input": {"depth_on": bool(cfg.INPUT.DEPTH_ON), "rgb_on": bool(cfg.INPUT.RGB_ON), "format": cfg.INPUT.FORMAT,
                              "pixel_mean": [float(v) for v in cfg.MODEL.PIXEL_MEAN], "pixel_std": [float(v) for v in cfg.MODEL.PIXEL_STD]}}
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    ok = sum("arch" in v for v in out.values())
    broken = sum("broken" in v for v in out.values())
    print(f"wrote configs.json: {len(out)} refiner configs, {ok} load, {len(out) - ok - broken} unsupported, {broken} broken in the reference")
    why = {}
    for v in out.values():
        r = v.get("unsupported") or v.get("broken")
        if r:
            why[r] = why.get(r, 0) + 1
    for k, n in sorted(why.items(), key=lambda kv: -kv[1]):
        print(f"  {n:4d}  {k}")


def gen_group_raster():
    """group_raster_*.npz: K = 20 / 199 centres as the reference itself enumerates them - find_instance_center on a centre
    map with K isolated peaks (raster order) - fed to group_pixels with large random offsets.  The HIP path can only be
    driven through the centre map, so these are the large-K grouping cases it is checked on."""
    post = load("ref_post", "maskrefiner/modeling/mask_refiner/post_processing.py")
    for name, h, w, k, seed, sigma in (("20", 96, 128, 20, 31, 5.0), ("199", 384, 512, 199, 32, 20.0)):
        rng = np.random.default_rng(seed)
        c = np.full((h, w), 0.1, np.float32)
        pts = set()
        while len(pts) < k:
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            if all(max(abs(y - py), abs(x - px)) > 3 for py, px in pts):      # one peak per 7x7 NMS window
                pts.add((y, x))
        for i, (y, x) in enumerate(sorted(pts)):
            c[y, x] = 0.5 + 0.002 * ((i * 37) % k)                             # distinct values, all kept by top-200
        ct = torch.as_tensor(c).reshape(1, h, w)
        centers = post.find_instance_center(ct.clone(), threshold=0.3, nms_kernel=7, top_k=200)
        assert centers.shape[0] == k
        offsets = torch.as_tensor(rng.normal(0, sigma, (2, h, w)), dtype=torch.float32)
        out = post.group_pixels(centers, offsets)
        save("group_raster_" + name, center=c, centers=centers.numpy(), offsets=offsets.numpy(), out=out.numpy().astype(np.int32))


def gen_post():
    post = load("ref_post", "maskrefiner/modeling/mask_refiner/post_processing.py")
    for n in ("cv2", "segmentation_models_pytorch"):
        sys.modules.setdefault(n, types.ModuleType(n))
    util = load("ref_eee_util", "explicit_error_estimation/util.py")
    gen = util.PerturbedInputOffsetGenerator(sigma=10)

    # ---------------------------------------------------------------- encode (a1) + fg-union (a2)
    def enc_case(name, masks):
        masks = np.ascontiguousarray(masks.astype(np.uint8))
        out = gen([m for m in masks]).numpy()
        fg = util.masks_to_fg_mask(masks)
        save("encode_" + name, masks=np.packbits(masks != 0, axis=-1), shape=np.array(masks.shape),
             value=np.array([int(masks.max())]), out=out, fg=fg)

    h, w = 96, 128
    rng = np.random.default_rng(1)
    _, m8 = synth.make_masks(rng, 8, h, w)
    enc_case("n8_96x128", m8 * 255)
    # border-touching, an empty mask, overlapping pair, half-integer centroid, single pixel
    m = np.zeros((7, h, w), np.uint8)
    m[0, 0:10, 0:14] = 1            # top-left corner (window clipped)
    m[1, 80:96, 100:128] = 1        # bottom-right corner
    # m[2] stays empty (skipped)
    m[3, 30:60, 40:90] = 1
    m[4, 45:70, 60:110] = 1         # overlaps m[3]; later wins in the offset planes
    m[5, 10:12, 50:52] = 1          # centroid at (10.5, 50.5): banker's rounding
    m[6, 70, 5] = 1                 # single pixel
    enc_case("edge_96x128", m)
    enc_case("n1_96x128", m[3:4] * 255)
    sc = synth.make_scene(7, 480, 640, 20)
    enc_case("n20_480x640", sc["masks"])
    # fg-union wrap-around: 256 overlapping masks of value 255 sum to 0 mod 256
    mw = np.zeros((256, 8, 16), np.uint8)
    mw[:, 2:5, 3:9] = 255
    mw[:100, 6, 10] = 255
    save("fgunion_wrap", masks=mw, fg=util.masks_to_fg_mask(mw))

    # ---------------------------------------------------------------- normalize_depth (8f rank 1)
    pre = load("ref_pre", "eval/preprocess_utils.py")          # imports cv2 at module top; normalize_depth is numpy only
    rng = np.random.default_rng(21)
    d16 = rng.integers(0, 3000, (60, 80)).astype(np.uint16)
    d16[:3, :5] = 0
    d16[10, :9] = [249, 250, 251, 1499, 1500, 1501, 65535, 1, 875]
    save("depthnorm_u16", depth=d16, out=pre.normalize_depth(d16.copy()), lo=np.array(250.0), hi=np.array(1500.0))
    d32 = rng.uniform(0, 2.5, (60, 80)).astype(np.float32)
    d32[:2, :4] = 0
    d32[5, :6] = [0.25, 0.2500001, 1.5, 1.4999999, 0.7431, 1e-8]
    save("depthnorm_f32", depth=d32, out=pre.normalize_depth(d32.copy(), 0.25, 1.5), lo=np.array(0.25), hi=np.array(1.5))

    # ---------------------------------------------------------------- find_instance_center (a8)
    def ctr_case(name, c):
        c = torch.as_tensor(c, dtype=torch.float32).reshape(1, *c.shape[-2:])
        out = post.find_instance_center(c.clone(), threshold=0.3, nms_kernel=7, top_k=200)
        save("centers_" + name, center=c.numpy(), out=out.numpy())

    enc = gen([x for x in (m8 * 255).astype(np.uint8)]).numpy()
    ctr_case("scene_96x128", enc[0])
    ctr_case("plateau", np.where(np.add.outer(np.arange(h), np.arange(w)) % 37 < 2, 0.5, 0.1))
    ctr_case("const", np.full((h, w), 0.5))
    ctr_case("below", np.full((h, w), 0.29))
    c = np.full((h, w), 0.1, np.float32)
    c[10, 10] = 0.3                                         # exactly the threshold: not a centre
    c[20, 20] = np.nextafter(np.float32(0.3), np.float32(1))
    c[40, 40] = 0.9
    c[40, 44] = 0.9                                         # tie inside one NMS window: both survive
    c[0, 0] = 0.7
    c[h - 1, w - 1] = 0.8
    ctr_case("threshold_ties", c)
    rng = np.random.default_rng(3)
    ctr_case("random_many", rng.random((h, w)).astype(np.float32))          # > 200 candidates
    cc = np.full((120, 160), 0.1, np.float32)
    vals = 0.31 + 0.002 * np.arange(15 * 20)
    cc[4::8, 4::8] = vals.reshape(15, 20)                                     # 300 isolated peaks, distinct
    ctr_case("peaks300_distinct", cc)
    cc2 = cc.copy()
    cc2[4::8, 4::8] = np.round(vals.reshape(15, 20), 2)                       # many equal values at the cut
    ctr_case("peaks300_ties", cc2)
    cc3 = np.full((120, 160), 0.1, np.float32)
    cc3[4::8, 4::8].flat[:200] = 0.5 + 0.001 * np.arange(200)                 # exactly 200 candidates
    ctr_case("peaks200_exact", cc3)
    sc_enc = gen([x for x in sc["masks"]]).numpy()
    ctr_case("scene_480x640", sc_enc[0])

    # ---------------------------------------------------------------- group_pixels (a9)
    def grp_case(name, centers, offsets):
        centers = torch.as_tensor(centers, dtype=torch.int64)
        offsets = torch.as_tensor(offsets, dtype=torch.float32)
        out = post.group_pixels(centers, offsets)
        save("group_" + name, centers=centers.numpy(), offsets=offsets.numpy(), out=out.numpy().astype(np.int32))

    rng = np.random.default_rng(5)
    grp_case("k1", [[40, 60]], rng.normal(0, 3, (2, h, w)))
    grp_case("ties_int", [[10, 10], [10, 30], [30, 10], [30, 30]], np.zeros((2, h, w)))   # exact ties -> first index
    ck = np.stack([rng.integers(0, h, 20), rng.integers(0, w, 20)], 1)
    grp_case("k20", ck, rng.normal(0, 5, (2, h, w)))
    ck = np.stack([rng.integers(0, h, 199), rng.integers(0, w, 199)], 1)
    grp_case("k199", ck, rng.normal(0, 20, (2, h, w)))
    _, ctr_s, off_s = synth.fake_head_outputs(sc_enc, sc["masks"], np.random.default_rng(11), noise=0.5)
    cs = post.find_instance_center(torch.as_tensor(ctr_s).clone(), 0.3, 7, 200)
    grp_case("scene_480x640", cs.numpy(), off_s)

    # ---------------------------------------------------------------- full panoptic (a8-a10)
    def pan_case(name, fg_logit, center, offsets):
        fg_logit = torch.as_tensor(fg_logit, dtype=torch.float32).reshape(1, *np.shape(fg_logit)[-2:])
        center = torch.as_tensor(center, dtype=torch.float32).reshape(1, *np.shape(center)[-2:])
        offsets = torch.as_tensor(offsets, dtype=torch.float32)
        fg = fg_logit.sigmoid().round()
        pan, ctr = post.get_panoptic_segmentation(
            fg, center.clone(), offsets, thing_ids=[0], label_divisor=1000, stuff_area=2048,
            void_label=-1, threshold=0.3, nms_kernel=7, top_k=200)
        save("panoptic_" + name, fg_logit=fg_logit.numpy(), center=center.numpy(), offsets=offsets.numpy(),
             fg=fg.numpy(), pan=pan.numpy(), centers=ctr.numpy())

    lg, ce, of = synth.fake_head_outputs(enc, m8 * 255, np.random.default_rng(12), noise=0.3)
    pan_case("scene_96x128", lg, ce, of)
    lg, ce, of = synth.fake_head_outputs(sc_enc, sc["masks"], np.random.default_rng(13), noise=0.5)
    pan_case("scene_480x640", lg, ce, of)
    # 511 vs 512 px instances
    fgl = np.full((h, w), -4.0, np.float32)
    ce = np.full((h, w), 0.1, np.float32)
    of = np.zeros((2, h, w), np.float32)
    fgl[10:26, 10:42] = 4.0            # 16x32 = 512 px  -> kept
    fgl[25, 41] = 4.0
    ce[18, 26] = 0.9
    fgl[50:66, 60:92] = 4.0            # 512 px, one removed -> 511 -> dropped
    fgl[50, 60] = -4.0
    ce[58, 76] = 0.8
    pan_case("area_511_512", fgl, ce, of)
    # K = 0 with a large fg blob -> label 1000; and with a small blob -> all void
    fgl = np.full((h, w), -4.0, np.float32)
    fgl[20:70, 30:100] = 4.0
    pan_case("k0_blob", fgl, np.full((h, w), 0.1, np.float32), of)
    fgl = np.full((h, w), -4.0, np.float32)
    fgl[20:40, 30:60] = 4.0
    pan_case("k0_small", fgl, np.full((h, w), 0.1, np.float32), of)
    # sigmoid().round() around 0: probes from SURVEY 8c
    fgl = np.zeros((h, w), np.float32)
    vals = np.array([-1.2e-7, -6e-8, 0.0, 6e-8, 1.2e-7, 2.4e-7, 1e-3, -1e-3], np.float32)
    fgl[:] = np.resize(vals, (h, w))
    ce = np.full((h, w), 0.1, np.float32)
    ce[48, 64] = 0.9
    pan_case("sigmoid_round", fgl, ce, of)


if __name__ == "__main__":
    main()
