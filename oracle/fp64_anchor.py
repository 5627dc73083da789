"""Float64-anchored adjudication of the north-star tolerance ("within 1e-4 (float) / bit-exact (label maps)").

Test infrastructure (imports oracle/): used by tests/test_gpu_loud_parity.py (the HIP path on the benchmarked plan), by
tests/test_fp64_anchor_cpu.py (the analysis itself, on perturbed oracle logits) and by bench.py's cpu_baseline leg
(`parity_vs_hip.within_stated_tolerance`).

Three pieces:
  * oracle64(): the oracle network (oracle/network_torch.py, the restatement of model.py / resnet.py) evaluated in
    float64, frame by frame (no layer couples frames: FrozenBN / eval-BatchNorm are affine, GroupNorm is per sample).
  * anchor_errors(): per tap and per head, max |x - fp64| for the HIP path and for the fp32 oracle.  The bar is
        max|HIP - fp64| <= RATIO * max|oracle_fp32 - fp64|      (RATIO = 1.5)
    i.e. the HIP path is no further from the exact result than the reference's own fp32 arithmetic is - plus the literal
    1e-4 on the fg / centre / error logits against the fp32 oracle.
  * explain_label_flips(): every pixel whose label differs between post-processing of the HIP logits and of the fp32
    oracle's logits is traced to a decision of post_processing.py that is a NEAR-TIE IN FLOAT64, with the near-tie margins
    derived from the stated tolerance (not from a measurement):
        A  foreground threshold (model.py:293 sigmoid().round()):   |fg64(p)| <= EPS_LOGIT
        B  argmin of the centre distance (post_processing.py:66-74): |d64(p, c_hip) - d64(p, c_oracle)| <= EPS_DIST
        C  centre present in one list only (post_processing.py:27-41): |c64 - 0.3| <= EPS_LOGIT, or another pixel of its
           7x7 window within 2 * EPS_LOGIT of it (NMS tie), or the k-th value of the top-k within 2 * EPS_LOGIT
        D  the 512-px area filter / running relabel (post_processing.py:141-150): an instance whose area straddles 512
           between the two maps differs in area only by pixels already explained by A - C; the labels after it shift
    A flip that is none of these raises.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import postproc_ref
from oracle.network_torch import ArchCfg, MaskRefinerNet

TAPS = ("res2", "res3", "res5", "y", "feat_eee_boundary", "z1", "feat_center")
HEADS = (("foreground", 0, 1), ("center", 1, 2), ("offset", 2, 4), ("eee_boundary", 4, 8))
STRIDE = 4                       # model.py:695-700: offsets are multiplied by the common stride after the x4 bilinear
TOL = 1e-4                       # BASELINE.json north_star, in head units (what the predictors emit)
RATIO = 1.5                      # HIP may be at most this much further from float64 than the fp32 oracle is
EPS_LOGIT = TOL                  # two values within TOL of the exact one straddle a threshold only if it is within TOL of it
EPS_DIST = 2.0 * np.sqrt(2.0) * TOL * STRIDE   # |dd_a - dd_b| <= 2 |doffset|, |doffset| <= sqrt(2) * TOL * STRIDE px


def build_net(sd, dtype=torch.float32, **kw):
    kw = dict(kw)
    if "hierarchy" in kw:
        kw["hierarchy"] = [list(l) for l in kw["hierarchy"]]
    if "fusion_target" in kw:
        kw["fusion_target"] = list(kw["fusion_target"])
    net = MaskRefinerNet(ArchCfg(**kw)).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    return net.to(dtype)


def oracle64(sd, image, offs, frames=None, want_taps=TAPS, **kw):
    """Yields (i, heads64 dict [1,C,H,W] float64, taps64 dict) per frame; image u8 [B,6,H,W], offs f32 [B,3,H,W]."""
    net = build_net(sd, torch.float64, **kw)
    offs = torch.as_tensor(offs)
    for i in (range(image.shape[0]) if frames is None else frames):
        taps = {}
        with torch.no_grad():
            out = net(image[i:i + 1].double(), offs[i:i + 1].double(), taps)
        yield i, out, {k: taps[k] for k in want_taps if k in taps}


class OracleStream:
    """The oracle over the frames of a batch, ONE FRAME AT A TIME (as the reference runs: batch 1, predictor.py:358), in
    float32 and - optionally - float64, on a worker thread: the consumer overlaps its own host work (the oracle's
    post-processing of every candidate's logits) and the GPU runs with the next frame's network evaluation.
    Iterating yields dict(i, out32, taps32, out64, taps64) in frame order; at most `ahead` finished frames are held."""

    def __init__(self, sd, image, offs, fp64=True, want_taps=TAPS, ahead=2, **kw):
        import queue
        import threading
        self.q = queue.Queue(maxsize=ahead)
        self.n = image.shape[0]
        offs = torch.as_tensor(offs)

        nthreads = torch.get_num_threads()

        def work():
            try:
                torch.set_num_threads(nthreads)              # the OpenMP thread count is per calling thread
                net32 = build_net(sd, torch.float32, **kw)
                net64 = build_net(sd, torch.float64, **kw) if fp64 else None
                for i in range(self.n):
                    t32, t64, o64 = {}, {}, None
                    with torch.no_grad():
                        o32 = net32(image[i:i + 1], offs[i:i + 1], t32)
                        if net64 is not None:
                            o64 = net64(image[i:i + 1].double(), offs[i:i + 1].double(), t64)
                    self.q.put({"i": i, "out32": o32, "taps32": {k: t32[k] for k in want_taps if k in t32},
                                "out64": o64, "taps64": {k: t64[k] for k in want_taps if k in t64}})
            except BaseException as e:                       # surfaced in the consumer
                self.q.put(e)

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def __iter__(self):
        for _ in range(self.n):
            item = self.q.get()
            if isinstance(item, BaseException):
                raise item
            yield item
        self.thread.join()


def cat_heads(out):
    """heads dict -> [B,8,H,W] in the plane order of quber_forward (fg, centre, offset y/x, 4 error classes)."""
    return torch.cat([out["foreground"], out["center"], out["offset"], out["eee_boundary"]], 1)


class AnchorErrors:
    """Running max |x - fp64| per tap / head for two candidates ('hip', 'oracle32'); taps relative to the tap's magnitude."""

    def __init__(self):
        self.err = {}

    def add(self, name, hip, o32, o64, scale=None):
        o64 = o64.double()
        s = 1.0 if scale is None else scale
        e = self.err.setdefault(name, {"hip": 0.0, "oracle32": 0.0, "hip_vs_oracle32": 0.0, "scale": s})
        e["hip"] = max(e["hip"], float((hip.double() - o64).abs().max()) / s)
        e["oracle32"] = max(e["oracle32"], float((o32.double() - o64).abs().max()) / s)
        e["hip_vs_oracle32"] = max(e["hip_vs_oracle32"], float((hip.double() - o32.double()).abs().max()) / s)

    def add_heads(self, hip_logits, o32_logits, o64_logits):
        """[b,8,H,W] each; recorded in head units (offset planes / STRIDE) and, for the offsets, raw pixels too."""
        for key, a, b in HEADS:
            s = float(STRIDE) if key == "offset" else 1.0
            self.add(key, hip_logits[:, a:b] / s, o32_logits[:, a:b] / s, o64_logits[:, a:b] / s)
        self.add("offset_px_raw", hip_logits[:, 2:4], o32_logits[:, 2:4], o64_logits[:, 2:4])

    def table(self):
        rows = ["| tensor | max abs HIP - fp64 | max abs oracle_fp32 - fp64 | ratio | max abs HIP - oracle_fp32 |", "|---|---|---|---|---|"]
        for k, e in self.err.items():
            rows.append(f"| {k} | {e['hip']:.2e} | {e['oracle32']:.2e} | {e['hip'] / max(e['oracle32'], 1e-300):.2f} | {e['hip_vs_oracle32']:.2e} |")
        return "\n".join(rows)

    def verdict(self, ratio=RATIO, literal=("foreground", "center", "eee_boundary", "offset")):
        """-> (ok, list of failures).  Bars: hip <= ratio * oracle32 for every entry; hip_vs_oracle32 <= TOL (head units) for the heads."""
        bad = []
        for k, e in self.err.items():
            if e["hip"] > ratio * e["oracle32"]:
                bad.append(f"{k}: |HIP - fp64| = {e['hip']:.2e} > {ratio} x |oracle_fp32 - fp64| = {e['oracle32']:.2e}")
            if k in literal and e["hip_vs_oracle32"] > TOL:
                bad.append(f"{k}: |HIP - oracle_fp32| = {e['hip_vs_oracle32']:.2e} > {TOL:.0e} (head units)")
        return not bad, bad


def decide(lg):
    """Post-processing decisions of one frame, by the oracle (postproc_ref).  lg f32 [8,H,W] ->
    dict(fg {0,1} [H,W], ctr int64 [K,2], cid: flat index of the centre each pixel groups to [H,W] (before the fg mask), pan f32 [H,W])."""
    fg = lg[0:1].sigmoid().round()
    ctr = postproc_ref.find_centers(lg[1:2])
    h, w = lg.shape[-2:]
    if ctr.shape[0]:
        grp = postproc_ref.group_pixels(ctr, lg[2:4])
        cid = (ctr[:, 0] * w + ctr[:, 1])[grp[0] - 1]
        ins = fg * grp
    else:
        cid = torch.full((h, w), -1, dtype=torch.int64)
        ins = torch.zeros_like(fg)
    return {"fg": fg[0], "ctr": ctr, "cid": cid, "pan": postproc_ref.merge(ins, fg)[0]}


def explain_label_flips(lg_hip, lg_o32, lg_64, pan_hip=None, top_k=200, threshold=0.3, nms_kernel=7, dec_hip=None, dec_o32=None,
                        eps_logit=None, eps_dist=None):
    """One frame.  lg_hip, lg_o32 f32 [8,H,W] (or their precomputed decide() results); lg_64 float64 [8,H,W]; pan_hip: the HIP
    path's own label map (checked equal to the oracle's post-processing of the HIP logits).  Returns a dict of counts;
    raises AssertionError on an unexplained flip."""
    EPS_LOGIT_, EPS_DIST_ = (EPS_LOGIT if eps_logit is None else eps_logit), (EPS_DIST if eps_dist is None else eps_dist)
    dh = dec_hip if dec_hip is not None else decide(lg_hip)
    do = dec_o32 if dec_o32 is not None else decide(lg_o32)
    h, w = dh["fg"].shape
    fg_h, ctr_h, cid_h, pan_h = dh["fg"], dh["ctr"], dh["cid"], dh["pan"]
    fg_o, ctr_o, cid_o, pan_o = do["fg"], do["ctr"], do["cid"], do["pan"]
    if pan_hip is not None:
        assert torch.equal(pan_hip.float(), pan_h), "HIP label map != oracle post-processing of the HIP logits"
    diff = pan_h != pan_o
    rep = {"pixels": h * w, "flipped": int(diff.sum()), "A_fg_threshold": 0, "B_argmin_tie": 0, "C_centre_list": 0,
           "D_area_or_relabel": 0, "centres_hip": int(ctr_h.shape[0]), "centres_oracle": int(ctr_o.shape[0]),
           "centres_changed": 0, "max_abs_fg64_at_A": 0.0, "max_dist_gap_at_B": 0.0, "max_centre_gap_at_C": 0.0}
    if rep["flipped"] == 0:
        return rep
    fg64, c64, off64 = lg_64[0], lg_64[1], lg_64[2:4]
    # ---- C: centres present in one list only must be near-ties of find_instance_center in float64 ----
    set_h = {int(v) for v in (ctr_h[:, 0] * w + ctr_h[:, 1]).tolist()}
    set_o = {int(v) for v in (ctr_o[:, 0] * w + ctr_o[:, 1]).tolist()}
    changed = set_h ^ set_o
    rep["centres_changed"] = len(changed)
    if changed:
        pad = (nms_kernel - 1) // 2
        cthr = torch.where(c64 > threshold, c64, torch.full_like(c64, -1.0))
        # k-th largest surviving value (top-k boundary), in float64
        pooled = F.max_pool2d(cthr[None, None], nms_kernel, 1, pad)[0, 0]
        surv = torch.where(cthr == pooled, cthr, torch.full_like(cthr, -1.0)).flatten()
        kth = float(torch.topk(surv, top_k).values[-1].clamp(min=0)) if surv.numel() >= top_k else 0.0
        for f in sorted(changed):
            y, x = divmod(f, w)
            v = float(c64[y, x])
            win = c64[max(0, y - pad):y + pad + 1, max(0, x - pad):x + pad + 1].clone()
            win[y - max(0, y - pad), x - max(0, x - pad)] = -np.inf
            gaps = [abs(v - threshold), abs(v - float(win.max())) / 2.0]
            if kth > 0:
                gaps.append(abs(v - kth) / 2.0)
            g = min(gaps)
            rep["max_centre_gap_at_C"] = max(rep["max_centre_gap_at_C"], g)
            assert g <= EPS_LOGIT_, (f"centre ({y},{x}) is in one centre list only but is no float64 near-tie: c64 = {v:.6f}, "
                                    f"|c-0.3| = {gaps[0]:.2e}, NMS gap/2 = {gaps[1]:.2e}")
    common = set_h & set_o
    # ---- per-pixel classes ----
    idx = torch.nonzero(diff)
    py, px = idx[:, 0], idx[:, 1]
    a = fg_h[py, px] != fg_o[py, px]
    if bool(a.any()):
        m = fg64[py[a], px[a]].abs()
        rep["A_fg_threshold"] = int(a.sum())
        rep["max_abs_fg64_at_A"] = float(m.max())
        assert float(m.max()) <= EPS_LOGIT_, f"foreground flip away from the threshold: |fg64| = {float(m.max()):.2e}"
    rest = ~a
    ch, co = cid_h[py, px], cid_o[py, px]
    b = rest & (ch != co)
    if bool(b.any()):
        is_common = torch.tensor([(int(u) in common) and (int(v) in common) for u, v in zip(ch[b].tolist(), co[b].tolist())])
        yb, xb = py[b], px[b]
        rep["C_centre_list"] = int((~is_common).sum())           # explained by a changed centre (asserted above)
        if bool(is_common.any()):
            yy, xx = yb[is_common].double(), xb[is_common].double()
            ly = yy + off64[0][yb[is_common], xb[is_common]]
            lx = xx + off64[1][yb[is_common], xb[is_common]]

            def dist(cf):
                cy, cx = (cf // w).double(), (cf % w).double()
                return torch.sqrt((cy - ly) ** 2 + (cx - lx) ** 2)

            gap = (dist(ch[b][is_common]) - dist(co[b][is_common])).abs()
            rep["B_argmin_tie"] = int(is_common.sum())
            rep["max_dist_gap_at_B"] = float(gap.max())
            assert float(gap.max()) <= EPS_DIST_, f"argmin flip between centres {float(gap.max()):.2e} px apart in float64 (bar {EPS_DIST_:.2e})"
    d = rest & (ch == co)
    if bool(d.any()):
        # same foreground decision, same centre: the label differs because an instance crossed the 512-px filter in one map
        # (or a centre changed) and the running relabel shifted.  Every such cause must exist and be explained by A - C.
        rep["D_area_or_relabel"] = int(d.sum())
        n_abc = rep["A_fg_threshold"] + rep["B_argmin_tie"] + rep["C_centre_list"]
        straddle = 0
        for f in common:
            ah = int(((cid_h == f) & (fg_h > 0)).sum())
            ao = int(((cid_o == f) & (fg_o > 0)).sum())
            if (ah >= postproc_ref.MIN_INSTANCE_AREA) != (ao >= postproc_ref.MIN_INSTANCE_AREA):
                straddle += 1
                assert abs(ah - ao) <= n_abc, f"instance at {divmod(f, w)}: areas {ah} / {ao} differ by more than the {n_abc} explained pixels"
        rep["instances_straddling_512"] = straddle
        assert straddle > 0 or changed, "labels shifted without an instance crossing the area filter or a changed centre"
    assert rep["A_fg_threshold"] + rep["B_argmin_tie"] + rep["C_centre_list"] + rep["D_area_or_relabel"] == rep["flipped"]
    return rep


def summarize(reports):
    tot = {k: sum(r.get(k, 0) for r in reports) for k in ("pixels", "flipped", "A_fg_threshold", "B_argmin_tie", "C_centre_list",
                                                          "D_area_or_relabel", "centres_changed")}
    for k in ("max_abs_fg64_at_A", "max_dist_gap_at_B", "max_centre_gap_at_C"):
        tot[k] = max([r[k] for r in reports] + [0.0])
    tot["label_map_equal_fraction"] = 1.0 - tot["flipped"] / max(tot["pixels"], 1)
    tot["frames"] = len(reports)
    tot["eps_logit"], tot["eps_dist_px"] = EPS_LOGIT, EPS_DIST
    return tot
