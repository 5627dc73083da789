"""Drop-in for the reference's ``eval/evaluation.py:multilabel_metrics`` (lines 57-274): Objects P / R / F, their
object-size-normalised variants, F@.75 detection counts, the IoU measures and the Boundary P / R / F.  All pairwise overlap
counts come from ONE pass of a HIP kernel over the two label maps instead of one numpy pass per pair; the boundary
true-positive counts of every pair (evaluation.py:21-54) come from three launches on bit planes (csrc/boundary.hip:
seg2bmap as an outside flood fill + 4-neighbour test, disk dilation as shifted ORs).

The overlap half reproduces the imported reference bit for bit (tests/golden/metrics_*.npz).  The boundary half restates
cv2.findContours / drawContours / dilate and skimage's disk from their published algorithms (neither library is in the
image): parity unpinned."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .assignment import munkres_assign

BACKGROUND_LABEL = 0
CAP = 256


def contingency(prediction, gt, device="cuda:0"):
    """-> (labels_pred, labels_gt, table[n_gt, n_pred]) with np.unique ordering; integer label maps of equal shape."""
    lib = _lib.load()
    prediction, gt = np.asarray(prediction), np.asarray(gt)
    assert prediction.shape == gt.shape
    remap = None
    lo = min(int(prediction.min()), int(gt.min())) if prediction.size else 0
    hi = max(int(prediction.max()), int(gt.max())) if prediction.size else 0
    if lo < 0 or hi > 65535:
        # the reference's np.unique accepts any integers (e.g. the -1 void label of panoptic_seg, or label * 1000 ids); the
        # kernel's LUT covers 0..65535, so such maps are renumbered densely first (order-preserving, mapped back below)
        vals = np.unique(np.concatenate([np.unique(prediction), np.unique(gt)]))
        if vals.size > 65536:
            raise ValueError("more than 65536 distinct labels")
        prediction, gt, remap = np.searchsorted(vals, prediction), np.searchsorted(vals, gt), vals
    p = torch.as_tensor(np.ascontiguousarray(prediction)).to(device=device, dtype=torch.int32).contiguous()
    g = torch.as_tensor(np.ascontiguousarray(gt)).to(device=device, dtype=torch.int32).contiguous()
    nbytes = lib.quber_contingency_workspace_bytes(CAP)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _lib.check(lib.quber_label_contingency(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), p.numel(), CAP,
                                           C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    o = 2 * 65536 * 4
    host = ws[o:o + CAP * CAP * 8 + 2 * CAP * 4 + 16].cpu().numpy()
    table = host[:CAP * CAP * 8].view(np.uint64).reshape(CAP, CAP)
    labels = host[CAP * CAP * 8:CAP * CAP * 8 + 2 * CAP * 4].view(np.int32).reshape(2, CAP)
    n_pred, n_gt, bad, _ = host[CAP * CAP * 8 + 2 * CAP * 4:].view(np.int32)
    if bad:
        raise ValueError("label values must lie in 0..65535")
    if n_pred > CAP or n_gt > CAP:
        raise ValueError(f"more than {CAP} distinct labels in a map")
    lp, lg = labels[0, :n_pred].astype(np.int64), labels[1, :n_gt].astype(np.int64)
    if remap is not None:
        lp, lg = remap[lp].astype(np.int64), remap[lg].astype(np.int64)
    return lp, lg, table[:n_gt, :n_pred].astype(np.int64)


def _degenerate(p, r, f, num_pred, num_gt, pct):
    return {'Objects F-measure': f, 'Objects Precision': p, 'Objects Recall': r,
            'Boundary F-measure': f, 'Boundary Precision': p, 'Boundary Recall': r,
            'Objects OSN F-measure': f, 'Objects OSN Precision': p, 'Objects OSN Recall': r,
            'Boundary OSN F-measure': f, 'Boundary OSN Precision': p, 'Boundary OSN Recall': r,
            'obj_detected': num_pred, 'obj_detected_075': 0., 'obj_gt': num_gt,
            'obj_detected_075_percentage': pct, 'obj_detected_075_percentage_normalized': pct}


def boundary_counts(prediction, gt, labels_pred, labels_gt, bound_th=0.003, device="cuda:0"):
    """-> (bound_counts_pred [n_pred], bound_counts_gt [n_gt], precision_tps [n_gt, n_pred], recall_tps [n_gt, n_pred]) as
    float64, for the given object labels (evaluation.py:165-175 and :21-54 for every pair)."""
    lib = _lib.load()
    prediction, gt = np.asarray(prediction), np.asarray(gt)
    h, w = prediction.shape
    n_pred, n_gt = len(labels_pred), len(labels_gt)
    bound_pix = int(bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(prediction.shape)))
    p = torch.as_tensor(np.ascontiguousarray(prediction)).to(device=device, dtype=torch.int32).contiguous()
    g = torch.as_tensor(np.ascontiguousarray(gt)).to(device=device, dtype=torch.int32).contiguous()
    labels = torch.as_tensor(np.concatenate([labels_pred, labels_gt]).astype(np.int32)).to(device)
    ws = torch.empty(lib.quber_boundary_workspace_bytes(h, w, n_pred + n_gt), dtype=torch.uint8, device=device)
    out = torch.empty(n_pred + n_gt + 2 * n_pred * n_gt, dtype=torch.int32, device=device)
    _lib.check(lib.quber_boundary_overlap(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), h, w, C.c_void_p(labels.data_ptr()),
                                          n_pred, n_gt, bound_pix, C.c_void_p(ws.data_ptr()), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    o = out.cpu().numpy().astype(np.float64)
    nm = n_pred + n_gt
    return (o[:n_pred], o[n_pred:nm], o[nm:nm + n_gt * n_pred].reshape(n_gt, n_pred),
            o[nm + n_gt * n_pred:].reshape(n_gt, n_pred))


def multilabel_metrics(prediction, gt, i=0, N=0, obj_detect_threshold=0.75, compute_boundary_stuff=True, verbose=False,
                       device="cuda:0"):
    lp, lg, table = contingency(prediction, gt, device)
    area_pred, area_gt = table.sum(axis=0), table.sum(axis=1)     # incl. the background row / column
    kp, kg = lp != BACKGROUND_LABEL, lg != BACKGROUND_LABEL
    num_pred, num_gt = int(kp.sum()), int(kg.sum())
    # edge cases of evaluation.py:104-162
    if num_pred == 0 and num_gt > 0:
        return _degenerate(1., 0., 0., num_pred, num_gt, 0.)
    if num_pred > 0 and num_gt == 0:
        return _degenerate(0., 1., 0., num_pred, num_gt, 0.)
    if num_pred == 0 and num_gt == 0:
        return _degenerate(1., 1., 1., num_pred, num_gt, 1.)
    tps = table[np.ix_(kg, kp)].astype(np.float64)                # obj_tps[i, j] = |gt_i & pred_j|
    ap, ag = area_pred[kp].astype(np.float64), area_gt[kg].astype(np.float64)
    union = ag[:, None] + ap[None, :] - tps
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tps / union
        P, R = tps / ap[None, :], tps / ag[:, None]
        F = (2 * P * R) / (P + R)
    F[np.isnan(F)] = 0
    assignments = munkres_assign(F.max() - F)
    idx = tuple(np.array(assignments).T)
    detected = sum(1 for a in assignments if F[a] > obj_detect_threshold)
    fg_pred = float(area_pred[lp >= 1].sum())                      # np.sum(prediction.clip(0,1) == 1)
    fg_gt = float(area_gt[lg >= 1].sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.sum(tps[idx]) / fg_pred
        recall = np.sum(tps[idx]) / fg_gt
        f_measure = (2 * precision * recall) / (precision + recall)
        if np.isnan(f_measure):
            f_measure = 0
        b = dict.fromkeys(("F", "P", "R", "Fo", "Po", "Ro"))
        if compute_boundary_stuff:                                   # evaluation.py:165-175, 200-206, 232-243
            bc_pred, bc_gt, ptps, rtps = boundary_counts(prediction, gt, lp[kp], lg[kg], device=device)
            bP, bR = ptps / bc_pred[None, :], rtps / bc_gt[:, None]
            bF = (2 * bP * bR) / (bP + bR)
            bF[np.isnan(bF)] = 0
            b["P"], b["R"] = np.sum(ptps[idx]) / np.sum(bc_pred), np.sum(rtps[idx]) / np.sum(bc_gt)
            b["F"] = (2 * b["P"] * b["R"]) / (b["P"] + b["R"])
            if np.isnan(b["F"]):
                b["F"] = 0
            b["Fo"], b["Po"], b["Ro"] = np.sum(bF[idx]) / max(num_pred, num_gt), np.sum(bP[idx]) / num_pred, np.sum(bR[idx]) / num_gt
        out = {'Objects F-measure': f_measure, 'Objects Precision': precision, 'Objects Recall': recall,
               'Boundary F-measure': b["F"], 'Boundary Precision': b["P"], 'Boundary Recall': b["R"],
               'Objects OSN F-measure': np.sum(F[idx]) / max(num_pred, num_gt),
               'Objects OSN Precision': np.sum(P[idx]) / num_pred, 'Objects OSN Recall': np.sum(R[idx]) / num_gt,
               'Boundary OSN F-measure': b["Fo"], 'Boundary OSN Precision': b["Po"], 'Boundary OSN Recall': b["Ro"],
               'obj_detected': num_pred, 'obj_detected_075': detected, 'obj_gt': num_gt,
               'obj_detected_075_percentage': detected / num_gt,
               'obj_detected_075_percentage_normalized': detected / max(num_gt, num_pred),
               'obj_mIOU_osn': np.mean(iou[idx]), 'obj_mIOU': np.sum(tps[idx]) / np.sum(union[idx])}
    return out
