"""Drop-in for the overlap part of the reference's ``eval/evaluation.py:multilabel_metrics`` (lines 57-274): Objects
P / R / F, their object-size-normalised variants, F@.75 detection counts and the IoU measures, with all pairwise
overlap counts produced by ONE pass of a HIP kernel over the two label maps instead of one numpy pass per pair.

The boundary measures need OpenCV contours and skimage disks (evaluation.py:21-54, utilities.py:672-697) and are not
built: ``compute_boundary_stuff`` must be False and the Boundary entries are None, exactly what the reference returns
for that flag."""
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


def multilabel_metrics(prediction, gt, i=0, N=0, obj_detect_threshold=0.75, compute_boundary_stuff=False, verbose=False,
                       device="cuda:0"):
    if compute_boundary_stuff:
        raise NotImplementedError("quber_amd: boundary measures need OpenCV / skimage (evaluation.py:21-54); not built")
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
        out = {'Objects F-measure': f_measure, 'Objects Precision': precision, 'Objects Recall': recall,
               'Boundary F-measure': None, 'Boundary Precision': None, 'Boundary Recall': None,
               'Objects OSN F-measure': np.sum(F[idx]) / max(num_pred, num_gt),
               'Objects OSN Precision': np.sum(P[idx]) / num_pred, 'Objects OSN Recall': np.sum(R[idx]) / num_gt,
               'Boundary OSN F-measure': None, 'Boundary OSN Precision': None, 'Boundary OSN Recall': None,
               'obj_detected': num_pred, 'obj_detected_075': detected, 'obj_gt': num_gt,
               'obj_detected_075_percentage': detected / num_gt,
               'obj_detected_075_percentage_normalized': detected / max(num_gt, num_pred),
               'obj_mIOU_osn': np.mean(iou[idx]), 'obj_mIOU': np.sum(tps[idx]) / np.sum(union[idx])}
    return out
