"""ORACLE (test infrastructure, never shipped): metrics of eval/evaluation.py:57-274 (multilabel_metrics), restated with
one numpy pass per (gt, prediction) pair like the reference.

The overlap half is pinned (below).  The boundary half (evaluation.py:21-54 boundary_overlap, utilities.py:672-697
seg2bmap) rests on cv2.findContours / drawContours / dilate and skimage.morphology.disk, none of which is in the image:
``seg2bmap`` / ``disk`` / ``boundary_overlap`` here restate the published algorithms with scipy.ndimage - **parity
unpinned**, checked on hand-derived cases (tests/test_oracle_golden.py).

Pinned: tests/golden/metrics_*.npz hold the dictionaries returned by the imported reference function (cv2 and
eval/utilities.py replaced by empty stand-ins for the import; neither is touched on the non-boundary path) and
tests/golden/munkres_expected.json the assignments of the vendored eval/munkres.py on 40 tie-heavy matrices."""
import numpy as np

from .assign_py import assign as munkres_assign      # the oracle's own solver (never the product's)


def seg2bmap(seg):
    """utilities.py:672-697: the pixels cv2.drawContours paints for cv2.findContours(seg, RETR_EXTERNAL, CHAIN_APPROX_NONE).
    Border following on 8-connected objects visits the object pixels that are 4-adjacent to the surrounding background;
    RETR_EXTERNAL keeps the components whose surrounding background is the one connected to the image frame."""
    from scipy import ndimage
    seg = np.asarray(seg).astype(bool)
    bg = np.pad(~seg, 1, constant_values=True)                      # the frame counts as background
    lab, _ = ndimage.label(bg, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outside = lab == lab[0, 0]
    nb = outside[:-2, 1:-1] | outside[2:, 1:-1] | outside[1:-1, :-2] | outside[1:-1, 2:]
    return (seg & nb).astype(np.uint8)


def disk(r):
    """skimage.morphology.disk: x^2 + y^2 <= r^2 on a (2r+1)^2 grid."""
    r = int(r)
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y <= r * r).astype(np.uint8)


def boundary_overlap(predicted_mask, gt_mask, bound_th=0.003):
    """evaluation.py:21-54 -> (precision_tps, recall_tps).  cv2.dilate's default border contributes nothing."""
    from scipy import ndimage
    bound_pix = bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(predicted_mask.shape))
    fg_b, gt_b = seg2bmap(predicted_mask).astype(bool), seg2bmap(gt_mask).astype(bool)
    st = disk(bound_pix).astype(bool)
    gt_d = ndimage.binary_dilation(gt_b, structure=st)
    fg_d = ndimage.binary_dilation(fg_b, structure=st)
    return int(np.sum(fg_b & gt_d)), int(np.sum(gt_b & fg_d))


def multilabel_metrics(prediction, gt, obj_detect_threshold=0.75, compute_boundary_stuff=False):
    lg = np.unique(gt)
    lg = lg[lg != 0]
    lp = np.unique(prediction)
    lp = lp[lp != 0]
    ng, npred = len(lg), len(lp)
    deg = None
    if npred == 0 and ng > 0:
        deg = (1., 0., 0., 0.)
    elif npred > 0 and ng == 0:
        deg = (0., 1., 0., 0.)
    elif npred == 0 and ng == 0:
        deg = (1., 1., 1., 1.)
    if deg is not None:
        p, r, f, pct = deg
        return {'Objects F-measure': f, 'Objects Precision': p, 'Objects Recall': r, 'Boundary F-measure': f,
                'Boundary Precision': p, 'Boundary Recall': r, 'Objects OSN F-measure': f, 'Objects OSN Precision': p,
                'Objects OSN Recall': r, 'Boundary OSN F-measure': f, 'Boundary OSN Precision': p,
                'Boundary OSN Recall': r, 'obj_detected': npred, 'obj_detected_075': 0., 'obj_gt': ng,
                'obj_detected_075_percentage': pct, 'obj_detected_075_percentage_normalized': pct}
    F, P, R = np.zeros((ng, npred)), np.zeros((ng, npred)), np.zeros((ng, npred))
    tps, iou, uni = np.zeros((ng, npred)), np.zeros((ng, npred)), np.zeros((ng, npred))
    bF, bP, bR, btps = np.zeros((ng, npred)), np.zeros((ng, npred)), np.zeros((ng, npred)), np.zeros((ng, npred, 2))
    if compute_boundary_stuff:
        bc_pred = np.array([np.sum(seg2bmap(prediction == pj)) for pj in lp], np.float64)
        bc_gt = np.array([np.sum(seg2bmap(gt == gi)) for gi in lg], np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i, gi in enumerate(lg):
            gm = gt == gi
            for j, pj in enumerate(lp):
                pm = prediction == pj
                if compute_boundary_stuff:
                    btps[i, j] = boundary_overlap(pm, gm)
                    bP[i, j] = btps[i, j][0] / bc_pred[j]
                    bR[i, j] = btps[i, j][1] / bc_gt[i]
                    bF[i, j] = (2 * bP[i, j] * bR[i, j]) / (bP[i, j] + bR[i, j])
                inter = np.int64(np.count_nonzero(gm & pm))
                union = np.int64(np.count_nonzero(gm | pm))
                iou[i, j], uni[i, j], tps[i, j] = inter / union, union, inter
                P[i, j] = inter / np.count_nonzero(pm)
                R[i, j] = inter / np.count_nonzero(gm)
                F[i, j] = (2 * P[i, j] * R[i, j]) / (P[i, j] + R[i, j])
        F[np.isnan(F)] = 0
        assign = munkres_assign(F.max() - F)
        idx = tuple(np.array(assign).T)
        det = sum(1 for a in assign if F[a] > obj_detect_threshold)
        precision = np.sum(tps[idx]) / np.sum(prediction.clip(0, 1) == 1)
        recall = np.sum(tps[idx]) / np.sum(gt.clip(0, 1) == 1)
        fm = (2 * precision * recall) / (precision + recall)
        if np.isnan(fm):
            fm = 0
        b = dict.fromkeys(("F", "P", "R", "Fo", "Po", "Ro"))
        if compute_boundary_stuff:                                     # evaluation.py:232-243
            bF[np.isnan(bF)] = 0
            b["P"] = np.sum(btps[idx][:, 0]) / np.sum(bc_pred)
            b["R"] = np.sum(btps[idx][:, 1]) / np.sum(bc_gt)
            b["F"] = (2 * b["P"] * b["R"]) / (b["P"] + b["R"])
            if np.isnan(b["F"]):
                b["F"] = 0
            b["Fo"], b["Po"], b["Ro"] = np.sum(bF[idx]) / max(npred, ng), np.sum(bP[idx]) / npred, np.sum(bR[idx]) / ng
        return {'Objects F-measure': fm, 'Objects Precision': precision, 'Objects Recall': recall,
                'Boundary F-measure': b["F"], 'Boundary Precision': b["P"], 'Boundary Recall': b["R"],
                'Objects OSN F-measure': np.sum(F[idx]) / max(npred, ng), 'Objects OSN Precision': np.sum(P[idx]) / npred,
                'Objects OSN Recall': np.sum(R[idx]) / ng, 'Boundary OSN F-measure': b["Fo"], 'Boundary OSN Precision': b["Po"],
                'Boundary OSN Recall': b["Ro"], 'obj_detected': npred, 'obj_detected_075': det, 'obj_gt': ng,
                'obj_detected_075_percentage': det / ng, 'obj_detected_075_percentage_normalized': det / max(ng, npred),
                'obj_mIOU_osn': np.mean(iou[idx]), 'obj_mIOU': np.sum(tps[idx]) / np.sum(uni[idx])}
