"""ORACLE (test infrastructure, never shipped): overlap metrics of eval/evaluation.py:57-274 (multilabel_metrics with
compute_boundary_stuff=False), restated with one numpy pass per (gt, prediction) pair like the reference.

Pinned: tests/golden/metrics_*.npz hold the dictionaries returned by the imported reference function (cv2 and
eval/utilities.py replaced by empty stand-ins for the import; neither is touched on the non-boundary path) and
tests/golden/munkres_expected.json the assignments of the vendored eval/munkres.py on 40 tie-heavy matrices."""
import numpy as np

from quber_amd.eval.assignment import munkres_assign


def multilabel_metrics(prediction, gt, obj_detect_threshold=0.75):
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
    with np.errstate(divide="ignore", invalid="ignore"):
        for i, gi in enumerate(lg):
            gm = gt == gi
            for j, pj in enumerate(lp):
                pm = prediction == pj
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
        return {'Objects F-measure': fm, 'Objects Precision': precision, 'Objects Recall': recall,
                'Boundary F-measure': None, 'Boundary Precision': None, 'Boundary Recall': None,
                'Objects OSN F-measure': np.sum(F[idx]) / max(npred, ng), 'Objects OSN Precision': np.sum(P[idx]) / npred,
                'Objects OSN Recall': np.sum(R[idx]) / ng, 'Boundary OSN F-measure': None, 'Boundary OSN Precision': None,
                'Boundary OSN Recall': None, 'obj_detected': npred, 'obj_detected_075': det, 'obj_gt': ng,
                'obj_detected_075_percentage': det / ng, 'obj_detected_075_percentage_normalized': det / max(ng, npred),
                'obj_mIOU_osn': np.mean(iou[idx]), 'obj_mIOU': np.sum(tps[idx]) / np.sum(uni[idx])}
