"""Evaluation metrics without per-sample host synchronisation (reference FABind/fabind/utils/utils.py:445-604
`evaluate_mean_pocket_cls_coord_multi_task`, utils/metrics.py:62-77 `pocket_metrics`).

The reference walks every sample of every batch in python (`for i, j in enumerate(batch_len)`, `.item()` per loss term,
boolean-mask indexing per complex): ~10 device synchronisations per complex.  Here a batch is reduced with masked tensor
arithmetic on the device it lives on, the partial sums stay there, and `compute()` reads them back once.  Same dictionary
keys and values as the reference (pinned against the reference's own loop: tests/golden/eval_metrics.npz)."""
import torch
import torch.nn.functional as F


def pearson_corrcoef(a, b):
    """torchmetrics.functional.pearson_corrcoef (published definition: covariance over the product of standard deviations)."""
    a, b = a - a.mean(), b - b.mean()
    return (a * b).sum() / (a.pow(2).sum().sqrt() * b.pow(2).sum().sqrt())


def pocket_metrics(pocket_coord_pred, pocket_coord):
    """utils/metrics.py:62-77: per-axis Pearson / RMSE / MAE averaged over x, y, z, mean centre distance, DCC (% < 4 A)."""
    pear = sum(pearson_corrcoef(pocket_coord_pred[:, k], pocket_coord[:, k]) for k in range(3)) / 3
    rmse = sum((pocket_coord_pred[:, k] - pocket_coord[:, k]).pow(2).mean().sqrt() for k in range(3)) / 3
    mae = sum((pocket_coord_pred[:, k] - pocket_coord[:, k]).abs().mean() for k in range(3)) / 3
    dist = F.pairwise_distance(pocket_coord_pred, pocket_coord, p=2)
    return {"pocket_pearson": pear, "pocket_rmse": rmse, "pocket_mae": mae, "pocket_center_avg_dist": dist.mean(),
            "pocket_center_DCC": (dist < 4).float().mean() * 100}


def _segment_mean(src, index, n):
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add_(0, index, src)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


class Evaluator:
    """Accumulates what the reference's evaluation loop accumulates; `update` issues no host synchronisation."""

    def __init__(self, args, com_coord_criterion, criterion, pocket_cls_criterion, pocket_coord_criterion, pred_dis=True):
        if not pred_dis:
            # the reference's pred_dis=False branch reads contact_by_pred_loss before assigning it (utils/utils.py:503-512)
            raise NotImplementedError("only pred_dis=True is a working configuration of the reference loop")
        self.args = args
        self.crit = (com_coord_criterion, criterion, pocket_cls_criterion, pocket_coord_criterion)
        self.rmsd, self.cdis, self.center_pred, self.center_true = [], [], [], []
        self.sums = None
        self.keep_less_5 = 0

    def update(self, out, com_coord, coords_center):
        """out: the model's 11-tuple (a3 of SURVEY section 8); com_coord = data.coords; coords_center = data.coords_center."""
        a = self.args
        com_crit, crit, cls_crit, cen_crit = self.crit
        (coords, cb, y_pred, y_by, logits, pocket_cls, mask, xyz, center_direct, dis_map, less5) = out[:11]
        coords, logits = coords.detach(), logits.detach()
        B = mask.shape[0]
        sd = ((coords - com_coord) ** 2).sum(dim=-1)
        self.rmsd.append(_segment_mean(sd, cb, B).sqrt())
        self.cdis.append((_segment_mean(coords, cb, B) - _segment_mean(com_coord, cb, B)).norm(dim=-1))
        zero = torch.zeros((), device=coords.device)
        has = dis_map.numel() > 0
        contact = a.pair_distance_loss_weight * crit(y_pred, dis_map) if has else zero
        by_pred = a.pair_distance_loss_weight * crit(y_by, dis_map) if has else zero
        cls_loss = a.pocket_cls_loss_weight * cls_crit(logits, pocket_cls.float())
        cen_loss = a.pocket_distance_loss_weight * cen_crit(center_direct.detach(), coords_center)
        com_loss = a.coord_loss_weight * com_crit(coords, com_coord)
        # per-complex pocket centre from the classifier: mean of the residues predicted positive, or -- when none is --
        # the soft (no-noise Gumbel-softmax) centre (utils/utils.py:541-559); all complexes at once
        m = mask.to(logits.dtype)
        prob = logits.sigmoid()
        pred_pos = (prob.round() == 1) & mask
        n_pos = pred_pos.sum(1)
        hard = (pred_pos.to(xyz.dtype).unsqueeze(-1) * xyz).sum(1) / n_pos.clamp(min=1).unsqueeze(-1)
        lp = torch.stack([torch.log(1.0 - prob), torch.log(prob)], -1) / a.gs_tau
        w = lp.softmax(-1)[..., 1] * m
        soft = (w.unsqueeze(-1) * xyz).sum(1) / w.sum(1, keepdim=True)
        self.center_pred.append(torch.where((n_pos > 0).unsqueeze(-1), hard, soft))
        self.center_true.append(coords_center)
        correct = ((prob.round().int() == pocket_cls.int()) & mask).sum()
        row = torch.stack([
            torch.as_tensor(float(B), device=coords.device), (n_pos == 0).sum().float(),
            y_pred.shape[0] * contact, y_by.shape[0] * by_pred, torch.as_tensor(float(y_pred.shape[0]), device=coords.device),
            coords.shape[0] * com_loss, torch.as_tensor(float(coords.shape[0]), device=coords.device),
            B * cls_loss, B * cen_loss, correct.float(), mask.sum().float()])
        self.sums = row if self.sums is None else self.sums + row
        self.keep_less_5 = self.keep_less_5 + less5

    def compute(self):
        """-> the reference's metrics dict (python floats); the one host synchronisation of an evaluation."""
        rmsd, cdis = torch.cat(self.rmsd), torch.cat(self.cdis)
        q = torch.tensor([0.25, 0.5, 0.75], device=rmsd.device, dtype=rmsd.dtype)
        pm = pocket_metrics(torch.cat(self.center_pred), torch.cat(self.center_true))
        vec = torch.cat([self.sums, torch.stack([rmsd.mean(), (rmsd < 2).float().mean(), (rmsd < 5).float().mean()]),
                         torch.quantile(rmsd, q), torch.stack([cdis.mean(), (cdis < 2).float().mean(), (cdis < 5).float().mean()]),
                         torch.quantile(cdis, q), torch.stack([pm[k] for k in ("pocket_pearson", "pocket_rmse", "pocket_mae",
                                                                               "pocket_center_avg_dist", "pocket_center_DCC")]),
                         torch.as_tensor(float(self.keep_less_5), device=rmsd.device).reshape(1)]).tolist()
        (n, skip, contact, by_pred, n_pair, com, n_atom, cls, cen, correct, n_res) = vec[:11]
        keys = ["rmsd", "rmsd < 2A", "rmsd < 5A", "rmsd 25%", "rmsd 50%", "rmsd 75%", "centroid_dis", "centroid_dis < 2A",
                "centroid_dis < 5A", "centroid_dis 25%", "centroid_dis 50%", "centroid_dis 75%"]
        metrics = {"samples": int(n), "skip_samples": int(skip), "keepNode < 5": int(vec[-1])}
        metrics.update({"contact_loss": contact / n_pair, "contact_by_pred_loss": by_pred / n_pair,
                        "com_coord_huber_loss": com / n_atom})
        metrics.update(dict(zip(keys, vec[11:23])))
        metrics.update({"pocket_cls_bce_loss": cls / n, "pocket_coord_mse_loss": cen / n, "pocket_cls_accuracy": correct / n_res})
        metrics.update(dict(zip(("pocket_pearson", "pocket_rmse", "pocket_mae", "pocket_center_avg_dist", "pocket_center_DCC"),
                                vec[23:28])))
        return metrics


@torch.no_grad()
def evaluate_mean_pocket_cls_coord_multi_task(accelerator, args, data_loader, model, com_coord_criterion, criterion,
                                              pocket_cls_criterion, pocket_coord_criterion, relative_k, device, pred_dis=False,
                                              info=None, saveFileName=None, use_y_mask=False,
                                              skip_y_metrics_evaluation=False, stage=1):
    """The reference's signature (utils/utils.py:446); one host synchronisation per evaluation instead of ~10 per complex."""
    ev = Evaluator(args, com_coord_criterion, criterion, pocket_cls_criterion, pocket_coord_criterion, pred_dis=pred_dis)
    for data in data_loader:
        data = data.to(device)
        out = model(data, stage=stage)
        ev.update(out, data.coords, data.coords_center)
    return ev.compute()
