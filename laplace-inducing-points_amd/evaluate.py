"""Predictive evaluation harness — the reference's ``scale_experiments/evaluate.py:40-231`` ("next" row N3):
MC predictive NLL / accuracy (``batch_nll`` ``:98-154``), Brier (``:40-43``), ECE (``:45-63``), OOD-AUROC
(``:65-93``), ``eval_dataset`` (``:157-184``), ``eval_dataset_extended`` (``:187-231``).
Logit samples come from ``predict_lla_scalable`` (HIP engine); the metrics are plain reductions on the device.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from .lla import predict_lla_dense, predict_lla_scalable


def brier_score(probs: torch.Tensor, labels: torch.Tensor) -> float:
    """``:40-43`` multi-class Brier score."""
    one_hot = torch.nn.functional.one_hot(labels.long(), probs.shape[-1]).to(probs.dtype)
    return float(((probs - one_hot) ** 2).sum(1).mean())


def ece(probs: torch.Tensor, labels: torch.Tensor, n_bins: int = 15) -> float:
    """``:45-63`` expected calibration error, histogram binning on [lo, hi)."""
    conf, pred = probs.max(1)
    acc = (pred == labels.long()).to(probs.dtype)
    edges = torch.linspace(0.0, 1.0, n_bins + 1, device=probs.device, dtype=probs.dtype)
    val = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        mask = (conf >= lo) & (conf < hi)
        if not bool(mask.any()):
            continue
        val += float((conf[mask].mean() - acc[mask].mean()).abs() * mask.to(probs.dtype).mean())
    return val


def ood_scores(probs: torch.Tensor) -> torch.Tensor:
    """``:65-67``: higher = more in-distribution-like."""
    return -probs.max(1).values


def roc_auc(labels: torch.Tensor, scores: torch.Tensor) -> float:
    """Area under the ROC curve by the rank statistic (ties get average ranks), = sklearn.roc_auc_score."""
    scores = scores.double().cpu()
    labels = labels.double().cpu()
    order = torch.argsort(scores)
    s = scores[order]
    ranks = torch.empty_like(s)
    i, n = 0, s.numel()
    r = torch.arange(1, n + 1, dtype=torch.float64)
    # average ranks over ties
    uniq, inv, cnt = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = torch.cumsum(cnt, 0).double()
    starts = ends - cnt.double() + 1
    ranks = ((starts + ends) / 2)[inv]
    full = torch.empty(n, dtype=torch.float64)
    full[order] = ranks
    pos = labels > 0.5
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((full[pos].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def batch_nll(state, x, y, Z, *, alpha, full_set_size, model_type, num_mc_samples, rng, scalable=True,
              return_mean=False):
    """``:98-154``: (NLL of the MC-averaged predictive, accuracy[, mean probabilities])."""
    if scalable == "marginals":
        # closed-form per-point predictive (lla.predict_lla_marginals): the metrics below only use per-point marginals,
        # so the S draws can be taken in the K-dimensional output space instead of through S tangent sweeps
        from .lla import predict_lla_marginals
        dist = predict_lla_marginals(state, x, Z, model_type=model_type, alpha=alpha, full_set_size=full_set_size)
        logit_samples = dist.sample(sample_shape=(num_mc_samples,), seed=rng).float()
    elif scalable:
        logit_samples = predict_lla_scalable(state, x, Z, model_type=model_type, alpha=alpha,
                                             full_set_size=full_set_size, num_samples=num_mc_samples, key=rng)
    else:
        dist = predict_lla_dense(state, x, Z, model_type=model_type, alpha=alpha, full_set_size=full_set_size)
        logit_samples = dist.sample(sample_shape=(num_mc_samples,), seed=rng).float()
    S = logit_samples.shape[0]
    log_probs = torch.log_softmax(logit_samples, dim=-1)                               # (S, B, C)
    y_int = y.reshape(-1).long().to(log_probs.device)
    log_p_true = torch.gather(log_probs, -1, y_int[None, :, None].expand(S, -1, 1)).squeeze(-1)   # (S, B)
    log_avg_prob = torch.logsumexp(log_p_true, dim=0) - math.log(S)
    nll = -log_avg_prob.mean()
    mean = torch.softmax(logit_samples, dim=-1).mean(0)                                # (B, C)
    acc = (mean.argmax(-1) == y_int).float().mean()
    if return_mean:
        return nll, acc, mean
    return nll, acc


def eval_dataset(state, dataloader: Iterable, Z, alpha, full_set_size, model_type, num_mc_samples, rng, scalable=True):
    """``:157-184``"""
    tot_nll = tot_correct = 0.0
    tot_N = 0
    for x_b, y_b in dataloader:
        rng = int(rng) + 1
        nll, acc = batch_nll(state, x_b, y_b, Z, alpha=alpha, full_set_size=full_set_size, model_type=model_type,
                             num_mc_samples=num_mc_samples, rng=rng, scalable=scalable)
        bs = x_b.shape[0]
        tot_nll += float(nll) * bs
        tot_correct += float(acc) * bs
        tot_N += bs
    return tot_nll / tot_N, tot_correct / tot_N


def eval_dataset_extended(state, dataloader: Iterable, Z, alpha, full_set_size, model_type, num_mc_samples, rng,
                          scalable=True):
    """``:187-231``: (NLL, accuracy, Brier, ECE, probs, labels)"""
    tot_nll = tot_correct = 0.0
    tot_N = 0
    all_probs, all_labels = [], []
    for x_b, y_b in dataloader:
        rng = int(rng) + 1
        nll, acc, mean = batch_nll(state, x_b, y_b, Z, alpha=alpha, full_set_size=full_set_size, model_type=model_type,
                                   num_mc_samples=num_mc_samples, rng=rng, scalable=scalable, return_mean=True)
        bs = x_b.shape[0]
        tot_nll += float(nll) * bs
        tot_correct += float(acc) * bs
        tot_N += bs
        all_probs.append(mean)
        all_labels.append(y_b.reshape(-1).to(mean.device))
    probs, labels = torch.cat(all_probs), torch.cat(all_labels)
    return tot_nll / tot_N, tot_correct / tot_N, brier_score(probs, labels), ece(probs, labels), probs, labels


def auroc_ood(state, id_probs: torch.Tensor, ood_loader: Iterable, Z, alpha, full_set_size, model_type,
              num_mc_samples, rng, scalable=True) -> float:
    """``:69-93``: AUROC of max-probability scores, in-distribution (label 0) vs OOD (label 1)."""
    ood = []
    for xb, yb in ood_loader:
        rng = int(rng) + 1
        dummy = torch.zeros(xb.shape[0], dtype=torch.long)
        _, _, mean = batch_nll(state, xb, dummy, Z, alpha=alpha, full_set_size=full_set_size, model_type=model_type,
                               num_mc_samples=num_mc_samples, rng=rng, scalable=scalable, return_mean=True)
        ood.append(mean)
    ood_probs = torch.cat(ood)
    scores = torch.cat([ood_scores(id_probs), ood_scores(ood_probs)])
    labels = torch.cat([torch.zeros(len(id_probs)), torch.ones(len(ood_probs))])
    return roc_auc(labels, scores)
