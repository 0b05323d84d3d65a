"""Oracle restatement (numpy) of the metric helpers of ``scale_experiments/evaluate.py:40-67`` (TEST INFRASTRUCTURE)."""
import numpy as np


def brier_score(probs, labels):
    one_hot = np.eye(probs.shape[-1])[labels.astype(int)]
    return np.mean(np.sum((probs - one_hot) ** 2, axis=1))


def ece(probs, labels, n_bins=15):
    confidences = probs.max(1)
    predictions = probs.argmax(1)
    accuracies = (predictions == labels)
    bin_edges = np.linspace(0.0, 1.0, n_bins + 1)
    ece_val = 0.0
    for lo, hi in zip(bin_edges[:-1], bin_edges[1:]):
        mask = (confidences >= lo) & (confidences < hi)
        if not np.any(mask):
            continue
        ece_val += np.abs(confidences[mask].mean() - accuracies[mask].mean()) * mask.mean()
    return ece_val


def mc_nll(logit_samples, y):
    """``batch_nll`` arithmetic (:127-151) on given logit samples (S, B, C)."""
    m = logit_samples.max(-1, keepdims=True)
    log_probs = logit_samples - m - np.log(np.exp(logit_samples - m).sum(-1, keepdims=True))
    S = logit_samples.shape[0]
    lp = np.take_along_axis(log_probs, y.astype(int)[None, :, None], axis=-1).squeeze(-1)
    mm = lp.max(0)
    log_avg = mm + np.log(np.exp(lp - mm).sum(0)) - np.log(S)
    probs = np.exp(log_probs).mean(0)
    return -log_avg.mean(), (probs.argmax(-1) == y).mean(), probs
