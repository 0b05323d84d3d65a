"""Oracle restatement of ``src/train_alpha.py:13-59`` (TEST INFRASTRUCTURE — see ``oracle/__init__.py``).
Literal: slogdet of I + rescale/alpha * W^T W, autograd for the gradient w.r.t. log alpha."""
import math

import torch

import lip_amd  # noqa: F401
from lip_amd.utils import flatten_nn_params

from .ggn import build_WTW, compute_W_vps


def log_marginal_likelihood(alpha, X, state, model_type, full_set_size=None):
    N = full_set_size or X.shape[0]
    rescale = N / X.shape[0]
    flat_p, _ = flatten_nn_params(state.params)
    D = flat_p.numel()
    W, WT = compute_W_vps(state, X, model_type, full_set_size=None)
    dummy = WT(torch.zeros(D, dtype=flat_p.dtype))
    d = dummy.numel()
    WTW = build_WTW(W, WT, tuple(dummy.shape), d, dtype=flat_p.dtype, block=1)
    alpha = torch.as_tensor(alpha, dtype=flat_p.dtype)
    _, logdet_lowrank = torch.linalg.slogdet(torch.eye(d, dtype=flat_p.dtype) + rescale / alpha * WTW)
    logdet_term = logdet_lowrank + D * torch.log(alpha)
    quad = -0.5 * alpha * torch.dot(flat_p, flat_p)
    norm = 0.5 * D * torch.log(alpha)
    return quad + norm - 0.5 * logdet_term


def grad_log_alpha(log_alpha, *lm_args):
    la = torch.tensor(float(log_alpha), dtype=torch.float64, requires_grad=True)
    val = log_marginal_likelihood(torch.exp(la), *lm_args)
    g, = torch.autograd.grad(val, la)
    return float(val), float(g)
