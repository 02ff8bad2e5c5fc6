"""Max-mixture GMM pose prior used inside the SMPLify-DC loop (reference:
tuch/smplify/prior.py:36-167, 8 Gaussians over the 69-D body pose).  [B,69] -> [B]; tiny,
stays on torch ops on the parameters' device (K8)."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn


class MaxMixturePrior(nn.Module):
    def __init__(self, prior_folder='prior', num_gaussians=6, dtype=torch.float32, epsilon=1e-16,
                 use_merged=True, gmm=None, **kwargs):
        """``gmm``: optional dict with 'means' [M,D], 'covars' [M,D,D], 'weights' [M]; otherwise
        ``prior_folder/gmm_{num_gaussians:02d}.pkl`` is read like the reference does (:56-76)."""
        super().__init__()
        if dtype not in (torch.float32, torch.float64):
            raise ValueError('Unknown float type {}'.format(dtype))
        self.num_gaussians = num_gaussians
        self.epsilon = epsilon
        self.use_merged = use_merged
        if gmm is None:
            path = os.path.join(prior_folder, 'gmm_{:02d}.pkl'.format(num_gaussians))
            if not os.path.exists(path):
                raise FileNotFoundError('The path to the mixture prior "{}" does not exist'.format(path))
            with open(path, 'rb') as f:
                gmm = pickle.load(f, encoding='latin1')
            if not isinstance(gmm, dict):
                gmm = {'means': gmm.means_, 'covars': gmm.covars_, 'weights': gmm.weights_}
        np_dtype = np.float32 if dtype == torch.float32 else np.float64
        means = np.asarray(gmm['means']).astype(np_dtype)
        covs = np.asarray(gmm['covars']).astype(np_dtype)
        weights = np.asarray(gmm['weights'])
        self.register_buffer('means', torch.tensor(means, dtype=dtype))
        self.register_buffer('covs', torch.tensor(covs, dtype=dtype))
        precisions = np.stack([np.linalg.inv(c) for c in covs]).astype(np_dtype)
        self.register_buffer('precisions', torch.tensor(precisions, dtype=dtype))
        # mixture weights folded with the Gaussian normalisers, relative to the tightest
        # component (reference :88-96)
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in np.asarray(gmm['covars'])])
        const = (2 * np.pi) ** (69 / 2.)
        nll_weights = np.asarray(weights / (const * (sqrdets / sqrdets.min())))
        self.register_buffer('nll_weights', torch.tensor(nll_weights, dtype=dtype).unsqueeze(0))
        self.register_buffer('log_nll_weights', torch.log(torch.tensor(nll_weights, dtype=dtype)).contiguous())
        self.register_buffer('weights', torch.tensor(weights, dtype=dtype).unsqueeze(0))
        self.register_buffer('pi_term', torch.log(torch.tensor(2 * np.pi, dtype=dtype)))
        cov_dets = [np.log(np.linalg.det(c.astype(np_dtype)) + epsilon) for c in covs]
        self.register_buffer('cov_dets', torch.tensor(cov_dets, dtype=dtype))
        self.random_var_dim = self.means.shape[1]

    def get_mean(self):
        return torch.matmul(self.weights, self.means)

    def merged_log_likelihood(self, pose, betas):
        """min_m [ 0.5 (p-mu_m)^T P_m (p-mu_m) - log w'_m ]  (reference :117-132)."""
        diff = pose.unsqueeze(1) - self.means
        quad = (torch.einsum('mij,bmj->bmi', self.precisions, diff) * diff).sum(-1)
        return (0.5 * quad - torch.log(self.nll_weights)).min(dim=1)[0]

    def log_likelihood(self, pose, betas, *args, **kwargs):
        """Per-component negative log-likelihood, then the arg-min component (reference :134-161)."""
        vals = []
        for m in range(self.num_gaussians):
            diff = pose - self.means[m]
            quad = torch.einsum('bi,bi->b', torch.einsum('bj,ji->bi', diff, self.precisions[m]), diff)
            cov_term = torch.log(torch.det(self.covs[m]) + self.epsilon)
            vals.append(quad + 0.5 * (cov_term + self.random_var_dim * self.pi_term))
        ll = torch.stack(vals, dim=1)
        idx = torch.argmin(ll, dim=1)
        return -torch.log(self.nll_weights[:, idx]) + ll[:, idx]

    def forward(self, pose, betas):
        return self.merged_log_likelihood(pose, betas) if self.use_merged else self.log_likelihood(pose, betas)
