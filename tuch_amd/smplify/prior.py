"""Max-mixture GMM pose prior used inside the SMPLify-DC loop.

Drop-in for ``tuch/smplify/prior.py`` (class name, constructor arguments, buffer names and
``forward(pose, betas)`` as in the reference, :36-167): 8 Gaussians over the 69-D body pose,
[B,69] -> [B].  Inside ``contact_fitting_loss`` the merged form is evaluated by the fused HIP kernel
(csrc/small_terms.hip); the torch expressions below serve every other caller.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn


def _read_mixture(prior_folder, num_gaussians):
    """means [M,D], covariances [M,D,D], weights [M] from gmm_{M:02d}.pkl (dict or sklearn GMM)."""
    path = os.path.join(prior_folder, 'gmm_{:02d}.pkl'.format(num_gaussians))
    if not os.path.exists(path):
        raise FileNotFoundError('The path to the mixture prior "{}" does not exist'.format(path))
    with open(path, 'rb') as handle:
        stored = pickle.load(handle, encoding='latin1')
    if isinstance(stored, dict):
        return stored['means'], stored['covars'], stored['weights']
    return stored.means_, stored.covars_, stored.weights_


class MaxMixturePrior(nn.Module):
    def __init__(self, prior_folder='prior', num_gaussians=6, dtype=torch.float32, epsilon=1e-16,
                 use_merged=True, gmm=None, **kwargs):
        """``gmm``: optional dict {'means', 'covars', 'weights'} instead of the pickle in ``prior_folder``."""
        super().__init__()
        if dtype not in (torch.float32, torch.float64):
            raise ValueError('Unknown float type {}'.format(dtype))
        np_dtype = np.float32 if dtype == torch.float32 else np.float64
        self.num_gaussians, self.epsilon, self.use_merged = num_gaussians, epsilon, use_merged
        if gmm is not None:
            mu, sigma, pi = gmm['means'], gmm['covars'], gmm['weights']
        else:
            mu, sigma, pi = _read_mixture(prior_folder, num_gaussians)
        mu64, sigma64, pi64 = np.asarray(mu), np.asarray(sigma), np.asarray(pi)
        sigma_t = sigma64.astype(np_dtype)

        def buf(name, array):
            self.register_buffer(name, torch.tensor(array, dtype=dtype))

        buf('means', mu64.astype(np_dtype))
        buf('covs', sigma_t)
        buf('precisions', np.stack([np.linalg.inv(c) for c in sigma_t]).astype(np_dtype))
        # mixture weights folded with each Gaussian's normaliser, relative to the tightest
        # component (reference :88-96): w_m / ((2 pi)^(69/2) * sqrt|S_m| / min_k sqrt|S_k|)
        root_det = np.sqrt(np.array([np.linalg.det(c) for c in sigma64]))
        folded = pi64 / ((2 * np.pi) ** (69 / 2.) * (root_det / root_det.min()))
        buf('nll_weights', np.asarray(folded)[None])
        self.register_buffer('log_nll_weights', torch.log(self.nll_weights[0]).contiguous())
        buf('weights', pi64[None])
        self.register_buffer('pi_term', torch.log(torch.tensor(2 * np.pi, dtype=dtype)))
        buf('cov_dets', [np.log(np.linalg.det(c) + epsilon) for c in sigma_t])
        self.random_var_dim = self.means.shape[1]

    def get_mean(self):
        """Mixture mean, [1,D]."""
        return self.weights @ self.means

    def merged_log_likelihood(self, pose, betas):
        """min_m [ 0.5 (p-mu_m)^T P_m (p-mu_m) - log w'_m ]  (reference :117-132)."""
        centred = pose[:, None, :] - self.means[None]
        mahalanobis = torch.einsum('bmi,mij,bmj->bm', centred, self.precisions, centred)
        return torch.min(0.5 * mahalanobis - torch.log(self.nll_weights), dim=1)[0]

    def log_likelihood(self, pose, betas, *args, **kwargs):
        """Per-component negative log-likelihood and the selected component's weight (reference :134-161)."""
        per_component = []
        for m in range(self.num_gaussians):
            centred = pose - self.means[m]
            mahalanobis = torch.einsum('bi,ij,bj->b', centred, self.precisions[m], centred)
            log_det = torch.log(torch.det(self.covs[m]) + self.epsilon)
            per_component.append(mahalanobis + 0.5 * (log_det + self.random_var_dim * self.pi_term))
        table = torch.stack(per_component, dim=1)
        chosen = torch.argmin(table, dim=1)
        return table[:, chosen] - torch.log(self.nll_weights[:, chosen])

    def forward(self, pose, betas):
        if self.use_merged:
            return self.merged_log_likelihood(pose, betas)
        return self.log_likelihood(pose, betas)
