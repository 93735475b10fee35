"""Acceptance metric of the hot path: the classifier two-sample test.

Same estimator and defaults as sbi/utils/metrics.py:56-175 (random forest with 100
trees, 5-fold CV, z-scored inputs, seed 1); 0.5 = indistinguishable samples.
"""

from __future__ import annotations

import numpy as np
import torch
from torch import Tensor


def c2st(X: Tensor, Y: Tensor, seed: int = 1, n_folds: int = 5, metric: str = "accuracy",
         classifier: str = "rf", z_score: bool = True, noise_scale=None, verbosity: int = 0) -> Tensor:
    """Cross-validated accuracy of a classifier telling X from Y."""
    from sklearn.ensemble import RandomForestClassifier
    from sklearn.model_selection import KFold, cross_val_score
    from sklearn.neural_network import MLPClassifier

    X, Y = X.detach().cpu(), Y.detach().cpu()
    if z_score:
        X_mean, X_std = torch.mean(X, dim=0), torch.std(X, dim=0)
        X_std[X_std == 0] = 1.0   # constant dims stay constant instead of turning into NaN
        X = (X - X_mean) / X_std
        Y = (Y - X_mean) / X_std
    if noise_scale is not None:
        X = X + noise_scale * torch.randn(X.shape)
        Y = Y + noise_scale * torch.randn(Y.shape)
    ndim = X.shape[-1]
    if classifier == "rf":
        clf = RandomForestClassifier(random_state=seed)
    elif classifier == "mlp":
        clf = MLPClassifier(activation="relu", hidden_layer_sizes=(10 * ndim, 10 * ndim), max_iter=1000,
                            solver="adam", early_stopping=True, n_iter_no_change=50, random_state=seed)
    else:
        raise ValueError(f"Invalid classifier: {classifier}. Use 'rf' or 'mlp'.")
    data = np.concatenate((X.numpy(), Y.numpy()))
    target = np.concatenate((np.zeros(X.shape[0]), np.ones(Y.shape[0])))
    shuffle = KFold(n_splits=n_folds, shuffle=True, random_state=seed)
    scores = cross_val_score(clf, data, target, cv=shuffle, scoring=metric, verbose=verbosity)
    return torch.from_numpy(np.atleast_1d(np.mean(scores).astype(np.float32)))


def check_c2st(x: Tensor, y: Tensor, alg: str, tol: float = 0.1) -> None:
    """Raise if the C2ST accuracy is further than ``tol`` from chance (0.5)."""
    score = c2st(x, y).item()
    print(f"c2st for {alg} is {score:.2f}.")
    assert (0.5 - tol) <= score <= (0.5 + tol), (
        f"{alg}'s c2st={score:.2f} is too far from the desired near-chance performance."
    )
