"""KNN validation estimator for the pretraining loop (reference: train_utils/knn.py:7-42).

The reference collects the un-projected backbone features of the training set on the host and fits
`sklearn.neighbors.KNeighborsClassifier()` (k = 5, uniform weights, Euclidean).  Here features never leave the GPU:
`GpuKNNClassifier` keeps them resident, gets all query-to-train distances from one exact-fp32 MFMA GEMM
(`focal_linear_fwd`: ||q||^2 + ||t||^2 - 2 q.t^T), takes the 5 nearest with torch.topk and votes; ties between classes go to
the smallest label, which is what sklearn's `mode` does.  Predictions equal sklearn's on the same features
(tests/test_kernels_gpu.py::test_gpu_knn_matches_sklearn)."""
import os
import sys

import torch

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd import ops  # noqa: E402


def extract_sample_features(args, classifier, aug_freq_loc_inputs, proj_head=False):
    """Concatenated per-modality features [B, M * D] (reference :7-19)."""
    if args.learn_framework not in {"FOCAL"}:
        raise Exception(f"Invalid learn framework ({args.learn_framework}) provided")
    mod_features = classifier(aug_freq_loc_inputs, class_head=False, proj_head=proj_head)
    return torch.cat([mod_features[mod] for mod in args.dataset_config["modality_names"]], dim=1)


class GpuKNNClassifier:
    def __init__(self, n_neighbors=5):
        self.k = n_neighbors
        self.x = self.y = self.x_sq = None
        self.n_classes = 0

    def fit(self, features, labels):
        self.x = features.detach().float().contiguous()
        self.y = labels.detach().to(self.x.device).long()
        self.x_sq = (self.x * self.x).sum(1)
        self.n_classes = int(self.y.max().item()) + 1
        return self

    @torch.no_grad()
    def predict(self, features, chunk=8192):
        q = features.detach().float().contiguous()
        k = min(self.k, self.x.shape[0])
        out = []
        for lo in range(0, q.shape[0], chunk):
            qc = q[lo:lo + chunk]
            dots, _ = ops.linear(qc, self.x, None, compute=torch.float32, y_dtype=torch.float32)  # [nq, nt] = q . t^T
            d2 = (qc * qc).sum(1, keepdim=True) + self.x_sq[None, :] - 2.0 * dots
            idx = torch.topk(d2, k, dim=1, largest=False).indices
            votes = torch.zeros(qc.shape[0], self.n_classes, device=q.device)
            votes.scatter_add_(1, self.y[idx], torch.ones_like(idx, dtype=votes.dtype))
            out.append(votes.argmax(1))  # first maximum = smallest label among tied classes
        return torch.cat(out)


def compute_knn(args, classifier, augmenter, data_loader_train):
    """Fit the estimator on the training set's features (reference :22-42)."""
    classifier.eval()
    feats, labels = [], []
    with torch.no_grad():
        for time_loc_inputs, y in data_loader_train:
            aug_freq_loc_inputs, _ = augmenter.forward("no", time_loc_inputs, y)
            feats.append(extract_sample_features(args, classifier, aug_freq_loc_inputs))
            labels.append(y.argmax(dim=1) if y.dim() > 1 else y)
    feats, labels = torch.cat(feats), torch.cat(labels).to(feats[0].device)
    from focal_amd import distributed as fdist
    if fdist.is_dist():  # the training loader hands every rank an equal share: the estimator is fitted on all of them
        feats, labels = fdist.all_gather_rows(feats), fdist.all_gather_rows(labels)
    return GpuKNNClassifier().fit(feats, labels)
