"""Drop-in for the scoring half of the reference's models/fusion_models/utils.py (:234-527).
The reference's eer_cos_* read exp/<run>/.../*.npy from site-local paths; the equivalents here take
an in-memory deeplip_amd.scoring.EmbeddingTable and a trial-list path."""
from deeplip_amd.scoring import (EmbeddingTable, cosine_scores, eer_cos, eer_from_scores,  # noqa: F401
                                 feature_fusion_scores, read_trial_list, roc_curve, score_fusion)
import numpy as np


def feature_normalize(data):
    """utils.py:524-527 (numpy, biased std) -- host-side helper kept verbatim in behaviour."""
    mu = np.mean(data, axis=0)
    std = np.std(data, axis=0)
    return (data - mu) / std


eer_cos_lomgrid = eer_cos
eer_cos_grid = eer_cos
