"""Drop-in for the scoring half of the reference's models/audio_models/utils.py (:234-527): the same ten entry
points with the same ONE-argument signature -- ``eer(exp_dir)``, ``eer_cos_lomgrid / eer_cos_grid(exp_dir)``,
``eer_plda_lomgrid / _grid(exp_dir)``, ``eer_cos_{lomgrid,grid}_scorefusion(exp_dir)``,
``eer_cos_{lomgrid,grid}_featurefusion(exp_dir)`` -- reading the reference's on-disk store
(``exp/<exp_dir>/test_{em,xv}_*/**.npy``, the trial lists, the lip-embedding ``.npz`` files, ``exp/plda.pkl``) and
``feature_normalize``.  Implementation and path overrides: deeplip_amd/scoring_entry.py (one table read, one scoring
launch per call instead of 40 000 np.load + sklearn calls).  The in-memory forms (``EmbeddingTable`` + trial path) stay
available as ``eer_cos`` / ``score_fusion`` / ``feature_fusion_scores``."""
from deeplip_amd.scoring import (EmbeddingTable, cosine_scores, eer_cos, eer_from_scores,  # noqa: F401
                                 feature_fusion_scores, read_trial_list, roc_curve, score_fusion)
from deeplip_amd.scoring_entry import AUDIO_DEFAULTS as _DEFAULTS, make_entry_points as _make, set_paths  # noqa: F401
import numpy as np


def feature_normalize(data):
    """utils.py:524-527 (numpy, biased std) -- host-side helper kept verbatim in behaviour."""
    mu = np.mean(data, axis=0)
    std = np.std(data, axis=0)
    return (data - mu) / std


_e = _make(_DEFAULTS)
eer = _e["eer"]
eer_cos_lomgrid = _e["eer_cos_lomgrid"]
eer_cos_grid = _e["eer_cos_grid"]
eer_plda_lomgrid = _e["eer_plda_lomgrid"]
eer_plda_grid = _e["eer_plda_grid"]
eer_cos_lomgrid_scorefusion = _e["eer_cos_lomgrid_scorefusion"]
eer_cos_grid_scorefusion = _e["eer_cos_grid_scorefusion"]
eer_cos_lomgrid_featurefusion = _e["eer_cos_lomgrid_featurefusion"]
eer_cos_grid_featurefusion = _e["eer_cos_grid_featurefusion"]
