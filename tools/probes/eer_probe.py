"""Where does the EER of a synthetic trial list land?  (tuning aid for SyntheticAVSet's `session` / `jitter` knobs: bench.py's C4
wants a non-degenerate EER so that its agreement with the oracle's EER means something.)
    python tools/probes/eer_probe.py            # on a GPU box"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from deeplip_amd import fusion, packing, scoring
from deeplip_amd.synthetic import SyntheticAVSet, synthetic_trials

dev = torch.device("cuda", 0)
packing.set_precision("f16x3")
video, audio, _ = bench.build_models(dev, 80)
for session, jitter in ((0.0, 0.3), (0.5, 0.3), (1.0, 1.0), (1.5, 1.5), (2.5, 2.5)):
    ds = SyntheticAVSet(32, 8, 1, 29, 80, 300, key="probe", session=session, jitter=jitter)
    n = len(ds)
    xa = torch.cat([audio.extract_embedding(torch.from_numpy(ds.audio(list(range(b, b + 64)))).unsqueeze(1).to(dev))[0] for b in range(0, n, 64)])
    xv = torch.cat([video.embed(torch.from_numpy(ds.video(list(range(b, b + 64)))[0]).to(dev)) for b in range(0, n, 64)])
    em = fusion.fuse_av(xa, xv)
    y, pairs = synthetic_trials(ds, 20000, 4000)
    out = {}
    for name, t in (("audio", xa), ("video", xv), ("fused", em)):
        tab = scoring.EmbeddingTable(ds.utt_ids, t.contiguous())
        ia, ib = tab.trial_indices(pairs)
        s = scoring.cosine_scores(tab.emb, ia, ib).cpu().numpy()
        eer, thr = scoring.eer_from_scores(y, s)
        out[name] = (round(float(eer), 4), round(float(thr), 4))
    print(f"session {session} jitter {jitter}: {out}", flush=True)
