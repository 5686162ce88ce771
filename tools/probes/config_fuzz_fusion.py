"""Config-space smoke of train_fusion.Trainer: fusion head x loss x {recorded, eager} in 'train' mode (two epochs of three steps), then 'av_test'
for audio arch x frames format x ragged / rectangular x use_plda at a tiny synthetic size.   python tools/probes/config_fuzz_fusion.py"""
import itertools
import math
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

os.chdir(tempfile.mkdtemp())
import train_fusion

bad = 0
small = {"data.n_spk": 6, "data.utt_per_spk": 4, "data.video_frames": 9, "data.audio_frames": 80, "train.bs": 12, "train.epoch": 2, "train.steps_per_epoch": 3,
         "data.test_speakers": 4, "data.test_utt_per_spk": 3, "data.trials": 60, "data.trial_targets": 12, "data.test_audio_frames": [60, 100],
         "data.test_video_frames": [5, 14], "test.write_store": False, "test.batch": 8}
for fus, loss, graph in itertools.product(["linear", "lowfer", "concat"], ["CrossEntropy", "LMCL"], [True, False]):
    tag = f"train fusion={fus} loss={loss} graph={graph}"
    try:
        tr = train_fusion.Trainer("train", overrides=dict(small, **{"model.fusion": fus, "train.loss": loss, "train.graph_step": graph}))
        losses = []
        for ep in range(2):
            tr.current_epoch = ep
            out = tr._train_epoch()
            losses.append(float(out[0] if isinstance(out, (tuple, list)) else out))
        if not all(math.isfinite(l) for l in losses):
            raise RuntimeError(f"loss {losses}")
        tr.close()
        print(f"{tag}: loss {losses[0]:.3f} -> {losses[1]:.3f}", flush=True)
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:220]}   <-- RAISED", flush=True)
        bad += 1
    torch.cuda.empty_cache()
for arch, frames, ragged, fus in itertools.product(["tdnn", "etdnn"], ["f32", "u8"], [True, False], ["linear", "concat"]):
    tag = f"av_test arch={arch} frames={frames} ragged={ragged} fusion={fus}"
    try:
        tr = train_fusion.Trainer("av_test", overrides=dict(small, **{"model.audio_config.arch": arch, "test.frames": frames, "data.test_ragged": ragged,
                                                                    "model.fusion": fus}))
        tabs = tr._extract(tr.lomgridtestset)
        ok = all(bool(torch.isfinite(t.emb).all()) for t in (tabs if isinstance(tabs, (list, tuple)) else tabs.values() if isinstance(tabs, dict) else [tabs])
                 if hasattr(t, "emb"))
        if not ok:
            raise RuntimeError("non-finite embeddings")
        tr.close()
        print(f"{tag}: ok", flush=True)
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:220]}   <-- RAISED", flush=True)
        bad += 1
    torch.cuda.empty_cache()
print(f"{bad} raised")
sys.exit(1 if bad else 0)
