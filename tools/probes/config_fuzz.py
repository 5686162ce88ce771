"""Config-space smoke of train_audio.Trainer: every arch x pooling x loss x {recorded, eager} x {frozen encoder or not} at a tiny synthetic size --
two epochs of two steps, then the test-mode extraction + EER.  Looks for combinations that raise or give a non-finite loss (the tests fix a
handful of combinations).   python tools/probes/config_fuzz.py"""
import itertools
import math
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

os.chdir(tempfile.mkdtemp())
import train_audio

bad = 0
combos = list(itertools.product(["tdnn", "etdnn", "resnet"], ["statistic", "attentive_statistic", "average"], ["LMCL", "AAMSoftmax", "CrossEntropy"],
                                [True, False], [False, True]))
for arch, pooling, loss, graph, frozen in combos:
    if (arch == "resnet") != (pooling == "average"):
        continue                                   # (the ResNet pools by averaging; the TDNNs by statistics)
    if frozen and not graph:
        continue
    ov = {"model.arch": arch, "data.n_spk": 6, "data.utt_per_spk": 4, "data.audio_frames": 120, "train.crop_frames": [80, 120], "train.bs": 12,
          "train.epoch": 2, "train.steps_per_epoch": 2, "train.loss": loss, "train.graph_step": graph, "data.test_speakers": 4,
          "data.test_utt_per_spk": 3, "data.trials": 60, "data.trial_targets": 12, "data.test_audio_frames": [60, 100], "test.write_store": False}
    if arch != "resnet":
        ov[f"model.{arch}.pooling"] = pooling
    if frozen:
        ov["train.freeze_encoder"] = True
    if pooling == "attentive_statistic":
        # At the config's lr 0.01 one LMCL step sharpens the attention to one-hot on synthetic data, a channel's weighted variance becomes
        # 0 (or -5e-7 in fp32) and its square root's gradient (or the root itself) is not finite -- in the REFERENCE class as well
        # (pooling.py:104; EXPERIMENTS R6.6).  The mechanics are checked at a learning rate both sides survive.
        ov["train.sgd.init_lr"] = 1.0e-4
    tag = f"arch={arch} pooling={pooling} loss={loss} graph={graph} frozen={frozen}"
    try:
        tr = train_audio.Trainer(overrides=ov)
        losses = []
        for ep in range(2):
            tr.current_epoch = ep
            losses.append(tr._train_epoch())
        if not all(math.isfinite(l) for l in losses):
            raise RuntimeError(f"loss {losses}")
        tab = tr._xvectors(tr.voxtestset, batch=4, normalize=False)
        if not bool(torch.isfinite(tab.emb).all()):
            raise RuntimeError("non-finite x-vectors")
        mode = tr.last_epoch_stats.get("step_mode")
        tr.close()
        print(f"{tag}: loss {losses[0]:.3f} -> {losses[1]:.3f} ({mode})", flush=True)
    except Exception as ex:
        print(f"{tag}: {type(ex).__name__}: {str(ex)[:220]}   <-- RAISED", flush=True)
        bad += 1
    torch.cuda.empty_cache()
print(f"{bad} raised")
sys.exit(1 if bad else 0)
