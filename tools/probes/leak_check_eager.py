"""Eager speech-encoder training steps (B = 64 x 200 frames: the on-load backward paths with their unwritten placeholder tensors and the attributes
that travel on them) -- device memory and host RSS after 80 and after 240 steps (the caching allocators' pools fill during the first epochs).   python tools/probes/leak_check_eager.py"""
import os
import resource
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

os.chdir(tempfile.mkdtemp())
import train_audio


def mem():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return dict(alloc=torch.cuda.memory_allocated() >> 20, reserved=torch.cuda.memory_reserved() >> 20, device_used=(total - free) >> 20,
                rss=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)


ta = train_audio.Trainer(overrides={"data.n_spk": 12, "data.utt_per_spk": 8, "train.bs": 64, "train.crop_frames": [200, 200], "data.audio_frames": 200,
                                    "train.steps_per_epoch": 10, "train.graph_step": False})
m0 = None
for ep in range(24):
    ta.current_epoch = ep
    ta._train_epoch()
    if ep == 7:
        m0 = mem()
    if ep % 4 == 3:
        print(ep, mem(), flush=True)
m1 = mem()
grow = {k: m1[k] - m0[k] for k in m0}
print(f"epoch 8 {m0} last {m1} growth {grow} MiB ({ta.last_epoch_stats['step_mode']})")
sys.exit(1 if max(grow["alloc"], grow["device_used"]) > 128 or grow["rss"] > 256 else 0)
