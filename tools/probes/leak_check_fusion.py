"""train_fusion.Trainer('train') over 16 small epochs, recorded and eager: device memory and host RSS after epoch 4 and after the last.
   python tools/probes/leak_check_fusion.py"""
import os
import resource
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch

os.chdir(tempfile.mkdtemp())
import train_fusion


def mem():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return dict(alloc=torch.cuda.memory_allocated() >> 20, reserved=torch.cuda.memory_reserved() >> 20, device_used=(total - free) >> 20,
                rss=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)


bad = False
for graph in (True, False):
    tr = train_fusion.Trainer("train", overrides={"data.n_spk": 6, "data.utt_per_spk": 4, "data.video_frames": 9, "data.audio_frames": 80, "train.bs": 12,
                                                  "train.steps_per_epoch": 6, "train.graph_step": graph})
    m0 = None
    for ep in range(16):
        tr.current_epoch = ep
        tr._train_epoch()
        if ep == 3:
            m0 = mem()
    m1 = mem()
    grow = {k: m1[k] - m0[k] for k in m0}
    print(f"graph_step={graph}: epoch 4 {m0} last {m1} growth {grow} MiB", flush=True)
    bad |= max(grow["alloc"], grow["reserved"], grow["device_used"]) > 64 or grow["rss"] > 128
    tr.close()
    del tr
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
