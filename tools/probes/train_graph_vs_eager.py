"""Eager loop vs recorded step of train_video.py: max |weight difference| after n steps, by parameter (dropout off)."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import train_video
root = os.path.dirname(os.path.abspath(train_video.__file__))
cfg = json.load(open(os.path.join(root, "conf", "video_config.json"))); cfg["tcn_dropout"] = 0.0
tmp = tempfile.mkdtemp(); json.dump(cfg, open(os.path.join(tmp, "cfg.json"), "w"))
def run(tag, steps, extra):
    argv = ["--save-path", os.path.join(tmp, tag), "--steps", str(steps), "--frames", "9", "--config-path", os.path.join(tmp, "cfg.json"), "--display", "100"] + extra
    train_video.main(argv)
    return torch.load(os.path.join(tmp, tag, "1.pt"), map_location="cpu")
init = None
for steps in (1, 2, 4):
    a = run(f"e{steps}", steps, []); b = run(f"f{steps}", steps, []); g = run(f"g{steps}", steps, ["--graph-step"])
    def md(u, v):
        worst = max(((float((u[k] - v[k]).abs().max()), k) for k in u if u[k].dtype.is_floating_point))
        return worst
    print(steps, "eager vs eager", md(a, b), "| eager vs graph", md(a, g), flush=True)
