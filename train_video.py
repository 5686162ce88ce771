#!/usr/bin/env python3
"""train_video.py -- lip-clip encoder entry point (ResNet-18 + MS-TCN) on the MI355X engine.

Re-creation of the reference's train_video.py surface: the same argparse flags (:31-68), JSON model
config (conf/video_config.json), ``get_model`` (:173-190), ``extract_feats`` (:99-106), ``train``
(:108-169: Adam 3e-4 / wd 1e-4, CosineAnnealingLR(T_max=5) stepped PER ITERATION, CrossEntropy,
per-epoch ``<save_path>/<epoch+1>.pt`` state-dict checkpoints), ``main``.  Data are synthetic clips
produced with ``pad_packed_collate`` semantics (zero-pad to the longest clip + lengths list,
models/video_models/dataset.py:123-139); ``--rgb`` feeds uint8 [B,T,3,88,88] frames through the
GPU ingest kernel (gray + (x/255-0.421)/0.165).

What trains: the FULL model, as in the reference (``model.train()``: batch-statistics BatchNorm in stem /
trunk / TCN, learnable PReLU slopes, dropout; Adam over all parameters) -- every forward and backward
step a ``dlip_*`` launch (deeplip_amd/autograd_video.py: conv dgrad / wgrad on the implicit-GEMM kernels,
train-mode BN, PReLU, max-pool, pooling kernels), pinned by a golden step captured from the reference
class (tests/test_train_video_gpu.py).  ``--head-only`` trains the classifier layer ``tcn.tcn_output`` on
frozen eval-mode features instead (the fast path on the fused inference kernels).  Under
``torch.distributed.run`` every rank draws its own batches and the gradients are averaged by bucketed
all-reduce per step over RCCL (the reference uses DataParallel: train_video.py:196).
Round 6: the optimisation step is RECORDED once per batch shape and replayed as one HIP graph by default (``--eager-step`` keeps the
loop of eager launches; deeplip_amd/train_plan.py), the next batch's host-to-device copies run behind the step in flight, metrics stay
on the device between ``--display`` points; ``--arith auto|f16x3|f32`` (default: $DLIP_ARITH, the config's "arith", auto) picks the
engine's arithmetic (deeplip_amd/arith.py).  ``--device cpu`` runs the plumbing only (config -> model -> batches -> optimiser/scheduler ->
checkpoint round trip) because the engine has no CPU arithmetic by design.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deeplip_amd import weightgen as wg  # noqa: E402
from models.video_models.model import Lipreading  # noqa: E402

SEED = 1


def load_args(argv=None):
    p = argparse.ArgumentParser(description="Lipreading on the deeplip_amd engine")
    p.add_argument("--dataset", default="lomgrid", help="dataset selection")
    p.add_argument("--num-classes", type=int, default=54, help="Number of classes (database/lomgrid_54SpeakerLabel.txt)")
    p.add_argument("--label-path", type=str, default=None, help="Path to txt file with labels")
    p.add_argument("--backbone-type", type=str, default="resnet", choices=["resnet", "shufflenet"])
    p.add_argument("--relu-type", type=str, default="prelu", choices=["relu", "prelu"])
    p.add_argument("--width-mult", type=float, default=1.0)
    p.add_argument("--lr", type=float, default=0.0003)
    p.add_argument("--maxepoch", type=int, default=1)
    p.add_argument("--tcn-kernel-size", type=int, nargs="+")
    p.add_argument("--tcn-num-layers", type=int, default=4)
    p.add_argument("--tcn-dropout", type=float, default=0.2)
    p.add_argument("--tcn-dwpw", default=False, action="store_true")
    p.add_argument("--tcn-width-mult", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=4)
    p.add_argument("--model-path", type=str, default=None, help="Pretrained model pathname (bare state_dict)")
    p.add_argument("--extract-feats", default=False, action="store_true")
    p.add_argument("--mouth-patch-path", type=str, default=None)
    p.add_argument("--mouth-embedding-out-path", type=str, default=None)
    p.add_argument("--config-path", type=str, default=os.path.join(ROOT, "conf/video_config.json"))
    p.add_argument("--display", type=int, default=1)
    p.add_argument("--save-path", type=str, default="exp/video/epoch")
    # build-owned
    p.add_argument("--device", default="gpu", choices=["gpu", "cpu"])
    p.add_argument("--gpus", type=int, default=1, help="GPUs of this node, one process each over RCCL (the reference: "
                   "nn.DataParallel over every visible GPU, train_video.py:206-207); N > 1 starts its own N-rank job")
    p.add_argument("--steps", type=int, default=2, help="synthetic iterations per epoch")
    p.add_argument("--frames", type=int, default=29)
    p.add_argument("--rgb", action="store_true",
                   help="feed uint8 RGB frames [B,T,3,S,S] (S = --frame-size) through the ingest kernel: model.train() batches get the "
                        "reference's train pipeline -- RandomCrop(88) + HorizontalFlip(0.5) per clip, dataloaders.py:13-17 -- everything "
                        "else the val pipeline's CenterCrop(88)")
    p.add_argument("--frame-size", type=int, default=96, help="side of the synthetic uint8 mouth crops (LRW ROIs are 96 x 96)")
    p.add_argument("--head-only", action="store_true", help="train tcn.tcn_output on frozen eval-mode features")
    p.add_argument("--graph-step", action="store_true",
                   help="(the default since round 6; kept for old command lines) record the optimisation step and replay it as one HIP graph")
    p.add_argument("--eager-step", action="store_true",
                   help="issue every launch of every step from Python instead of replaying a recorded step (deeplip_amd.train_plan: one "
                        "recorded HIP graph per batch shape, recorded the second time a shape is met; the default for full-model training)")
    p.add_argument("--data-cache", type=int, default=0, metavar="N",
                   help="synthetic source: generate N batches once, keep them in pinned host memory and cycle through them (the numpy "
                        "generator makes ~30 clips/s; a run that measures the TRAINER rather than the generator uses this)")
    from deeplip_amd import arith
    arith.add_argument(p)
    return p.parse_args(argv)


def load_json(path):
    with open(path) as f:
        return json.load(f)


def get_model(args):
    """train_video.py:173-190."""
    j = load_json(args.config_path)
    args.backbone_type, args.width_mult, args.relu_type = j["backbone_type"], j["width_mult"], j["relu_type"]
    tcn_options = {"num_layers": j["tcn_num_layers"], "kernel_size": j["tcn_kernel_size"], "dropout": j["tcn_dropout"],
                   "dwpw": j["tcn_dwpw"], "width_mult": j["tcn_width_mult"]}
    return Lipreading(num_classes=args.num_classes, tcn_options=tcn_options, backbone_type=args.backbone_type,
                      relu_type=args.relu_type, width_mult=args.width_mult, extract_feats=args.extract_feats)


def pad_packed_collate(batch):
    """dataset.py:123-139 semantics: sort by length (desc), zero-pad to the longest, lengths list."""
    batch = sorted(batch, key=lambda x: x[0].shape[0], reverse=True)
    lengths = [a.shape[0] for a, _ in batch]
    data = np.zeros((len(batch), lengths[0]) + batch[0][0].shape[1:], dtype=batch[0][0].dtype)
    for i, (a, _) in enumerate(batch):
        data[i, :a.shape[0]] = a
    return torch.from_numpy(data), lengths, torch.LongTensor([b for _, b in batch])


def synthetic_batch(args, it, rgb=False):
    r = np.random.Generator(np.random.PCG64([SEED, it]))
    items = []
    for i in range(args.batch_size):
        spk = int(r.integers(args.num_classes))
        T = args.frames if i == 0 else int(r.integers(max(2, args.frames // 3), args.frames + 1))
        size = args.frame_size if rgb else 88
        clip = wg.video_input(1, T, size, key=f"tv.{it}.{i}", speakers=[spk])[0, 0]          # [T,S,S] normalised gray
        if rgb:
            g = np.clip((clip * 0.165 + 0.421) * 255.0, 0, 255).astype(np.uint8)
            clip = np.repeat(g[:, None], 3, axis=1)                                            # [T,3,S,S] uint8: the loader's frames
        items.append((clip, spk))
    return pad_packed_collate(items)


def extract_feats(model, clip_thw, device):
    """:99-106: model(FloatTensor(data)[None,None], lengths=[T]) with extract_feats=True."""
    model.eval()
    x = torch.as_tensor(clip_thw, dtype=torch.float32)[None, None].to(device)
    y = model(x, lengths=[x.shape[2]])
    from deeplip_amd import _lib
    _lib.check_range(sync=True)           # the caller consumes y next: a range report of this very forward surfaces now
    return y


def train(model, args, device):
    from deeplip_amd import autograd as ag, dist as ddist, ops
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    head = model.tcn.tcn_output
    full = not args.head_only and device.type != "cpu"
    if not full:
        for p in model.parameters():
            p.requires_grad = False
        for p in head.parameters():
            p.requires_grad = True
    params = [p for p in model.parameters() if p.requires_grad]
    # DP: gradients live in flat buckets whose all-reduces (RCCL) start while backward is still running
    buckets = ddist.GradBuckets(params) if (ddist.active() and device.type != "cpu") else None
    # (with GradBuckets the recorded step contains the bucket all-reduces -- RCCL collectives captured into the graph: exercised at one
    # rank by tests/test_rccl_gpu.py; DLIP_GRAPH_WITH_BUCKETS=0 keeps data-parallel runs on the eager loop)
    graph_step = not bool(getattr(args, "eager_step", False)) and full and (buckets is None or os.environ.get("DLIP_GRAPH_WITH_BUCKETS", "1") != "0")
    if graph_step:      # a recorded step reads its learning rate from a device tensor (the scheduler updates it in place)
        optimizer = torch.optim.Adam(params, lr=torch.tensor(float(args.lr), device=device), weight_decay=1e-4, capturable=True, fused=True)
    else:
        optimizer = torch.optim.Adam(params, lr=args.lr, weight_decay=1e-4)                  # (:112-113)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=5, eta_min=4e-08)      # (:114)
    last = None
    aug_rng = random.Random(SEED + rank)      # the train pipeline's crop / flip draws (the reference: the `random` module, preprocess.py)
    frontend = None
    if device.type == "cpu" and rank == 0:
        print("[plumbing] --device cpu exercises config / collate / optimizer / scheduler / checkpoint plumbing ONLY (BASELINE config C1): the "
              "engine has no CPU arithmetic path by design, so no forward runs and no loss or logits exist here; the same command on "
              "--device gpu computes them (C1-size run: tests/test_entrypoints.py::test_train_video_gpu_two_steps[c1-size]).")
    plan = None
    if graph_step:
        from deeplip_amd.train_plan import ShapeKeyedSteps, grad_witness, step_state

        def one_step(xb, lb, ln):
            if buckets is None:
                optimizer.zero_grad(set_to_none=True)
            else:
                buckets.zero()                       # the gradients are views into the buckets: cleared in place
            lg = model(xb, lengths=ln)
            ls = ag.margin_ce_loss(lg, lb, 1.0, 0.0)
            ls.backward()                            # bucket all-reduces start from the hooks as the gradients land
            if buckets is not None:
                buckets.finish()
            optimizer.step()
            return ls, lg

        # One recorded step per batch shape (pad_packed_collate pads to the batch's longest clip, dataset.py:123-139: a run meets a
        # few shapes); a shape's first step runs eagerly, its second records.  In a job of several ranks the first replay is
        # checked against an eager step from the same state and the run falls back to eager steps on any doubt (train_plan.py).
        # (with GradBuckets the branches stay on one stream: a bucket's all-reduce is enqueued behind the CURRENT stream of the hook
        # that completes it, and would not wait for gradients another branch stream is still writing)
        plan = ShapeKeyedSteps(one_step, eager_steps=1, device=device, branch_streams=buckets is None,
                               state=step_state([model], [optimizer], buckets), witness=grad_witness([model], buckets))
    source = _BatchSource(args, world, rank, device)
    copy_stream = torch.cuda.Stream(device=device) if device.type != "cpu" else None

    def stage(i):
        """Batch i on the device: host batch (generated, or from the --data-cache ring) -> asynchronous copies on the copy stream,
        behind the step that is running.  The reference's loop does `.cuda()` in front of every forward (train_video.py:125)."""
        nonlocal frontend
        inputs, lengths, labels = source.get(i)
        cur = torch.cuda.current_stream(device)
        with torch.cuda.stream(copy_stream):
            lab = labels.to(device, non_blocking=True)
            ln = torch.as_tensor(lengths, dtype=torch.int32).to(device, non_blocking=True)
            raw = inputs.to(device, non_blocking=True)
            cp = None
            if args.rgb and full:
                cp = torch.from_numpy(ops.draw_clip_params(inputs.shape[0], inputs.shape[-2], inputs.shape[-1], 88, rng=aug_rng)).to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return raw, lab, ln, cp, lengths, ev

    total_steps = int(args.maxepoch) * args.steps
    staged = stage(0) if device.type != "cpu" and total_steps > 0 else None
    acc = torch.zeros(3, dtype=torch.float64, device=device) if device.type != "cpu" else None      # sum loss*n, correct, n: read at display time only
    last_loss = last_shape = None
    # throughput of the loop as it runs: counted from the 4th step on (a shape's first step is eager, its second records and replays)
    import time
    mark_at, t_mark, t_end = min(3, max(total_steps - 1, 0)), None, None
    for epoch in range(int(args.maxepoch)):
        if acc is not None:
            acc.zero_()
        model.train() if full else model.eval()                                                # (:129)
        for it in range(args.steps):
            gi = epoch * args.steps + it
            if device.type == "cpu":
                inputs, lengths, labels = source.get(gi)
                print(f"[plumbing] batch {tuple(inputs.shape)} lengths {lengths} labels {labels.tolist()} lr {sched.get_last_lr()}")
                optimizer.step(); sched.step()
                continue
            if gi == mark_at:
                torch.cuda.synchronize(device)
                t_mark = time.perf_counter()
            raw, labels, ln, cp, lengths, ev = staged
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            for t in (raw, labels, ln, cp):
                if t is not None:
                    t.record_stream(cur)
            if args.rgb:
                # uint8 frames -> normalised 88 x 88 gray clips on the GPU: RandomCrop + HorizontalFlip per clip while the model
                # trains (dataloaders.py:13-17), CenterCrop otherwise (:19-24); padding frames = zeros of the normalised clip
                from deeplip_amd.frontend import VideoFrontend
                frontend = frontend or VideoFrontend(88)
                x = frontend(raw, clip_params=cp, lengths=ln)
            else:
                x = raw.unsqueeze(1)                                                                    # :125
            if plan is not None:
                loss, logits = plan.step(x.contiguous(), labels, ln)
            else:
                optimizer.zero_grad(set_to_none=buckets is None)
                if full:
                    logits = model(x, lengths=lengths)                          # (:140) whole graph on the engine
                else:
                    with torch.no_grad():
                        pooled = model.classifier_features(x, lengths)         # frozen stem + trunk + MS-TCN
                    logits = ag.linear(pooled, head.weight, head.bias)          # tcn_output (model.py:27)
                loss = ag.margin_ce_loss(logits, labels, 1.0, 0.0)              # nn.CrossEntropyLoss (:115,143)
                loss.backward()
                if buckets is not None:
                    buckets.finish()                                            # wait for the bucket all-reduces, average
                optimizer.step()
            sched.step()                                                        # per-iteration (:147)
            # metrics stay on the device (a recorded step's outputs are rewritten by the next replay: folded in right behind it)
            n_b = labels.shape[0]
            _, pred = torch.max(torch.softmax(logits.detach(), 1), 1)           # (:145)
            acc[0] += loss.detach().double() * n_b
            acc[1] += (pred == labels).sum()
            acc[2] += n_b
            last_loss, last_shape = loss.detach().clone(), tuple(logits.shape)
            staged = stage(gi + 1) if gi + 1 < total_steps else None            # the next batch's copies run behind this step
            if it % args.display == 0 or it + 1 == args.steps:
                if plan is not None:
                    plan.finish()                                               # waits; a range report of the f16x3 arithmetic surfaces here
                a0, a1, a2 = acc.tolist()
                if rank == 0 and it % args.display == 0:
                    print(f"epoch {epoch} it {it} loss {a0 / a2:.4f} acc {a1 / a2:.3f} lr {float(sched.get_last_lr()[0]):.2e}"
                          f"{' (replayed)' if plan is not None and plan.last.recorded else ''}", flush=True)
        if device.type != "cpu" and epoch + 1 == int(args.maxepoch):
            torch.cuda.synchronize(device)
            t_end = time.perf_counter()                                                        # (the checkpoint below is not a step)
        if rank == 0:
            os.makedirs(args.save_path, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(args.save_path, f"{epoch + 1}.pt"))  # (:169)
    last = (float(last_loss), last_shape) if last_loss is not None else None
    train.last_stats = None
    if t_mark is not None and total_steps > mark_at and t_end is not None:
        dt = t_end - t_mark
        n = (total_steps - mark_at) * args.batch_size * world
        train.last_stats = {"clips_per_s": n / dt, "ms_per_step": 1e3 * dt / (total_steps - mark_at), "steps_timed": total_steps - mark_at,
                            "step_mode": plan.mode if plan is not None else "eager", "global_batch": args.batch_size * world}
        if rank == 0:
            print("train: {:.1f} clips/s, {:.2f} ms/step over the last {} steps [{}]".format(
                train.last_stats["clips_per_s"], train.last_stats["ms_per_step"], total_steps - mark_at, train.last_stats["step_mode"]), flush=True)
    if rank == 0 and device.type != "cpu":
        how = ("recorded steps: " + str(plan.summary())) if plan is not None else "eager steps"
        print(f"train: {how}", flush=True)
    train.last_plan = plan
    return last


class _BatchSource:
    """The synthetic loader behind train(): ``get(i)`` -> (inputs, lengths, labels) of iteration i as pad_packed_collate hands them
    over (dataset.py:123-139).  ``--data-cache N``: the first N batches are generated once, pinned, and walked cyclically."""

    def __init__(self, args, world, rank, device):
        self.args, self.world, self.rank = args, world, rank
        self.n = int(getattr(args, "data_cache", 0) or 0)
        self.cache = {}
        self.pin = device.type != "cpu"

    def get(self, i):
        k = i % self.n if self.n > 0 else i
        if k in self.cache:
            return self.cache[k]
        b = synthetic_batch(self.args, k * self.world + self.rank, self.args.rgb)
        if self.n > 0:
            b = (b[0].pin_memory() if self.pin else b[0], b[1], b[2])
            self.cache[k] = b
        return b


def main(argv=None):
    args = load_args(argv)
    from deeplip_amd import arith
    mode = arith.configure(args.arith, load_json(args.config_path).get("arith"))     # before the first weight pack
    if args.device == "gpu" and args.gpus > 1:
        from deeplip_amd import launch
        rc = launch.maybe_self_launch(os.path.abspath(__file__), list(sys.argv[1:] if argv is None else argv), args.gpus)
        if rc is not None:      # this process was the launcher of the N-rank job; nothing here touched the GPU
            sys.exit(rc)
    # the reference seeds once (train_video.py:70-73); under DP every rank needs its own dropout masks
    torch.manual_seed(SEED + int(os.environ.get("RANK", "0"))); np.random.seed(SEED)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))) if args.device == "gpu" else torch.device("cpu")
    if args.device == "gpu" and not torch.cuda.is_available():
        raise RuntimeError("no ROCm GPU visible: use --device cpu for the plumbing-only run")
    model = get_model(args)
    if args.model_path and os.path.exists(args.model_path):
        model.load_state_dict(torch.load(args.model_path, map_location="cpu"))
    else:
        sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, prefix="video.")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(device)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    in_job = "RANK" in os.environ            # any torch.distributed.run job joins its process group, a one-rank one included
    if in_job and args.device == "gpu":
        import torch.distributed as dist
        from deeplip_amd import dist as ddist
        torch.cuda.set_device(device)
        ddist.init_from_env(device)
        ddist.broadcast_params(list(model.parameters()) + list(model.buffers()))
    if args.extract_feats and args.mouth_patch_path:
        out = extract_feats(model, np.load(args.mouth_patch_path)["data"], device)
        if args.mouth_embedding_out_path:
            np.savez(args.mouth_embedding_out_path, data=out.cpu().numpy())
        return out
    res = train(model, args, device)
    if in_job and args.device == "gpu":
        import torch.distributed as dist
        dist.barrier()
    # checkpoint round trip (bare state_dict, as the reference saves it)
    ck = os.path.join(args.save_path, f"{int(args.maxepoch)}.pt")
    model.load_state_dict(torch.load(ck, map_location="cpu"))
    if int(os.environ.get("RANK", "0")) == 0:
        print("done:", res, "checkpoint", ck)
    if in_job and args.device == "gpu":
        import torch.distributed as dist
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
