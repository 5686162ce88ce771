#!/usr/bin/env python3
"""train_audio.py -- x-vector (TDNN / E-TDNN) entry point on the MI355X engine.

Re-creation of the reference's train_audio.py surface that the A+V hot path uses: config schema
(conf/audio_config.yaml: data / model / train / test), ``Trainer`` with ``extract_test_xv`` (x-vector
extraction + F.normalize, train_audio.py:343-373), ``model_average`` (checkpoint averaging,
:216-232), ``_adjust_margin`` (:141-145), ``save`` / ``load``; modes ``test`` (extract + cosine EER)
and ``train``.  ``train`` is the reference's loop (train_audio.py:167-199): ``model.train()``, forward
through every TDNN layer with batch-statistics BatchNorm, LMCL / CrossEntropy, ``loss.backward()`` through the
whole encoder (conv dgrad / wgrad, BN and pooling backward as dlip_* launches -- deeplip_amd/autograd.py),
SGD over model + criterion parameters, MultiStepLR, margin schedule, per-epoch checkpoints and checkpoint
averaging.  ``train.freeze_encoder: true`` keeps the older criterion-only step on frozen x-vectors.

Data parallelism (the reference wraps the model in nn.DataParallel over ``gpus_id``, train_audio.py:80-83): launched
under ``torch.distributed.run`` every rank draws its own batches, replicas start from rank 0's weights, gradients are
averaged by bucketed all-reduces over RCCL that overlap the backward pass (deeplip_amd.dist.GradBuckets), metrics are
summed over ranks and rank 0 alone writes checkpoints.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deeplip_amd import _lib, dist as ddist, ops, scoring, weightgen as wg  # noqa: E402
from deeplip_amd.synthetic import SyntheticAVSet, synthetic_trials  # noqa: E402
from models.audio_models import tdnn  # noqa: E402
from models.audio_models.loss import AAMSoftmax, LMCL, CrossEntropy  # noqa: E402


class Trainer(object):
    def __init__(self, config="conf/audio_config.yaml", overrides=None):
        with open(os.path.join(ROOT, config)) as f:
            opts = yaml.safe_load(f)
        for k, v in (overrides or {}).items():
            d = opts
            *path, leaf = k.split(".")
            for p in path:
                d = d[p]
            d[leaf] = v
        self.train_opts, self.model_opts = opts["train"], opts["model"]
        self.data_opts, self.test_opts = opts["data"], opts["test"]
        if not torch.cuda.is_available():
            raise RuntimeError("train_audio.py needs a ROCm GPU: the deeplip_amd engine has no CPU path")
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        torch.cuda.set_device(self.device)
        self.rank, self.world = ddist.init_from_env(self.device)
        arch = self.model_opts["arch"]
        if arch in ("tdnn", "etdnn"):
            self.model = tdnn.SpeakerEmbNet(self.model_opts)
        elif arch == "resnet":                                               # train_audio.py:64-66 (`import models.resnet`)
            import models.resnet as resnet
            self.model = resnet.SpeakerEmbNet(self.model_opts)
        else:
            raise NotImplementedError("Other models are not implemented!")   # train_audio.py:67-68
        d = self.data_opts
        # the resnet takes [B,1,F,T] with F = the feature dimension of the data section (train_audio.py:183-184)
        F_ = self.model_opts[arch]["input_dim"] if arch != "resnet" else int(d.get("feat_dim", 40))
        self.trainset = SyntheticAVSet(d["n_spk"], d["utt_per_spk"], 0, 1, F_, d["audio_frames"], key="atrain")
        self.voxtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], 0, 1, F_, d["audio_frames"], key="atest")
        sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in self.model.state_dict().items()}, prefix="audio.")
        self.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        self.model.eval().to(self.device)
        E = self.model_opts[arch]["embedding_dim"]
        if self.train_opts["loss"] == "LMCL":
            self.init_margin, self.end_margin = self.train_opts["margin"]
            self.criterion = LMCL(E, d["n_spk"], self.train_opts["scale"], self.init_margin).to(self.device)
        elif self.train_opts["loss"] == "AAMSoftmax":                       # a stub upstream (loss.py:62-67); ArcFace here
            self.init_margin, self.end_margin = self.train_opts["margin"]
            self.criterion = AAMSoftmax(E, d["n_spk"], self.train_opts["scale"], self.init_margin).to(self.device)
        else:
            self.criterion = CrossEntropy(E, d["n_spk"]).to(self.device)
        o = self.train_opts["sgd"]
        self.freeze_encoder = bool(self.train_opts.get("freeze_encoder", False))
        groups = [{"params": self.criterion.parameters()}] if self.freeze_encoder else \
                 [{"params": self.model.parameters()}, {"params": self.criterion.parameters()}]   # train_audio.py:112
        self.optim = torch.optim.SGD(groups, o["init_lr"], momentum=o["momentum"], weight_decay=o["weight_decay"])
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optim, milestones=self.train_opts["lr_decay_step"], gamma=0.1)
        self.epoch, self.current_epoch = self.train_opts["epoch"], 0
        self.log_time = time.asctime(time.localtime(time.time())).replace(" ", "_")[4:]
        if ddist.active():          # one directory name for the job (rank 0's clock), identical replicas
            name = [self.log_time]
            torch.distributed.broadcast_object_list(name, 0)
            self.log_time = name[0]
        ddist.broadcast_params(list(self.model.parameters()) + list(self.criterion.parameters()))
        self.buckets = None         # built on the first training step (gradient views need the final placement)

    def _adjust_margin(self):
        if isinstance(self.criterion, (LMCL, AAMSoftmax)):
            self.criterion.margin = self.init_margin if self.current_epoch <= 5 else self.end_margin

    def _train_epoch(self):
        """train_audio.py:167-199 on the synthetic set: full-encoder step (or criterion-only with freeze_encoder)."""
        bs = self.train_opts["bs"]
        rng = np.random.Generator(np.random.PCG64([self.current_epoch, 5, self.rank]))    # every rank its own batches
        tot = n = correct = 0.0
        self.model.train(not self.freeze_encoder)
        if ddist.active() and self.buckets is None:
            params = [p for g in self.optim.param_groups for p in g["params"]]
            self.buckets = ddist.GradBuckets(params)
        t0 = time.perf_counter()
        steps = self.train_opts.get("steps_per_epoch", 2)
        for _ in range(steps):
            idx = rng.integers(0, len(self.trainset), bs)
            x = torch.from_numpy(self.trainset.audio(idx)).to(self.device)
            lab = torch.from_numpy(self.trainset.labels(idx)).to(self.device)
            self.optim.zero_grad(set_to_none=self.buckets is None)
            if self.freeze_encoder:
                with torch.no_grad():
                    emb = self.model(x)
            else:
                emb = self.model(x)                       # output of bn2 / LeakyReLU (train_audio.py:187)
            loss, logits = self.criterion(emb, lab)
            loss.backward()                               # bucket all-reduces start as the gradients land
            if self.buckets is not None:
                self.buckets.finish()
            self.optim.step()
            correct += float((torch.max(logits, dim=1)[1] == lab).sum())
            tot += float(loss.detach()) * len(idx); n += len(idx)
        _lib.check_range(sync=True)                       # f16x3 packing: an overflow in this epoch is an error, not a NaN
        tot, correct, n = ddist.allreduce_metrics([tot, correct, n], self.device)
        self.last_epoch_stats = {"loss": tot / n, "acc": correct / n, "utt_per_s": n / (time.perf_counter() - t0),
                                 "steps": steps, "bs": bs * self.world}
        self.model.eval()
        return tot / n

    def _train(self):
        for epoch in range(self.current_epoch + 1, self.epoch + 1):
            self.current_epoch = epoch
            self._adjust_margin()
            loss = self._train_epoch()
            st = self.last_epoch_stats
            print("Epoch {} loss {:.4f} acc {:.3f} ({:.0f} utt/s, {} steps of {})".format(epoch, loss, st["acc"], st["utt_per_s"],
                                                                                     st["steps"], st["bs"]), flush=True)
            self.lr_scheduler.step()
            self.save()
            if ddist.active():
                torch.distributed.barrier()     # rank 0's checkpoint is on disk before anybody averages / resumes

    def save(self, filename=None):
        path = "exp/{}/{}".format(self.log_time, filename or "net_{}.pth".format(self.current_epoch))
        if self.rank != 0:          # replicas are identical: one writer
            return path
        os.makedirs(os.path.dirname(path), exist_ok=True)
        # keys carry the DataParallel 'module.' prefix like the reference's checkpoints (train_audio.py:262)
        torch.save({"epoch": self.current_epoch, "state_dict": {"module." + k: v for k, v in self.model.state_dict().items()},
                    "criterion": self.criterion.state_dict(), "optimizer": self.optim.state_dict()}, path)
        return path

    def load(self, resume):
        """Resume from one of this trainer's checkpoints or from a reference one (train_audio.py:234-296): there
        ``criterion`` is the pickled criterion MODULE (train_audio.py:264), and ``net_avg.pth`` (written by
        model_average, :229-232) has neither ``criterion`` nor ``epoch``."""
        try:
            ck = torch.load(resume, map_location="cpu", weights_only=True)       # tensors and plain containers only
        except Exception as ex:   # noqa: BLE001 -- torch raises UnpicklingError for anything beyond that
            # a reference checkpoint pickles the criterion MODULE: loading it executes the pickle, i.e. arbitrary code.
            # Allowed only on request (train.allow_pickled_checkpoints: True / DLIP_ALLOW_PICKLED_CHECKPOINTS=1).
            if not (self.train_opts.get("allow_pickled_checkpoints") or os.environ.get("DLIP_ALLOW_PICKLED_CHECKPOINTS") == "1"):
                raise RuntimeError(f"{resume} holds pickled objects (a reference-style checkpoint stores the criterion module); loading it "
                                   "runs code from the file. Set train.allow_pickled_checkpoints: True (or "
                                   "DLIP_ALLOW_PICKLED_CHECKPOINTS=1) if you trust it.") from ex
            ck = torch.load(resume, map_location="cpu", weights_only=False)
        self.model.load_state_dict({k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in ck["state_dict"].items()})
        crit = ck.get("criterion")
        if crit is not None:
            if isinstance(crit, torch.nn.Module):
                crit = crit.state_dict()
            self.criterion.load_state_dict(crit)
        self.current_epoch = int(ck.get("epoch", self.current_epoch))

    def model_average(self, avg_num=4):
        """train_audio.py:216-232: average the state dicts of the last ``avg_num`` epoch checkpoints."""
        paths = ["exp/{}/net_{}.pth".format(self.log_time, e) for e in range(self.current_epoch - avg_num + 1, self.current_epoch + 1)]
        paths = [p for p in paths if os.path.exists(p)]
        avg = None
        for p in paths:
            sd = torch.load(p, map_location="cpu")["state_dict"]
            avg = {k: v.clone().double() for k, v in sd.items()} if avg is None else {k: avg[k] + v.double() for k, v in sd.items()}
        own = self.model.state_dict()
        avg = {k.replace("module.", "", 1): (v / len(paths)).to(own[k.replace("module.", "", 1)].dtype) for k, v in avg.items()}
        self.model.load_state_dict(avg)
        if self.rank == 0 and paths:      # the file the reference's extract_* methods load (train_audio.py:229-232,300,346)
            torch.save({"state_dict": {"module." + k: v for k, v in avg.items()}}, "exp/{}/net_avg.pth".format(self.log_time))
        return len(paths)

    def extract_test_xv(self, batch=64):
        """x-vectors of the test set, L2-normalised (train_audio.py:343-373) -> EmbeddingTable."""
        rows = []
        with torch.no_grad():
            for b0 in range(0, len(self.voxtestset), batch):
                idx = list(range(b0, min(len(self.voxtestset), b0 + batch)))
                xv, _ = self.model.extract_embedding(torch.from_numpy(self.voxtestset.audio(idx)).to(self.device))
                rows.append(ops.l2_normalize(xv))
        self.table = scoring.EmbeddingTable(self.voxtestset.utt_ids, torch.cat(rows))
        _lib.check_range(sync=True)       # a range report of the LAST batch must surface here, not at some later call
        return self.table

    def eer(self):
        y, pairs = synthetic_trials(self.voxtestset, self.data_opts["trials"], self.data_opts["trial_targets"])
        ia, ib = self.table.trial_indices(pairs)
        return scoring.eer_from_scores(y, scoring.cosine_scores(self.table.emb, ia, ib).cpu().numpy())


def _self_launch(gpus, config, overrides, key="train.gpus_id"):
    """Outside a torch.distributed job and asked for N > 1 GPUs (--gpus, or the config's gpus_id list): start the N-rank job
    of this same command and return its exit code (deeplip_amd/launch.py); runs before anything touches the GPU."""
    from deeplip_amd import launch
    if launch.in_job():
        return None
    if gpus is None:
        with open(os.path.join(ROOT, config) if not os.path.isabs(config) else config) as f:
            d = yaml.safe_load(f)
        ids = overrides.get(key)
        if ids is None:
            for part in key.split("."):
                d = d.get(part, {}) if isinstance(d, dict) else {}
            ids = d
        gpus = len(ids) if isinstance(ids, (list, tuple)) else 1
    return launch.maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], gpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="test", choices=["train", "test"])    # reference: hard-coded at :485
    ap.add_argument("--config", default="conf/audio_config.yaml")
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node, one process each (default: len(train.gpus_id), as the reference sizes nn.DataParallel: "
                         "train_audio.py:80-83)")
    a = ap.parse_args()
    rc = _self_launch(a.gpus, a.config, {k: yaml.safe_load(v) for k, v in (kv.split("=", 1) for kv in a.set)})
    if rc is not None:
        sys.exit(rc)
    tr = Trainer(a.config, {k: yaml.safe_load(v) for k, v in (kv.split("=", 1) for kv in a.set)})
    if a.mode == "train":
        tr._train()
        tr.model_average(min(4, tr.epoch))
    tr.extract_test_xv()
    eer, thr = tr.eer()
    if tr.rank == 0:
        print("EER: {:.6f}%".format(eer * 100))
    if ddist.active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
