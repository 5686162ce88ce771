#!/usr/bin/env python3
"""train_audio.py -- x-vector (TDNN / E-TDNN) entry point on the MI355X engine.

Re-creation of the reference's train_audio.py surface that the A+V hot path uses: config schema
(conf/audio_config.yaml: data / model / train / test), ``Trainer`` with the reference's method names --
``__call__`` (:473-483), ``_train`` / ``_train_epoch``, ``_adjust_margin`` (:141-145), ``model_average`` (checkpoint averaging,
:216-232), ``extract_train_xv`` (:234-258), ``save`` / ``load`` / ``load_finetune`` (:260-296), ``train_plda`` (:298-341),
``extract_test_xv`` (x-vector extraction + F.normalize, :343-373), ``extract_test_xv_lomgrid`` / ``_grid`` (:375-437) -- each
extraction leaving the reference's ``exp/<run>/<set>/<utt>.npy`` store, and ``__main__`` scoring it through
``models.audio_models.utils.eer*(log_time)`` as the reference's does (:485-543); modes ``train``, ``test``, ``av_test``, ``av_fusion``.  ``train`` is the reference's loop (train_audio.py:167-199): ``model.train()``, forward
through every TDNN layer with batch-statistics BatchNorm, LMCL / CrossEntropy, ``loss.backward()`` through the
whole encoder (conv dgrad / wgrad, BN and pooling backward as dlip_* launches -- deeplip_amd/autograd.py),
SGD over model + criterion parameters, MultiStepLR, margin schedule, per-epoch checkpoints and checkpoint
averaging.  ``train.freeze_encoder: true`` keeps the older criterion-only step on frozen x-vectors.

Round 6: a batch is cut to ONE crop length drawn from a short ladder over ``train.crop_frames`` (the collate's random crop,
models/audio_models/datasets.py:112-115) and the optimisation step is recorded once per crop length (and margin) and replayed
(``train.graph_step``, ``--eager-step``); ragged test lists go through ONE extractor per trainer; ``--arith`` / ``model.arith`` pick the
arithmetic (deeplip_amd/arith.py: auto = f16x3 with an in-process f32 re-run -- and calibration -- of what leaves its range).

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

from deeplip_amd import _lib, arith, dist as ddist, ops, scoring, weightgen as wg  # noqa: E402
from deeplip_amd.synthetic import SyntheticAVSet, synthetic_trials  # noqa: E402
from models.audio_models import tdnn  # noqa: E402
from models.audio_models.loss import AAMSoftmax, LMCL, CrossEntropy  # noqa: E402


class Trainer(object):
    def __init__(self, config="conf/audio_config.yaml", overrides=None, arith_mode=None):
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
        # the arithmetic of the engine (--arith > $DLIP_ARITH > model.arith > auto), before the first weight pack
        self.arith = arith.configure(arith_mode, self.model_opts.get("arith"))
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
        # test lists hold utterances of differing duration, as the reference's do (data.test_ragged; train_audio.py:343-373 feeds them
        # one at a time at their own lengths); the resnet's padded convolutions have no ragged batch
        rag = dict(ragged=bool(d.get("test_ragged", False)) and arch != "resnet", audio_range=tuple(d.get("test_audio_frames", (137, 412))))
        self.voxtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], 0, 1, F_, d["audio_frames"], key="atest", **rag)
        # the reference's three A+V evaluation lists (train_audio.py:119-139: lomgriddevloader / lomgridtestloader / gridtestloader)
        self.lomgriddevset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], 0, 1, F_, d["audio_frames"], key="alomdev", **rag)
        self.lomgridtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], 0, 1, F_, d["audio_frames"], key="alomgrid", **rag)
        self.gridtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], 0, 1, F_, d["audio_frames"], key="agrid", **rag)
        self.resume = self.train_opts.get("resume", "exp/none/net_avg.pth")
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
        # train.graph_step (default on): the optimisation step is recorded once per crop length and replayed as one HIP graph
        # (deeplip_amd/train_plan.py); a recorded step reads its learning rate from a device tensor, which MultiStepLR updates in
        # place, and takes the fused SGD kernel.  train.graph_step: false keeps the loop of eager launches.
        self.graph_step = bool(self.train_opts.get("graph_step", True)) and os.environ.get("DLIP_GRAPH_STEP", "1") != "0"
        self.optim = self._make_sgd(groups, o)
        self._steps = None          # ShapeKeyedSteps, built with the first recorded step
        self._extractor = None      # one RaggedExtractor for every list and epoch of this trainer (its recorded plans are kept)
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optim, milestones=self.train_opts["lr_decay_step"], gamma=0.1)
        self.epoch, self.current_epoch = self.train_opts["epoch"], 0
        self.log_time = time.asctime(time.localtime(time.time())).replace(" ", "_")[4:]
        if ddist.active():          # one directory name for the job (rank 0's clock), identical replicas
            name = [self.log_time]
            torch.distributed.broadcast_object_list(name, 0)
            self.log_time = name[0]
        ddist.broadcast_params(list(self.model.parameters()) + list(self.criterion.parameters()))
        self.buckets = None         # built on the first training step (gradient views need the final placement)

    def _make_sgd(self, groups, o):
        if self.graph_step:
            return torch.optim.SGD(groups, lr=torch.tensor(float(o["init_lr"]), device=self.device), momentum=o["momentum"],
                                   weight_decay=o["weight_decay"], fused=True)
        return torch.optim.SGD(groups, o["init_lr"], momentum=o["momentum"], weight_decay=o["weight_decay"])

    def crop_ladder(self):
        """The crop lengths of the training batches.  The reference's collate cuts every batch to ONE random length of 200 .. 400
        frames (models/audio_models/datasets.py:112-115); here that length is drawn from the short geometric ladder of
        deeplip_amd.ragged.rung_tops over ``train.crop_frames`` (neighbouring lengths within ``train.crop_ladder_waste`` = 10 % of
        each other, 8 rungs for 200 .. 400): one recorded step per rung.  No ``crop_frames``: every batch at ``data.audio_frames``."""
        from deeplip_amd.ragged import rung_tops
        cf = self.train_opts.get("crop_frames")
        Ta = int(self.data_opts["audio_frames"])
        if not cf:
            return [Ta]
        lo, hi = int(cf[0]), min(int(cf[1]), Ta)
        lo = min(lo, hi)
        need = (self.model.frames_consumed() + 2) if hasattr(self.model, "frames_consumed") else 1
        if lo < need:
            raise ValueError(f"train.crop_frames: the encoder needs utterances of >= {need} frames")
        return rung_tops(lo, hi, float(self.train_opts.get("crop_ladder_waste", 0.10)), 4 if hi % 4 == 0 else 1)

    def _one_step(self, x, lab):
        """One optimisation step, launches only (what a recorded step replays): train_audio.py:185-200."""
        if self.buckets is None:
            self.optim.zero_grad(set_to_none=True)
        else:
            self.buckets.zero()
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
        return loss, logits

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
        steps = self.train_opts.get("steps_per_epoch", 2)
        ladder = self.crop_ladder()
        Ta = int(self.data_opts["audio_frames"])
        recorded = self.graph_step and (self.buckets is None or os.environ.get("DLIP_GRAPH_WITH_BUCKETS", "1") != "0")
        if recorded and self._steps is None:
            from deeplip_amd.train_plan import ShapeKeyedSteps, grad_witness, step_state
            mods = [self.criterion] if self.freeze_encoder else [self.model, self.criterion]
            # (with GradBuckets the branches of a step stay on one stream: see train_video.py)
            self._steps = ShapeKeyedSteps(self._one_step, eager_steps=1, device=self.device, branch_streams=self.buckets is None,
                                          state=step_state(mods, [self.optim], self.buckets), witness=grad_witness(mods, self.buckets))
        acc = torch.zeros(3, dtype=torch.float64, device=self.device)          # loss * n, correct, n: read once per epoch
        # ONE copy stream for the trainer's lifetime: torch's caching allocator keeps a pool per stream, so a stream per epoch left every
        # epoch's batch buffers reserved for ever (tools/probes/leak_check_eager.py: +22 MiB reserved per 10-step epoch at B = 64, the allocated
        # bytes flat)
        copy_stream = self.__dict__.get("_copy_stream")
        if copy_stream is None:
            copy_stream = self._copy_stream = torch.cuda.Stream(device=self.device)
        cache = self.__dict__.setdefault("_batch_cache", {})
        n_cache = int(self.train_opts.get("data_cache", 0) or 0)

        def stage(i):
            """Batch i -> the device, copies on their own stream behind the running step.  ``train.data_cache: N``: the synthetic
            source's first N batches are generated once (pinned) and walked cyclically -- a run that measures the trainer, not numpy."""
            k = (self.current_epoch, i) if n_cache <= 0 else i % n_cache
            hb = cache.get(k) if n_cache > 0 else None
            if hb is None:
                idx = rng.integers(0, len(self.trainset), bs)
                T = int(ladder[int(rng.integers(0, len(ladder)))])
                t0_ = int(rng.integers(0, Ta - T + 1))
                xb = np.ascontiguousarray(self.trainset.audio(idx)[:, :, t0_:t0_ + T])       # the batch's one crop (datasets.py:112-115)
                hb = (torch.from_numpy(xb).pin_memory(), torch.from_numpy(self.trainset.labels(idx)).pin_memory())
                if n_cache > 0:
                    cache[k] = hb
            with torch.cuda.stream(copy_stream):
                x, lab = hb[0].to(self.device, non_blocking=True), hb[1].to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return x, lab, ev

        staged = stage(0) if steps > 0 else None
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for i in range(steps):
            x, lab, ev = staged
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            x.record_stream(cur); lab.record_stream(cur)
            if recorded:
                # keyed by the crop length (the shape) AND the margin the criterion bakes into the recorded launches
                loss, logits = self._steps.step(x, lab, key=float(getattr(self.criterion, "margin", 0.0)))
            else:
                loss, logits = self._one_step(x, lab)
            acc[0] += loss.detach().double() * x.shape[0]
            acc[1] += (torch.max(logits.detach(), dim=1)[1] == lab).sum()
            acc[2] += x.shape[0]
            staged = stage(i + 1) if i + 1 < steps else None
        if recorded:
            self._steps.finish()
        tot, correct, n = acc.tolist()
        _lib.check_range(sync=True)                       # f16x3 packing: an overflow in this epoch is an error, not a NaN
        dt = time.perf_counter() - t0
        tot, correct, n = ddist.allreduce_metrics([tot, correct, n], self.device)
        self.last_epoch_stats = {"loss": tot / n, "acc": correct / n, "utt_per_s": n / dt, "steps": steps, "bs": bs * self.world,
                                 "step_mode": (self._steps.mode if recorded else "eager"), "crop_ladder": [int(t) for t in ladder]}
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
        import pickle
        try:
            ck = torch.load(resume, map_location="cpu", weights_only=True)       # tensors and plain containers only
        except pickle.UnpicklingError as ex:   # what torch raises for anything beyond that; a missing / truncated file propagates as itself
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

    def _xvectors(self, dataset, batch=64, normalize=True):
        """The embedding the reference extracts (train_audio.py:362-366): CrossEntropy -> the 1st fc layer's output; LMCL (and
        the ArcFace stand-in) -> the 2nd fc layer's, L2-normalised (``F.normalize``) for the test lists."""
        rows = []
        ce = self.train_opts["loss"] == "CrossEntropy"
        self.model.eval()
        with torch.no_grad():
            if dataset.ragged:
                # utterances at their own lengths (train_audio.py:343-373 feeds them one by one): length-bucketed batches,
                # every row equal to the one-at-a-time result (deeplip_amd/extract.py).  ONE extractor per trainer: the plans it
                # records (a dozen padded shapes x two input sets) serve every list and every epoch's evaluation.
                D = self.model.embedding_dim
                ex = self._ragged_extractor(batch)
                both, _ = ex.run(dataset, 0, len(dataset), 2 * D)
                self.extract_stats = dict(ex.stats)
                xv, x_a = both[:, :D].contiguous(), both[:, D:].contiguous()
                rows.append(x_a if ce else ops.l2_normalize(xv) if normalize else xv)
            else:
                for b0 in range(0, len(dataset), batch):
                    idx = list(range(b0, min(len(dataset), b0 + batch)))
                    xv, x_a = self.model.extract_embedding(torch.from_numpy(dataset.audio(idx)).to(self.device))
                    rows.append(x_a if ce else ops.l2_normalize(xv) if normalize else xv)
        table = scoring.EmbeddingTable(dataset.utt_ids, torch.cat(rows))
        _lib.check_range(sync=True)       # a range report of the LAST batch must surface here, not at some later call
        return table

    def _ragged_extractor(self, batch):
        """The trainer's RaggedExtractor (built on first use, per batch size).  Its recorded plans address the model's packed
        weights: a plan whose weights changed since (training, a checkpoint load) is stale -- the extractor is rebuilt then."""
        from deeplip_amd import holders
        from deeplip_amd.extract import RaggedExtractor
        from deeplip_amd import packing
        key = (int(batch), holders.PACK_GEN[0], packing.state_version(self.model, self.device))
        if self._extractor is not None and self._extractor[0] != key:
            self._extractor[1].close()
            self._extractor = None
        if self._extractor is None:
            ex = RaggedExtractor(lambda a, l: torch.cat(self.model.extract_embedding(a, lengths=l), dim=1), None, self.device, batch=batch,
                                 audio_min_frames=self.model.frames_consumed() + 2)
            self._extractor = (key, ex)
        return self._extractor[1]

    def close(self):
        """Release the recorded extraction plans (their arenas) and the recorded training steps."""
        if self._extractor is not None:
            self._extractor[1].close()
            self._extractor = None
        self._steps = None

    def _load_for_extract(self, avg_only=False):
        """train_audio.py:235-236,300-301 (net_avg.pth if the run has one) / :345-347,377-379 (``resume`` unless fine-tuning)."""
        avg = "exp/{}/net_avg.pth".format(self.log_time)
        if avg_only:
            if os.path.exists(avg):
                self.load(avg)
        elif os.path.exists(self.resume) and self.train_opts.get("train_type") != "finetune":
            self.load(self.resume)

    def _store(self, sub, table, flat=False):
        """One ``[1, D]`` .npy per utterance under exp/<run>/<sub>/ (train_audio.py:367-369); rank 0 writes."""
        if self.rank == 0 and self.test_opts.get("write_store", True):
            root = "exp/{}/{}".format(self.log_time, sub)
            if flat:                      # train_plda stores by basename (train_audio.py:320-322)
                scoring.EmbeddingTable([os.path.basename(u) for u in table.utt_ids], table.emb).save_npy_tree(root)
            else:
                table.save_npy_tree(root)
        if ddist.active():
            torch.distributed.barrier()

    def extract_test_xv(self, batch=64):
        """x-vectors of the test set, L2-normalised (train_audio.py:343-373) -> EmbeddingTable (+ exp/<run>/test_xv/)."""
        self._load_for_extract()
        self.table = self._xvectors(self.voxtestset, batch)
        self._store("test_xv", self.table)
        self._point_scoring("eer", self.voxtestset, "task")
        return self.table

    def extract_train_xv(self, batch=64):
        """train_audio.py:234-258: embeddings of the training list, NOT normalised, under exp/<run>/train_xv/."""
        self._load_for_extract(avg_only=True)
        self.train_table = self._xvectors(self.trainset, batch, normalize=False)
        self._store("train_xv", self.train_table)
        return self.train_table

    def extract_test_xv_lomgrid(self, batch=64):
        """train_audio.py:375-405 -> exp/<run>/test_xv_lomgrid/."""
        self._load_for_extract()
        self.lomgrid_table = self._xvectors(self.lomgridtestset, batch)
        self._store("test_xv_lomgrid", self.lomgrid_table)
        self._point_scoring("lomgrid", self.lomgridtestset, "trial_lomgrid")
        return self.lomgrid_table

    def extract_test_xv_grid(self, batch=64):
        """train_audio.py:407-437 -> exp/<run>/test_xv_grid/."""
        self._load_for_extract()
        self.grid_table = self._xvectors(self.gridtestset, batch)
        self._store("test_xv_grid", self.grid_table)
        self._point_scoring("grid", self.gridtestset, "trial_grid")
        return self.grid_table

    def _point_scoring(self, name, dataset, fname):
        """The synthetic trial list of a test set, written in the reference's format, and this run's scoring calls pointed at
        it (the reference hard-codes ``task.txt`` / ``data/trial/A_*_trial_2w``: models/audio_models/utils.py:237,254,271)."""
        from deeplip_amd import scoring_entry as se
        if not self.test_opts.get("write_store", True):
            return
        path = "exp/{}/{}.txt".format(self.log_time, fname)
        if self.rank == 0:
            y, pairs = synthetic_trials(dataset, self.data_opts["trials"], self.data_opts["trial_targets"])
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                fh.writelines("{} {} {}\n".format(int(l), a, b) for l, (a, b) in zip(y, pairs))
        if ddist.active():
            torch.distributed.barrier()
        for fn in (("eer",) if name == "eer" else ("eer_cos_" + name, "eer_plda_" + name)):
            se.set_paths(fn, trial=path)

    def train_plda(self, n_principal_components=20):
        """train_audio.py:298-341: x-vectors of the LombardGRID development list -> exp/<run>/dev_xv_lomgrid/<basename>.npy, labels
        from the speaker id in the file name (``s<k>_...``), a PLDA model with ``n_principal_components=20`` fitted on them
        (host: the `plda` package's algorithm, deeplip_amd/plda.py) -> ``exp/plda.pkl``."""
        from deeplip_amd.plda import PLDA
        self._load_for_extract(avg_only=True)
        dev = self._xvectors(self.lomgriddevset)
        self._store("dev_xv_lomgrid", dev, flat=True)
        labels = [int(os.path.basename(u).split("_")[0].replace("s", "")) for u in dev.utt_ids]      # :334
        X = dev.emb.cpu().numpy()
        self.plda = PLDA.fit(X, labels, n_principal_components=min(n_principal_components, X.shape[1], max(2, len(set(labels)) - 1)))
        if self.rank == 0:
            self.plda.save("exp/plda.pkl")
        if ddist.active():
            torch.distributed.barrier()
        return self.plda

    def load_finetune(self, resume, param_groups=None):
        """train_audio.py:276-296: load an encoder checkpoint, freeze the encoder and rebuild optimizer + schedule over the
        criterion's parameters alone (``param_groups`` is overwritten there too)."""
        print("loading model from {}".format(resume))
        self.log_time = resume.split("/")[1]
        self.load(resume)
        for p in self.model.parameters():
            p.requires_grad = False
        self.freeze_encoder = True
        param_groups = [{"params": self.criterion.parameters()}]
        kind = self.train_opts.get("type", "sgd")
        if kind == "sgd":
            self.optim = self._make_sgd(param_groups, self.train_opts["sgd"])
        elif kind == "adam":
            o = self.train_opts["adam"]
            if self.graph_step:
                self.optim = torch.optim.Adam(param_groups, lr=torch.tensor(float(o["init_lr"]), device=self.device),
                                              weight_decay=o["weight_decay"], capturable=True, fused=True)
            else:
                self.optim = torch.optim.Adam(param_groups, o["init_lr"], weight_decay=o["weight_decay"])
        else:
            raise NotImplementedError(kind)
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optim, milestones=self.train_opts["lr_decay_step"], gamma=0.1)
        self.buckets = None
        self._steps = None              # the recorded steps addressed the old optimizer's state

    def __call__(self):
        """train_audio.py:473-483."""
        if self.rank == 0:
            print("[LOG Time: {}]".format(self.log_time))
            os.makedirs("exp/{}".format(self.log_time), exist_ok=True)
        self._train()

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
    ap.add_argument("--mode", default="test", choices=["train", "test", "av_test", "av_fusion"])    # reference: hard-coded at :486
    ap.add_argument("--config", default="conf/audio_config.yaml")
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--run", default=None, help="name of an existing exp/<run>/ directory to score (av_fusion; the reference edits log_time by hand)")
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node, one process each (default: len(train.gpus_id), as the reference sizes nn.DataParallel: "
                         "train_audio.py:80-83)")
    ap.add_argument("--eager-step", action="store_true",
                    help="training: issue every launch of every step from Python instead of replaying one recorded HIP graph per crop "
                         "length (train.graph_step: false)")
    arith.add_argument(ap)
    a = ap.parse_args()
    rc = _self_launch(a.gpus, a.config, {k: yaml.safe_load(v) for k, v in (kv.split("=", 1) for kv in a.set)})
    if rc is not None:
        sys.exit(rc)
    ov = {k: yaml.safe_load(v) for k, v in (kv.split("=", 1) for kv in a.set)}
    if a.eager_step:
        ov["train.graph_step"] = False
    tr = Trainer(a.config, ov, arith_mode=a.arith)
    from models.audio_models import utils           # the scoring entry points, called as train_audio.py:499-543 calls them

    def report(fn):
        if tr.rank == 0:            # the store is on disk and complete (barrier in _store): one rank scores and prints
            eer, threshold = fn(tr.log_time)
            print("EER: {:.6f}%".format(eer * 100))

    if a.mode == "train":
        tr()
        tr.model_average(min(4, tr.epoch))
        tr.extract_test_xv()
        report(utils.eer)
    elif a.mode == "test":
        tr.extract_test_xv()
        report(utils.eer)
    else:                           # av_test: cosine / PLDA on the x-vectors (train_audio.py:504-520)
        if a.run:
            tr.log_time = a.run     # av_fusion scores the stores an earlier run left under exp/<run>/ (:521-543: extraction commented out)
        if a.mode == "av_test" and tr.test_opts.get("train_plda"):
            tr.train_plda()
        for name in ("lomgrid", "grid"):
            if not tr.test_opts.get("eval_" + name, True):
                continue
            if a.mode == "av_test":
                getattr(tr, "extract_test_xv_" + name)()
            if tr.test_opts.get("use_cos", True):
                report(getattr(utils, "eer_cos_" + name + ("_featurefusion" if a.mode == "av_fusion" else "")))
            if tr.test_opts.get("use_plda"):
                report(getattr(utils, "eer_plda_" + name))
    if tr.rank == 0 and arith.STATS["f32_reruns"]:
        print("arith auto: {} batch(es) computed again in exact f32".format(arith.STATS["f32_reruns"]))
    tr.close()
    if ddist.active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
