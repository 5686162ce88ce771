#!/usr/bin/env python3
"""train_fusion.py -- audio-visual fusion trainer / tester on the MI355X engine.

Entry point of the reference (train_fusion.py) re-created runnable: same config schema
(conf/fusion_config.yaml: data / model / train / test), same ``Trainer`` method names
(``__call__/_train/_train_epoch/model_average/feature_normalize/extract_test_xv_lomgrid/extract_test_xv_grid/
save/load/load_finetune``) and the same flow -- frozen audio + video encoders, trainable fusion
head + criterion (train_fusion.py:120,198-201), test-time fusion = z-norm + concat
(:353-358), cosine EER over a trial list -- with the upstream breakages fixed and
three structural changes:
  * batched: the reference's per-utterance / per-clip Python loops (:267-281,:346-349) become one
    encoder launch set per batch + a segmented clip-group mean on the device;
  * embeddings stay in HBM (``EmbeddingTable``) through extraction; rank 0 then writes the reference's on-disk store
    (one .npy per utterance, :361-364) once, and ``__main__`` scores it through ``models.fusion_models.utils.eer_*(log_time)``
    exactly as the reference's does (:430-469);
  * data parallel = one process per GPU over RCCL (gradient all-reduce of the trainable tail, one
    all-gather of test embeddings), replacing nn.DataParallel (:91-93).

Round 6: ``--mode train`` runs the frozen encoders as a two-deep recorded pipeline (copies of batch i + 2 behind the encoders' work on
batch i + 1 behind the head's step on batch i) and the head's forward + backward + all-reduce + SGD as ONE recorded step
(``train.graph_step``, ``--eager-step``); extraction keeps one RaggedExtractor per trainer (``test.batch``, ``test.frames: u8|f32``);
``--arith`` / ``model.arith`` pick the arithmetic (auto = the benchmarked f16x3, out-of-range batches computed again in f32).

    python train_fusion.py --mode train            # single GPU
    python train_fusion.py --mode train --gpus 8   # starts its own 8-rank job (deeplip_amd/launch.py); so does a config
                                                   # whose train.gpus_id lists 8 devices, as in the reference
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_fusion.py --mode train
    python train_fusion.py --mode av_test | av_fusion
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml
from torch import optim
from torch.optim import lr_scheduler

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deeplip_amd import arith, dist as ddist, fusion, ops, scoring  # noqa: E402
from deeplip_amd.synthetic import SyntheticAVSet, synthetic_trials  # noqa: E402
from models.audio_models import tdnn  # noqa: E402
from models.audio_models.loss import LMCL, CrossEntropy  # noqa: E402
from models.fusion_models import LBP, model_fusion  # noqa: E402
from models.video_models.model import Lipreading  # noqa: E402


class Trainer(object):
    def __init__(self, mode, config="conf/fusion_config.yaml", overrides=None, dry=False, arith_mode=None):
        with open(os.path.join(ROOT, config) if not os.path.isabs(config) else config) as f:
            opts = yaml.safe_load(f)
        for k, v in (overrides or {}).items():      # e.g. {"train.bs": 8}
            d = opts
            *path, leaf = k.split(".")
            for p in path:
                d = d[p]
            d[leaf] = v
        self.train_opts, self.model_opts = opts["train"], opts["model"]
        self.data_opts, self.test_opts = opts["data"], opts["test"]
        self.mode = mode
        self._extractor = None      # one RaggedExtractor for every list / epoch of this trainer (keeps its recorded plans)
        self._host_cache = {} if self.test_opts.get("cache_host_batches", False) else None
        self._steps = self._enc_pipe = self.buckets = None
        self.graph_step = bool(self.train_opts.get("graph_step", True)) and os.environ.get("DLIP_GRAPH_STEP", "1") != "0" and not dry
        if not dry:                 # the arithmetic of the engine (--arith > $DLIP_ARITH > model.arith > auto), before the first weight pack
            self.arith = arith.configure(arith_mode, self.model_opts.get("arith"))

        self.rank, self.world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
        local = int(os.environ.get("LOCAL_RANK", 0))
        self.dry = bool(dry)
        if self.dry:
            # --dry: a REHEARSAL of the data-parallel protocol on CPU ranks over gloo (tests/test_launch_cpu.py), like bench.py's
            # --dry-launch.  Nothing of the engine runs and nothing is measured: the frozen encoders are replaced by seeded rows,
            # the head and criterion by stand-ins (_DryHead); what is real is everything AROUND the arithmetic -- the launcher,
            # the job name broadcast, per-rank sampling, the flat gradient all-reduce, the optimizer / scheduler, metric
            # reduction, rank 0's checkpoints.
            self.device = torch.device("cpu")
            ddist.init_from_env(None)
        else:
            if not torch.cuda.is_available():
                raise RuntimeError("train_fusion.py needs a ROCm GPU: the deeplip_amd engine has no CPU path")
            torch.cuda.set_device(local)
            self.device = torch.device("cuda", local)
            if "RANK" in os.environ and not dist.is_initialized():     # any torch.distributed.run job, a one-rank one included
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group("nccl", device_id=self.device)

        d = self.data_opts
        acfg = self.model_opts["audio_config"]
        feat_dim = acfg[acfg["arch"]]["input_dim"]
        self.trainset = SyntheticAVSet(d["n_spk"], d["utt_per_spk"], d["clips_per_utt"], d["video_frames"], feat_dim,
                                       d["audio_frames"], key="train")
        # test lists: utterances and clips of differing duration, as the reference's are (data.test_ragged; its loop takes each
        # at its own length, train_fusion.py:334-349) -- 1 .. test_clips_per_utt clip files per utterance
        rag = dict(ragged=bool(d.get("test_ragged", False)), audio_range=tuple(d.get("test_audio_frames", (137, 412))),
                   video_range=tuple(d.get("test_video_frames", (11, 75))))
        tclips = d.get("test_clips_per_utt", d["clips_per_utt"]) if rag["ragged"] else d["clips_per_utt"]
        self.lomgridtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], tclips,
                                             d["video_frames"], feat_dim, d["audio_frames"], key="lomgrid", **rag)
        self.gridtestset = SyntheticAVSet(d["test_speakers"], d["test_utt_per_spk"], tclips,
                                          d["video_frames"], feat_dim, d["audio_frames"], key="grid", **rag)
        n_spk = self.trainset.n_spk
        if self.dry:
            return self._init_dry(acfg, n_spk)

        if acfg["arch"] in ("tdnn", "etdnn"):
            self.model_audio = tdnn.SpeakerEmbNet(acfg)
        else:
            raise NotImplementedError("Other models are not implemented!")
        vcfg = self.model_opts["video_config"]
        if vcfg["arch"] == "tcn":
            t = vcfg["tcn"]
            tcn_options = {"num_layers": t["tcn_num_layers"], "kernel_size": t["tcn_kernel_size"],
                           "dropout": t["tcn_dropout"], "dwpw": t["tcn_dwpw"], "width_mult": t["tcn_width_mult"]}
            self.model_video = Lipreading(num_classes=n_spk, tcn_options=tcn_options, backbone_type=t["backbone_type"],
                                          relu_type=t["relu_type"], width_mult=t["width_mult"],
                                          extract_feats=t["extract_feats"])
        else:
            raise NotImplementedError("Other models are not implemented!")

        self.embedding_dim = acfg[acfg["arch"]]["embedding_dim"]
        kind = self.model_opts.get("fusion", "linear")
        if kind == "linear":      # the commented-out upstream choice (train_fusion.py:82)
            self.model_fusion = model_fusion.model_fusion(self.embedding_dim * 2, 512, n_spk, extract_feats=False)
            fused_dim = 512
        elif kind == "lowfer":    # LBP.LowFER as shipped: cat[e1, sigmoid(e2), sigmoid(e2)*e1]
            self.model_fusion = LBP.LowFER(self.embedding_dim, self.embedding_dim, 512)
            fused_dim = 3 * self.embedding_dim
        elif kind == "concat":
            self.model_fusion = torch.nn.Identity()
            fused_dim = 2 * self.embedding_dim
        else:
            raise NotImplementedError(kind)
        self.fusion_kind = kind
        for m in (self.model_audio, self.model_video, self.model_fusion):
            m.to(self.device)

        if self.train_opts["loss"] == "CrossEntropy":
            self.criterion = CrossEntropy(fused_dim, n_spk).to(self.device)
        elif self.train_opts["loss"] == "LMCL":
            self.init_margin, self.end_margin = self.train_opts["audio_config"]["margin"]
            self.criterion = LMCL(fused_dim, n_spk, self.train_opts["audio_config"]["scale"], self.init_margin).to(self.device)
        else:
            raise NotImplementedError("Other loss function has not been implemented yet!")

        self._init_optim()

    def _init_dry(self, acfg, n_spk):
        self.embedding_dim = acfg[acfg["arch"]]["embedding_dim"]
        self.model_audio = self.model_video = None
        self.fusion_kind = "dry"
        torch.manual_seed(1234 + self.rank)              # ranks start DIFFERENT: the broadcast below has something to do
        self.model_fusion = _DryHead(2 * self.embedding_dim, n_spk)
        self.criterion = _DryCriterion()
        self._init_optim()

    def _init_optim(self):
        param_groups = [{"params": list(self.model_fusion.parameters())}, {"params": self.criterion.parameters()}]
        if self.train_opts["optimizer"] == "sgd":
            o = self.train_opts["sgd"]
            if self.graph_step:     # a recorded step reads its learning rate from a device tensor (MultiStepLR updates it in place)
                self.optim = optim.SGD(param_groups, lr=torch.tensor(float(o["init_lr"]), device=self.device), momentum=o["momentum"],
                                       weight_decay=o["weight_decay"], fused=True)
            else:
                self.optim = optim.SGD(param_groups, o["init_lr"], momentum=o["momentum"], weight_decay=o["weight_decay"])
        else:
            raise NotImplementedError(self.train_opts["optimizer"])
        self.epoch = self.train_opts["epoch"]
        self.resume_audio = self.train_opts["audio_config"]["resume"]
        self.resume_video = self.train_opts["video_config"]["resume"]
        self.resume_fusion = self.train_opts["resume"]
        self.log_time = time.asctime(time.localtime(time.time())).replace(" ", "_")[4:]
        if ddist.active():          # one run directory for the job: rank 0's clock
            name = [self.log_time]
            dist.broadcast_object_list(name, 0)
            self.log_time = name[0]
        self.lr_scheduler = lr_scheduler.MultiStepLR(self.optim, milestones=self.train_opts["lr_decay_step"], gamma=0.1)
        self.current_epoch = 0
        if not self.dry:
            self.load_finetune()
        # replicas start identical (DataParallel broadcast equivalent)
        if ddist.active():
            for p in list(self.model_fusion.parameters()) + list(self.criterion.parameters()):
                dist.broadcast(p.data, 0)

    # ------------------------------------------------------------------ training
    def _adjust_margin(self):
        if isinstance(self.criterion, LMCL):
            self.criterion.margin = self.init_margin if self.current_epoch <= 5 else self.end_margin

    def _train(self):
        for epoch in range(self.current_epoch + 1, self.epoch + 1):
            self.current_epoch = epoch
            self._adjust_margin()
            self._train_epoch()
            self.lr_scheduler.step()
            if self.rank == 0:
                self.save()

    def __call__(self):
        if self.rank == 0:
            os.makedirs("exp/{}".format(self.log_time), exist_ok=True)
        self._train()

    def feature_normalize(self, data):
        """train_fusion.py:233-238 (per-row z-norm, unbiased std) -- one HIP launch."""
        return fusion.feature_normalize(data)

    def _embed_batch(self, dataset, idx):
        """Frozen encoders on a batch of utterances -> (xv_audio [B,512], em_video [B,512])."""
        if self.dry:                                     # seeded rows shaped like the encoders' outputs (speaker centre + noise)
            rows = np.stack([np.random.Generator(np.random.PCG64([int(i), 99])).normal(size=2 * self.embedding_dim) * 0.5
                             + np.random.Generator(np.random.PCG64([int(dataset.utts[i][0]), 7])).normal(size=2 * self.embedding_dim)
                             for i in idx]).astype(np.float32)
            t = torch.from_numpy(rows)
            return t[:, :self.embedding_dim].contiguous(), t[:, self.embedding_dim:].contiguous()
        with torch.no_grad():
            audio = torch.from_numpy(dataset.audio(idx)).to(self.device)
            xv_audio, _ = self.model_audio.extract_embedding(audio)      # train_fusion.py:262,338
            clips, ptr = dataset.video(idx)
            clip_means = self.model_video.embed(torch.from_numpy(clips).to(self.device))   # mean over T (:274,:348)
            em_video = ops.group_mean(clip_means, torch.from_numpy(ptr).to(self.device))   # mean over clip files (:275,:349)
        return xv_audio, em_video

    def _fuse(self, xv_audio, em_video):
        if self.fusion_kind == "lowfer":
            return self.model_fusion(xv_audio, em_video)
        return self.model_fusion(torch.cat([xv_audio, em_video], dim=1).contiguous())

    def _allreduce_grads(self):
        if not ddist.active():
            return
        ddist.allreduce_grads([p for g in self.optim.param_groups for p in g["params"]], self.world)   # one bucket (3.4 MB)

    def _head_step(self, xv_audio, em_video, labels):
        """The trainable part of a step (train_fusion.py:291-299): fusion head + criterion forward, backward, gradient all-reduce of
        the 3.4 MB head, SGD -- launches only, so that it can be recorded once and replayed (deeplip_amd.train_plan)."""
        if self.buckets is None:
            self.optim.zero_grad(set_to_none=True)
        else:
            self.buckets.zero()
        output = self._fuse(xv_audio, em_video)
        loss, logits = self.criterion(output, labels)
        loss.backward()
        if self.buckets is not None:
            self.buckets.finish()
        self.optim.step()
        return loss, logits

    def _train_epoch(self):
        if self.dry or not self.graph_step:
            return self._train_epoch_eager()
        return self._train_epoch_recorded()

    def _train_epoch_recorded(self):
        """_train_epoch (train_fusion.py:241-315) as two replayed pieces per step: the frozen encoders' plan (deeplip_amd.pipeline:
        the batch's host-to-device copies and the encoders run BEHIND the previous step's head; a batch that leaves the f16x3 range is
        computed again in f32 under arith auto) and the head's recorded step (forward + backward + all-reduce + SGD, one HIP graph)."""
        from deeplip_amd.pipeline import ExtractPipeline, pin
        from deeplip_amd.train_plan import ShapeKeyedSteps, grad_witness, step_state
        self.model_fusion.train()
        self.model_audio.eval()
        self.model_video.eval()
        bs = self.train_opts["bs"]
        steps = self.train_opts.get("steps_per_epoch", max(1, len(self.trainset) // (bs * self.world)))
        rng = np.random.Generator(np.random.PCG64([self.current_epoch, 17]))
        D = self.embedding_dim
        heads = [m for m in (self.model_fusion, self.criterion) if isinstance(m, torch.nn.Module)]
        if ddist.active() and self.buckets is None:
            self.buckets = ddist.GradBuckets([p for g in self.optim.param_groups for p in g["params"]])     # one 3.4 MB bucket
        if self._steps is None:
            self._steps = ShapeKeyedSteps(self._head_step, eager_steps=1, device=self.device, branch_streams=False,
                                          state=step_state(heads, [self.optim], self.buckets), witness=grad_witness(heads, self.buckets))
        cache = self.__dict__.setdefault("_batch_cache", {})
        n_cache = int(self.train_opts.get("data_cache", 0) or 0)

        def host_batch(it):
            # speaker-balanced sampling as the reference's sampler (datasets.py:161-164): idx % n_spk
            glob = rng.integers(0, len(self.trainset), bs * self.world)
            k = it % n_cache if n_cache > 0 else None
            if k is not None and k in cache:
                return cache[k]
            idx = glob[self.rank * bs:(self.rank + 1) * bs]
            clips, ptr = self.trainset.video(idx)
            hb = (pin(torch.from_numpy(self.trainset.audio(idx))), pin(torch.from_numpy(clips)), pin(torch.from_numpy(ptr)),
                  pin(torch.from_numpy(self.trainset.labels(idx))))
            if k is not None:
                cache[k] = hb
            return hb

        def encoders(audio, clips, ptr, labels):
            xv_audio, _ = self.model_audio.extract_embedding(audio)                      # train_fusion.py:262
            em_video = ops.group_mean(self.model_video.embed(clips), ptr)                # :267-281: mean over T, then over clip files
            return xv_audio, em_video, labels

        first = host_batch(0)
        key = tuple(tuple(t.shape) for t in first)
        if self._enc_pipe is not None and self._enc_pipe[0] != key:
            self._enc_pipe[1].close()
            self._enc_pipe = None
        if self._enc_pipe is None:
            with torch.no_grad():
                self._enc_pipe = (key, ExtractPipeline(encoders, *(t.to(self.device) for t in first), device=self.device))
        pipe = self._enc_pipe[1]
        DEPTH = pipe.depth                                   # batches in flight: one in the encoders, one in its copies
        slots = [(torch.empty((bs, D), device=self.device), torch.empty((bs, D), device=self.device),
                  torch.empty((bs,), dtype=torch.int64, device=self.device)) for _ in range(DEPTH)]
        acc = torch.zeros(3, dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream(self.device)
        with torch.no_grad():
            for j in range(min(DEPTH, steps)):
                pipe.submit(first if j == 0 else host_batch(j), slots[j % DEPTH], 0)
        for it in range(steps):
            pipe.wait_next()                                 # batch `it` is through the encoders (and inside the arithmetic's range)
            xv_audio, em_video, labels = slots[it % DEPTH]
            loss, logits = self._steps.step(xv_audio, em_video, labels, key=float(getattr(self.criterion, "margin", 0.0)))
            acc[0] += loss.detach().double() * bs
            acc[1] += (torch.max(logits.detach(), dim=1)[1] == labels).sum()
            acc[2] += bs
            if it + DEPTH < steps:
                # batch it + DEPTH takes the input set and the slot batch `it` just left: its copies start now, behind the encoders'
                # work on batch it + 1; the slot is rewritten only after the head's step above has read it (ordered on `cur`)
                pipe.run_stream.wait_stream(cur)
                with torch.no_grad():
                    pipe.submit(host_batch(it + DEPTH), slots[it % DEPTH], 0)
        pipe.finish()
        self._steps.finish()
        sum_loss, correct, sum_samples = acc.tolist()
        dt = time.perf_counter() - t0
        tot = ddist.allreduce_metrics([sum_loss, correct, sum_samples], self.device)
        self.last_epoch_stats = {"loss": tot[0] / tot[2], "acc": tot[1] / tot[2], "pairs_per_s": tot[2] / dt, "steps": steps,
                                 "step_mode": self._steps.mode, "ms_per_step": 1e3 * dt / max(steps, 1)}
        if self.rank == 0:
            print("Epoch {} loss {:.4f} acc {:.2f}% | {:.1f} A+V pairs/s on {} GPU(s) [{}]".format(
                self.current_epoch, tot[0] / tot[2], 100.0 * tot[1] / tot[2], tot[2] / dt, self.world, self._steps.mode), flush=True)
        return tot[0] / tot[2], tot[1] / tot[2]

    def _train_epoch_eager(self):
        self.model_fusion.train()
        if not self.dry:
            self.model_audio.eval()
            self.model_video.eval()
        bs = self.train_opts["bs"]
        steps = self.train_opts.get("steps_per_epoch", max(1, len(self.trainset) // (bs * self.world)))
        rng = np.random.Generator(np.random.PCG64([self.current_epoch, 17]))
        sum_loss = sum_samples = correct = 0.0
        t0 = time.perf_counter()
        for it in range(steps):
            # speaker-balanced sampling as the reference's sampler (datasets.py:161-164): idx % n_spk
            glob = rng.integers(0, len(self.trainset), bs * self.world)
            idx = glob[self.rank * bs:(self.rank + 1) * bs]
            labels = torch.from_numpy(self.trainset.labels(idx)).to(self.device)
            self.optim.zero_grad()
            xv_audio, em_video = self._embed_batch(self.trainset, idx)
            output = self._fuse(xv_audio, em_video)
            loss, logits = self.criterion(output, labels)
            _, prediction = torch.max(logits, dim=1)
            loss.backward()
            self._allreduce_grads()
            self.optim.step()
            n = float(len(idx))
            sum_loss += float(loss.detach()) * n
            sum_samples += n
            correct += float((prediction == labels).sum())
        tot = ddist.allreduce_metrics([sum_loss, correct, sum_samples], self.device)
        dt = time.perf_counter() - t0
        self.last_epoch_stats = {"loss": tot[0] / tot[2], "acc": tot[1] / tot[2], "pairs_per_s": tot[2] / dt, "steps": steps,
                                 "step_mode": "eager", "ms_per_step": 1e3 * dt / max(steps, 1)}
        if self.rank == 0:
            print("Epoch {} loss {:.4f} acc {:.2f}% | {:.1f} A+V pairs/s on {} GPU(s)".format(
                self.current_epoch, tot[0] / tot[2], 100.0 * tot[1] / tot[2], tot[2] / dt, self.world), flush=True)
        return tot[0] / tot[2], tot[1] / tot[2]

    # ------------------------------------------------------------------ checkpoints
    def save(self, filename=None):
        path = "exp/{}/{}".format(self.log_time, filename or "net_{}.pth".format(self.current_epoch))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"epoch": self.current_epoch, "state_dict": self.model_fusion.state_dict(),
                    "criterion": self.criterion.state_dict(), "optimizer": self.optim.state_dict(), "writer_rank": self.rank}, path)
        return path

    def load(self, resume):
        ckpt = torch.load(resume, map_location="cpu")
        self.model_fusion.load_state_dict(ckpt["state_dict"])
        self.criterion.load_state_dict(ckpt["criterion"])
        self.current_epoch = ckpt["epoch"]

    def model_average(self, avg_num=2):
        """train_fusion.py:158-175: average the fusion head's state dicts of the last ``avg_num`` epoch checkpoints
        (``net_<epoch - i>.pth``), write ``net_avg.pth`` (``epoch 0``, the last checkpoint's optimizer state) and load the
        average into the head.  Sums in fp64, rounded once (the reference adds fp32 tensors in place, which also mutates the
        first checkpoint's tensors); integer buffers (BatchNorm's ``num_batches_tracked``) are averaged with integer
        division as ``v / avg_num`` on a LongTensor did when the reference was written."""
        sums, ckpt = {}, None
        for i in range(avg_num):
            ckpt = torch.load("exp/{}/net_{}.pth".format(self.log_time, self.epoch - i), map_location="cpu")
            for k, v in ckpt["state_dict"].items():
                sums[k] = sums[k] + v.double() if k in sums else v.double().clone()
        own = self.model_fusion.state_dict()
        avg = {k: (torch.div(v, avg_num, rounding_mode="floor") if not own[k].dtype.is_floating_point else v / avg_num).to(own[k].dtype)
               for k, v in sums.items()}
        if self.rank == 0:
            torch.save({"epoch": 0, "state_dict": avg, "optimizer": ckpt["optimizer"]}, "exp/{}/net_avg.pth".format(self.log_time))
        self.model_fusion.load_state_dict(avg)
        return avg

    def load_finetune(self):
        """train_fusion.py:191-215: load pretrained audio (keys carry DataParallel's 'module.' prefix) and
        video checkpoints if they exist, then freeze both encoders.  Without files the encoders keep
        the deterministic synthetic initialisation (there are no checkpoints in this environment)."""
        from deeplip_amd import weightgen as wg
        if os.path.exists(self.resume_audio):
            ck = torch.load(self.resume_audio, map_location="cpu")
            self.model_audio.load_state_dict({k.replace("module.", ""): v for k, v in ck["state_dict"].items()})
        else:
            sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in self.model_audio.state_dict().items()}, prefix="audio.")
            self.model_audio.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        if os.path.exists(self.resume_video):
            self.model_video.load_state_dict(torch.load(self.resume_video, map_location="cpu"))
        else:
            sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in self.model_video.state_dict().items()}, prefix="video.")
            self.model_video.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        for m in (self.model_audio, self.model_video):
            for p in m.parameters():
                p.requires_grad = False
            m.eval()

    # ------------------------------------------------------------------ test-time extraction + scoring
    def _ragged_extractor(self, batch):
        """The trainer's RaggedExtractor, built on first use and kept: the plans it records (a dozen padded shapes x two input sets
        per encoder) serve both test lists and every later evaluation.  Rebuilt when the encoders' weights changed since (its plans
        address their packed weights) or the batch size did."""
        from deeplip_amd import holders, packing
        from deeplip_amd.extract import RaggedExtractor
        key = (int(batch), holders.PACK_GEN[0], packing.state_version(self.model_audio, self.device), packing.state_version(self.model_video, self.device))
        if self._extractor is not None and self._extractor[0] != key:
            self._extractor[1].close()
            self._extractor = None
        if self._extractor is None:
            ex = RaggedExtractor(lambda a, l: self.model_audio.extract_embedding(a, lengths=l)[0],          # train_fusion.py:338
                                 lambda v, l: self.model_video.embed(v, lengths=l),                        # :346-348 (mean over T)
                                 self.device, batch=batch, audio_min_frames=self.model_audio.frames_consumed() + 2,
                                 max_arena_bytes=int(self.test_opts.get("max_arena_gb", 64)) << 30)
            self._extractor = (key, ex)
        return self._extractor[1]

    def close(self):
        """Release the recorded plans (extraction arenas, the training pipeline) this trainer holds."""
        if self._extractor is not None:
            self._extractor[1].close()
            self._extractor = None
        if self._enc_pipe is not None:
            self._enc_pipe[1].close()
            self._enc_pipe = None
        self._steps = None

    def _extract(self, dataset, batch=None):
        """Sharded over ranks; returns (EmbeddingTable of fused [N,1024], audio-only table, video-only table).  The reference
        walks the list one utterance at a time with a `.to(device)` in front of every forward (train_fusion.py:338-358); here
        the rank's shard streams through deeplip_amd.pipeline.ExtractPipeline: batches of `batch` utterances (``test.batch``), the
        host-to-device copies of batch i+1 behind the replay of batch i (two input sets, one recorded plan each), the embeddings
        left in HBM.  ``test.frames: u8``: the lip clips travel as the uint8 RGB frames a loader holds ([B,T,3,88,88], BASELINE.json's
        input shape; a quarter of the bytes) and are normalised by the stem's pre-pass."""
        from deeplip_amd import _lib
        from deeplip_amd.pipeline import ExtractPipeline, pin
        batch = int(batch or self.test_opts.get("batch", 64))
        u8 = str(self.test_opts.get("frames", "f32")).lower() == "u8"
        lo, hi = ddist.shard_range(len(dataset))
        D = self.embedding_dim
        n_loc = hi - lo
        xa = torch.empty((n_loc, D), device=self.device)
        xv = torch.empty((n_loc, D), device=self.device)
        if n_loc and dataset.ragged:
            # utterances / clips of differing length: length-bucketed batches, one recorded plan per padded shape, every row
            # equal to the reference's one-at-a-time result (deeplip_amd/extract.py, tests/test_ragged_gpu.py)
            ex = self._ragged_extractor(batch)
            hc = None
            if self._host_cache is not None:           # test.cache_host_batches: a second pass over a list re-uses its pinned batches
                hc = self._host_cache.setdefault(id(dataset), {})
            xa, xv = ex.run(dataset, lo, hi, D, u8=u8, host_cache=hc)                                      # :349 inside (clip-group mean)
            self.extract_stats = dict(ex.stats)
        elif n_loc:
            cpu = dataset.clips                        # clips per utterance (constant over a synthetic set)

            def host_batch(b0):
                idx = list(range(b0, min(hi, b0 + batch)))
                clips, ptr = dataset.video(idx)
                ptr_full = np.full((batch + 1,), ptr[-1], dtype=np.int32)      # a short batch: empty groups behind its last one
                ptr_full[:len(ptr)] = ptr
                if u8:
                    from deeplip_amd.synthetic import frames_u8_from_clips
                    clips = frames_u8_from_clips(clips, rgb=True)
                return (pin(torch.from_numpy(dataset.audio(idx))), pin(torch.from_numpy(clips)), pin(torch.from_numpy(ptr_full)))

            def step(audio, clips, ptr):
                xv_audio, _ = self.model_audio.extract_embedding(audio)                # train_fusion.py:338
                em_video = ops.group_mean(self.model_video.embed(clips), ptr)          # :346-349: mean over T, then over clip files
                return xv_audio, em_video

            first = host_batch(lo)
            full = tuple(torch.zeros((batch,) + tuple(first[0].shape[1:])) if i == 0 else
                         torch.zeros((batch * cpu,) + tuple(first[1].shape[1:]), dtype=first[1].dtype) if i == 1 else first[2].clone() for i in range(3))
            for dst, src in zip(full[:2], first[:2]):
                dst[:src.shape[0]] = src                                               # the plan is recorded on representative values
            with torch.no_grad():
                pipe = ExtractPipeline(step, *(t.to(self.device) for t in full))
                batches = (first if b0 == lo else host_batch(b0) for b0 in range(lo, hi, batch))
                pipe.run(batches, (xa, xv))
                pipe.finish()                          # synchronises; a range report of the last batch surfaces here
                pipe.close()
        em = fusion.fuse_av(xa, xv) if n_loc else torch.empty((0, 2 * D), device=self.device)  # :353-358
        n = len(dataset)
        _lib.check_range(sync=True)
        em, xa, xv = (ddist.gather_rows(t, n) for t in (em, xa, xv))
        return (scoring.EmbeddingTable(dataset.utt_ids, em), scoring.EmbeddingTable(dataset.utt_ids, xa),
                scoring.EmbeddingTable(dataset.utt_ids, xv))

    def _write_store(self, name, dataset, tables):
        """What the reference leaves on disk for its ``utils.eer_*`` functions, so that they can be called with the run's
        name alone as its ``__main__`` does (train_fusion.py:430-469):
          exp/<run>/test_em/test_em_<name>/<utt>.npy   fused [1,1024] rows (train_fusion.py:362-364)
          exp/<run>/test_xv_<name>/<utt>.npy           speech x-vectors [1,512] (train_audio.py:375-405 writes these)
          exp/<run>/embedding_<name>/<pattern>_c<k>.npz  lip embeddings, ``data`` [1,T,512] per clip file (train_video.py:212);
                                                       T = 1 here: the clip's frame mean, which is all the readers use
          exp/<run>/trial_<name>.txt, video_trial_<name>.txt   the synthetic trial list in both of the reference's formats
        and the default paths of this run's scoring calls pointed at them (the reference hard-codes its site's)."""
        from deeplip_amd import scoring_entry as se
        if not self.test_opts.get("write_store", True):
            return
        root = "exp/{}".format(self.log_time)
        trial, vtrial = os.path.join(root, "trial_{}.txt".format(name)), os.path.join(root, "video_trial_{}.txt".format(name))
        vdir = os.path.join(root, "datasets_{}".format(name)) + "/"
        kind = "spk/utt" if name == "lomgrid" else "utt"
        if self.rank == 0:
            tables[0].save_npy_tree(os.path.join(root, "test_em", "test_em_" + name))
            tables[1].save_npy_tree(os.path.join(root, "test_xv_" + name))
            for u, row in zip(dataset.utt_ids, tables[2].emb.cpu().numpy()):
                f = (vdir + se._pattern(kind, u) + "_c0.npz").replace("datasets", "embedding")
                os.makedirs(os.path.dirname(f), exist_ok=True)
                np.savez_compressed(f, data=row[None, None, :])
            y, pairs = synthetic_trials(dataset, self.data_opts["trials"], self.data_opts["trial_targets"])
            with open(trial, "w") as fh:
                fh.writelines("{} {} {}\n".format(int(l), a, b) for l, (a, b) in zip(y, pairs))
            with open(vtrial, "w") as fh:
                fh.writelines("{}\t{}\n".format(se._pattern(kind, a), se._pattern(kind, b)) for a, b in pairs)
        if ddist.active():
            dist.barrier()
        for fn in ("eer_cos_{}", "eer_plda_{}", "eer_cos_{}_scorefusion", "eer_cos_{}_featurefusion"):
            se.set_paths(fn.format(name), trial=trial, video_dir=vdir, video_trial=vtrial)

    def extract_test_xv_lomgrid(self):
        self.lomgrid_tables = self._extract(self.lomgridtestset)
        self._write_store("lomgrid", self.lomgridtestset, self.lomgrid_tables)
        return self.lomgrid_tables[0]

    def extract_test_xv_grid(self):
        self.grid_tables = self._extract(self.gridtestset)
        self._write_store("grid", self.gridtestset, self.grid_tables)
        return self.grid_tables[0]

    def eer_cos(self, dataset, tables, mode="cos"):
        y, pairs = synthetic_trials(dataset, self.data_opts["trials"], self.data_opts["trial_targets"])
        ia, ib = tables[0].trial_indices(pairs)
        if mode == "cos":              # utils.eer_cos_* (utils.py:251-283)
            s = scoring.cosine_scores(tables[0].emb, ia, ib)
        elif mode == "scorefusion":    # utils.eer_cos_*_scorefusion (:331-381)
            s = scoring.score_fusion(tables[1].emb, tables[2].emb, ia, ib)
        else:                          # utils.eer_cos_*_featurefusion (:433-521)
            s = scoring.feature_fusion_scores(tables[1].emb, tables[2].emb, ia, ib)
        return scoring.eer_from_scores(y, s.cpu().numpy())


class _DryHead(torch.nn.Module):
    """--dry only: a stock-torch stand-in for the fusion head (the engine's head needs the GPU).  Never built otherwise."""

    def __init__(self, d, n):
        super().__init__()
        self.fc = torch.nn.Linear(d, n)

    def forward(self, x):
        return self.fc(x)


class _DryCriterion(torch.nn.Module):
    def forward(self, output, labels):
        return torch.nn.functional.cross_entropy(output, labels), output


def _self_launch(gpus, config, overrides, key):
    """One command starts every GPU (the reference: gpus_id -> nn.DataParallel).  Outside a torch.distributed job and asked
    for N > 1 GPUs -- by --gpus or by the length of the config's gpus_id list -- this process becomes the launcher of an
    N-rank job of the same command (deeplip_amd/launch.py) and returns its exit code; nothing has touched the GPU yet."""
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
    ap.add_argument("--mode", default="av_test", choices=["train", "av_test", "av_fusion"])   # reference: hard-coded at :424
    ap.add_argument("--config", default="conf/fusion_config.yaml")
    ap.add_argument("--set", nargs="*", default=[], help="overrides, e.g. train.bs=8 data.trials=2000")
    ap.add_argument("--dry", action="store_true",
                    help="rehearse the data-parallel protocol of --mode train on CPU ranks over gloo: stand-in arithmetic, nothing of the "
                         "engine runs, nothing is measured (tests)")
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node to use, one process each (default: len(train.gpus_id) of the config, as the "
                         "reference sizes nn.DataParallel: train_fusion.py:88-93)")
    ap.add_argument("--eager-step", action="store_true",
                    help="--mode train: issue every launch of every step from Python instead of replaying the recorded encoder plan + "
                         "the recorded head step (train.graph_step: false)")
    arith.add_argument(ap)
    args = ap.parse_args()
    ov = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        ov[k] = yaml.safe_load(v)
    if args.eager_step:
        ov["train.graph_step"] = False
    rc = _self_launch(args.gpus, args.config, ov, "train.gpus_id")
    if rc is not None:
        sys.exit(rc)
    if args.dry:
        if args.mode != "train":
            ap.error("--dry rehearses --mode train only")
        trainer = Trainer("train", args.config, ov, dry=True)
        trainer()
        # every rank leaves what it holds after the last step: replicas must be bit-identical (the test compares the files)
        os.makedirs("exp/{}".format(trainer.log_time), exist_ok=True)     # every rank: rank 0's save() may not have run yet / here
        torch.save({k: v.clone() for k, v in trainer.model_fusion.state_dict().items()},
                   "exp/{}/dry_rank{}.pt".format(trainer.log_time, trainer.rank))
        if trainer.rank == 0:
            print("DRY (stand-in arithmetic, nothing of the engine ran): {} rank(s), run {}".format(trainer.world, trainer.log_time), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    trainer = Trainer(args.mode, args.config, ov, arith_mode=args.arith)
    from models.fusion_models import utils          # the scoring entry points, called as train_fusion.py:430-469 calls them

    def report(fn):
        # every rank extracted its shard and holds the gathered tables; the store is on disk: rank 0 scores and prints
        if trainer.rank == 0:
            eer, threshold = fn(trainer.log_time)
            print("EER: {:.6f}%".format(eer * 100))

    if args.mode == "train":
        trainer()
        if trainer.test_opts.get("model_average", False):      # commented out upstream (train_fusion.py:428)
            if ddist.active():
                dist.barrier()
            trainer.model_average(min(2, trainer.epoch))
        trainer.extract_test_xv_lomgrid()
        report(utils.eer_cos_lomgrid_featurefusion)
    else:
        sets = [(s, getattr(trainer, "extract_test_xv_" + s)) for s in ("lomgrid", "grid") if trainer.test_opts["eval_" + s]]
        for name, extract in sets:
            extract()
            if trainer.test_opts["use_cos"]:
                report(getattr(utils, "eer_cos_" + name + ("" if args.mode == "av_test" else "_scorefusion")))
            if trainer.test_opts["use_plda"]:
                report(getattr(utils, "eer_plda_" + name))
    if trainer.rank == 0 and arith.STATS["f32_reruns"]:
        print("arith auto: {} batch(es) computed again in exact f32".format(arith.STATS["f32_reruns"]))
    trainer.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
