#!/usr/bin/env python3
"""bench.py -- fused audio-visual lip-biometric embedding throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 outside a torch.distributed job starts that job itself (deeplip_amd/launch.py: a child
`python -m torch.distributed.run ... bench.py <same arguments>`, started before anything touches the GPU), relays rank 0's
JSON line and exits with the job's return code -- one command starts every GPU, as the reference's `python train_fusion.py`
does (train_fusion.py:88-93).  `--dry-launch` rehearses exactly that launch on CPU ranks (gloo, a stand-in step): it tests
the launcher, the rendezvous, the exchange and the one-line contract where no second GPU exists, and measures nothing.

One "step" = one pass of the hot path over one batch of synthetic A+V pairs PER RANK
(BASELINE.json configs[1] clip batch, one utterance per clip -- SURVEY.md section 8d C2/C3/C4):
    video  [B,1,29,88,88] -> Lipreading(extract_feats) -> temporal mean -> [B,512]
    audio  [B,1,80,300]   -> E-TDNN extract_embedding             -> [B,512]
    fuse   z-norm + concat (train_fusion.py:353-358)               -> [B,1024]
    (N>1)  all-gather of the [B,1024] rows over RCCL: the enrol/verify exchange step
Inputs and weights are synthetic (name-keyed generator, seed 1), already resident in HBM when the
timed region starts.  Rank 0 prints ONE JSON line.  metric = lip-clips/sec (whole job).

Arithmetic modes (--precision): "f16x3" (default) evaluates every conv/linear product of fp32
values as hi*hi + hi*lo + lo*hi on the f16 matrix core with fp32 accumulation (fp32-grade: measured
~1e-6 vs the CPU reference, bar 1e-4); "f32" is the exact fp32-MFMA path.  Both are timed in one
run; the other one is reported under "alt_mode".

roofline: the path is a dense contraction (MFMA-bound).  `achieved` is for the dominant kernel (the
implicit-GEMM conv instance that carries most FLOPs): ALGORITHMIC FLOPs (2*M*N*K) of its launches
in a step / the sum of their durations, measured with HIP events around every launch inside the
timed region, on the stream the kernels run on.  `peak`: 157.3 TFLOP/s (fp32 MFMA) for "f32";
2500/3 = 833.3 TFLOP/s for "f16x3" (dense f16 MFMA peak over the 3 MFMAs each product costs).
`step_achieved` applies BASELINE.md's whole-step formula (clips/s x 20.90 GFLOP).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deeplip_amd import launch as _launch

if __name__ == "__main__" and not _launch.in_job():
    # N > 1 asked of a plain process: become the launcher NOW -- before torch is imported, long before any torch.cuda call
    def _relay(line):       # stdout carries rank 0's JSON line and nothing else; library chatter of the ranks goes to stderr
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    _n = _launch.argv_gpus(sys.argv[1:])
    if _n > 1:
        sys.exit(_launch.self_launch(os.path.abspath(__file__), sys.argv[1:], _n, relay=_relay))

LIVE_PMC = None       # {"kernels": pmc.summarise(...), "steps": n, "seconds": s} from this run's own counter passes, or {"skipped": why}


def _live_pmc_passes(argv):
    """`roofline.traffic` measured IN THIS RUN: two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, as the guide
    prescribes) of this very command, run as CHILD processes NOW -- before this process has touched the GPU -- on 5 + 2 replayed steps
    each.  Single process only (N = 1 outside a job); --no-pmc skips them; any failure leaves the committed collection
    (profiles/traffic_latest.json, sha-stamped) as the source, and the line says which it was."""
    import shutil
    import subprocess
    import tempfile
    if "--pmc-child" in argv or "--no-pmc" in argv or "--dry-launch" in argv or "--eager" in argv:
        return {"skipped": "disabled for this invocation"}
    # Never from inside a profiler: under `rocprofv3 ... -- python3 bench.py` the tool's preloaded library may already have initialised
    # the GPU in THIS process, and a process that holds the GPU must not start (fork + exec) another program on this pool.
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return {"skipped": "this process runs under a profiler"}
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"skipped": "rocprofv3 not on PATH"}
    steps, warm = 5, 2
    t0 = time.perf_counter()
    dirs = []
    passthrough = []
    for flag in ("--batch", "--audio-dim", "--precision"):
        if flag in argv:
            i = argv.index(flag)
            passthrough += argv[i:i + 2]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="dlip_pmc_")
            dirs.append(d)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", str(steps),
                   "--warmup", str(warm), "--no-cpu-baseline", "--single-mode", "--no-configs", "--no-kernel-events", "--no-h2d", "--no-spans",
                   "--pmc-child"] + passthrough
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
            if r.returncode != 0:
                return {"skipped": f"the {counter} pass exited with {r.returncode}"}
        from deeplip_amd import pmc
        kernels = pmc.summarise(dirs, steps + warm)
        if not kernels:
            return {"skipped": "the passes wrote no counter rows"}
        return {"kernels": kernels, "steps": steps + warm, "seconds": round(time.perf_counter() - t0, 1)}
    except Exception as ex:   # noqa: BLE001 -- evidence gathering must never take the measurement down
        return {"skipped": f"{type(ex).__name__}: {ex}"[:200]}
    finally:
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__" and not _launch.in_job() and _launch.argv_gpus(sys.argv[1:]) <= 1:
    LIVE_PMC = _live_pmc_passes(sys.argv[1:])       # (child processes; this one has not imported torch yet)

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (spec)
DTYPE_NAME = {"f32": "f32", "f16x3": "f16x3"}
DTYPE_NOTE = {"f32": "fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 fma chains",
              "f16x3": "fp32 values carried as hi+lo fp16 pairs, 3 f16 MFMAs per product, fp32 accumulate (fp32-grade results)"}
GFLOP_PER_FUSED_CLIP = 20.90      # BASELINE.md section 2 (18.337 video + 2.563 audio)
REGIONS = 3                       # timed regions of K steps each; the line reports the median one
TCN_OPTS = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
ETDNN_CONTEXT = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]


class EventHook:
    """Brackets MFMA-kernel launches with HIP events on the current stream.  ``only`` restricts the
    bracketing to one kernel instance (the dominant one inside the timed region: an event record is a
    queue packet of its own, so bracketing all ~32 launches of a step costs the step a few percent);
    without events the hook still tallies FLOPs per instance (how the dominant one is found)."""

    def __init__(self):
        self.records = []   # (name, flops, ev0, ev1)
        self.flops = {}     # name -> algorithmic FLOPs seen (events or not)
        self.enabled = False
        self.only = None

    def begin(self, name, flops):
        self.flops[name] = self.flops.get(name, 0.0) + flops
        if not self.enabled or (self.only is not None and name != self.only):
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, flops, e0, e1)

    def end(self, tok):
        if tok is not None:
            tok[3].record()
            self.records.append(tok)

    def summary(self):
        agg = {}
        for name, flops, e0, e1 in self.records:
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
        return agg


def build_models(device, audio_dim):
    from deeplip_amd import weightgen as wg
    from models.audio_models.tdnn import SpeakerEmbNet
    from models.video_models.model import Lipreading
    video = Lipreading(hidden_dim=256, backbone_type="resnet", num_classes=57, relu_type="prelu",
                       tcn_options=TCN_OPTS, width_mult=1.0, extract_feats=True)
    et = {"input_dim": audio_dim, "hidden_dim": [512] * 9 + [1500], "context": ETDNN_CONTEXT, "tdnn_layers": 10,
          "embedding_dim": 512, "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
    audio = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
    sds = []
    for m, pre in ((video, "video."), (audio, "audio80.")):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = wg.fill_state_dict(shapes, prefix=pre)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval().to(device)
        sds.append(sd)
    return video, audio, sds


AUDIO_STREAM = None   # --eager --audio-stream (round-1 path)
SINGLE_STREAM = False


def kernel_source_sha():
    """sha256 (16 hex digits) over the sources the dominant kernel is compiled from (its .hip file and the headers it includes):
    what profiles/traffic_latest.json is stamped with."""
    from deeplip_amd import build
    return build.dominant_kernel_sha()


def local_step(video, audio, xv, xa, sequential=False):
    """The per-rank part of a step: launches only (what a StepPlan records).  The speech encoder is issued on a second
    HIP stream (deeplip_amd.fusion.embed_av: a fork / join the recorded plan keeps as two branches of its graph), so
    the tail of one encoder's launches overlaps the head of the other's; --single-stream issues them in sequence."""
    from deeplip_amd import fusion
    return fusion.embed_av(audio, video, xa, xv, two_streams=not (SINGLE_STREAM or sequential))


def exchange(fused, world):
    """The enrol/verify exchange: every rank gets every rank's fused rows (RCCL all-gather over xGMI)."""
    if not (dist.is_available() and dist.is_initialized()):
        return fused
    out = torch.empty((world * fused.shape[0], fused.shape[1]), device=fused.device, dtype=fused.dtype)
    dist.all_gather_into_tensor(out, fused)
    return out


def step(video, audio, xv, xa, world):
    from deeplip_amd import fusion
    if AUDIO_STREAM is not None:
        cur = torch.cuda.current_stream()
        AUDIO_STREAM.wait_stream(cur)
        with torch.cuda.stream(AUDIO_STREAM):
            xv_audio, _ = audio.extract_embedding(xa)
        em_video = video.embed(xv)
        cur.wait_stream(AUDIO_STREAM)
    else:
        em_video = video.embed(xv)
        xv_audio, _ = audio.extract_embedding(xa)
    fused = fusion.fuse_av(xv_audio, em_video)
    if dist.is_available() and dist.is_initialized():
        out = torch.empty((world * fused.shape[0], fused.shape[1]), device=fused.device, dtype=fused.dtype)
        dist.all_gather_into_tensor(out, fused)
        return out
    return fused


def _host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(sds, audio_dim, budget_s=12.0, sample=8):
    """The CPU oracle (port of the reference path) timed on this box's host cores.  torch's intra-op
    pool collapses when oversubscribed, so the thread count is calibrated first (a few short passes
    at 8..cores threads) and the best one is used and reported."""
    from deeplip_amd import weightgen as wg
    from oracle import deeplip_oracle as O
    vsd, asd = O.to_torch_sd(sds[0]), O.to_torch_sd(sds[1])
    cores = _host_cores()
    xv = torch.from_numpy(wg.video_input(sample, key="bench.video"))
    xa = torch.from_numpy(wg.audio_input(sample, audio_dim, 300, key="bench.audio"))
    cands = sorted({c for c in (8, 16, 32, 64, 128, cores) if c <= cores}) or [cores]
    best_t, best_n = 1e30, cands[0]
    for n in cands:
        torch.set_num_threads(n)
        O.fused_av_embedding(vsd, asd, xv[:2], xa[:2])
        t0 = time.perf_counter()
        O.fused_av_embedding(vsd, asd, xv[:2], xa[:2])
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_t, best_n = dt, n
        if dt > 4 * best_t:
            break
    torch.set_num_threads(best_n)
    ref = O.fused_av_embedding(vsd, asd, xv, xa)   # warm-up + parity reference
    iters, t0 = 0, time.perf_counter()
    while True:
        O.fused_av_embedding(vsd, asd, xv, xa)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 50:
            break
    return {"value": round(sample * iters / el, 3), "unit": "lip-clips/sec", "cores": best_n,
            "kind": "port",
            "sample": f"{iters} passes of the oracle fused A+V embed on {sample} clips [{sample},1,29,88,88] + "
                      f"[{sample},{audio_dim},300] (torch-CPU fp32, {best_n} of {cores} usable cores, {el:.1f}s)"}, ref, xv, xa


def _timed_replay(run, steps, warmup, sync):
    """warmup + steps calls of ``run`` bracketed by HIP events on the current stream -> ms per call."""
    for _ in range(warmup):
        run()
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    sync()
    return e0.elapsed_time(e1) / steps


def extra_configs(args, device, video, audio, xv, xa, peak, StepPlan, c5_all=None):
    """The other BASELINE.json configurations, measured beside the headline: every entry is
    plan replay over inputs resident in HBM, HIP-event timed; failures are recorded, never fatal to the headline.
      C2 video-only clip embed [B,1,29,88,88]; C3 speech-encoder embed [256,1,F,300] (configs[2]);
      C4 fused extraction of a 256-utterance test list + 20 000 cosine trials shaped like trial_grid_v1.txt + EER;
      C5 one rank's DP training step of the fusion head (frozen encoders, bs 60): where its time goes;
      F2 (not a BASELINE configuration): one optimisation step of the FULL lip-clip model (B = 32, replayed
      HIP graph) and of the full speech encoder (B = 256) -- the trainers' throughput where the driver measures it."""
    from deeplip_amd import _lib, fusion, ops, scoring, weightgen as wg
    from deeplip_amd.synthetic import SyntheticAVSet, synthetic_trials
    out = {}
    sync = torch.cuda.synchronize
    B = xv.shape[0]
    st, wu = max(5, args.steps // 2), 2

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as ex:   # noqa: BLE001 -- a broken side configuration must not take the headline down
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    def c2():
        plan = StepPlan(lambda v: video.embed(v), xv)
        ms = _timed_replay(plan.run, st, wu, sync)
        plan.close()
        tf = B / ms * 18.337           # clips per ms x GFLOP per 29-frame clip = TFLOP/s
        return {"workload": f"video-only embed [{B},1,29,88,88] -> [{B},512]", "value": round(1e3 * B / ms, 1), "unit": "lip-clips/sec",
                "ms_per_step": round(ms, 4), "step_tflops": round(tf, 1), "step_frac": round(tf / peak, 4)}

    def c3():
        Ba = 256
        xa3 = torch.from_numpy(wg.audio_input(Ba, args.audio_dim, 300, key="bench.audio.c3")).unsqueeze(1).to(device)
        hook = EventHook()
        ops.LAUNCH_HOOK = hook
        try:
            audio.extract_embedding(xa3)
            hook.enabled = True
            hook.records = []
            audio.extract_embedding(xa3)
            sync()
        finally:
            ops.LAUNCH_HOOK = None
        agg = hook.summary()
        dom = max(agg.items(), key=lambda kv: kv[1]["flops"])
        plan = StepPlan(lambda a_: audio.extract_embedding(a_)[0], xa3)
        ms = _timed_replay(plan.run, st, wu, sync)
        plan.close()
        gf = 2.563 + (0.085 if args.audio_dim == 80 else 0.0)      # GFLOP per 300-frame utterance
        tf = Ba / ms * gf
        dtf = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
        return {"workload": f"speech-encoder embed [{Ba},1,{args.audio_dim},300] -> [{Ba},512] (BASELINE configs[2])",
                "value": round(1e3 * Ba / ms, 1), "unit": "utt/sec", "ms_per_step": round(ms, 4), "step_tflops": round(tf, 1),
                "step_frac": round(tf / peak, 4),
                "dominant_kernel": {"kernel": dom[0], "launches": dom[1]["launches"], "tflops": round(dtf, 1), "frac": round(dtf / peak, 4)}}

    def c4():
        from deeplip_amd.pipeline import ExtractPipeline
        from deeplip_amd.synthetic import frames_u8_from_clips
        ds = SyntheticAVSet(32, 8, 1, 29, args.audio_dim, 300, key="bench.c4")        # 256 utterances, 32 speakers, 1 clip each
        n = len(ds)
        host = []
        for b0 in range(0, n, B):
            idx = list(range(b0, min(n, b0 + B)))
            host.append((torch.from_numpy(frames_u8_from_clips(ds.video(idx)[0], rgb=True)).pin_memory(),
                         torch.from_numpy(ds.audio(idx)).unsqueeze(1).pin_memory()))
        # uint8 RGB frames [B,29,3,88,88] + mel [B,1,F,300] from pinned host memory, copies on their own stream behind the
        # replay of the previous batch (deeplip_amd/pipeline.py); the list is walked `passes` times so that >= 2048 utterances
        # flow through (the timing does not care that rows repeat; the scored table is one pass)
        pipe = ExtractPipeline(lambda fr, a_: local_step(video, audio, fr, a_), host[0][0].to(device), host[0][1].to(device))
        passes = max(1, -(-2048 // n))
        table = torch.empty((passes * n, 1024), device=device)
        pipe.run(host, table); pipe.finish()                                            # warm-up pass
        sync()
        t0 = time.perf_counter()
        pipe.run(host * passes, table)
        pipe.finish()
        extract_s = time.perf_counter() - t0
        pipe.close()
        table = table[:n].clone()
        y, pairs = synthetic_trials(ds, 20000, 4000)
        tab = scoring.EmbeddingTable(ds.utt_ids, table)
        ia, ib = tab.trial_indices(pairs)
        ms = _timed_replay(lambda: scoring.cosine_scores(table, ia, ib), 20, 3, sync)
        scores = scoring.cosine_scores(table, ia, ib).cpu().numpy()
        eer, thr = scoring.eer_from_scores(y, scores)
        h2d_mb = sum(t.numel() * t.element_size() for t in host[0]) / host[0][0].shape[0] / 1e6
        return {"workload": f"fused A+V extraction of {passes * n} utterances ({passes} passes over a {n}-utterance list; uint8 RGB frames "
                            f"[{B},29,3,88,88] + mel [{B},1,{args.audio_dim},300] copied from pinned host memory inside the loop, double-buffered "
                            "behind the compute) + 20000 cosine trials (4000 target / 16000 non-target, trial_grid_v1.txt shape) + EER",
                "extract_utt_per_s": round(passes * n / extract_s, 1), "h2d_mb_per_pair": round(h2d_mb, 3),
                "h2d_gb_per_s": round(passes * n / extract_s * h2d_mb / 1e3, 2),
                "trials_per_s": round(20000 / (ms * 1e-3), 0), "scoring_ms": round(ms, 4), "eer": round(float(eer), 6),
                "eer_note": "random-init weights, but the synthetic list's speakers share their generators across utterances and carry "
                            "session variability beside them (SyntheticAVSet session / jitter = 1): the EER sits mid-range, so its agreement "
                            "with the oracle's EER on the same table is a real check of the scoring path",
                "threshold": round(float(thr), 6), "_table": table, "_trials": (y, ia.cpu().numpy(), ib.cpu().numpy(), scores)}

    def c4_ragged(rect):
        """C4's extraction on a RAGGED list -- what the reference's trial lists are (25 834 utterances of differing duration,
        BASELINE.md section 1; 1-3 clip files per utterance): audio 137 .. 412 frames, clips 11 .. 75 frames, length-bucketed
        batches through one recorded plan per padded shape (deeplip_amd/extract.py).  Reported in VALID (un-padded) work per second
        -- GFLOP of the frames that belong to utterances, at BASELINE.md's per-frame figures -- beside the same figure of the
        rectangular list above (`rect`): padding, short last batches and plan switches are what the ratio pays for."""
        from deeplip_amd.extract import RaggedExtractor
        n_spk, per = 32, 32                                                          # 1024 utterances, ~2050 clips
        ds = SyntheticAVSet(n_spk, per, 3, 29, args.audio_dim, 300, key="bench.c4r", ragged=True)

        # the timing does not care what the pixels are: every item is cut from one of a few generated maximum-length items
        pool_a = [wg.audio_input(1, args.audio_dim, 412, key=f"bench.c4r.a{i}", speakers=[i])[0] for i in range(8)]
        pool_v = [wg.video_input(1, 75, 88, key=f"bench.c4r.v{i}", speakers=[i])[0, 0] for i in range(4)]
        ds.audio_item = lambda i: pool_a[i % 8][:, :int(ds.audio_len[i])]
        ds.clip_item = lambda c: pool_v[c % 4][:int(ds.clip_len[c])]
        ex = RaggedExtractor(lambda a, l: audio.extract_embedding(a, lengths=l)[0], lambda v, l: video.embed(v, lengths=l),
                             device, batch=B, clip_batch=B, max_arena_bytes=176 << 30)   # every rung's plans stay recorded (~100 GB of 288)
        cache = {}
        n = len(ds)
        try:
            ex.run(ds, 0, n, 512, u8=True, host_cache=cache)                         # warm-up pass: records the plans, pins the host batches
            sync()
            t0 = time.perf_counter()
            passes = 2
            for _ in range(passes):
                xa_r, xv_r = ex.run(ds, 0, n, 512, u8=True, host_cache=cache)
            sync()
            el = (time.perf_counter() - t0) / passes
            stats = dict(ex.stats)
        finally:
            ex.close()
        gf_v, gf_a = 18.337 / 29.0, (2.563 + (0.085 if args.audio_dim == 80 else 0.0)) / 300.0        # GFLOP per valid frame
        valid_gf = stats["valid_video_frames"] * gf_v + stats["valid_audio_frames"] * gf_a
        rect_gf_per_s = rect["extract_utt_per_s"] * (18.337 + gf_a * 300.0)
        return {"workload": f"{n} utterances (audio {int(ds.audio_len.min())}..{int(ds.audio_len.max())} frames), {len(ds.clip_len)} lip clips "
                            f"({int(ds.clip_len.min())}..{int(ds.clip_len.max())} frames, 1-3 per utterance), uint8 RGB frames + mel from pinned host "
                            f"memory, batches of {B} utterances / {B} clips sorted by length and padded to a ladder of lengths (padding <= 10 %)",
                "extract_utt_per_s": round(n / el, 1), "clips_per_s": round(len(ds.clip_len) / el, 1),
                "valid_video_frames_per_s": round(stats["valid_video_frames"] / el, 0), "valid_audio_frames_per_s": round(stats["valid_audio_frames"] / el, 0),
                "valid_gflop_per_s": round(valid_gf / el, 1), "rectangular_gflop_per_s": round(rect_gf_per_s, 1),
                "valid_work_rate_vs_rectangular": round(valid_gf / el / rect_gf_per_s, 4), **stats}

    def e1_av_test(ragged):
        """The product surface, measured as a user drives it: train_fusion.Trainer('av_test') on a ragged synthetic list -> the
        reference's on-disk store -> models.fusion_models.utils.eer_cos_lomgrid(run) (train_fusion.py:317-420,423-451).  Extraction is
        the second pass over the list (plans recorded, host batches pinned: test.cache_host_batches), in VALID work per second like
        C4_ragged_extraction, whose rate it is held against."""
        import train_fusion
        from models.fusion_models import utils
        ov = {"data.test_speakers": 32, "data.test_utt_per_spk": 32, "data.test_clips_per_utt": 3, "data.utt_per_spk": 2,
              "model.audio_config.etdnn.input_dim": args.audio_dim, "test.batch": B, "test.frames": "u8", "test.cache_host_batches": True,
              "test.max_arena_gb": 176, "test.eval_grid": False}

        def run():
            from deeplip_amd import scoring_entry as se
            tr = train_fusion.Trainer("av_test", overrides=ov)
            try:
                ds = tr.lomgridtestset
                pool_a = [wg.audio_input(1, args.audio_dim, 412, key=f"bench.e1.a{i}", speakers=[i])[0] for i in range(8)]
                pool_v = [wg.video_input(1, 75, 88, key=f"bench.e1.v{i}", speakers=[i])[0, 0] for i in range(4)]
                ds.audio_item = lambda i: pool_a[i % 8][:, :int(ds.audio_len[i])]       # (the timing does not care what the pixels are)
                ds.clip_item = lambda c: pool_v[c % 4][:int(ds.clip_len[c])]
                tr._extract(ds)                                                          # records the plans, pins the host batches
                sync()
                t0 = time.perf_counter()
                tables = tr._extract(ds)
                sync()
                t_ex = time.perf_counter() - t0
                tr.lomgrid_tables = tables
                t0 = time.perf_counter()
                tr._write_store("lomgrid", ds, tables)
                t_store = time.perf_counter() - t0
                t0 = time.perf_counter()
                eer, thr = utils.eer_cos_lomgrid(tr.log_time)
                t_score = time.perf_counter() - t0
                stats = dict(tr.extract_stats)
                n = len(ds)
                gf_v, gf_a = 18.337 / 29.0, (2.563 + (0.085 if args.audio_dim == 80 else 0.0)) / 300.0
                valid_gf = stats["valid_video_frames"] * gf_v + stats["valid_audio_frames"] * gf_a
                return {"workload": f"train_fusion.Trainer('av_test'): {n} utterances / {len(ds.clip_len)} lip clips of differing length "
                                    f"(batches of {B}, uint8 RGB frames + mel from pinned host memory) -> fused [N,1024] table -> the reference's "
                                    ".npy store -> utils.eer_cos_lomgrid(run) over 20000 trials",
                        "extract_utt_per_s": round(n / t_ex, 1), "clips_per_s": round(len(ds.clip_len) / t_ex, 1),
                        "valid_gflop_per_s": round(valid_gf / t_ex, 1),
                        "vs_C4_ragged_extraction": round(valid_gf / t_ex / ragged["valid_gflop_per_s"], 4) if ragged and "valid_gflop_per_s" in ragged else None,
                        "store_write_s": round(t_store, 3), "score_from_store_s": round(t_score, 3), "eer": round(float(eer), 6),
                        "eer_note": "near 0.5 by construction: this leg cuts every item from a small pool of generated clips / utterances (the timing "
                                    "does not care what the pixels are), so the list carries no speaker structure; C4_fusion_scoring is the EER check",
                        "arith": tr.arith, "f32_reruns": stats.get("f32_reruns", 0), "plans_recorded": stats.get("plans_recorded")}
            finally:
                tr.close()
                se._process_paths.clear()
        return _in_tmp(run)

    def e2_train_video(f2):
        """train_video.py as a user runs it: main() with the reference's flags, full-model training at B = 32 x 29 frames -- the loop's
        own clips/s (from its 4th step on: recorded steps replayed, the next batch's copies behind the running step)."""
        import train_video

        def run():
            train_video.main(["--batch-size", "32", "--frames", "29", "--steps", "27", "--maxepoch", "1", "--data-cache", "2", "--display", "100",
                              "--save-path", os.path.join(os.getcwd(), "ck")])
            st = dict(train_video.train.last_stats)
            st = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}
            st["workload"] = "train_video.main(): full Lipreading training (forward + backward + Adam + per-iteration cosine LR), 32 clips x 29 frames per step"
            if f2 and "clips_per_s" in f2:
                st["vs_F2_train_video_step"] = round(st["clips_per_s"] / f2["clips_per_s"], 4)
            return st
        return _in_tmp(run)

    def e3_train_audio(f2):
        """train_audio.Trainer._train_epoch(): full E-TDNN training (LMCL, SGD) at B = 256 x 300 frames, the second epoch (recorded
        step replayed)."""
        import train_audio
        ov = {"train.bs": 256, "train.steps_per_epoch": 10, "train.crop_frames": [300, 300], "data.audio_frames": 300, "train.data_cache": 2,
              "data.utt_per_spk": 8, "data.test_speakers": 2, "data.test_utt_per_spk": 2}

        def run():
            tr = train_audio.Trainer(overrides=ov)
            try:
                tr.current_epoch = 1
                tr._train_epoch()
                tr.current_epoch = 2
                tr._train_epoch()
                st = dict(tr.last_epoch_stats)
                out_ = {"workload": "train_audio.Trainer._train_epoch(): full E-TDNN training step (forward + backward + SGD, LMCL), 256 utterances "
                                    "x 300 frames x 24 features", "utt_per_s": round(st["utt_per_s"], 1),
                        "ms_per_step": round(1e3 * st["bs"] / st["utt_per_s"], 3), "step_mode": st["step_mode"], "loss": round(st["loss"], 4),
                        "arith": tr.arith}
                if f2 and "utt_per_s" in f2:
                    out_["vs_F2_train_audio_step"] = round(out_["utt_per_s"] / f2["utt_per_s"], 4)
                return out_
            finally:
                tr.close()
        return _in_tmp(run)

    def f2_video():
        """one optimisation step of the FULL lip-clip model (ResNet-18 + MS-TCN, Adam) at the reference's shapes,
        B = 32 clips x 29 frames, recorded once and replayed as one HIP graph (deeplip_amd.train_plan)."""
        import gc
        from deeplip_amd import autograd as ag
        from deeplip_amd.train_plan import TrainStepGraph
        from models.video_models.model import Lipreading
        Bt = 32
        tcn = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": False, "width_mult": 1}
        net = Lipreading(num_classes=54, relu_type="prelu", tcn_options=tcn, extract_feats=False)
        sdv = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="video.")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()})
        net.to(device).train()
        opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(3e-4, device=device), weight_decay=1e-4, capturable=True, fused=True)
        xb = torch.from_numpy(wg.video_input(Bt, frames=29, key="bench.vtrain")).to(device)
        lb = torch.from_numpy(wg.labels(Bt, 54)).to(device)
        ln = torch.full((Bt,), 29, dtype=torch.int32, device=device)

        def one(x_, l_, n_):
            opt.zero_grad(set_to_none=True)
            ls = ag.margin_ce_loss(net(x_, lengths=n_), l_)
            ls.backward()
            opt.step()
            return ls

        plan = TrainStepGraph(one, eager_steps=1, device=device)
        for _ in range(3):
            plan.step(xb, lb, ln)
        plan.finish()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            ls = plan.step(xb, lb, ln)
        plan.finish()
        ms = (time.perf_counter() - t0) / n * 1e3
        res = {"workload": f"full Lipreading training step (forward + backward + Adam), {Bt} clips x 29 frames, replayed HIP graph",
               "clips_per_s": round(1e3 * Bt / ms, 1), "ms_per_step": round(ms, 3), "loss": round(float(ls.detach()), 4),
               "tflops_at_3x_forward": round(3 * 20.567 * Bt / ms, 1)}
        del plan, net, opt
        gc.collect(); torch.cuda.empty_cache()
        return res

    def f2_audio():
        """one optimisation step of the full speech encoder (E-TDNN, LMCL, SGD), B = 256 x 300 frames."""
        import gc
        from models.audio_models.loss import LMCL
        from models.audio_models.tdnn import SpeakerEmbNet
        Bt = 256
        ctx = [[-2, -1, 0, 1, 2], [0], [-2, 0, 2], [0], [-3, 0, 3], [0], [-4, 0, 4], [0], [0], [0]]
        et = {"input_dim": 24, "hidden_dim": [512] * 9 + [1500], "context": ctx, "tdnn_layers": 10, "embedding_dim": 512,
              "pooling": "statistic", "attention_hidden_size": 64, "bn_first": True}
        net = SpeakerEmbNet({"arch": "etdnn", "etdnn": et})
        sda = wg.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, prefix="audio.")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sda.items()})
        net.to(device).train()
        crit = LMCL(512, 57, 30, 0.2).to(device)
        opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], 0.01, momentum=0.9, weight_decay=1e-5)
        xb = torch.from_numpy(wg.audio_input(Bt, 24, 300, key="bench.atrain")).to(device)
        lb = torch.from_numpy(wg.labels(Bt, 57)).to(device)

        def one():
            opt.zero_grad()
            ls, _ = crit(net(xb), lb)
            ls.backward()
            opt.step()
            return ls

        for _ in range(2):
            one()
        sync()
        n = 8
        t0 = time.perf_counter()
        for _ in range(n):
            ls = one()
        sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        _lib.check_range(sync=True)
        res = {"workload": f"full E-TDNN training step (forward + backward + SGD, LMCL), {Bt} utterances x 300 frames x 24 features",
               "utt_per_s": round(1e3 * Bt / ms, 1), "ms_per_step": round(ms, 3), "loss": round(float(ls.detach()), 4)}
        del net, opt, crit
        gc.collect(); torch.cuda.empty_cache()
        return res

    guarded("C2_video_embed", c2)
    guarded("C3_audio_embed", c3)
    guarded("C4_fusion_scoring", c4)
    if "error" not in out["C4_fusion_scoring"]:
        guarded("C4_ragged_extraction", lambda: c4_ragged(out["C4_fusion_scoring"]))
    in_job = dist.is_available() and dist.is_initialized()
    if c5_all is not None:
        out["C5_fusion_train_step"] = c5_all     # a job of several ranks: EVERY rank runs the trainer's epoch at the end of main()
    else:
        guarded("C5_fusion_train_step", lambda: c5_entry(args))
    guarded("F2_train_video_step", f2_video)
    guarded("F2_train_audio_step", f2_audio)
    if not args.no_entry_points and not in_job:
        # the three entry points measured by calling them (the product surface, in the arithmetic their configs name: auto)
        guarded("E1_train_fusion_av_test", lambda: e1_av_test(out.get("C4_ragged_extraction")))
        guarded("E2_train_video_main", lambda: e2_train_video(out.get("F2_train_video_step")))
        guarded("E3_train_audio_epoch", lambda: e3_train_audio(out.get("F2_train_audio_step")))
    return out


def _in_tmp(fn):
    """Run an entry-point leg in a scratch directory (the trainers write exp/<run>/ under the working directory)."""
    import shutil
    import tempfile
    cwd, tmp = os.getcwd(), tempfile.mkdtemp(prefix="dlip_bench_")
    from deeplip_amd import arith, packing
    prec, mode = packing.PRECISION, arith.MODE
    os.chdir(tmp)
    try:
        return fn()
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
        arith.configure(mode)                # the trainers configure the arithmetic themselves (model.arith: auto)
        packing.set_precision(prec)

def c5_entry(args):
    """One rank's DP training step of the fusion head THROUGH THE ENTRY POINT: train_fusion.Trainer('train')._train_epoch() --
    the frozen encoders' recorded plan (copies behind compute) + the head's recorded step (Linearfusion + LMCL forward, backward,
    all-reduce of the 3.4 MB head at N > 1, SGD: one HIP graph), conf/fusion_config.yaml:87-99 at bs 60."""
    import train_fusion
    from deeplip_amd import arith
    if os.environ.get("DLIP_BENCH_DP_HANG") == "1":      # test hook (tests/test_rccl_gpu.py): a rank that never comes back -> the watchdog's turn
        time.sleep(10 ** 6)
    sync = torch.cuda.synchronize
    st_, wu = max(5, args.steps // 2), 2
    ov = {"train.bs": 60, "train.loss": "LMCL", "train.steps_per_epoch": 12, "train.data_cache": 2, "train.epoch": 2,
          "model.audio_config.etdnn.input_dim": args.audio_dim, "data.utt_per_spk": 4}

    def run():
        tr = train_fusion.Trainer("train", overrides=ov)
        try:
            tr.current_epoch = 1
            tr._train_epoch()                                    # eager head step, recording, first replays; the batches get pinned
            tr.current_epoch = 2
            tr._train_epoch()
            st = dict(tr.last_epoch_stats)
            # where a step's time goes: the two recorded pieces replayed alone
            pipe, steps = tr._enc_pipe[1], tr._steps.last
            enc_ms = _timed_replay(lambda: pipe.plans[0].run(check_reports=False), st_, wu, sync)
            head_ms = None
            if steps.recorded:
                xs = [t.clone() for t in steps.static]
                head_ms = _timed_replay(lambda: steps.step(*xs), 20, 3, sync)
            return {"workload": "train_fusion.Trainer('train')._train_epoch(): 60 A+V pairs per step, frozen encoders (recorded plan, copies "
                                "behind compute), Linearfusion + LMCL head as one recorded step (forward + backward + SGD; its 3.4 MB "
                                "gradient bucket all-reduced inside at N > 1), conf/fusion_config.yaml:87-99",
                    "pairs_per_s": round(st["pairs_per_s"], 1), "ms_per_step": round(st["ms_per_step"], 4), "step_mode": st["step_mode"],
                    "ms": {"encoders_frozen_extract": round(enc_ms, 4), "head_step_recorded": round(head_ms, 4) if head_ms is not None else None},
                    "loss": round(st["loss"], 4), "arith": tr.arith, "f32_reruns": arith.STATS["f32_reruns"],
                    "verified": steps.verified}
        finally:
            tr.close()
    return _in_tmp(run)



def last_leg_under_watchdog(res, rank, timeout_s, key, fn):
    """Run ``fn()`` -- a leg EVERY rank of the job takes part in -- as the last thing before the line is printed, under a watchdog.
    Everything measured so far is in ``res`` (rank 0's line; None on the other ranks).  If the leg does not come back within
    ``timeout_s`` (one rank failing inside a collective leaves the others waiting for it) the watchdog prints the line as it stands,
    marks the leg, and ends THIS process with code 0; every rank has its own watchdog, so the job ends clean and the record
    survives.  Returns what ``fn`` returned (or an error record), already stored under ``res["configs"][key]`` on rank 0."""
    import threading

    def bail():
        if rank == 0 and res is not None:
            res.setdefault("configs", {})[key] = {"error": f"this leg did not finish within {timeout_s} s; the line was printed by the watchdog"}
            print(json.dumps(res), flush=True)
        os._exit(0)

    dog = threading.Timer(float(timeout_s), bail)
    dog.daemon = True
    dog.start()
    try:
        out = fn()
    except Exception as ex:   # noqa: BLE001
        out = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # the ranks leave the leg together or not at all (a rank that failed above would otherwise run ahead into the teardown)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    dog.cancel()
    if res is not None:
        res.setdefault("configs", {})[key] = out
    return out


def dry_launch(args, world, rank):
    """Rehearsal of the N-rank launch where there is no second GPU: CPU ranks, backend gloo, a stand-in for the per-rank step
    (a [B,1024] row block filled with the rank id), then the REAL protocol of a scaling run -- exchange() after every step,
    barrier-bracketed wall time, MAX over ranks, per-rank self-check records, ONE JSON line from rank 0.  It measures nothing
    (value is null) and touches no kernel: what it proves is that `python bench.py --gpus N` starts N ranks that find each
    other and that a failing rank fails the command."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    if rank == args.dry_fail_rank:
        sys.exit(3)
    B = args.batch
    rows = torch.full((B, 1024), float(rank))

    def sync_all():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        exchange(rows, world)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allrows = exchange(rows, world)
    sync_all()
    my_elapsed = time.perf_counter() - t0
    tmax = torch.tensor([my_elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ok = all(float(allrows[r * B, 0]) == float(r) for r in range(world))      # every rank's block arrived in rank order
    mine = {"rank": rank, "backend": "gloo", "world_size": dist.get_world_size() if world > 1 else 1, "pid": os.getpid(),
            "launched_by": os.environ.get("DLIP_LAUNCHED_BY"), "exchange_ok": bool(ok)}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    res = None
    if rank == 0:
        res = {"metric": "lip-clips/sec (fused A+V embed)", "value": None, "unit": "lip-clips/sec", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * float(tmax.item()) / max(1, args.steps), 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None, "data": "none",
               "dry_launch": True, "config": {"workload": "launcher rehearsal: CPU ranks over gloo, stand-in step, "
                                                           "real exchange protocol; not a measurement"},
               "ranks": ranks}
    if args.dp_leg:
        # rehearsal of the scaling run's LAST leg (the DP training epoch every rank takes part in) and of its watchdog: a stand-in
        # collective; --dry-hang-rank R makes rank R never come back, which leaves the others waiting inside the all-reduce
        def leg():
            if rank == args.dry_hang_rank:
                time.sleep(10 ** 6)
            t = torch.tensor([float(rank + 1)])
            if world > 1:
                dist.all_reduce(t)
            return {"stand_in_allreduce": float(t.item()), "ranks": world}

        last_leg_under_watchdog(res, rank, args.dp_leg_timeout, "C5_fusion_train_step", leg)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="A+V pairs per rank per step (configs[1]: 64)")
    ap.add_argument("--audio-dim", type=int, default=80, help="mel bins F of the [B,1,F,300] audio input")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="implicit-GEMM arithmetic: exact fp32 MFMA, or split fp16 pairs (3 f16 MFMAs per product)")
    ap.add_argument("--single-mode", action="store_true", help="do not also time the other arithmetic mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--audio-stream", action="store_true",
                    help="run the speech encoder on a second HIP stream beside the lip-clip encoder (kernel tails overlap; "
                         "per-kernel event durations then include the time shared with the other stream)")
    ap.add_argument("--single-stream", action="store_true", help="issue the two encoders in sequence on one stream (default: the "
                    "speech encoder on a second stream, fork / join recorded into the step plan)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this run "
                    "(then profiles/traffic_latest.json is quoted, if it belongs to this kernel build)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dp-leg", action="store_true", help="inside a torch.distributed job: run the DP training leg even with --no-configs (tests)")
    ap.add_argument("--dp-leg-timeout", type=int, default=240, help="N > 1: seconds the DP training leg (C5 on every rank) may take before the "
                    "watchdog prints the line without it")
    ap.add_argument("--no-entry-points", action="store_true", help="skip the entry-point lines (E1..E3: the trainers driven as a user drives them)")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configurations (C2..C5) reported under `configs`")
    ap.add_argument("--eager", action="store_true",
                    help="issue every launch of every step from Python (round-1 behaviour) instead of replaying a recorded "
                         "step plan (dlip_plan_run); for A/B runs on one box")
    ap.add_argument("--dbg", action="append", default=[], metavar="KEY=VALUE",
                    help="development: dlip_debug_set(KEY, VALUE) before anything is launched (tile / split / window / tile-order "
                         "choices of the convolution kernels, include/deeplip_hip.h); for whole-step A/B runs on one box -- the line "
                         "then carries `debug`")
    ap.add_argument("--no-spans", dest="spans", action="store_false", help="do not let the timed plan's conv launches time themselves in-kernel")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false", help="skip the H2D-inclusive leg (value_h2d_inclusive)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal on CPU ranks (gloo): stand-in step + the real exchange, one JSON line, no measurement")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="with --dry-launch: this rank exits 3 (tests return-code propagation)")
    ap.add_argument("--dry-hang-rank", type=int, default=-1, help="with --dry-launch --dp-leg: this rank never comes back from the last leg "
                    "(tests the watchdog: the line is printed without the leg, every rank ends with code 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # inside a job the launcher's world size is the truth; outside one, N > 1 was already turned into a job at import
        # time (top of this file) -- reaching this line with a mismatch means an inconsistent hand-made launch
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (self-launching) "
                 "or torch.distributed.run --nproc-per-node N")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # Inside a torch.distributed.run job the process group is ALWAYS initialised -- a one-rank job included, so that a single-GPU
    # box can exercise the RCCL path end to end (init with device_id, all_gather_into_tensor, barrier, MAX all-reduce):
    # `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (tests/test_rccl_gpu.py).
    dist_on = world > 1 or "RANK" in os.environ
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    from deeplip_amd import _lib, arith, fusion, ops, packing, weightgen as wg
    from deeplip_amd.plan import StepPlan
    global SINGLE_STREAM
    SINGLE_STREAM = args.single_stream
    for kv in args.dbg:
        from deeplip_amd import _lib as _dl
        key, val = kv.split("=")
        _dl.debug_set(int(key), int(val))
    if args.audio_stream:
        global AUDIO_STREAM
        AUDIO_STREAM = torch.cuda.Stream(device=device)
    B = args.batch
    # per-rank shard of the synthetic utterance list (weak scaling: B pairs per rank)
    spk = (np.arange(B) + rank * B) % 33
    xv = torch.from_numpy(wg.video_input(B, speakers=spk, key=f"bench.video.r{rank}")).to(device)
    xa = torch.from_numpy(wg.audio_input(B, args.audio_dim, 300, speakers=spk, key=f"bench.audio.r{rank}")).unsqueeze(1).to(device)

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def h2d_inclusive(video, audio):
        """The same step with the inputs NOT resident: every batch comes from pinned host memory inside the timed loop -- uint8 RGB
        frames [B,29,3,88,88] (BASELINE.json's input shape; normalised by the stem's pre-pass) + mel [B,1,F,300] -- through
        deeplip_amd.pipeline.ExtractPipeline (copies on their own stream, two input sets, one plan per set).  Whole job, barrier-
        bracketed, MAX over ranks, like `value`; every rank pulls over its own PCIe link."""
        from deeplip_amd.pipeline import ExtractPipeline
        from deeplip_amd.synthetic import frames_u8_from_clips
        ring = []
        xv_h, xa_h = xv.cpu().numpy(), xa.cpu()
        for r in range(3):                                    # three distinct pinned host batches, walked cyclically
            clips = np.roll(xv_h, r, axis=0)
            ring.append((torch.from_numpy(frames_u8_from_clips(clips, rgb=True)).pin_memory(), torch.roll(xa_h, r, 0).pin_memory()))
        pipe = ExtractPipeline(lambda fr, a_: local_step(video, audio, fr, a_), ring[0][0].to(device), ring[0][1].to(device))
        nb = max(args.steps, 32)
        table = torch.empty((nb * B, 1024), device=device)
        pipe.run([ring[i % 3] for i in range(6)], table); pipe.finish()
        sync_all()
        t0 = time.perf_counter()
        pipe.run([ring[i % 3] for i in range(nb)], table)
        pipe.finish()
        if dist_on:
            exchange(table[:B], world)                       # one exchange of rows closes the job, as a scoring run would
        sync_all()
        el = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        pipe.close()
        mb = sum(t.numel() * t.element_size() for t in ring[0]) / B / 1e6
        v = world * B * nb / float(el.item())
        return {"value": round(v, 2), "batches": nb, "ms_per_step": round(1e3 * float(el.item()) / nb, 4), "h2d_mb_per_pair": round(mb, 3),
                "h2d_gb_per_s_per_gpu": round(v / world * mb / 1e3, 2),
                "input": f"uint8 RGB frames [{B},29,3,88,88] + fp32 mel [{B},1,{args.audio_dim},300] from pinned host memory, double-buffered"}

    def measure(precision):
        """W warm-up + K timed steps in one arithmetic mode -> (result fields, models, state dicts).
        The steady-state loop replays a recorded step plan (one dlip_plan_run per step; --eager issues the
        launches from Python instead) and runs the all-gather exchange after it when N > 1."""
        arith.configure(precision)              # the headline runs the mode it names, no fallback: "f16x3" raises where "auto" would re-run
        video, audio, sds = build_models(device, args.audio_dim)
        hook = EventHook()
        ops.LAUNCH_HOOK = hook                  # events off: tallies the algorithmic FLOPs per kernel instance
        run_stream = torch.cuda.Stream(device=device)
        torch.cuda.synchronize()
        with torch.cuda.stream(run_stream):
            exchange(local_step(video, audio, xv, xa), world)     # one eager step: kernel mix, packing, workspaces
            sync_all()
            dominant = max(hook.flops.items(), key=lambda kv: kv[1])[0] if hook.flops else None
            plan = None
            if args.eager:
                run_step = lambda: step(video, audio, xv, xa, world)
                if not args.no_kernel_events:   # eager: the dominant instance is bracketed inside the timed region
                    hook.only, hook.enabled = dominant, True
            else:
                ops.LAUNCH_HOOK = None
                plan = StepPlan(lambda v, a: local_step(video, audio, v, a), xv, xa, stream=run_stream, spans=args.spans)
                run_step = lambda: exchange(plan.run(), world)
            for _ in range(args.warmup):
                run_step()
            sync_all()
            # The timed region: EXACTLY K steps between barrier + synchronize on both sides -- run REGIONS times back to back, and the
            # line reports the MEDIAN region (value, ms_per_step, the in-kernel spans: all of that one region) with the fastest and
            # slowest beside it (value_min / value_max, roofline.frac_regions): the boxes of this pool differ by +-4-7 %, and within one
            # box the chip's clock moves with temperature -- one sample said little (BENCH r02 -> r03 -> r04: 15 951 / 15 083 / 16 072).
            n_regions = 1 if args.eager else REGIONS
            regions = []
            for _r in range(n_regions):
                if plan is not None and args.spans:
                    plan.span_summary()             # reset: only this region's replays count
                hook.records = []
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                sync_all()
                t0 = time.perf_counter()
                ev0.record()
                for _ in range(args.steps):
                    run_step()
                ev1.record()
                sync_all()
                el = time.perf_counter() - t0
                # what the REPLAYED launches of the region measured about themselves (in-kernel 100 MHz clock, first workgroup
                # in -> last workgroup out, per launch; dlip_span_scope_*)
                rep = plan.span_summary() if (plan is not None and args.spans) else None
                tmax = torch.tensor([el], device=device, dtype=torch.float64)
                if dist_on:
                    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                regions.append({"elapsed": float(tmax.item()), "mine": el, "gpu_ms": ev0.elapsed_time(ev1), "replayed": rep,
                                "stream_us": dict(plan.last_stream_us) if plan is not None else {}})
            hook.enabled = False
            by_time = sorted(range(n_regions), key=lambda i: regions[i]["elapsed"])
            med = regions[by_time[n_regions // 2]]
            elapsed, my_elapsed, gpu_ms, replayed = med["elapsed"], med["mine"], med["gpu_ms"], med["replayed"]
            if plan is not None:
                plan.last_stream_us = med["stream_us"]
            region_values = [world * B * args.steps / r["elapsed"] for r in regions]
            value = world * B * args.steps / elapsed
            ranks = None
            n1_alone = None
            if dist_on:
                # self-check of a scaling run: what every rank saw (RCCL's own world size, its wall time over the same K
                # steps, the all-gather of [B,1024] rows timed on its own)
                fused_probe = torch.zeros((B, 1024), device=device)
                ag_ms = _timed_replay(lambda: exchange(fused_probe, world), 20, 3, sync_all)
                mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "rccl_world_size": dist.get_world_size(),
                        "clips_per_s": round(B * args.steps / my_elapsed, 1), "ms_per_step": round(1e3 * my_elapsed / args.steps, 4),
                        "allgather_us": round(1e3 * ag_ms, 1)}
                ranks = [None] * world
                dist.all_gather_object(ranks, mine)
                # the N = 1 figure of THIS job: rank 0 replays the same K steps alone (no exchange) while the others wait
                # at the barrier -- lets a scaling run be checked against the single-GPU bench line it sits next to
                n1 = None
                if rank == 0:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        plan.run() if plan is not None else step(video, audio, xv, xa, 1)
                    torch.cuda.synchronize()
                    n1 = round(B * args.steps / (time.perf_counter() - t1), 2)
                sync_all()
                n1_alone = n1
            # roofline of the mode: exact fp32 MFMA peak, or the f16 dense peak / 3 (three f16 MFMAs
            # per fp32-grade product: the ceiling of the split algorithm in ALGORITHMIC FLOP/s)
            peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS / 3.0
            step_tf = value / world * GFLOP_PER_FUSED_CLIP / 1e3
            roof = {"bound": "mfma", "achieved": None, "peak": round(peak, 1), "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "step_achieved": round(step_tf, 2), "step_frac": round(step_tf / peak, 4)}
            if precision != "f32":
                roof["peak_note"] = ("algorithmic (2*M*N*K) FLOP/s ceiling of the split-fp16 scheme = 2500 TFLOP/s dense f16 MFMA / 3 "
                                     "MFMAs per product (conv / linear layers and the stem)")
            if not args.no_kernel_events and dominant is not None:
                ops.LAUNCH_HOOK = hook
                if plan is not None:
                    # A replayed plan has no per-launch host hook: the same launches are issued eagerly for K more
                    # steps right behind the timed region, the dominant instance bracketed by HIP events on the
                    # launch stream (its duration does not depend on how the launch was issued; the committed
                    # rocprofv3 kernel trace of this command, which sees the replayed launches, is the cross-check).
                    hook.records, hook.only, hook.enabled = [], dominant, True
                    for _ in range(args.steps):
                        local_step(video, audio, xv, xa, sequential=True)     # one stream: a launch's events time that launch alone
                    sync_all()
                    hook.enabled = False
                    roof["events_region"] = f"{args.steps} eagerly issued steps directly after the timed plan-replay region"
                else:
                    roof["events_region"] = "inside the timed region"
                name, a = dominant, hook.summary()[dominant]
                tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
                roof.update({"achieved": round(tf, 2), "frac": round(tf / peak, 4), "kernel": name,
                             "launches_per_step": a["launches"] // args.steps,
                             "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2),
                             "gflop_per_launch": round(a["flops"] / a["launches"] / 1e9, 3),
                             "timed_by": "HIP events on the launch stream, " + roof["events_region"]})
                # per-instance breakdown of a step: 3 extra, untimed steps with every MFMA launch bracketed
                hook.records, hook.only, hook.enabled = [], None, True
                for _ in range(3):
                    local_step(video, audio, xv, xa, sequential=True)
                sync_all()
                hook.enabled = False
                roof["kernels"] = {k: {"launches_per_step": v["launches"] // 3, "ms_per_step": round(v["ms"] / 3, 4),
                                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                   for k, v in sorted(hook.summary().items())}
                roof["kernels_ms_sum"] = round(sum(v["ms_per_step"] for v in roof["kernels"].values()), 4)
                roof["kernels_note"] = ("3 untimed eager single-stream steps after the timed region, every MFMA launch bracketed by HIP events; "
                                        "the timed plan overlaps the two encoders on two streams, so ms_per_step can be below kernels_ms_sum")
            if replayed:
                # cross-check carried by the driver's own run: the dominant instance as timed INSIDE the replayed, two-stream timed
                # region (a launch shares the chip with the other encoder's launches there, so its span is longer than alone)
                roof["replayed_spans"] = {"note": "in-kernel spans of every MFMA launch of the timed plan-replay region (mean over "
                                                  f"its {args.steps} replays; two streams overlap, so a launch's span includes time shared with the other encoder)",
                                          "kernels": replayed}
                if dominant in replayed:
                    roof["frac_regions"] = [round(r["replayed"][dominant]["tflops"] / peak, 4) if r["replayed"] and dominant in r["replayed"] else None
                                            for r in regions]
                    rd = replayed[dominant]
                    roof["replayed_dominant"] = {"kernel": dominant, **rd, "frac": round(rd["tflops"] / peak, 4)}
                    # THE figure the line leads with: the dominant kernel as timed by the very launches `value` was measured on
                    # (in-kernel spans of the timed two-stream replay).  The single-stream HIP-event figure -- the same launches
                    # alone on the chip, directly behind the timed region -- stays beside it.
                    if roof.get("achieved") is not None:
                        roof.update({"achieved_single_stream": roof["achieved"], "frac_single_stream": roof["frac"],
                                     "avg_launch_us_single_stream": roof["avg_launch_us"], "single_stream_timed_by": roof.pop("timed_by")})
                    roof.update({"achieved": rd["tflops"], "frac": round(rd["tflops"] / peak, 4), "kernel": dominant,
                                 "avg_launch_us": rd["avg_launch_us"], "launches_per_step": rd["launches_per_step"],
                                 "timed_by": f"in-kernel spans (100 MHz clock, first workgroup in -> last workgroup out) of the {args.steps} "
                                             "timed plan replays themselves; the two encoders overlap on two streams there"})
                # do the spans add up to the step?  Per stream of the recorded plan (stream0 = the lip-clip encoder's, the critical
                # path; stream1 = the speech encoder's) the sum of its timed launches' spans per replay, against ms_per_step
                per_stream = {k: round(v / 1e3, 4) for k, v in sorted(plan.last_stream_us.items())}
                if per_stream:
                    crit = max(per_stream.values())
                    step_ms = 1e3 * elapsed / args.steps
                    roof["replayed_spans"]["per_stream_ms"] = per_stream
                    roof["replayed_spans"]["critical_stream_ms"] = round(crit, 4)
                    roof["replayed_spans"]["critical_stream_over_ms_per_step"] = round(crit / step_ms, 4)
                    roof["replayed_spans"]["reconciliation_note"] = (
                        "the critical stream's spans cover every MFMA launch on it (stem incl. its pre-pass, window, ring and rows kernels); "
                        "what they leave of ms_per_step is the element-wise tail (pool finishers, z-norm + concat, the range / span "
                        "bookkeeping launches) and the gaps between launches of a replayed graph")
            if plan is not None:
                roof["plan_launches"] = plan.launches
                plan.close()
            h2d = None
            if args.h2d and not args.eager and precision == args.precision:
                h2d = h2d_inclusive(video, audio)
        ops.LAUNCH_HOOK = None
        # HBM traffic per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot be collected from
        # inside this process, so it comes from the committed PMC passes of this same command (tools/collect_profiles.sh
        # -> profiles/traffic_latest.json) -- and only if that file was produced by THIS kernel source (sha over
        # conv_igemm_f16x3_dma.hip and the headers it includes, stamped into it); otherwise null rather than a stale number.
        live = LIVE_PMC if (isinstance(LIVE_PMC, dict) and "kernels" in LIVE_PMC and precision == args.precision) else None
        if live is not None:
            k = live["kernels"].get(roof.get("kernel", ""))
            if k and "hbm_read_bytes_per_launch" in k and "hbm_write_bytes_per_launch" in k:
                from deeplip_amd import build as _build
                roof["traffic"] = round(k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"])
                roof["traffic_read_write"] = [round(k["hbm_read_bytes_per_launch"]), round(k["hbm_write_bytes_per_launch"])]
                roof["traffic_source"] = ("measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, FETCH x2 gfx950 "
                                          f"correction) of this command over {live['steps']} replayed steps each, run as child processes before the "
                                          f"timed regions ({live['seconds']} s); {k['launches']} launches of the kernel counted")
                roof["traffic_collected"] = {"utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "box": _build.box_id(), "same_box_as_this_run": True,
                                             "in_run": True}
        elif isinstance(LIVE_PMC, dict) and precision == args.precision:
            roof["traffic_in_run"] = LIVE_PMC.get("skipped")
        try:
            if roof.get("traffic") is not None:
                raise StopIteration      # (measured in this run: the committed collection is not consulted)
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            meta = tr.get("_meta", {})
            key = roof.get("kernel", "")
            if meta.get("kernel_sha") != kernel_source_sha():
                roof["traffic_source"] = "profiles/traffic_latest.json is from another build of the kernel (source sha mismatch): ignored"
            elif key in tr and "hbm_read_bytes_per_launch" in tr[key]:
                roof["traffic"] = round(tr[key]["hbm_read_bytes_per_launch"] + tr[key]["hbm_write_bytes_per_launch"])
                from deeplip_amd import build as _build
                same_box = meta.get("box") == _build.box_id()
                roof["traffic_source"] = ("profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per "
                                          "launch, FETCH x2 gfx950 correction; stamped with this kernel source's sha); collected "
                                          f"{meta.get('collected_utc', 'at an unrecorded time')} on {'THIS box' if same_box else 'box ' + str(meta.get('box', '?'))}"
                                          + (" by tools/collect_profiles.sh in the same call as this line" if same_box else ""))
                roof["traffic_collected"] = {"utc": meta.get("collected_utc"), "box": meta.get("box"), "same_box_as_this_run": same_box}
        except Exception:
            pass
        fields = {"value": round(value, 2), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                  "value_min": round(min(region_values), 2), "value_median": round(value, 2), "value_max": round(max(region_values), 2),
                  "value_regions": [round(v, 2) for v in region_values],
                  "step_frac_regions": [round(v / world * GFLOP_PER_FUSED_CLIP / 1e3 / peak, 4) for v in region_values],
                  "dtype": DTYPE_NAME[precision], "dtype_note": DTYPE_NOTE[precision], "gpu_ms_per_step_hip_events": round(gpu_ms / args.steps, 4),
                  "roofline": roof, "ranks": ranks, "n1_alone": n1_alone, "h2d": h2d, "peak": peak}
        return fields, video, audio, sds

    def parity(precision, video, audio, ref, cxv, cxa):
        arith.configure(precision)            # the pack cache is keyed by the arithmetic mode
        got = fusion.fuse_av(audio.extract_embedding(cxa.unsqueeze(1).to(device))[0], video.embed(cxv.to(device)))
        torch.cuda.synchronize()
        err = float((got.cpu() - ref).abs().max() / ref.abs().max())
        gn = torch.nn.functional.normalize(got.cpu().double()); rn = torch.nn.functional.normalize(ref.double())
        sc_err = float(((gn @ gn.t()) - (rn @ rn.t())).abs().max())
        return {"fused_emb_rel_err_vs_cpu": float(f"{err:.3e}"), "cos_score_abs_err_vs_cpu": float(f"{sc_err:.3e}"),
                "tolerance": 1e-4}

    main_fields, video, audio, sds = measure(args.precision)
    configs = None
    if rank == 0 and not args.no_configs:
        configs = extra_configs(args, device, video, audio, xv, xa, main_fields["peak"], StepPlan, {"pending": True} if dist_on else None)
    if dist_on:
        dist.barrier()
    alt = None
    if not args.single_mode:
        alt_prec = "f32" if args.precision != "f32" else "f16x3"
        alt_fields, avideo, aaudio, _ = measure(alt_prec)
        alt = (alt_prec, alt_fields, avideo, aaudio)
    arith.configure("f32")

    res = None
    if rank == 0:
        res = {
            "metric": "lip-clips/sec (fused A+V embed)", "value": main_fields["value"], "unit": "lip-clips/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_fields["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": main_fields["dtype"], "data": "synthetic",
            **({"debug": args.dbg} if args.dbg else {}),
            "config": {"workload": f"fused A+V embed: video [{B},1,29,88,88] 3D-stem+ResNet-18 + audio [{B},1,{args.audio_dim},300] "
                                   f"E-TDNN -> z-norm concat [{B},1024] per rank per step (BASELINE configs[1] clip batch)",
                       "global_batch": world * B, "parallelism": f"dp{world}", "weights": "random-init (name-keyed, seed 1)",
                       "gpu_ms_per_step_hip_events": main_fields["gpu_ms_per_step_hip_events"],
                       "dtype_note": main_fields["dtype_note"]},
            "roofline": main_fields["roofline"],
            # `value` is the MEDIAN of REGIONS back-to-back timed regions of exactly K steps each (barrier + synchronize on both
            # sides of every one); the fastest / slowest region beside it, and the whole-step fraction of each
            "value_min": main_fields["value_min"], "value_median": main_fields["value_median"], "value_max": main_fields["value_max"],
            "value_regions": main_fields["value_regions"], "step_frac_regions": main_fields["step_frac_regions"],
        }
        if main_fields["h2d"] is not None:
            # `value` has the inputs resident in HBM when the timed region starts (the contract); this is the same job with
            # every batch copied from host memory inside the loop
            res["value_h2d_inclusive"] = main_fields["h2d"]["value"]
            res["h2d_inclusive"] = main_fields["h2d"]
        if main_fields["ranks"] is not None:
            res["ranks"] = main_fields["ranks"]
            res["n1_value_rank0_alone"] = main_fields["n1_alone"]
        c4_private = None
        if configs is not None:
            c4 = configs.get("C4_fusion_scoring", {})
            c4_private = (c4.pop("_table", None), c4.pop("_trials", None))
            res["configs"] = configs
        if alt is not None:
            res["alt_mode"] = {k: alt[1][k] for k in ("dtype", "dtype_note", "value", "value_min", "value_max", "ms_per_step", "roofline")}
        if not args.no_cpu_baseline:
            cb, ref, cxv, cxa = cpu_baseline(sds, args.audio_dim)
            res["cpu_baseline"] = cb
            res["parity"] = parity(args.precision, video, audio, ref, cxv, cxa)
            if c4_private and c4_private[0] is not None:
                # C4's "EER delta": the oracle's trial scoring + EER (sklearn / scipy semantics restated) on the SAME
                # embedding table the GPU scored -- the checker beside the GPU number, never the thing timed
                from oracle import deeplip_oracle as O
                y, ia, ib, gpu_scores = c4_private[1]
                osc = O.cosine_trial_scores(c4_private[0].cpu().numpy(), ia, ib)
                oeer, _ = O.eer(list(y), list(osc))
                c4 = res["configs"]["C4_fusion_scoring"]
                c4["oracle_eer"] = round(float(oeer), 6)
                c4["abs_delta_eer"] = float(f"{abs(float(oeer) - c4['eer']):.3e}")
                c4["max_abs_score_err_vs_oracle"] = float(f"{float(np.abs(osc.reshape(-1) - gpu_scores.reshape(-1)).max()):.3e}")
            if alt is not None:
                res["alt_mode"]["parity"] = parity(alt[0], alt[2], alt[3], ref, cxv, cxa)

    if dist_on and (not args.no_configs or args.dp_leg):
        # BASELINE configs[4] inside a scaling run, LAST and under a watchdog: train_fusion.Trainer's DP epoch on EVERY rank, the
        # head's 3.4 MB gradient bucket all-reduced over RCCL INSIDE its recorded step -- a code path no multi-GPU node has run yet
        # (a captured collective on a second communicator; first replay checked against an eager step, train_plan.py).  Everything
        # measured above is already in `res`: if this leg hangs (one rank failing inside a collective would leave the others
        # waiting) the watchdog prints the line as it stands, says so, and ends the process -- the scaling record survives.
        def dp_leg():
            out = c5_entry(args)
            out["ranks"] = world
            return out

        last_leg_under_watchdog(res, rank, args.dp_leg_timeout, "C5_fusion_train_step", dp_leg)
    if rank == 0:
        print(json.dumps(res), flush=True)

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
