"""The nccl (= RCCL) backend for real, on the one GPU a test box has: ONE-rank torch.distributed.run jobs.

RCCL refuses two ranks on one device, so the world size is 1 -- but everything else is the production path: the launcher of
`deeplip_amd.launch` (a child process, never an exec), `init_process_group("nccl", device_id=...)`, all_gather_into_tensor /
all_reduce / broadcast / barrier issued through the communicator, the bucketed gradient exchange behind a real backward, and
bench.py's scaling-run protocol (`ranks[]`, `n1_value_rank0_alone`).  What a second GPU would add is bytes on xGMI, not code."""
import json
import os
import sys

import pytest

from deeplip_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, argv):
    lines = []
    rc = launch.self_launch(os.path.join(ROOT, script), argv, 1, relay=lines.append)
    js = [json.loads(l) for l in lines if l.lstrip().startswith("{")]
    return rc, js, lines


def test_dist_api_on_rccl_one_rank():
    rc, js, lines = _run("tests/rccl_rank.py", [])
    assert rc == 0, "".join(lines)[-2000:]
    r = js[-1]
    assert r["backend"] == "nccl" and r["active"] and r["world"] == 1
    assert r["gather_ok"] and r["score_err"] < 1e-6 and r["metrics"] == [1.0, 2.0]
    assert r["buckets"] >= 2 and r["reduced_elements"] > 0 and r["flat_reduced"] > 0
    assert r["n_grads"][0] == r["n_grads"][1] and r["grad_max_rel_diff"] < 1e-5
    assert abs(r["loss_plain"] - r["loss_bucketed"]) < 1e-5 * abs(r["loss_plain"])


def test_bench_scaling_protocol_on_rccl_one_rank():
    rc, js, lines = _run("bench.py", ["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-configs", "--single-mode", "--no-h2d"])
    assert rc == 0, "".join(lines)[-2000:]
    assert len(js) == 1
    r = js[0]
    assert r["n_gpus"] == 1 and r["value"] > 1000
    rk = r["ranks"]
    assert len(rk) == 1 and rk[0]["rccl_world_size"] == 1 and rk[0]["allgather_us"] > 0
    assert r["n1_value_rank0_alone"] > 1000


def test_bench_dp_training_leg_on_rccl_one_rank(monkeypatch):
    """The leg a scaling run ends with (BASELINE configs[4] on every rank: train_fusion.Trainer's epoch, the head's gradient bucket
    all-reduced INSIDE its recorded step on the capture group) as a one-rank job, with the first-replay verification forced on: the
    line says the recorded step ran, was checked against an eager step from the same state, and what the check measured."""
    monkeypatch.setenv("DLIP_VERIFY_GRAPH", "1")
    rc, js, lines = _run("bench.py", ["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-configs", "--single-mode", "--no-h2d", "--dp-leg"])
    assert rc == 0, "".join(lines)[-2000:]
    c5 = js[-1]["configs"]["C5_fusion_train_step"]
    assert "error" not in c5, c5
    assert c5["ranks"] == 1 and c5["step_mode"] == "graph" and c5["pairs_per_s"] > 1000
    assert c5["verified"]["outputs_rel_err"] <= 1e-6 and c5["verified"]["witness_rel_err"] <= 1e-6


def test_bench_watchdog_prints_the_line_when_the_dp_leg_hangs(monkeypatch):
    """The DP training leg runs LAST and under a watchdog: a rank that hangs in it (here: by a test hook) must not cost the scaling
    record -- after --dp-leg-timeout seconds the line is printed as it stands, with the leg marked, and the job ends with code 0."""
    monkeypatch.setenv("DLIP_BENCH_DP_HANG", "1")
    rc, js, lines = _run("bench.py", ["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-configs", "--single-mode", "--no-h2d", "--dp-leg",
                                      "--dp-leg-timeout", "5"])
    assert rc == 0, "".join(lines)[-2000:]
    assert len(js) == 1 and js[0]["value"] > 1000
    assert "did not finish" in js[0]["configs"]["C5_fusion_train_step"]["error"]


def test_train_audio_dp_on_rccl_one_rank(tmp_path):
    """BASELINE config C5's mechanism (DP training, bucketed all-reduce behind backward) as a job on the real backend."""
    over = ["data.test_speakers=4", "data.test_utt_per_spk=3", "data.trials=200", "data.trial_targets=40", "data.audio_frames=120",
            "data.n_spk=6", "data.utt_per_spk=3", "train.bs=8", "train.epoch=1"]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        rc, js, lines = _run("train_audio.py", ["--mode", "train", "--config", os.path.join(ROOT, "conf/audio_config.yaml"), "--set", *over])
    finally:
        os.chdir(cwd)
    text = "".join(lines)
    assert rc == 0, text[-2000:]
    assert "Epoch 1 loss" in text and "EER" in text


def test_train_fusion_dp_on_rccl_one_rank(tmp_path):
    """BASELINE config C5 itself -- `train_fusion.py --mode train`: the fusion head on frozen encoders, its gradients one flat
    all-reduce (deeplip_amd.dist.allreduce_grads), epoch metrics reduced over ranks, then sharded extraction + ragged gather +
    trial scoring -- as a one-rank job on the real backend."""
    over = ["train.bs=16", "train.epoch=1", "train.steps_per_epoch=2", "data.n_spk=6", "data.utt_per_spk=4", "data.test_speakers=4",
            "data.test_utt_per_spk=3", "data.trials=300", "data.trial_targets=60", "data.video_frames=9", "data.audio_frames=120"]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        rc, js, lines = _run("train_fusion.py", ["--mode", "train", "--config", os.path.join(ROOT, "conf/fusion_config.yaml"), "--set", *over])
    finally:
        os.chdir(cwd)
    text = "".join(lines)
    assert rc == 0, text[-2000:]
    assert "EER" in text


def test_train_video_dp_on_rccl_one_rank(tmp_path):
    """`train_video.py` (full lip-clip model training) as a one-rank job: parameters broadcast, gradients in flat buckets whose
    all-reduces start from autograd hooks while the backward still runs (deeplip_amd.dist.GradBuckets on RCCL's stream)."""
    rc, js, lines = _run("train_video.py", ["--save-path", str(tmp_path / "ck"), "--steps", "2", "--frames", "9"])
    text = "".join(lines)
    assert rc == 0, text[-2000:]
    assert "done:" in text and os.path.exists(tmp_path / "ck" / "1.pt")


def test_train_video_dp_recorded_step_on_rccl_one_rank(tmp_path):
    """The same with --graph-step: the bucket all-reduces are RCCL collectives captured INTO the recorded step graph (launched from
    autograd hooks during the captured backward) and replayed with it."""
    rc, js, lines = _run("train_video.py", ["--save-path", str(tmp_path / "ck"), "--steps", "4", "--frames", "9", "--graph-step"])
    text = "".join(lines)
    assert rc == 0, text[-3000:]
    assert text.count("(replayed)") == 3 and "done:" in text


def test_step_plan_recorded_behind_collectives_on_the_same_stream_survives_the_watchdog():
    """Regression test of round 4's intermittent: "Process group watchdog thread terminated with exception: HIP error: operation
    not permitted on an event last recorded in a capturing stream" (one full-suite run in five, test_bench_scaling_protocol_...).
    The probe issues collectives with the stream it hands to StepPlan current, then records a plan whose capture stays open for
    0.3 s -- three polls of the process group's watchdog.  Recorded on that stream (StepPlan until round 4) the job aborts at the
    first plan; StepPlan records on a private stream now (deeplip_amd/plan.py)."""
    rc, js, lines = _run("tools/probes/capture_race.py", ["--child", "4", "long", "same"])
    text = "".join(lines)
    assert rc == 0, text[-2000:]
    assert "RESULT survived=4 of 4 error=none" in text, text[-2000:]
