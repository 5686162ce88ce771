"""One rank of a torch.distributed.run job on a GPU: everything deeplip_amd.dist does, through the REAL backend (nccl = RCCL).

Started by tests/test_rccl_gpu.py with --nproc-per-node 1 (a GPU box has one card; RCCL refuses two ranks on one device), so the
exchange is trivial in size but not in mechanism: communicator init bound to the device, all_gather_into_tensor /
all_reduce / broadcast enqueued on RCCL's stream, the bucketed gradient exchange overlapped with a real backward of the speech
encoder on the engine.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from deeplip_amd import dist as ddist, scoring, weightgen as wg
    from models.audio_models import tdnn
    from models.audio_models.loss import LMCL
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    rank, world = ddist.init_from_env(dev)
    out = {"rank": rank, "world": world, "backend": dist.get_backend(), "active": ddist.active()}

    # ragged gather of embedding rows + trial-sharded scoring with the HIP scorer
    n_utt = 37
    g = torch.Generator().manual_seed(3)
    table = torch.nn.functional.normalize(torch.randn(n_utt, 64, generator=g)).to(dev)
    lo, hi = ddist.shard_range(n_utt)
    full = ddist.gather_rows(table[lo:hi].contiguous(), n_utt)
    out["gather_ok"] = bool(torch.equal(full, table))
    ia = (torch.arange(101, dtype=torch.int32) % n_utt).to(dev)
    ib = ((torch.arange(101, dtype=torch.int32) * 7) % n_utt).to(dev)
    sc = ddist.score_trials_sharded(lambda a, b: scoring.cosine_scores(full, a, b), ia, ib)
    ref = (table[ia.long()] * table[ib.long()]).sum(1)
    out["score_err"] = float((sc - ref).abs().max())
    out["metrics"] = ddist.allreduce_metrics([1.0, 2.0 + rank], dev)

    # data-parallel training step of the speech encoder: bucketed all-reduce behind backward == the plain gradients at world 1
    import yaml
    opts = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conf", "audio_config.yaml")))["model"]
    model = tdnn.SpeakerEmbNet(opts)
    sd = wg.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, prefix="audio.")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev).train()
    a = opts[opts["arch"]]
    crit = LMCL(a["embedding_dim"], 6, 30.0, 0.2).to(dev)
    ddist.broadcast_params(list(model.parameters()) + list(crit.parameters()))
    x = torch.from_numpy(wg.audio_input(8, a["input_dim"], 120, key="rccl.audio")).to(dev)
    lab = (torch.arange(8) % 6).to(dev)

    def grads(buckets):
        for p in list(model.parameters()) + list(crit.parameters()):
            if buckets is None:
                p.grad = None
        if buckets is not None:
            buckets.zero()
        loss, _ = crit(model(x), lab)
        loss.backward()
        n = buckets.finish() if buckets is not None else 0
        return float(loss.detach()), n, [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]

    l0, _, g0 = grads(None)
    params = [p for p in list(model.parameters()) + list(crit.parameters()) if p.requires_grad]
    bk = ddist.GradBuckets(params, bucket_bytes=256 << 10)
    l1, n_red, g1 = grads(bk)
    out.update(loss_plain=l0, loss_bucketed=l1, buckets=len(bk.buckets), reduced_elements=n_red,
               n_grads=[len(g0), len(g1)],
               grad_max_rel_diff=max(float((u - v).abs().max() / (v.abs().max() + 1e-30)) for u, v in zip(g0, g1)))
    n_flat = ddist.allreduce_grads(params)
    out["flat_reduced"] = n_flat
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
