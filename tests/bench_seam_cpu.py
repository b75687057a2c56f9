"""CPU stand-in for bench.Seam (tests/test_bench_multi_cpu.py): the N-rank launcher, the weight-blob broadcast and the
max-over-ranks / sum-over-ranks reduction of bench.py run for real over gloo; only the GPU work (run_config) is replaced by a
deterministic fake whose numbers the test can predict.  TEST INFRASTRUCTURE."""
import torch

import bench
from bert_vits2_amd import models, sharding, synth


class CpuSeam(bench.Seam):
    backend = "gloo"

    def device(self, local):
        return torch.device("cpu")

    def init_pg(self, dev, rank, world):
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def load_model(self, hp, rank, dev):
        model = models.from_hparams(hp)
        sd = None
        if rank == 0:
            sd = synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5)
            model.load_state_dict(sd, strict=False)
        t = sharding.distribute_weights(model, dev, src=0)          # real gloo broadcast of the packed blob
        assert model._host_blob is not None and model._host_blob.numel() > 1 << 20
        return model, sd, t

    def run_config(self, num, model, hp, dev, rank, world, steps, warmup, overrides, full_profile=False, solo=False, collective=False, **kw):
        cfg = bench.CONFIGS[num]
        B, T = overrides.get("batch") or cfg["batch"], overrides.get("symbols") or cfg["symbols"]
        _, lengths = bench.make_batch(cfg, B, T, rank)               # the real per-rank shard of the workload
        frames = 3 * sum(lengths)
        digest = float(model._host_blob[256:4096].double().sum())    # every rank must hold rank 0's blob
        return dict(config=num, B=B, T=T, Ty=3 * T, gen_dtype=cfg["dtype"], flow_dtype=cfg["flow"], graph=bool(cfg["graph"]),
                    dt=(0.008 if solo else 0.010) * steps * (1 + rank), steps=steps, audio_per_step=frames * hp.total_upsample / hp.sampling_rate,
                    lengths=lengths,
                    roofline=dict(bound="mfma", kernel="fake", achieved=1.0 + rank, peak=10.0, unit="TFLOP/s", frac=(1.0 + rank) / 10,
                                  avg_launch_us=1.0, traffic=None, blob_digest=digest))

    def sync(self):
        pass

    def device_name(self, dev):
        return "cpu"


seam = CpuSeam()
