#!/usr/bin/env python3
"""bench.py — audio-seconds/sec on the SynthesizerTrn.infer() hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full ``infer()`` (phase A + the reference's one host sync + phase B) over one batch of synthetic
utterances already resident in HBM.  N=1 workload = BASELINE config 2: B=1, T=128 symbols, fp32, durations pinned to
3 frames/symbol (T_y=384, 196 608 samples = 4.458 s of 44.1 kHz audio).  N>1: utterances are independent units, so each
rank runs the same per-GPU workload on its own utterance (weak scaling, no data-path collective); the only collective is
the one-time RCCL broadcast of the packed weight blob from rank 0, timed separately.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bert_vits2_amd import hparams as H, models, sharding, synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix = fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 MFMA
PEAK_HBM_GBPS = 8000.0             # same guide: HBM3E
CONFIGS = {2: dict(batch=1, symbols=128, dtype="f32", flow="f32"), 3: dict(batch=32, symbols=128, dtype="bf16", flow="f16", graph=1),
           5: dict(batch=8, symbols=512, dtype="bf16", flow="f16")}
GEN_FLOP_PER_FRAME = 651.6e6       # Generator algorithmic FLOPs per latent frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="BASELINE.json config: 2 = B=1 T=128 fp32 (the metric's config, default), 3 = B=32 T=128 bf16 Generator, "
                         "5 = long-form B=8 T=512")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (overrides the config's)")
    ap.add_argument("--symbols", type=int, default=None, help="symbols per utterance (overrides the config's)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None, help="Generator arithmetic (overrides the config's)")
    ap.add_argument("--flow-dtype", choices=("f32", "f16"), default=None, help="transformer-flow conv arithmetic (overrides the config's)")
    ap.add_argument("--graph", type=int, default=None, choices=(0, 1),
                    help="replay each phase as a captured hipGraph (default: the config's setting)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--full-profile", action="store_true", help="extra untimed pass timing every MFMA kernel family")
    return ap.parse_args()


def log(msg):
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


def usable_cores() -> int:
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores even inside a quota-limited container, and oversubscribed OpenMP teams are catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(hp, sd, batch, kw, iters, budget_s=45.0):
    """The oracle restatement (same aten CPU kernels and per-call weight_norm fold as the reference's infer) timed on
    this box's host cores — a reported baseline, kind "port" (the Python reference cannot travel to the GPU box).
    Bounded: stops after ``iters`` timed runs or ``budget_s`` seconds, whichever comes first."""
    from oracle import bv2_oracle as O
    nthreads = min(usable_cores(), 64)
    torch.set_num_threads(nthreads)
    B, T = batch["x"].shape
    nw, nz = synth.synthetic_noise(B, T, 3 * T + 8, hp.inter_channels)
    run = lambda: O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"],
                          batch["bert"], batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    t_start = time.perf_counter()
    out = run()                                     # warm-up
    log(f"cpu baseline warm-up {time.perf_counter() - t_start:.2f}s on {nthreads} threads")
    ts = []
    while len(ts) < iters and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        out = run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    audio_s = float(out["y_lengths"].sum()) * hp.total_upsample / hp.sampling_rate
    return dict(value=round(audio_s / med, 3), unit="audio-seconds/sec", cores=nthreads, kind="port",
                sample=f"{len(ts)} timed runs (median) of the same workload (B={B}, T={T}, T_y={int(out['y_lengths'].max())}, "
                       f"{audio_s:.3f} s audio) after 1 warm-up, torch CPU fp32", ms_per_step=round(med * 1e3, 2))


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes of tools/collect_traffic.py (rocprofv3 cannot wrap the timed run
    itself without perturbing it, so the counters come from separate passes of this same command, committed under
    profiles/): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE.  None when no such profile exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"].get(kernel)
            if k:
                return dict(bytes_per_launch=round(k["traffic_bytes"]), fetch_bytes=round(k["fetch_bytes"]),
                            write_bytes=round(k["write_bytes"]), source=os.path.relpath(path, ROOT))
        except Exception:
            continue
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback for the product path"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)     # RCCL

    hp = H.default_v23()
    cfgd = CONFIGS[args.config]
    B = args.batch if args.batch is not None else cfgd["batch"]
    T = args.symbols if args.symbols is not None else cfgd["symbols"]
    gen_dtype = args.dtype or cfgd["dtype"]
    flow_dtype = args.flow_dtype or cfgd["flow"]
    kw = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=0.0, length_scale=1.0)

    # ---- weights: rank 0 folds/packs the seeded synthetic checkpoint, every other rank receives the blob over RCCL
    model = models.from_hparams(hp)
    sd = None
    if rank == 0:
        sd = synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5)
        model.load_state_dict(sd, strict=False)
    log(f"rank {rank}/{world}: packing / distributing weights")
    t_bcast = sharding.distribute_weights(model, dev, src=0)
    log("weights attached")
    if gen_dtype == "bf16":
        model.set_generator_dtype(torch.bfloat16)
    if flow_dtype == "f16":
        model.set_flow_dtype(torch.float16)
    use_graph = bool(cfgd.get("graph", 0)) if args.graph is None else bool(args.graph)

    # ---- this rank's utterances (weak scaling: same per-GPU work, different utterances)
    batch = synth.synthetic_batch([T] * B, first_index=rank * B)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    call = lambda: model.infer(dbatch["x"], dbatch["x_lengths"], dbatch["sid"], dbatch["tone"], dbatch["language"],
                               dbatch["bert"], dbatch["ja_bert"], dbatch["en_bert"], **kw)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    model.enable_graphs(use_graph)
    for i in range(args.warmup):
        out = call()
        if i == 0:
            torch.cuda.synchronize()
            log("first infer() done")
    model.profile(0 if use_graph else 2)   # HIP events around the Generator's kernel launches only (dominant family)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    prof_steps = min(args.steps, 150)              # the event pool holds 8192 launches (~40 Generator launches per step)
    for i in range(args.steps):
        if i == prof_steps and not use_graph:
            model.profile_pause()
        o, attn, y_mask, _rest = call()
        frames += y_mask.shape[2] * B              # pinned durations: every frame of every utterance is valid
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f}s for {args.steps} steps")
    prof = model.profile_report() if not use_graph else None
    model.profile(0)
    Ty = y_mask.shape[2]
    if use_graph:
        # events cannot be recorded inside a captured graph: the roofline leg times the same launches in an eager pass
        # of the same K steps right after the timed (graph-replayed) region
        model.enable_graphs(False)
        call()
        model.profile(2)
        torch.cuda.synchronize()
        for _ in range(prof_steps):
            call()
        torch.cuda.synchronize()
        prof = model.profile_report()
        model.profile(0)

    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ftot = torch.tensor([frames], dtype=torch.float64, device=dev)
        dist.all_reduce(ftot, op=dist.ReduceOp.SUM)
        frames = float(ftot.item())
    audio_s = frames * hp.total_upsample / hp.sampling_rate
    value = audio_s / dt

    if rank == 0:
        # ---- roofline of the dominant kernel family (the MFMA implicit-GEMM conv that runs the Generator)
        roof = None
        if prof:
            dom = max(prof, key=lambda r: r["total_ms"])
            psteps = prof_steps                      # steps whose launches were event-timed
            gen_ms = sum(r["total_ms"] for r in prof) / psteps
            # the roof that binds the dominant kernel: its layer-wise arithmetic intensity against the machine balance
            peak_tf = PEAK_BF16_MFMA_TFLOPS if "bf16" in dom["name"] else PEAK_FP32_MFMA_TFLOPS
            ai = dom["flops"] / max(dom["bytes"], 1.0)
            secs = dom["total_ms"] * 1e-3
            if ai >= peak_tf * 1e12 / (PEAK_HBM_GBPS * 1e9):
                ach, peak, unit, bound = dom["flops"] / secs / 1e12, peak_tf, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = dom["bytes"] / secs / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
            roof = dict(bound=bound, kernel=dom["name"], achieved=round(ach, 3), peak=peak, unit=unit,
                        frac=round(ach / peak, 4), arithmetic_intensity_flop_per_byte=round(ai, 1),
                        traffic=(pmc_traffic(dom["name"]) or {}).get("bytes_per_launch"),
                        traffic_detail=pmc_traffic(dom["name"]),
                        alg_bytes_per_launch=round(dom["bytes"] / dom["launches"]),
                        launches_per_step=dom["launches"] / psteps,
                        avg_launch_us=round(dom["total_ms"] * 1e3 / dom["launches"], 2),
                        flops_per_launch=dom["flops"] / dom["launches"],
                        generator_ms_per_step=round(gen_ms, 4),
                        generator_tflops=round(sum(r["flops"] for r in prof) / psteps / (gen_ms * 1e-3) / 1e12, 3),
                        families=[dict(name=r["name"], launches=r["launches"] / psteps,
                                       ms_per_step=round(r["total_ms"] / psteps, 4),
                                       tflops=round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 3),
                                       alg_GBps=round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)) for r in prof])
        full = None
        if args.full_profile:
            model.profile(3)               # one row per launch site and shape (untimed pass; events perturb the run)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            full = [dict(name=r["name"], launches=r["launches"] / 3, ms_per_step=round(r["total_ms"] / 3, 4),
                         us_per_launch=round(r["total_ms"] * 1e3 / r["launches"], 2),
                         tflops=round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 3)) for r in model.profile_report()]
            model.profile(0)
            for r in sorted(full, key=lambda r: -r["ms_per_step"]):
                log(f"  {r['ms_per_step']:8.4f} ms/step  {r['launches']:5.0f} x {r['us_per_launch']:8.2f} us  "
                    f"{r['tflops']:7.2f} TF  {r['name']}")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle port)")
            cpu = cpu_baseline(hp, sd, batch, kw, args.cpu_iters)
        line = dict(
            metric="audio-seconds/sec (44.1 kHz), SynthesizerTrn.infer(), 128-phoneme utterance", value=round(value, 2),
            unit="audio-seconds/sec", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype=gen_dtype if gen_dtype == "f32" or flow_dtype == "f32" else "bf16+f16", data="synthetic",
            config=dict(workload=f"BASELINE config {args.config}: B={B} x T={T} symbols per GPU, "
                                 f"{'bf16 Generator (fp32 accumulate)' if gen_dtype == 'bf16' else 'fp32 Generator'}, "
                                 f"{'fp16 flow convs (fp32 accumulate / LayerNorm / softmax)' if flow_dtype == 'f16' else 'fp32 flow'}, "
                                 f"fp32 text encoder / durations / spline, T_y={Ty} frames "
                                 f"({Ty * hp.total_upsample} samples, {Ty * hp.total_upsample / hp.sampling_rate:.3f} s) per utterance, "
                                 f"transformer flow, synthetic seeded weights, durations pinned to 3 frames/symbol",
                        utterances_per_gpu=B, symbols=T, frames=Ty, parallelism=f"utterance-sharded x{world}",
                        rtf=round(dt / audio_s, 6), x_realtime_per_gpu=round(value / world, 2),
                        hipgraph=use_graph, weight_broadcast_ms=round(t_bcast * 1e3, 3)),
            roofline=roof, cpu_baseline=cpu)
        if full:
            line["kernel_families_untimed_pass"] = full
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
