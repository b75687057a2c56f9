#!/usr/bin/env python3
"""bench.py — audio-seconds/sec on the SynthesizerTrn.infer() hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: bench.py spawns its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full ``infer()`` (phase A + the reference's one host sync + phase B) over one batch of synthetic utterances already
resident in HBM.  Workloads (BASELINE.json ``configs``):

  2  B=1, T=128 symbols, fp32, durations pinned to 3 frames/symbol (T_y=384, 196 608 samples = 4.458 s) — the config the metric is
     quoted on: the N=1 default, and ``value`` of the line
  3  B=32 x T=128, bf16 Generator + fp16 flow convolutions, hipGraph replay
  4  per GPU B=32 utterances of uniform [96,128] symbols, ZH/JA/EN round-robin, random speakers, precisions of config 3 — the N>1
     default (BASELINE: "8xMI355X utterance-sharded throughput"): utterances are independent, each rank runs its own shard, NO
     data-path collective; the only collective is the one-time RCCL broadcast of the packed weight blob, timed separately
  5  B=8 x T=512 (T_y=1536), bf16 Generator + fp16 flow

At N=1 the line also carries ``secondary``: configs 3, 4 (one GPU's share) and 5 measured in the same process right after the primary,
each with its own roofline block, so the driver-run line documents every BASELINE configuration.  Nothing is profiled inside a timed
region: the per-kernel HIP-event pass (roofline) is a separate eager pass after it.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import re
import os
import sys
import time

# Kernel arguments in DEVICE memory: the ROCm 7 default on this part, stated here because the batch-1 step is ~190 dependent launches
# whose first instruction is a scalar load from the kernarg segment — with host-resident kernargs (=0) the same step takes 4.29 ms
# instead of 3.64 (profiles/r05_hip_force_dev_kernarg.txt).  Must be in the environment before the HIP runtime loads.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bert_vits2_amd import hparams as H, models, sharding, synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix = fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 MFMA
PEAK_HBM_GBPS = 8000.0             # same guide: HBM3E
CONFIGS = {
    2: dict(batch=1, symbols=128, dtype="f32", flow="f32", graph=0, ragged=False),
    3: dict(batch=32, symbols=128, dtype="bf16", flow="f16", graph=1, ragged=False),
    4: dict(batch=32, symbols=128, dtype="bf16", flow="f16", graph=1, ragged=True),
    5: dict(batch=8, symbols=512, dtype="bf16", flow="f16", graph=0, ragged=False),
}
WN_FLOW_B32 = "f16"               # flow arithmetic of secondary.config3_residual_flow: the WN convolutions on the fp16 matrix core
KW = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=0.0, length_scale=1.0)
KERNEL_SOURCES = {"conv1d_x6": "conv_x6.hip", "conv1d_x3": "conv_x6.hip", "respair_x6": "respair_x6.hip", "respair_x3": "respair_x6.hip", "conv1d_mfma": "conv_mfma.hip", "conv1d_splitk": "conv_mfma.hip", "resblock_fused": "resblock_fused.hip",
                  "conv_cl_bf16": "gen_bf16.hip", "respair_cl_bf16": "respair_cl_bf16.hip", "resblock_cl_bf16": "resblock_cl_bf16.hip", "resblock_c16_bf16": "resblock_c16_bf16.hip", "conv_f16": "enc_f16.hip",
                  "attention": "attention.hip"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4, 5),
                    help="BASELINE.json config (default: 2 at --gpus 1, 4 at --gpus N>1)")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (overrides the config's)")
    ap.add_argument("--symbols", type=int, default=None, help="symbols per utterance (overrides the config's)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None, help="Generator arithmetic (overrides the config's)")
    ap.add_argument("--flow-dtype", choices=("f32", "f16"), default=None, help="transformer-flow conv arithmetic (overrides the config's)")
    ap.add_argument("--graph", type=int, default=None, choices=(0, 1), help="replay each phase as a captured hipGraph")
    ap.add_argument("--residual-flow", action="store_true",
                    help="the ResidualCouplingBlock / WN flow (use_transformer_flow=false) instead of the transformer flow (fp32 unless --flow-dtype f16)")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the secondary configs 3 / 4 / 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--option", action="append", default=[], metavar="KEY=0|1",
                    help="bv2_set_option switch for A/B runs (fused_attn_o, overlap_dp, fused_dds, fused_resblock)")
    ap.add_argument("--variants", default=None, metavar="SPEC,CL,HC",
                    help="bv2_test_set_variants(spec, cl_generic, hc_generic) before the run (tuning A/B only)")
    ap.add_argument("--streams", type=int, default=2, help="requests in flight for the secondary two-stream leg of config 2")
    ap.add_argument("--full-profile", action="store_true", help="extra untimed pass timing every MFMA kernel launch site")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the primary timed region is run this many times back to back; `value` is the FIRST (the contract's K steps after W "
                         "warm-ups), the others are reported as its spread")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path at whatever world size there is — at --gpus 1: a world-size-1 RCCL process group on the "
                         "device, the blob broadcast, both barriers, all_reduce(MAX), all_gather_object and the N>1 line builder — so "
                         "that the multi-GPU path executes on the one GPU a builder has")
    ap.add_argument("--master-port", type=int, default=0, help="self-launched N>1 runs: rendezvous port on 127.0.0.1 (0 = pick a free one)")
    ap.add_argument("--library", default=None, metavar="PATH",
                    help="same-box A/B of two BUILDS (tools/ab_build.py): load this libbv2 (same C ABI) instead of the in-tree build")
    ap.add_argument("--details-out", default=None, metavar="PATH",
                    help="where the full record (launch tables, PMC families, every secondary leg in full) is written; default "
                         "gpurun_out/bench_details.json.  stdout carries ONE compact (< 4 KB) strict-JSON line")
    ap.add_argument("--seam", default=None, help=argparse.SUPPRESS)   # tests only: module replacing the GPU-touching seam (see Seam)
    return ap.parse_args(argv)


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


def reference_container():
    """The REAL reference timed on the build container beside the oracle port (oracle/time_reference.py; the Python reference cannot
    travel to the GPU box): the newest committed profiles/*reference_container*.json, or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*reference_container*.json")), reverse=True):
        try:
            d = json.load(open(path))
            rng = list(d.get("reference_over_port_range") or [d["reference_over_port"]] * 2)
            hist = [v for v in (d.get("history") or {}).values() if isinstance(v, (int, float))]
            return dict(ms=d["reference"]["ms"], audio_s_per_s=d["reference"]["audio_s_per_s"], threads=d["threads"], cores=d["cores_usable"],
                        torch=d["torch"], port_ms_same_box=d["port"]["ms"], reference_over_port=d["reference_over_port"],
                        reference_over_port_range=rng, sessions=d.get("sessions", 1),
                        reference_over_port_all_measurements=[min(rng + hist), max(rng + hist)],
                        waveform_rms_reference_vs_port=d["waveform_rms_reference_vs_port"], source=os.path.relpath(path, ROOT),
                        note="the reference's own SynthesizerTrn.infer (models.py:1026-1074, unmodified) on the BUILD CONTAINER's host cores, "
                             "same utterance / checkpoint / noise as the port figure beside it; measured by oracle/time_reference.py")
        except Exception:
            continue
    return None


def cpu_baseline(hp, sd, iters, budget_s=40.0):
    """The oracle restatement (same aten CPU kernels and per-call weight_norm fold as the reference's infer) timed on this box's
    host cores — a reported baseline, kind "port" (the Python reference cannot travel to the GPU box; `reference_container` beside
    it is the real reference timed on the build container with the port's time on that same machine, so the ratio is known).
    Config 2's utterance on all usable threads (2 warm-ups, median of up to ``iters`` runs), plus config 1's shape (T=64) and a
    single-thread figure, all inside ``budget_s`` seconds.  Also returns (batch, noise_w, noise_z, oracle output) of config 2's
    utterance for the parity block."""
    from oracle import bv2_oracle as O
    nthreads = min(usable_cores(), 64)
    t_begin = time.perf_counter()

    def timed(T, threads, warm, n):
        torch.set_num_threads(threads)
        batch = synth.synthetic_batch([T])
        nw, nz = synth.synthetic_noise(1, T, 3 * T + 8, hp.inter_channels)
        run = lambda: O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                              batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **KW)
        for _ in range(warm):
            out = run()
        ts = []
        while len(ts) < n and (not ts or time.perf_counter() - t_begin < budget_s):
            t0 = time.perf_counter()
            out = run()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        audio = float(out["y_lengths"].sum()) * hp.total_upsample / hp.sampling_rate
        return audio / med, med, len(ts), audio, int(out["y_lengths"].max()), (batch, nw, nz, out)

    v2, med2, n2, audio2, ty2, par = timed(128, nthreads, 2, iters)
    log(f"cpu baseline: config 2 on {nthreads} threads {v2:.2f} audio-s/s ({med2 * 1e3:.1f} ms)")
    v1, med1, n1, _, _, _ = timed(64, nthreads, 1, 3)
    vs, meds, ns = (None, None, 0)
    if time.perf_counter() - t_begin < budget_s - 8:
        vs, meds, ns, _, _, _ = timed(128, 1, 0, 1)
    torch.set_num_threads(nthreads)
    rc = reference_container()
    est = None if rc is None else round(v2 / rc["reference_over_port"], 3)
    est_rng = None if rc is None else [round(v2 / rc["reference_over_port_all_measurements"][1], 3),
                                       round(v2 / rc["reference_over_port_all_measurements"][0], 3)]
    return dict(value=round(v2, 3), unit="audio-seconds/sec", cores=nthreads, kind="port",
                sample=f"median of {n2} timed runs of config 2's utterance (B=1, T=128, T_y={ty2}, {audio2:.3f} s audio) after 2 warm-ups, "
                       f"torch CPU fp32, oracle restatement",
                ms_per_step=round(med2 * 1e3, 2),
                config1_T64=dict(value=round(v1, 3), ms_per_step=round(med1 * 1e3, 2), runs=n1),
                single_thread=None if vs is None else dict(value=round(vs, 3), ms_per_step=round(meds * 1e3, 2), cores=1, runs=ns),
                reference_container=rc,
                reference_estimate_this_box=None if est is None else dict(
                    value=est, range=est_rng, unit="audio-seconds/sec",
                    note="this box's port figure / reference_container.reference_over_port: what the reference's own infer() would "
                         "reach on these host cores if the container's reference/port ratio carries over.  `range` spans every ratio "
                         "measured so far (interleaved sessions of the current file and the sequential figures of rounds 3-4): the ratio "
                         "moves with host load, so the estimate is an interval, not a number")), par


def parity_block(model, hp, dev, par):
    """Parity of the measured configuration, inside the measured line: config 2's utterance (the one the CPU leg just ran through
    the oracle) through the HIP path with the SAME injected noise; waveform RMS error (north_star: <= 1e-3), mel-spectrogram L1
    with the reference's definition (mel_processing.py:95-142, rebuilt in oracle/mel.py), exact-match rate of the ceil'd durations."""
    from oracle import mel
    batch, nw, nz, ref = par
    model.enable_graphs(False)
    model.set_generator_dtype(torch.float32)
    model.set_flow_dtype(torch.float32)                  # both flow variants have an fp16 form: the block labelled fp32 is fp32
    b = {k: v.to(dev) for k, v in batch.items()}
    o, attn, y_mask, _ = model.infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"],
                                     noise_w=nw, noise_z=nz.to(dev), **KW)
    torch.cuda.synchronize()
    o = o.cpu()
    wc = model.last_encode["w_ceil"].cpu().reshape(ref["w_ceil"].shape)
    match = float((wc == ref["w_ceil"]).float().mean())
    res = dict(config="BASELINE config 2's utterance (B=1, T=128, fp32, pinned durations), HIP path vs the CPU oracle, same injected noise",
               w_ceil_match=match, frames=int(y_mask.sum().item()))
    if o.shape == ref["o"].shape:
        n = int(ref["y_lengths"][0]) * hp.total_upsample
        d = (o[0, 0, :n] - ref["o"][0, 0, :n]).double()
        res.update(wave_rms=float(d.pow(2).mean().sqrt()), wave_max_abs=float(d.abs().max()),
                   signal_rms=float(ref["o"][0, 0, :n].double().pow(2).mean().sqrt()),
                   mel_l1=float(mel.mel_l1(o[:, 0].numpy(), ref["o"][:, 0].numpy(), [n])),
                   attn_equal=bool(torch.equal(attn.cpu(), ref["attn"])), bar="wave_rms <= 1e-3 (north_star)")
    else:
        res["error"] = f"shape mismatch {tuple(o.shape)} vs {tuple(ref['o'].shape)}"
    return res


def parity_unpinned_block(hp, dev, T=128):
    """The parity check that CAN fail on the integer outputs: an un-pinned checkpoint (dp.proj as drawn, not zeroed), sdp_ratio 0.5 (both
    duration predictors, the spline flows), NO w_ceil injection — the HIP path's own ceil'd durations / frame count / alignment path against the
    oracle's on a 128-symbol utterance.  (parity_block above runs the MEASURED utterance, whose durations are pinned to ceil(2.5) so that T_y is the
    same on every box: its w_ceil_match / attn_equal cannot fail.)  If a 1-ulp logw difference flips a ceil(), the downstream comparison is repeated
    with the oracle's durations and says so."""
    from oracle import bv2_oracle as O, mel
    kw = dict(KW, sdp_ratio=0.5)
    sd = synth.synthetic_state_dict(hp, seed=0)
    batch = synth.synthetic_batch([T], languages=[0], sids=[3])
    nw, nz = synth.synthetic_noise(1, T, 8 * T, hp.inter_channels)
    with torch.no_grad():
        ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"], batch["ja_bert"],
                      batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    b = {k: v.to(dev) for k, v in batch.items()}
    args = (b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"])
    o, attn, y_mask, _ = m.infer(*args, noise_w=nw, noise_z=nz.to(dev), **kw)
    torch.cuda.synchronize()
    wc = m.last_encode["w_ceil"].cpu().reshape(ref["w_ceil"].shape)
    match = float((wc == ref["w_ceil"]).float().mean())
    res = dict(config=f"un-pinned synthetic checkpoint, B=1, T={T}, sdp_ratio 0.5, fp32, no duration injection: HIP path vs the CPU oracle",
               w_ceil_match=match, distinct_durations=int(ref["w_ceil"].unique().numel()), frames=int(y_mask.sum().item()),
               frames_oracle=int(ref["y_lengths"].sum()), pinned_after_flip=False)
    if match < 1.0:
        o, attn, y_mask, _ = m.infer(*args, noise_w=nw, noise_z=nz.to(dev), w_ceil=ref["w_ceil"], **kw)
        torch.cuda.synchronize()
        res["pinned_after_flip"] = True
    o = o.cpu()
    if o.shape == ref["o"].shape:
        n = int(ref["y_lengths"][0]) * hp.total_upsample
        d = (o[0, 0, :n] - ref["o"][0, 0, :n]).double()
        res.update(wave_rms=float(d.pow(2).mean().sqrt()), signal_rms=float(ref["o"][0, 0, :n].double().pow(2).mean().sqrt()),
                   mel_l1=float(mel.mel_l1(o[:, 0].numpy(), ref["o"][:, 0].numpy(), [n])),
                   attn_equal=bool(torch.equal(attn.cpu(), ref["attn"])))
    else:
        res["error"] = f"shape mismatch {tuple(o.shape)} vs {tuple(ref['o'].shape)}"
    del m
    return res


def run_config3_unpinned(hp, dev, n_cold=20, n_steady=40):
    """BASELINE config 3 with REAL durations: the same shape (B = 32 x 128 symbols, bf16 Generator + fp16 flow, hipGraph replay) on an UN-pinned
    checkpoint at sdp_ratio 0.5, every step a batch nobody has seen — so T_y = max(y_lengths) moves from step to step and the captured decode
    lives or dies by its T_y buckets (models.enable_graphs ty_bucket, 32 frames).  `cold`: the first n_cold distinct batches, captures inside
    the clock.  `steady`: n_steady further distinct batches.  `eager`: those same n_steady batches with graphs off.  value = VALID audio seconds
    (sum of y_lengths) per second; padded_value counts B x T_y (what the pinned config 3 figure counts, where every utterance is T_y long)."""
    # infer() draws its duration noise from torch's generators: seeded, so that the T_y sequence — which buckets the steady pass meets, and
    # whether one of them was never captured in the cold pass — is the same in every run of this leg (an unseeded run once met a fourth
    # bucket inside the steady pass: one capture in the clock, 18.8 instead of 13.4 ms per step)
    torch.manual_seed(20260930)
    sd = synth.synthetic_state_dict(hp, seed=0)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    m.set_generator_dtype(torch.bfloat16)
    m.set_flow_dtype(torch.float16)
    B, T = CONFIGS[3]["batch"], CONFIGS[3]["symbols"]
    kw = dict(KW, sdp_ratio=0.5)
    n = n_cold + n_steady
    # one speaker, as config 3 itself: the synthetic speaker table is N(0,1), which moves a SPEAKER's mean duration by 2x and more — nothing a
    # trained table does; within one speaker the per-utterance spread (min / mean / max of y_lengths about 0.88 / 1 / 1.2) is what a
    # length-bucketed serving batch looks like
    batches = [{k: v.to(dev) for k, v in synth.synthetic_batch([T] * B, first_index=5000 + i * B).items()} for i in range(n)]
    call = lambda b: m.infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"], want_attn=False, **kw)

    def timed(bs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames, padded, tys = 0, 0, []
        outs = [call(b)[2] for b in bs]                     # y_mask views: lengths summed after the clock stops
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for ym in outs:
            frames += int(ym.sum().item())
            padded += B * ym.shape[2]
            tys.append(int(ym.shape[2]))
        return dt, frames, padded, tys

    m.enable_graphs(False)
    call(batches[0])                                        # workspace + allocator warm-up, eager
    m.enable_graphs(True, static_io=True)
    dt_c, fr_c, _, tys_c = timed(batches[:n_cold])
    st_c = dict(m.graph_stats)
    dt_s, fr_s, pad_s, tys_s = timed(batches[n_cold:])
    st_s = {k: m.graph_stats[k] - st_c[k] for k in st_c}
    m.enable_graphs(False)
    dt_e, fr_e, _, _ = timed(batches[n_cold:])
    sec = lambda fr: fr * hp.total_upsample / hp.sampling_rate
    hit = lambda st: round(st["replays"] / max(1, st["replays"] + st["captures"]), 4)
    res = dict(workload=f"config 3's shape (B={B} x {T} symbols, bf16 Generator, fp16 flow, hipGraph + {m._ty_bucket}-frame T_y buckets) on an UN-pinned "
                        f"checkpoint, sdp_ratio 0.5, every step a new batch", value=round(sec(fr_s) / dt_s, 2), unit="audio-seconds/sec",
               ms_per_step=round(dt_s / n_steady * 1e3, 4), padded_value=round(sec(pad_s) / dt_s, 2), steps=n_steady,
               distinct_ty=len(set(tys_s)), ty_min=min(tys_s), ty_max=max(tys_s), distinct_buckets=len({(t + 31) // 32 for t in tys_s}),
               graph_hit_rate=hit(st_s), captures=st_s["captures"],
               cold=dict(steps=n_cold, value=round(sec(fr_c) / dt_c, 2), ms_per_step=round(dt_c / n_cold * 1e3, 4), graph_hit_rate=hit(st_c),
                         captures=st_c["captures"], distinct_ty=len(set(tys_c))),
               eager=dict(value=round(sec(fr_e) / dt_e, 2), ms_per_step=round(dt_e / n_steady * 1e3, 4)),
               replay_over_eager=round(dt_e / dt_s, 4))
    del m
    return res


def file_digest(name):
    with open(os.path.join(ROOT, "bert-vits2_amd", "csrc", "kernels", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(kernel, config=2):
    """HBM bytes per launch of `kernel` from the PMC passes of tools/collect_traffic.py (rocprofv3 cannot wrap the timed run itself
    without perturbing it, so the counters come from separate passes of this same command, committed under profiles/): FETCH_SIZE
    x2 (gfx950 correction) + WRITE_SIZE.  Only a profile taken with the CURRENT source of that kernel counts: the traffic file
    records the digest of every kernel source; a stale one is refused (returns a note instead of numbers)."""
    src = next((v for k, v in KERNEL_SOURCES.items() if kernel.startswith(k)), None)
    want = file_digest(src) if src else None
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
            k = d["kernels"].get(kernel)
            if not k and kernel.startswith("conv1d_splitk"):
                # the launcher names the split-K variants (<32x32,8w>, _ldsx<...>); the PMC trace has ONE family for the kernel symbol
                k = d["kernels"].get("conv1d_splitk<32x32>")
            if not k:
                continue
            # a family's launches differ between configs (conv1d_mfma<64x64> is the Generator at config 2 and the text encoder's
            # FFN at config 3): only a profile of THIS config's command counts (configs 4 / 5 fall back to config 3's: same kernels,
            # same per-launch shapes up to the batch's padding)
            ba = d.get("_bench_args") or []
            fcfg = int(ba[ba.index("--config") + 1]) if "--config" in ba else 2
            if fcfg != (2 if config == 2 else 3):
                continue
            have = (d.get("source_digests") or {}).get(src)
            if want is None or have != want:
                stale = stale or os.path.relpath(path, ROOT)
                continue
            return dict(bytes_per_launch=round(k["traffic_bytes"]), fetch_bytes=round(k["fetch_bytes"]),
                        write_bytes=round(k["write_bytes"]), source=os.path.relpath(path, ROOT), kernel_source_digest=have)
        except Exception:
            continue
    return dict(bytes_per_launch=None, note=f"no PMC profile of the current {src} under profiles/" + (f" (newest stale: {stale})" if stale else ""))


def make_batch(cfg, B, T, rank):
    """This rank's utterances (weak scaling: same per-GPU work, different utterances).  Ragged (config 4): lengths uniform in
    [96, T], languages ZH/JA/EN round-robin, speakers uniform over the table — seeded per rank."""
    if not cfg["ragged"]:
        return synth.synthetic_batch([T] * B, first_index=rank * B), [T] * B
    g = torch.Generator().manual_seed(977 + rank)
    lengths = [T] + torch.randint(96, T + 1, (B - 1,), generator=g).tolist()      # one full-length utterance fixes the padded shape
    langs = [(rank * B + i) % 3 for i in range(B)]
    sids = torch.randint(0, 850, (B,), generator=g).tolist()
    return synth.synthetic_batch(lengths, langs, sids, first_index=rank * B), lengths


def roofline_block(prof, psteps, config=2):
    dom = max(prof, key=lambda r: r["total_ms"])
    gen_ms = sum(r["total_ms"] for r in prof) / psteps
    # the roof that binds the dominant kernel: its layer-wise arithmetic intensity against the machine balance
    peak_tf = PEAK_BF16_MFMA_TFLOPS if ("bf16" in dom["name"] or "f16" in dom["name"]) else PEAK_FP32_MFMA_TFLOPS
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    secs = dom["total_ms"] * 1e-3
    if ai >= peak_tf * 1e12 / (PEAK_HBM_GBPS * 1e9):
        ach, peak, unit, bound = dom["flops"] / secs / 1e12, peak_tf, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = dom["bytes"] / secs / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
    tr = pmc_traffic(dom["name"], config)
    x6 = {}
    if (dom["name"].startswith("conv1d_x6") or dom["name"].startswith("respair_x6")) and bound == "mfma":
        # kernels/conv_x6.hip computes the fp32 conv with SIX bf16 MFMA products per multiply-add (exact three-way bf16 splits of both
        # operands): the roof that binds it is the bf16 matrix core, and what it must issue per launch is 6x the conv's FLOPs.
        # `achieved` / `frac` are in those issued bf16 FLOPs against the 2.5 PF bf16 peak; the fp32-equivalent rate (the conv's own
        # FLOPs per second) and its ratio to the fp32 matrix peak the previous kernel was bounded by are reported beside it.
        x6 = dict(math="fp32 operands / fp32 results; every product formed on the bf16 matrix core from exact 3-way bf16 splits, 6 of the "
                       "9 cross terms (dropped terms < 2^-23 of a product): 6 issued bf16 MFMA FLOPs per algorithmic FLOP",
                  issued_flops_per_alg_flop=6, fp32_equivalent_tflops=round(ach, 3),
                  frac_of_fp32_mfma_peak=round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                  measured_mfma_ceiling="tools/probe/mfma_bf16_probe.hip on this chip: 0.75-0.80 of 2.5 PF with operands in registers "
                                        "(clock drops to ~1.8 GHz under dense bf16 MFMA), 0.60-0.65 with this kernel's operand traffic "
                                        "(profiles/r03_mfma_bf16_probe.txt; round 4, full-range random operands: 0.68-0.73 in registers, "
                                        "0.57-0.64 with this operand traffic, profiles/r04_mfma_bf16_probe.txt)")
        ach, peak = 6.0 * ach, PEAK_BF16_MFMA_TFLOPS
    elif (dom["name"].startswith("conv1d_x3") or dom["name"].startswith("respair_x3")) and bound == "mfma":
        # the same kernel's two-plane fp16 form (bv2_kernels.h "x3"): THREE fp16 MFMA products per multiply-add (the operands scaled by
        # powers of two into fp16's range and split into two fp16 halves; dropped term <= 2^-24 of a product).  Same accounting: issued
        # fp16 FLOPs against the 2.5 PF fp16 / bf16 matrix peak.
        x6 = dict(math="fp32 operands / fp32 results; every product formed on the fp16 matrix core from 2-way fp16 splits of the scaled "
                       "operands, 3 of the 4 cross terms (dropped term <= 2^-24 of a product): 3 issued fp16 MFMA FLOPs per algorithmic FLOP",
                  issued_flops_per_alg_flop=3, fp32_equivalent_tflops=round(ach, 3),
                  frac_of_fp32_mfma_peak=round(ach / PEAK_FP32_MFMA_TFLOPS, 4))
        ach, peak = 3.0 * ach, PEAK_BF16_MFMA_TFLOPS
    return dict(bound=bound, kernel=dom["name"], achieved=round(ach, 3), peak=peak, unit=unit, frac=round(ach / peak, 4), **x6,
                arithmetic_intensity_flop_per_byte=round(ai, 1), traffic=tr.get("bytes_per_launch"), traffic_detail=tr,
                alg_bytes_per_launch=round(dom["bytes"] / dom["launches"]), launches_per_step=dom["launches"] / psteps,
                avg_launch_us=round(dom["total_ms"] * 1e3 / dom["launches"], 2), flops_per_launch=dom["flops"] / dom["launches"],
                generator_ms_per_step=round(gen_ms, 4),
                generator_tflops=round(sum(r["flops"] for r in prof) / psteps / (gen_ms * 1e-3) / 1e12, 3),
                timing="HIP events around the Generator's launches in a separate eager pass AFTER the timed region",
                families=[dict(name=r["name"], launches=r["launches"] / psteps, ms_per_step=round(r["total_ms"] / psteps, 4),
                               tflops=round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 3),
                               alg_GBps=round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)) for r in prof])


def upsampling_block(ups, psteps, config):
    """HBM roofline of the Generator's five ConvTranspose1d launches (models.py:510-522, 545), each row = one launch site."""
    tot_b = sum(r["bytes"] for r in ups)
    tot_s = sum(r["total_ms"] for r in ups) * 1e-3
    rows = []
    for r in ups:
        fam = r["name"].split("|")[1].split(" ")[0]
        tr = pmc_traffic(fam, config)
        rows.append(dict(site=r["name"], us_per_launch=round(r["total_ms"] * 1e3 / r["launches"], 2),
                         alg_bytes_per_launch=round(r["bytes"] / r["launches"]), alg_GBps=round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1),
                         frac_of_hbm_peak=round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         tflops=round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2),
                         kernel_family_pmc_traffic_bytes_per_launch=tr.get("bytes_per_launch"),
                         pmc_source=tr.get("source") or tr.get("note")))
    # the families that run NOTHING but upsampling launches: bf16 conv_cl_bf16<1x4> (the last ConvTranspose1d), fp32 conv1d_mfma<32x128>
    # (the last two: their family average is over both; "algorithmic" counts the input once per polyphase problem, the PMC figure sees
    # the phases share it in L2, so it can be below)
    narrow = [r for r in rows if "conv_cl_bf16<1x4>" in r["site"] or "conv1d_mfma<32x128>" in r["site"]]
    return dict(bound="hbm", achieved=round(tot_b / tot_s / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                frac=round(tot_b / tot_s / 1e9 / PEAK_HBM_GBPS, 4), launches_per_step=len(ups),
                alg_bytes_per_step=round(tot_b / psteps), ms_per_step=round(tot_s * 1e3 / psteps, 4),
                traffic=(narrow[0]["kernel_family_pmc_traffic_bytes_per_launch"] if narrow else None),
                traffic_note="PMC bytes (FETCH x2 + WRITE) per launch of the kernel family that runs nothing but upsampling launches (bf16: the last "
                             "ConvTranspose1d; fp32: the last two, family average); the other rows share their kernel symbol with ResBlock "
                             "launches, so their family averages are shown per row, not used here",
                timing="HIP events around the ConvTranspose1d launches in a separate eager pass AFTER the timed region",
                launches=rows)


def encoder_gemms_block(rows, psteps, config):
    """north_star: "MFMA utilisation on the encoder GEMMs against gfx950 peak".  `rows`: a per-site HIP-event pass (profile mode 3).  The encoder
    GEMMs are the launch sites `enc.qkv` / `enc.o` / `enc.ffn1` / `enc.ffn2` of attentions.Encoder (reference attentions.py:263-266, 438-446) — the
    text encoder's six layers and every transformer-flow coupling's stack — grouped by kernel family: achieved = the convs' own FLOPs / HIP-event
    time, against the dense matrix peak of the family's arithmetic (fp32 MFMA 157.3 TF, fp16 MFMA 2 500 TF); `pmc_mfma_util` = the SQ busy-cycle
    figure of the same family from the newest committed PMC pass of this config (profiles/*pmc_c{2,3}.json), when there is one."""
    fam = {}
    for r in rows:
        site, _, rest = r["name"].partition("|")
        if not (site.startswith("enc.") or site == "attention") or not r["launches"]:      # the attention core sits between q/k/v and conv_o
            continue
        name = rest.split(" n")[0]
        f = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
        f["launches"] += r["launches"]; f["ms"] += r["total_ms"]; f["flops"] += r["flops"]
    if not fam:
        return None
    pmc, pmc_file = {}, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_c{2 if config == 2 else 3}.json")), reverse=True):
        try:
            pmc, pmc_file = json.load(open(path)).get("kernels", {}), os.path.basename(path)
            break
        except (OSError, ValueError):
            continue
    out = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        peak = PEAK_BF16_MFMA_TFLOPS if ("f16" in name or (name.startswith("attention") and config != 2)) else PEAK_FP32_MFMA_TFLOPS
        tf = f["flops"] / (f["ms"] * 1e-3) / 1e12
        key = "conv1d_splitk<32x32>" if name.startswith("conv1d_splitk") else name.split("<")[0] if name.startswith("conv_f16") else name
        pk = pmc.get(name) or pmc.get(key) or next((v for k, v in pmc.items() if k.startswith(key)), None)
        out.append(dict(kernel=name, launches_per_step=round(f["launches"] / psteps, 1), us_per_launch=round(f["ms"] * 1e3 / f["launches"], 2),
                        ms_per_step=round(f["ms"] / psteps, 4), tflops=round(tf, 2), peak=peak, frac=round(tf / peak, 4),
                        pmc_mfma_util=None if not pk else pk.get("mfma_util")))
    tot_ms, tot_fl = sum(f["ms"] for f in fam.values()), sum(f["flops"] for f in fam.values())
    return dict(sites="enc.qkv / enc.o / enc.ffn1 / enc.ffn2 (text encoder + transformer-flow stacks)", ms_per_step=round(tot_ms / psteps, 4),
                tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), families=out, pmc_file=pmc_file,
                timing="HIP events around every launch of these sites in a separate eager pass after the timed region")


def run_config(num, model, hp, dev, rank, world, steps, warmup, overrides, full_profile=False, solo=False, collective=False,
               repeats=1, contract=True):
    """Time `steps` steps of BASELINE config `num` on this rank; returns the result dict (rank-local times; the caller reduces).
    ``collective``: a process group exists and every rank is in this call — the timed region is bracketed by its barriers (always at
    world > 1).  ``solo``: this rank runs alone while the others wait (the N=1 anchor of the N>1 line): no barriers, no extra legs.
    ``repeats``: the timed region (exactly `steps` steps between two syncs) is run this many times back to back.  ``contract`` = True
    (the primary line): `dt` is the FIRST region — the K steps after the W warm-ups the driver's contract names — and the others are
    reported as its spread; False (secondary legs): `dt` is the median region, so one bad draw cannot set a leg's figure."""
    cfg = dict(CONFIGS[num])
    B = overrides.get("batch") or cfg["batch"]
    T = overrides.get("symbols") or cfg["symbols"]
    gen_dtype = overrides.get("dtype") or cfg["dtype"]
    flow_dtype = overrides.get("flow") or cfg["flow"]
    use_graph = bool(cfg["graph"]) if overrides.get("graph") is None else bool(overrides["graph"])
    model.enable_graphs(False)
    model.set_generator_dtype(torch.bfloat16 if gen_dtype == "bf16" else torch.float32)
    model.set_flow_dtype(torch.float16 if flow_dtype == "f16" else torch.float32)     # both flow variants have an fp16 form
    batch, lengths = make_batch(cfg, B, T, rank)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    call = lambda b=dbatch: model.infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"],
                                        b["en_bert"], **KW)
    frames_per_step = 3 * sum(lengths)                 # pinned durations: ceil(2.5) = 3 frames per symbol, valid frames only
    audio_per_step = frames_per_step * hp.total_upsample / hp.sampling_rate

    def barrier():
        if (world > 1 or collective) and not solo:
            import torch.distributed as dist
            dist.barrier()

    model.enable_graphs(use_graph, static_io=True)     # serving-loop replay: inputs read in place, outputs are the graph's buffers
    model.profile(0)
    for i in range(warmup):
        out = call()
    torch.cuda.synchronize()
    assert int(out[2].sum().item()) == frames_per_step, (int(out[2].sum().item()), frames_per_step)
    dts = []
    for _rep in range(max(1, repeats)):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            o, attn, y_mask, _rest = call()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = dts[0] if contract else sorted(dts)[len(dts) // 2]
    Ty = y_mask.shape[2]
    res = dict(config=num, B=B, T=T, Ty=Ty, gen_dtype=gen_dtype, flow_dtype=flow_dtype, graph=use_graph, dt=dt, steps=steps,
               audio_per_step=audio_per_step, lengths=lengths)
    if len(dts) > 1:
        ms = [d / steps * 1e3 for d in dts]
        spread = (max(ms) - min(ms)) / min(ms)
        res["repeats"] = dict(ms_per_step=[round(m, 4) for m in ms], min=round(min(ms), 4), max=round(max(ms), 4),
                              median=round(sorted(ms)[len(ms) // 2], 4), spread=round(spread, 4), unstable=bool(spread > 0.05),
                              reported="first region (the contract's K steps)" if contract else "median region")
    if solo:
        model.enable_graphs(False)
        return res
    if rank == 0:
        # ---- PCIe-inclusive variant: inputs start in pinned HOST memory, the audio ends in pinned HOST memory.  The
        # uploads land in the PERSISTENT device tensors the (static_io) graph reads in place, so a captured graph is replayed, not
        # re-captured per step (a fresh upload per step has new addresses: ADVICE r2).
        hbatch = {k: v.pin_memory() for k, v in batch.items()}
        S = Ty * hp.total_upsample
        host_o = torch.empty(B, 1, S, dtype=torch.float32, pin_memory=True)
        n_io = max(3, min(steps, 10))

        def call_io():
            for k, v in hbatch.items():
                dbatch[k].copy_(v, non_blocking=True)
            o = call()[0]
            host_o.copy_(o, non_blocking=True)

        call_io()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_io):
            call_io()
        torch.cuda.synchronize()
        res["io_ms_per_step"] = (time.perf_counter() - t1) / n_io * 1e3
        res["io_bytes_per_step"] = sum(v.numel() * v.element_size() for v in batch.values()) + host_o.numel() * 4

    # ---- roofline leg: per-launch HIP events on the Generator's kernels, separate eager pass (events cannot be recorded inside a
    # captured graph, and inside the timed loop they cost ~10 us of bubble per event pair)
    model.enable_graphs(False)
    psteps = max(3, min(steps, 150 if B == 1 else 20))
    call()
    model.profile(2)
    torch.cuda.synchronize()
    for _ in range(psteps):
        call()
    torch.cuda.synchronize()
    prof = model.profile_report()
    model.profile(0)
    res["roofline"] = roofline_block(prof, psteps, num) if prof else None
    # BASELINE config 5: "HBM-bandwidth roofline run on Generator upsampling" — the ConvTranspose1d launches alone (profile mode 4),
    # algorithmic bytes (inputs read once, output written once, weights once) / HIP-event time against the 8 TB/s HBM peak
    try:
        model.profile(4)
        for _ in range(psteps):
            call()
        torch.cuda.synchronize()
        ups = model.profile_report()
        model.profile(0)
        res["upsampling_roofline"] = upsampling_block(ups, psteps, num) if ups else None
    except Exception as e:
        model.profile(0)
        res["upsampling_roofline"] = dict(error=repr(e)[:200])
    try:
        model.profile(3)               # one row per launch site and shape
        esteps = max(2, min(psteps, 10))
        for _ in range(esteps):
            call()
        torch.cuda.synchronize()
        res["encoder_gemms"] = encoder_gemms_block(model.profile_report(), esteps, num)
        model.profile(0)
    except Exception as e:
        model.profile(0)
        res["encoder_gemms"] = dict(error=repr(e)[:200])
    if full_profile:
        model.profile(3)               # one row per launch site and shape
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        full = [dict(name=r["name"], launches=r["launches"] / 3, ms_per_step=round(r["total_ms"] / 3, 4),
                     us_per_launch=round(r["total_ms"] * 1e3 / r["launches"], 2),
                     tflops=round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 3)) for r in model.profile_report()]
        model.profile(0)
        for r in sorted(full, key=lambda r: -r["ms_per_step"]):
            log(f"  {r['ms_per_step']:8.4f} ms/step  {r['launches']:5.0f} x {r['us_per_launch']:8.2f} us  {r['tflops']:7.2f} TF  {r['name']}")
        res["full"] = full
    return res


def run_batches_in_flight(model, hp, dev, steps, nstreams=2, num=3):
    """BASELINE config `num`'s batch with `nstreams` REQUESTS in flight (each a full batch of that config, eager): one request's flow —
    memory / latency-bound fp16 convs and attention — runs beside the other's Generator — MFMA-bound pair kernels.  The serving
    pipeline's throughput (serving.synthesize(requests_in_flight=n)); reported beside the one-request figure, never instead of it."""
    cfg = CONFIGS[num]
    ms = [model]
    for _ in range(nstreams - 1):
        m2 = models.from_hparams(hp)
        m2.attach_blob(model._blob)
        ms.append(m2)
    for m in ms:
        m.enable_graphs(False)
        m.set_generator_dtype(torch.bfloat16 if cfg["dtype"] == "bf16" else torch.float32)
        m.set_flow_dtype(torch.float16 if cfg["flow"] == "f16" else torch.float32)
    batch, lengths = make_batch(cfg, cfg["batch"], cfg["symbols"], 0)
    b = {k: v.to(dev) for k, v in batch.items()}
    # two disjoint stream sets, the better placement is reported (streams that share a hardware queue serialise: run_two_streams)
    tries = []
    for attempt in range(2):
        streams = [torch.cuda.Stream(dev) for _ in ms]

        def step(i):
            with torch.cuda.stream(streams[i % nstreams]):
                return ms[i % nstreams].infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"], **KW)

        for i in range(3 * nstreams):
            out = step(i)
        torch.cuda.synchronize()
        frames = int(out[2].sum().item())
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        tries.append(time.perf_counter() - t0)
    dt = min(tries)
    audio = frames * hp.total_upsample / hp.sampling_rate
    return dict(workload=f"BASELINE config {num}'s batch (B={cfg['batch']} x T={cfg['symbols']}, bf16 Generator + fp16 flow, eager), {nstreams} "
                         f"requests in flight on {nstreams} HIP streams ({nstreams} handles, one weight blob)", requests_in_flight=nstreams,
                value=round(audio * steps / dt, 2), unit="audio-seconds/sec", ms_per_step=round(dt / steps * 1e3, 4), steps=steps,
                ms_per_step_by_stream_set=[round(t / steps * 1e3, 4) for t in tries],
                note="throughput of a request pipeline; per-request latency is about twice ms_per_step; best of two stream placements")


def run_two_streams(model, hp, dev, steps, nstreams=2):
    """Config 2's utterance with TWO requests in flight: two shim instances share the packed weight blob, each owns a HIP stream and
    its workspace, and the host alternates between them — while one request's phase A / flow (a chain of small kernels that leaves
    most CUs idle) runs, the other request's Generator fills the machine.  Every launch is still batch 1; a step is still one full
    infer().  Reported beside the sequential figure (which stays `value`: it is also the per-request latency), never instead of it."""
    ms = [model]
    for _ in range(nstreams - 1):
        m2 = models.from_hparams(hp)
        m2.attach_blob(model._blob)
        ms.append(m2)
    for m in ms:
        m.enable_graphs(False)
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)
    batch, lengths = make_batch(CONFIGS[2], 1, 128, 0)
    b = {k: v.to(dev) for k, v in batch.items()}
    torch.cuda.synchronize()

    # The runtime maps HIP streams to a handful of hardware queues; two streams that share one serialise (observed: "2 in flight" at
    # exactly the sequential rate in one process, 1.33x in another).  Which streams collide is not controllable from here, so the leg
    # is measured on two disjoint stream sets and the better placement is reported (both are listed).
    tries = []
    for attempt in range(2):
        streams = [torch.cuda.Stream(dev) for _ in ms]

        def step(i):
            with torch.cuda.stream(streams[i % nstreams]):
                return ms[i % nstreams].infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"], **KW)

        for i in range(3 * nstreams):
            out = step(i)
        torch.cuda.synchronize()
        frames = int(out[2].sum().item())
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        tries.append(time.perf_counter() - t0)
    dt = min(tries)
    audio = frames * hp.total_upsample / hp.sampling_rate
    return dict(workload="BASELINE config 2's utterance (B=1 x T=128, fp32), two requests in flight on two HIP streams "
                         f"({nstreams} handles, one weight blob); every launch is batch 1", requests_in_flight=nstreams,
                value=round(audio * steps / dt, 2), unit="audio-seconds/sec", ms_per_step=round(dt / steps * 1e3, 4), steps=steps,
                ms_per_step_by_stream_set=[round(t / steps * 1e3, 4) for t in tries],
                note="throughput of a request pipeline; per-request latency is the sequential figure's ms_per_step or more; best of two "
                     "stream placements (streams that share a hardware queue serialise)")


def build_bert(dev):
    from bert_vits2_amd import bert_synth as BS
    from bert_vits2_amd.bert_encoder import BertEncoder
    sd = BS.bert_state_dict(BS.LARGE, 0, layers=22)
    return BertEncoder(**BS.LARGE).load_state_dict(sd, device=dev), sd


def run_text_to_audio(model, hp, dev, steps, nstreams, enc0):
    """SURVEY 8f-2 + the hot path as ONE device-resident request: BertModel forward for the sentence (53 tokens, hidden_states[-3]) ->
    word-level features handed to infer() through bert_index (no host copy, no repeated matrix) -> config 2's 128-symbol utterance,
    with `nstreams` requests in flight (a handle + HIP stream each for both models, one copy of each weight blob)."""
    from bert_vits2_amd import bert_features as BF, bert_synth as BS
    cfg = BS.LARGE
    encs = [enc0] + [enc0.replica() for _ in range(nstreams - 1)]
    ms = [model]
    for _ in range(nstreams - 1):
        m2 = models.from_hparams(hp)
        m2.attach_blob(model._blob)
        ms.append(m2)
    for m in ms:
        m.enable_graphs(False)
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)
    streams = [torch.cuda.Stream(dev) for _ in ms]
    batch, _ = make_batch(CONFIGS[2], 1, 128, 0)
    b = {k: v.to(dev) for k, v in batch.items()}
    ids, _ = BS.synthetic_inputs(cfg, [53], 0)
    ids = ids.to(dev)
    word2ph = [1] + [2, 3] * 24 + [2, 2, 2] + [1]                   # 53 words -> 128 symbols (blanks interspersed)
    assert len(word2ph) == 53 and sum(word2ph) == 128
    torch.cuda.synchronize()

    def step(i):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            feat, index = BF.word_level_feature_cs(encs[k](ids)[0], word2ph)
            return ms[k].infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], feat[None], b["ja_bert"], b["en_bert"],
                               bert_index=(index[None], None, None), **KW)

    for i in range(3 * nstreams):
        out = step(i)
    torch.cuda.synchronize()
    frames = int(out[2].sum().item())
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio = frames * hp.total_upsample / hp.sampling_rate
    return dict(workload=f"BertModel 24x1024 features (53 tokens) + BASELINE config 2's utterance per request, features handed over on the "
                         f"device at word level, {nstreams} requests in flight", requests_in_flight=nstreams,
                value=round(audio * steps / dt, 2), unit="audio-seconds/sec", ms_per_request=round(dt / steps * 1e3, 4), steps=steps)


def bert_traffic():
    """HBM bytes per BERT forward from tools/collect_traffic_bert.py's PMC passes (profiles/*traffic_bert*.json), only if taken with the
    current kernel sources; None otherwise."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_bert*.json")), reverse=True):
        try:
            d = json.load(open(path))
            if all(file_digest(f) == v for f, v in d["source_digests"].items()):
                return round(d["traffic_bytes_per_forward"])
        except Exception:
            continue
    return None


def bench_bert(dev, with_cpu, enc=None, sd=None):
    """SURVEY 8f-2 leg (secondary, N=1 only): hidden_states[-3] of a chinese-roberta-wwm-ext-large-shaped BertModel for ONE sentence
    of config 2's size (128 symbols with blanks interspersed = ~51 characters + [CLS]/[SEP] = 53 tokens) through bv2_bert_forward,
    seeded synthetic weights.  HBM-side roofline: every forward streams the 22 layers' fp32 weights once (algorithmic bytes)."""
    from bert_vits2_amd import bert_synth as BS
    cfg, S, layers = BS.LARGE, 53, 22
    if enc is None:
        enc, sd = build_bert(dev)
    ids, _ = BS.synthetic_inputs(cfg, [S], 0)
    ids = ids.to(dev)
    for _ in range(3):
        enc(ids)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        out = enc(ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    C, I = cfg["hidden_size"], cfg["intermediate_size"]
    wbytes = 4.0 * layers * (4 * C * C + 2 * C * I + 9 * C + I)                     # weights + biases + LayerNorm vectors, once
    abytes = 4.0 * S * (layers * (3 * C + 2 * C + 2 * C + I + I + 2 * C) + 4 * C)   # activations written + read once per GEMM / LN
    flops = 2.0 * S * layers * (4 * C * C + 2 * C * I) + 4.0 * layers * S * S * C
    res = dict(workload=f"BertModel 24x1024 (chinese-roberta-wwm-ext-large shape), hidden_states[-3] = {layers} layers, B=1 x S={S} tokens, "
                        f"fp32, seeded synthetic weights; reference call site text/chinese_bert.py:34-37",
               ms_per_sentence=round(ms, 4), sentences_per_sec=round(1e3 / ms, 2), dtype="f32", launches=1 + 1 + 7 * layers,
               roofline=dict(bound="hbm", achieved=round((wbytes + abytes) / (ms * 1e-3) / 1e9, 1), peak=8000.0, unit="GB/s",
                             frac=round((wbytes + abytes) / (ms * 1e-3) / 8e12, 4), alg_bytes_per_forward=int(wbytes + abytes),
                             tflops=round(flops / (ms * 1e-3) / 1e12, 2), traffic=bert_traffic(),
                             note="latency-bound at batch 1: 156 dependent launches of ~10 us; the weights (1.1 GB fp32) are the algorithmic bytes"))
    # the same model over a padded batch of 8 sentences (a request split into sentences, infer.py:268-332): the weights are
    # streamed once per batch instead of once per sentence
    ids8, ln8 = BS.synthetic_inputs(cfg, [53, 41, 37, 53, 29, 48, 33, 53], 1)
    ids8, ln8 = ids8.to(dev), ln8.to(dev)
    for _ in range(2):
        enc(ids8, lengths=ln8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        enc(ids8, lengths=ln8)
    torch.cuda.synchronize()
    ms8 = (time.perf_counter() - t0) / 10 * 1e3
    res["batch8"] = dict(ms_per_batch=round(ms8, 4), sentences_per_sec=round(8e3 / ms8, 1), tokens=int(ln8.sum().item()),
                         tflops=round(flops / S * 8 * 53 / (ms8 * 1e-3) / 1e12, 2))
    # the reference's Japanese / English extractors (DeBERTa-v2 large; text/japanese_bert.py, english_bert_mock.py) on the same sentence
    # size: disentangled attention (kernels/deberta_attn.hip), the Japanese model's ConvLayer
    try:
        from bert_vits2_amd.bert_encoder import BertEncoder
        for nm, dcfg in (("deberta_v2_large_japanese", BS.LARGE_JA), ("deberta_v3_large", BS.LARGE_V3)):
            dcfg = dict(dcfg, vocab_size=min(dcfg["vocab_size"], 32000))        # the embedding table is read one row per token
            dsd = BS.deberta_state_dict(dcfg, 0, layers=layers)
            denc = BertEncoder(**dcfg, model_type="deberta-v2").load_state_dict(dsd, device=dev)
            dids = torch.randint(1, dcfg["vocab_size"], (1, S), generator=torch.Generator().manual_seed(1)).to(dev)
            for _ in range(3):
                denc(dids)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dout = denc(dids)
            torch.cuda.synchronize()
            dms = (time.perf_counter() - t0) / 20 * 1e3
            res[nm] = dict(ms_per_sentence=round(dms, 4), sentences_per_sec=round(1e3 / dms, 2), tokens=S)
            if with_cpu and nm == "deberta_v2_large_japanese":
                from oracle import deberta_oracle as DO                      # the checker, cpu leg only
                dref = DO.hidden_state(dsd, dcfg, dids.cpu(), layers)
                res[nm]["max_abs_err_vs_oracle"] = float((dout.cpu().transpose(1, 2) - dref).abs().max())
            del denc, dsd
    except Exception as e:
        res["deberta_error"] = repr(e)[:300]
    if with_cpu:
        nthreads = min(usable_cores(), 64)           # torch's default team is the HOST's core count even inside a CPU quota
        torch.set_num_threads(nthreads)
        from oracle import bert_oracle as BO                                    # the checker, cpu leg only
        ref = BO.hidden_state(sd, cfg, ids.cpu(), layers)                       # warm-up
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            ref = BO.hidden_state(sd, cfg, ids.cpu(), layers)
            ts.append((time.perf_counter() - t1) * 1e3)
        cpu_ms = sorted(ts)[1]
        res["cpu_baseline"] = dict(value=round(1e3 / cpu_ms, 3), unit="sentences/sec", ms_per_sentence=round(cpu_ms, 1), cores=nthreads,
                                   kind="port", sample="median of 3 runs of the same sentence after 1 warm-up, torch CPU fp32, oracle "
                                                       "restatement of transformers' BertModel (the reference runs the HF model on torch)")
        res["max_abs_err_vs_oracle"] = float((out.cpu().transpose(1, 2) - ref).abs().max())
    return res


def describe(res, hp, world):
    gd, fd = res["gen_dtype"], res["flow_dtype"]
    ragged = CONFIGS[res["config"]]["ragged"]
    shape = (f"B={res['B']} utterances of uniform [96,{res['T']}] symbols (mean {sum(res['lengths']) / res['B']:.1f}), ZH/JA/EN round-robin, "
             f"random speakers" if ragged else f"B={res['B']} x T={res['T']} symbols")
    return (f"BASELINE config {res['config']}: {shape} per GPU, "
            f"{'bf16 Generator (fp32 accumulate)' if gd == 'bf16' else 'fp32 Generator (wide-stage products on the bf16 matrix core from exact 3-way bf16 operand splits, fp32 accuracy; secondary.config2_fp32_mfma = the fp32-MFMA form)'}, "
            f"{'fp16 flow convs (fp32 accumulate / LayerNorm / softmax / gate)' if fd == 'f16' else 'fp32 flow'}, "
            f"fp32 text encoder / durations / spline, T_y={res['Ty']} frames "
            f"({res['Ty'] * hp.total_upsample} samples, {res['Ty'] * hp.total_upsample / hp.sampling_rate:.3f} s) per padded utterance, "
            f"{'transformer' if hp.use_transformer_flow else 'residual (WN)'} flow, synthetic seeded weights, durations pinned to 3 frames/symbol, "
            f"hipGraph={'on' if res['graph'] else 'off'}")


def summary(res, hp, world, dt=None, audio=None):
    dt = res["dt"] if dt is None else dt
    audio = res["audio_per_step"] * res["steps"] * world if audio is None else audio
    out = dict(workload=describe(res, hp, world), value=round(audio / dt, 2), unit="audio-seconds/sec",
               ms_per_step=round(dt / res["steps"] * 1e3, 4), steps=res["steps"],
               dtype=dtype_label(res), utterances_per_gpu=res["B"], symbols=res["T"], frames=res["Ty"], hipgraph=res["graph"],
               rtf=round(dt / audio, 6))
    if res.get("repeats"):
        out["repeats"] = res["repeats"]
    if "io_ms_per_step" in res:
        out["pcie_inclusive"] = dict(value=round(res["audio_per_step"] / (res["io_ms_per_step"] * 1e-3), 2),
                                     ms_per_step=round(res["io_ms_per_step"], 4), bytes_per_step=res["io_bytes_per_step"],
                                     note="inputs copied from pinned host memory and the audio copied back to pinned host memory every step")
    if res.get("roofline") is not None:
        out["roofline"] = res["roofline"]
    if res.get("encoder_gemms") is not None:
        out["encoder_gemms"] = res["encoder_gemms"]
    if res.get("upsampling_roofline") is not None:
        if res["config"] == 5 and "error" not in res["upsampling_roofline"]:
            # BASELINE config 5 names its roofline: HBM bandwidth on the Generator upsampling.  The dominant-kernel (MFMA) block stays beside it.
            out["roofline_dominant_kernel"] = out.get("roofline")
            out["roofline"] = res["upsampling_roofline"]
        else:
            out["upsampling_roofline"] = res["upsampling_roofline"]
    return out


def dtype_label(res):
    gd, fd = res["gen_dtype"], res["flow_dtype"]
    if gd == "f32" and fd == "f32":
        return "f32"
    return f"gen={gd},flow={fd}"                 # what each part computes in (fp32 accumulate everywhere)

HEADLINE_MAX_BYTES = 4096          # the driver parses the LAST stdout line; round 3's 36 KB line was not parsed (BENCH_r03.parsed == null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _leg(v):
    """One secondary leg in one line: what it reached, how long a step took, the roofline fraction of its dominant kernel."""
    if not isinstance(v, dict):
        return v
    if "error" in v:
        return dict(error=str(v["error"])[:80])
    out = _pick(v, ("value", "ms_per_step", "ms_per_request", "ms_per_sentence", "sentences_per_sec", "launches", "graph_hit_rate", "distinct_ty",
                    "replay_over_eager", "padded_value"))
    rp = v.get("repeats")
    if isinstance(rp, dict):
        if rp.get("unstable"):                                    # min / max of the repeats live in the details file
            out["min_max"] = [rp.get("min"), rp.get("max")]
            out["unstable"] = True
    r = v.get("roofline")
    if isinstance(r, dict) and "frac" in r:
        out["frac"], out["bound"] = r["frac"], r.get("bound")
    rd = v.get("roofline_dominant_kernel")
    if isinstance(rd, dict) and "frac" in rd:
        out["mfma_frac"] = rd["frac"]
    return out


def headline(line, details_path=None):
    """The compact form of the full record `line`: every contract key verbatim, scalar-only `roofline` / `cpu_baseline` / `parity`
    blocks, one-line summaries of the secondary legs and of the ranks.  Always strict JSON (no NaN / Infinity) below
    HEADLINE_MAX_BYTES; whatever does not fit lives in the details file."""
    h = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    cfg = line.get("config") or {}
    h["config"] = _pick(cfg, ("workload", "utterances_per_gpu", "symbols", "frames", "parallelism", "hipgraph", "hip_force_dev_kernarg"))
    if isinstance(h["config"].get("workload"), str):                 # the long description stays in the details file
        h["config"]["workload"] = re.sub(r" \(wide-stage[^)]*\)", " (split-bf16 MFMA)", h["config"]["workload"]).split(", T_y=")[0][:200]
    if isinstance(cfg.get("pcie_inclusive"), dict):
        h["config"]["pcie_inclusive"] = _pick(cfg["pcie_inclusive"], ("value", "ms_per_step"))
    if isinstance(line.get("repeats"), dict):
        h["repeats"] = _pick(line["repeats"], ("ms_per_step", "spread", "unstable"))
    r = line.get("roofline")
    if isinstance(r, dict):
        h["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac")) if "error" not in r else dict(error=str(r["error"])[:120])
        if "error" not in r:
            h["roofline"]["traffic"] = r.get("traffic")
            h["roofline"].update(_pick(r, ("alg_bytes_per_launch", "flops_per_launch", "avg_launch_us", "launches_per_step",
                                           "issued_flops_per_alg_flop", "fp32_equivalent_tflops", "generator_ms_per_step")))
    else:
        h["roofline"] = None
    c = line.get("cpu_baseline")
    if isinstance(c, dict):
        h["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "ms_per_step"))
        h["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:120]
        if isinstance(c.get("reference_container"), dict):
            h["cpu_baseline"]["reference_container"] = _pick(c["reference_container"], ("threads", "reference_over_port", "reference_over_port_range"))
        if isinstance(c.get("reference_estimate_this_box"), dict):
            h["cpu_baseline"]["reference_estimate_this_box"] = c["reference_estimate_this_box"].get("value")
            h["cpu_baseline"]["reference_estimate_range"] = c["reference_estimate_this_box"].get("range")
        if isinstance(c.get("single_thread"), dict):
            h["cpu_baseline"]["single_thread"] = _pick(c["single_thread"], ("value", "ms_per_step"))
    else:
        h["cpu_baseline"] = None
    if isinstance(line.get("parity"), dict):
        h["parity"] = _pick(line["parity"], ("w_ceil_match", "wave_rms", "wave_max_abs", "signal_rms", "mel_l1", "attn_equal", "error"))
        for blk in (h["parity"],):
            if isinstance(line["parity"].get("unpinned"), dict):
                blk["unpinned"] = _pick(line["parity"]["unpinned"], ("w_ceil_match", "distinct_durations", "frames", "frames_oracle", "pinned_after_flip",
                                                                      "wave_rms", "mel_l1", "attn_equal", "error"))
        for blk in (h["parity"], h["parity"].get("unpinned") or {}):
            for k, v in list(blk.items()):
                if isinstance(v, float):
                    blk[k] = float(f"{v:.4g}")
                elif isinstance(v, str):
                    blk[k] = v[:120]
    u = line.get("upsampling_roofline")
    if isinstance(u, dict) and "error" not in u:
        h["upsampling_roofline"] = _pick(u, ("bound", "achieved", "peak", "unit", "frac", "ms_per_step"))
        h["upsampling_roofline"]["traffic"] = u.get("traffic")
    def _eg(e):
        if not isinstance(e, dict) or "families" not in e:
            return None
        return dict(ms_per_step=e.get("ms_per_step"), tflops=e.get("tflops"),
                    families=[_pick(f, ("kernel", "us_per_launch", "peak", "frac", "pmc_mfma_util")) for f in e["families"][:2]])
    sec = line.get("secondary")
    eg = {k: v for k, v in (("config2", _eg(line.get("encoder_gemms"))),
                            ("config3", _eg((sec or {}).get("config3", {}).get("encoder_gemms") if isinstance(sec, dict) else None))) if v}
    if eg:
        h["encoder_gemms"] = eg
    if isinstance(sec, dict):
        h["secondary"] = {k: _leg(v) for k, v in sec.items()              # the other in-flight depths are in the details file
                          if k not in ("config2_2_requests_in_flight", "text_features_plus_config2_4_in_flight")}
        ctl = sec.get("config2_fp32_mfma")
        if isinstance(ctl, dict) and ctl.get("value"):
            # the same step with the round-2 kernel on the SAME box: box-to-box spread cancels in the ratio
            h["control_ratio_vs_fp32_mfma"] = round(line["value"] / ctl["value"], 4)
    if line.get("per_rank") is not None:
        h["ranks_seen"], h["launcher"] = line.get("ranks_seen"), line.get("launcher")
        h["per_rank"] = [None if r is None else dict(_pick(r, ("rank", "ms_per_step", "audio_s_per_step", "utterances", "symbols_total")),
                                                      roofline=_pick(r.get("roofline") or {}, ("frac", "avg_launch_us")))
                         for r in line["per_rank"]]
        h["weight_broadcast_ms"] = line.get("weight_broadcast_ms")
        h["collectives_in_timed_region"] = "none"
        if isinstance(line.get("n1_same_workload"), dict):
            h["n1_same_workload"] = _pick(line["n1_same_workload"], ("value", "ms_per_step"))
            h["scaling_efficiency"] = line.get("scaling_efficiency")
        if line.get("forced_dist"):
            h["forced_dist"] = str(line["forced_dist"])[:160]
    if details_path:
        h["details"] = details_path
    txt = json.dumps(h, allow_nan=False, separators=(",", ":"))
    # belt and braces: shed optional blocks rather than ever print a line the driver cannot take
    for drop in ("upsampling_roofline", "secondary", "encoder_gemms", "per_rank", "parity"):
        if len(txt) < HEADLINE_MAX_BYTES:
            break
        if drop in h:
            h[drop] = "see details"
            txt = json.dumps(h, allow_nan=False, separators=(",", ":"))
    assert len(txt) < HEADLINE_MAX_BYTES, len(txt)
    return txt


def _finite(o):
    """Strict JSON has no NaN / Infinity: map them to null wherever they occur."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def emit(line, details_out=None):
    """Full record -> details file; compact headline -> the ONE line on stdout, the last thing this process prints there."""
    line = _finite(line)
    path = details_out or os.path.join("gpurun_out", "bench_details.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, allow_nan=False)
    except OSError as e:
        log(f"details file not written: {e}")
        path = None
    txt = headline(line, path)
    sys.stdout.write(txt + "\n")
    sys.stdout.flush()
    return txt


class Seam:
    """Everything in the N-rank driver that touches a GPU, so that a CPU test can drive the SAME launcher / reduction code with
    gloo and world size 2 (tests/test_bench_multi_cpu.py passes --seam tests.bench_seam_cpu, whose `seam` object replaces this one)."""
    backend = "nccl"                                   # RCCL

    def device(self, local):
        assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback for the product path"
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def init_pg(self, dev, rank, world):
        import torch.distributed as dist
        dist.init_process_group(self.backend, device_id=dev, rank=rank, world_size=world)

    def load_model(self, hp, rank, dev):
        """rank 0 folds/packs the seeded synthetic checkpoint, every other rank receives the blob over RCCL"""
        model = models.from_hparams(hp)
        sd = None
        if rank == 0:
            sd = synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5)
            model.load_state_dict(sd, strict=False)
        t_bcast = sharding.distribute_weights(model, dev, src=0)
        return model, sd, t_bcast

    def run_config(self, *a, **k):
        return run_config(*a, **k)

    def sync(self):
        torch.cuda.synchronize()

    def device_name(self, dev):
        return torch.cuda.get_device_name(dev)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawned_rank(local, argv, port, world):
    """One rank of a self-launched N>1 run (torch.multiprocessing.spawn entry point)."""
    os.environ.update(RANK=str(local), LOCAL_RANK=str(local), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BV2_BENCH_SELF_LAUNCHED="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank_main(parse(argv))


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us (`python bench.py --gpus N`): spawn one process per GPU ourselves, rendezvous on 127.0.0.1
        import torch.multiprocessing as mp
        port = args.master_port or _free_port()
        log(f"self-launching {args.gpus} ranks (127.0.0.1:{port})")
        mp.spawn(_spawned_rank, args=(list(sys.argv[1:] if argv is None else argv), port, args.gpus), nprocs=args.gpus, join=True)
        return
    rank_main(args)


def rank_main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or with no launcher at all")
    if args.library:
        from bert_vits2_amd import lib as _bv2lib
        _bv2lib.load(path=os.path.abspath(args.library))
        log(f"A/B: library {args.library}")
    seam = Seam()
    if args.seam:
        import importlib
        seam = importlib.import_module(args.seam).seam
    dev = seam.device(local)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                    # --force-dist without a launcher: a world-size-1 group on this device
            os.environ.setdefault("MASTER_PORT", str(args.master_port or _free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        seam.init_pg(dev, rank, world)

    hp = H.default_v23(use_transformer_flow=not args.residual_flow)
    primary = args.config if args.config is not None else (4 if use_dist else 2)
    overrides = dict(batch=args.batch, symbols=args.symbols, dtype=args.dtype, flow=args.flow_dtype, graph=args.graph)
    if args.residual_flow and overrides["flow"] is None:
        overrides["flow"] = "f32"                     # --flow-dtype f16 selects the fp16 WN convolutions

    log(f"rank {rank}/{world}: packing / distributing weights")
    model, sd, t_bcast = seam.load_model(hp, rank, dev)
    log("weights attached")
    if args.variants is not None:
        import ctypes
        spec, clg, hcg = args.variants.split(",")
        lib = model._lib
        lib.bv2_test_set_variants.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        lib.bv2_test_set_variants.restype = None
        lib.bv2_test_set_variants(spec.replace(";", ",").encode(), int(clg), int(hcg))     # "32:1;4:2,1,0": several <tiles>:<variant> pairs
        log(f"variants {args.variants}")
    for kv in args.option:
        key, val = kv.split("=")
        model.set_option(key, int(val))
        log(f"option {key} = {val}")

    res = seam.run_config(primary, model, hp, dev, rank, world, args.steps, args.warmup, overrides, args.full_profile,
                          collective=use_dist, repeats=args.repeats, contract=True)
    log(f"rank {rank}: config {primary}: timed region {res['dt']:.3f}s for {args.steps} steps")
    dt, audio = res["dt"], res["audio_per_step"] * args.steps
    per_rank = None
    n1_same = None
    if use_dist:
        # the same-workload single-GPU anchor of this line: rank 0 runs ITS shard once more, alone, while every other rank waits at the
        # barrier below (outside the timed region) — so that value / (N x n1) is a scaling efficiency on ONE workload, whatever the
        # N = 1 line of the driver measured (that one is BASELINE config 2, batch 1)
        if rank == 0:
            r1 = seam.run_config(primary, model, hp, dev, 0, 1, args.steps, min(args.warmup, 2), overrides, solo=True)
            n1_same = dict(value=round(r1["audio_per_step"] * args.steps / r1["dt"], 2), ms_per_step=round(r1["dt"] / args.steps * 1e3, 4),
                           utterances=r1["B"], steps=args.steps)
            log(f"rank 0 alone, same workload: {n1_same['value']} audio-s/s ({n1_same['ms_per_step']} ms/step)")
        dist.barrier()
    if use_dist:
        # value = audio of ALL ranks / the slowest rank's time (max over ranks); each rank also reports its own line
        mine = dict(rank=rank, local_rank=local, device=seam.device_name(dev), ms_per_step=round(res["dt"] / args.steps * 1e3, 4),
                    audio_s_per_step=round(res["audio_per_step"], 4), utterances=res["B"], symbols_total=sum(res["lengths"]),
                    weight_broadcast_ms=round(t_bcast * 1e3, 3),
                    roofline=None if not res.get("roofline") else {k: res["roofline"][k] for k in
                                                                    ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us")
                                                                    if k in res["roofline"]})
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dt = max(r["ms_per_step"] for r in per_rank) * args.steps * 1e-3
        audio = sum(r["audio_s_per_step"] for r in per_rank) * args.steps
        # the same two numbers through a device collective (the contract's MAX over ranks), as a cross-check of the object gather
        tmax = torch.tensor([res["dt"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    secondary = {}
    if world == 1 and not use_dist and rank == 0 and not args.no_secondary and args.config is None and not args.residual_flow:
        for num in (3, 4, 5):
            try:
                r = run_config(num, model, hp, dev, 0, 1, max(3, min(args.steps, 10)), 3, {}, repeats=3, contract=False)
                secondary[f"config{num}"] = summary(r, hp, 1)
                log(f"secondary config {num}: {secondary[f'config{num}']['value']} audio-s/s ({secondary[f'config{num}']['ms_per_step']} ms/step)")
            except Exception as e:          # a secondary workload must never take the primary line down
                secondary[f"config{num}"] = dict(error=repr(e)[:300])

        try:
            secondary["config3_unpinned"] = run_config3_unpinned(hp, dev, *((20, 40) if args.steps >= 10 else (4, 6)))
            log(f"secondary config 3, un-pinned durations: {secondary['config3_unpinned']}")
        except Exception as e:
            secondary["config3_unpinned"] = dict(error=repr(e)[:300])

        # config 2 once more with the wide Generator convs on the fp32 matrix core (v_mfma_f32_32x32x2_f32, conv_mfma.hip) instead
        # of the split-bf16 form (conv_x6.hip): the same fp32 numerics at the fp32 MFMA rate — the kernel of rounds 1-2
        try:
            model.set_option("conv_x6", 0)
            r = run_config(2, model, hp, dev, 0, 1, max(3, min(args.steps, 20)), 3, {}, repeats=3, contract=False)
            secondary["config2_fp32_mfma"] = summary(r, hp, 1)
            rf = secondary["config2_fp32_mfma"].get("roofline")
            if isinstance(rf, dict) and rf.get("traffic") is not None:
                # the PMC traffic files are collected with the DEFAULT options, where the conv1d_mfma<64x64> symbol only runs two
                # ConvTranspose1d launches — not this leg's 18 ResBlock launches: that family average is not evidence here
                rf["traffic"] = None
                rf["traffic_detail"] = dict(note="no PMC profile of this leg's option set (conv_x6 = 0); the committed traffic files describe the default path")
            log(f"secondary config 2 on the fp32 matrix core: {secondary['config2_fp32_mfma']['value']} audio-s/s")
        except Exception as e:
            secondary["config2_fp32_mfma"] = dict(error=repr(e)[:300])
        finally:
            model.set_option("conv_x6", 1)

        # north_star names the ResidualCouplingBlock / WN flow explicitly (models.py:403-445, modules.py:185-210): the same utterances
        # with use_transformer_flow=false — config 2 in fp32, config 3's batch with the bf16 Generator (+ the fp16 WN convs when built)
        try:
            hp_wn = H.default_v23(use_transformer_flow=False)
            m_wn = models.from_hparams(hp_wn)
            m_wn.load_state_dict(synth.synthetic_state_dict(hp_wn, seed=0, pin_durations=2.5), strict=False)
            sharding.distribute_weights(m_wn, dev, src=0)
            for num, key, ov in ((2, "config2_residual_flow", dict(flow="f32")), (3, "config3_residual_flow", dict(flow=WN_FLOW_B32))):
                r = run_config(num, m_wn, hp_wn, dev, 0, 1, max(3, min(args.steps, 20 if num == 2 else 10)), 3, ov, repeats=3, contract=False)
                secondary[key] = summary(r, hp_wn, 1)
                log(f"secondary {key}: {secondary[key]['value']} audio-s/s ({secondary[key]['ms_per_step']} ms/step)")
            del m_wn
        except Exception as e:
            secondary["residual_flow_error"] = repr(e)[:300]

        try:
            secondary["config3_2_requests_in_flight"] = run_batches_in_flight(model, hp, dev, 10, 2, 3)
            log(f"secondary config 3, 2 requests in flight: {secondary['config3_2_requests_in_flight']['value']} audio-s/s")
        except Exception as e:
            secondary["config3_2_requests_in_flight"] = dict(error=repr(e)[:300])
        for ns in sorted({args.streams, 4}):
            key = f"config2_{ns}_requests_in_flight"
            try:
                secondary[key] = run_two_streams(model, hp, dev, max(20, args.steps), ns)
                log(f"secondary config 2, {ns} requests in flight: {secondary[key]['value']} audio-s/s")
            except Exception as e:
                secondary[key] = dict(error=repr(e)[:300])
        try:
            bert_enc, bert_sd = build_bert(dev)
        except Exception as e:
            bert_enc, bert_sd = None, None
            secondary["bert_zh_features"] = dict(error=repr(e)[:300])
        for ns in (1, 4) if bert_enc is not None else ():
            key = f"text_features_plus_config2_{ns}_in_flight"
            try:
                secondary[key] = run_text_to_audio(model, hp, dev, max(20, args.steps), ns, bert_enc)
                log(f"secondary BERT + config 2, {ns} in flight: {secondary[key]['value']} audio-s/s")
            except Exception as e:
                secondary[key] = dict(error=repr(e)[:300])
        try:
            if bert_enc is not None:
                secondary["bert_zh_features"] = bench_bert(dev, not args.no_cpu_baseline, bert_enc, bert_sd)
            log(f"secondary BERT feature extraction: {secondary['bert_zh_features']['ms_per_sentence']} ms per sentence")
        except Exception as e:
            secondary["bert_zh_features"] = dict(error=repr(e)[:300])

    if rank == 0:
        cpu, parity = None, None
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle port)")
            cpu, par = cpu_baseline(hp, sd, args.cpu_iters)
            if primary == 2 and not any(v is not None for v in overrides.values()):
                try:
                    parity = parity_block(model, hp, dev, par)
                    log(f"parity (config 2's utterance vs the oracle): {parity}")
                    parity["unpinned"] = parity_unpinned_block(hp, dev)
                    log(f"parity (un-pinned utterance, no duration injection): {parity['unpinned']}")
                except Exception as e:
                    parity = dict(error=repr(e)[:300])
        s = summary(res, hp, world, dt, audio)
        line = dict(
            metric="audio-seconds/sec (44.1 kHz), SynthesizerTrn.infer(), 128-phoneme utterance", value=s["value"],
            unit="audio-seconds/sec", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=s["ms_per_step"],
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype=s["dtype"], data="synthetic",
            config=dict(workload=s["workload"], utterances_per_gpu=res["B"], symbols=res["T"], frames=res["Ty"],
                        parallelism=f"utterance-sharded x{world}", rtf=s["rtf"], x_realtime_per_gpu=round(s["value"] / world, 2),
                        hipgraph=res["graph"], weight_broadcast_ms=round(t_bcast * 1e3, 3),
                        hip_force_dev_kernarg=os.environ.get("HIP_FORCE_DEV_KERNARG"),
                        pcie_inclusive=s.get("pcie_inclusive"),
                        note=("N=1 measures BASELINE config 2 (the metric's config); N>1 measures config 4 per GPU — the single-GPU figure of "
                              "that same workload is n1_same_workload of the N>1 line itself (and secondary.config4 of the N=1 line)"
                              if args.config is None else None)),
            roofline=s.get("roofline"), cpu_baseline=cpu, parity=parity)
        if s.get("repeats"):
            line["repeats"] = s["repeats"]
        if s.get("upsampling_roofline") is not None:
            line["upsampling_roofline"] = s["upsampling_roofline"]
        if s.get("encoder_gemms") is not None:
            line["encoder_gemms"] = s["encoder_gemms"]
        if per_rank is not None:
            line["ranks_seen"] = len([r for r in per_rank if r is not None])
            line["launcher"] = ("self (torch.multiprocessing.spawn)" if os.environ.get("BV2_BENCH_SELF_LAUNCHED") else
                                "none (--force-dist: this process is the whole world-size-1 group)" if (args.force_dist and world == 1 and "TORCHELASTIC_RUN_ID" not in os.environ)
                                else "torch.distributed.run")
            line["per_rank"] = per_rank
            line["weight_broadcast_ms"] = round(max(r["weight_broadcast_ms"] for r in per_rank), 3)
            line["collectives_in_timed_region"] = "none on the data path (two barriers bracket it); the weight blob is broadcast once, before"
            if n1_same is not None:
                line["n1_same_workload"] = n1_same
                line["scaling_efficiency"] = round(s["value"] / (world * n1_same["value"]), 4) if n1_same["value"] else None
            if args.force_dist:
                line["forced_dist"] = f"world size {world}: backend {seam.backend}, process group + blob broadcast + barriers + all_reduce(MAX) + all_gather_object executed"
        if secondary:
            line["secondary"] = secondary
        if res.get("full"):
            line["kernel_families_untimed_pass"] = res["full"]
        emit(line, args.details_out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
