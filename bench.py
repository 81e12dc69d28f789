#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec (24 kHz) at batch 32, 256-token utterances, per BASELINE.json.

One "step" = one pass of the whole synthesis hot path over one batch of synthetic utterances: preprocess (mel + text
conditioning) -> 31 flow-matching Euler steps x 2 CFG branches through the 22-block DiT -> vocoder -> int16 PCM, with the
inputs (int16 reference clips, int32 ids, fp32 noise) already resident in HBM when the timed region starts and the PCM left in
HBM.  The same step timed as SURVEY.md 8(d) words the metric -- pinned HOST inputs -> H2D -> ... -> PCM D2H to pinned host memory,
~46 MB over PCIe per step, ~0.1 % of the step -- is measured on two extra steps and reported as `pcie_inclusive` (rounds 1-3
reported that figure as `value`; `--pcie` times the K steps that way).
Work per utterance follows
SURVEY.md 8(d): T = 256 ids (96 reference + 160 target), 6.0 s reference clip (144,000 samples ->
563 frames), 1037 generated frames, N = 1600 frames, 265,472 output samples = 11.061 s of audio.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: starts the N ranks itself, as a child
                                                            `python -m torch.distributed.run` on 127.0.0.1, and relays their output)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, rank 0 packs the weights and RCCL-broadcasts the flat buffer over xGMI,
every rank synthesises its own batch of 32 (weak scaling, no data-path collective).
Prints ONE JSON line on rank 0.  The CPU oracle is used only for the reported cpu_baseline leg.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vietvoice_tts_amd import pack, sharding  # noqa: E402,F401
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights  # noqa: E402
from vietvoice_tts_amd.runtime import HipSynth  # noqa: E402

SEED = 9527                    # reference default random_seed (model_config.py:33)
REF_SAMPLES = 144000           # 6.0 s
TEXT_TOKENS = 256
GEN_FRAMES = 1037
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def synth_reference_clip(i: int, n: int = REF_SAMPLES) -> torch.Tensor:
    """Band-limited noise: 32 sinusoids 80-7600 Hz, random phase, reference normalisation (peak 29491)."""
    g = torch.Generator().manual_seed(SEED + i)
    t = torch.arange(n, dtype=torch.float64) / 24000.0
    f = 80.0 + (7600.0 - 80.0) * torch.rand(32, generator=g, dtype=torch.float64)
    ph = 2 * math.pi * torch.rand(32, generator=g, dtype=torch.float64)
    x = torch.sin(2 * math.pi * f[:, None] * t[None, :] + ph[:, None]).sum(0)
    x = x - x.mean()
    x = x * (29491.0 / x.abs().max())
    return x.to(torch.int16)


def make_inputs(spec: ModelSpec, B: int, rank: int, device):
    g = torch.Generator().manual_seed(SEED + 1000 * rank)
    audio = torch.stack([synth_reference_clip(rank * B + i) for i in range(B)])
    ids = torch.randint(1, spec.vocab_size, (B, TEXT_TOKENS), generator=g, dtype=torch.int32)
    n_ref = REF_SAMPLES // spec.hop_length + 1
    N = n_ref + GEN_FRAMES
    noise = torch.randn(B, N, spec.n_mel, generator=g, dtype=torch.float32)
    d = dict(audio=audio.to(device), audio_len=torch.full((B,), REF_SAMPLES, dtype=torch.int32, device=device),
             ids=ids.to(device), text_len=torch.full((B,), TEXT_TOKENS, dtype=torch.int32, device=device),
             seq_len=torch.full((B,), N, dtype=torch.int32, device=device), noise=noise.to(device))
    if torch.device(device).type == "cuda":          # host-side originals (pinned) for the H2D leg of the timed step
        d["host"] = {k: t.pin_memory() for k, t in (("audio", audio), ("ids", ids), ("noise", noise))}
    d["seq_len_host"] = [N] * B                        # the caller knows the frame counts: no read-back inside the Euler-step call
    return d, N


def mixed_unit_plan(spec: ModelSpec, total: int, g: torch.Generator):
    """The seeded unit list of BASELINE configs[3]: per unit text tokens (64-512), reference-clip samples (3-9 s), reference /
    generated / total frames by the reference's rules.  Draws from g (the first two draws of make_mixed_inputs' stream)."""
    toks = torch.randint(64, 513, (total,), generator=g)
    secs = 3.0 + 6.0 * torch.rand(total, generator=g)
    samples = (secs * spec.sample_rate).to(torch.int64) // 256 * 256
    ref_frames = samples // spec.hop_length + 1
    gen_frames = torch.clamp((toks.float() * 0.62 * 6.48).to(torch.int64), min=94)     # ~62 % of the ids are target text
    gen_frames = torch.minimum(gen_frames, 1875 - ref_frames)                            # 20 s chunk cap of the reference
    frames = (ref_frames + gen_frames).tolist()
    return toks, samples, ref_frames, gen_frames, frames


def make_mixed_inputs(spec: ModelSpec, per_rank: int, rank: int, world: int, device):
    """BASELINE configs[3]: 32*world units, text 64-512 tokens, reference clips 3-9 s, LPT-sharded by frame cost;
    this rank's shard becomes one ragged batch (per-item lengths on the device, masks in every kernel)."""
    g = torch.Generator().manual_seed(SEED + 77)
    total = per_rank * world
    toks, samples, ref_frames, gen_frames, frames = mixed_unit_plan(spec, total, g)
    mine = sharding.shard_units([sharding.unit_cost(f, spec.dim) for f in frames], world)[rank]
    batches = []
    pad_frac = float(os.environ.get("VV_BENCH_PAD_FRAC", "1.0"))     # 1.0 = one ragged batch per rank (rows are packed on the device)
    for grp in sharding.plan_batches([frames[u] for u in mine], per_rank, pad_frac=pad_frac, min_units=4):
        units = [mine[j] for j in grp]
        B = len(units)
        S, T = int(samples[units].max()), int(toks[units].max())
        audio = torch.zeros(B, S, dtype=torch.int16)
        ids = torch.zeros(B, T, dtype=torch.int32)
        for j, u in enumerate(units):
            audio[j, : int(samples[u])] = synth_reference_clip(1000 + u, int(samples[u]))
            ids[j, : int(toks[u])] = torch.randint(1, spec.vocab_size, (int(toks[u]),), generator=g, dtype=torch.int32)
        seq = torch.tensor([frames[u] for u in units], dtype=torch.int32)
        N = int(seq.max())
        noise = torch.randn(B, N, spec.n_mel, generator=g, dtype=torch.float32)
        d = dict(audio=audio.to(device), audio_len=samples[units].to(torch.int32).to(device), ids=ids.to(device),
                 text_len=toks[units].to(torch.int32).to(device), seq_len=seq.to(device), noise=noise.to(device))
        d["host"] = {k: t.pin_memory() for k, t in (("audio", audio), ("ids", ids), ("noise", noise))}
        d["seq_len_host"] = [int(v) for v in seq]
        d["gen_frames"] = [int(gen_frames[u]) for u in units]
        batches.append((d, N, int(gen_frames[units].max())))
    audio_s = float(gen_frames[mine].sum()) * spec.hop_length / spec.sample_rate
    fill = float(sum(frames[u] for u in mine)) / sum(b[0]["seq_len"].numel() * b[1] for b in batches)
    return batches, audio_s, len(mine), fill


def longform(a):
    """BASELINE configs[4]: one 4096-character text through the drop-in TTSEngine (chunking, ragged GPU batches of 8
    chunks, hipGraph-replayed vocoder step, host cross-fade).  Wall clock includes all host plumbing."""
    import tempfile
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    sent = "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ và ngắm hoa nở bên đường nhé. "
    text = (sent * 60)[:4096]
    with tempfile.TemporaryDirectory() as d:
        cfg = ModelConfig(model_cache_dir=d, synthetic_model=True, model_spec=a.spec, acoustic_dtype=a.dtype, nfe_step=a.nfe,
                          max_batch_chunks=8, use_hip_graph=True)
        eng = TTSEngine(cfg)
        for _ in range(a.warmup):
            eng.synthesize(text)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = 0
        for _ in range(a.steps):
            wave, _secs = eng.synthesize(text)
            tot += wave.size
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        chunks = len(eng._last_plan) if hasattr(eng, "_last_plan") else None
        eng.cleanup()
    audio_s = tot / 24000.0
    print(json.dumps({"metric": "audio-seconds/sec (24 kHz), long-form 4k-char text, 8 chunks in flight (BASELINE configs[4])",
                      "value": round(audio_s / el, 3), "unit": "audio-seconds/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
                      "ms_per_step": round(el / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": a.dtype, "data": "synthetic voice pack + random-init weights",
                      "config": {"workload": "longform: 4096 chars, chunked by the reference rule, max_batch_chunks=8, hipGraph vocoder",
                                 "chunks": chunks, "spec": a.spec}}), flush=True)


SERVE_SENTENCES = [
    "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ và ngắm hoa nở bên đường nhé.",
    "Bản tin thời sự tối nay có những nội dung chính sau đây, mời quý vị và các bạn cùng theo dõi.",
    "Ngày xửa ngày xưa, ở một ngôi làng nhỏ bên bờ sông, có một cậu bé rất thích nghe kể chuyện.",
    "Cảm ơn các bạn đã lắng nghe, hẹn gặp lại các bạn trong chương trình lần sau.",
    "Xin vui lòng giữ máy trong giây lát, chúng tôi sẽ kết nối bạn với nhân viên hỗ trợ.",
    "Thật vậy sao, tôi chưa từng nghe điều đó bao giờ, bạn kể thêm cho tôi nghe đi.",
]
SERVE_VOICES = [dict(), dict(gender="male", group="news", area="southern", emotion="serious"), dict(gender="female", group="story", area="northern", emotion="happy"),
                dict(gender="male", group="audiobook", area="central", emotion="neutral"), dict(gender="female", group="interview", area="southern", emotion="surprised")]


def serve_requests(n: int):
    """The seeded request list of --workload serve: 1-3 sentences (the reference's sample texts are 51-148 characters,
    models/reference_samples.csv), five built-in voices, two speeds."""
    g = torch.Generator().manual_seed(SEED + 5)
    reqs = []
    for i in range(n):
        k = 1 + int(torch.randint(0, 3, (1,), generator=g))
        idx = torch.randperm(len(SERVE_SENTENCES), generator=g)[:k].tolist()
        reqs.append(dict(text=" ".join(SERVE_SENTENCES[j] for j in idx), speed=(0.9, 1.2)[int(torch.randint(0, 2, (1,), generator=g))],
                         voice=SERVE_VOICES[int(torch.randint(0, len(SERVE_VOICES), (1,), generator=g))], serial=i))
    return reqs


def serve(a):
    """SURVEY 8(f) N2 measured: what the batching front end SERVES.  16 client threads issue 96 requests (closed loop: a client sends its
    next request when the previous one returned) against ONE TTSEngine(model_spec=full, bf16) behind BatchingFrontend, next to the same
    requests issued one at a time through TTSEngine.synthesize -- the reference REST layer's behaviour: one shared engine, requests
    serialised (/root/reference/vietvoicetts/api/tts_engine.py:79-87).  Wall clock includes all host plumbing (voice selection, text
    cleaning, chunk plan, cross-fade)."""
    import tempfile
    import threading
    from vietvoice_tts_amd.batching import BatchingFrontend
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    reqs = serve_requests(a.requests)

    def pct(v, q):
        v = sorted(v)
        return v[min(len(v) - 1, int(round(q * (len(v) - 1))))]

    out = {}
    with tempfile.TemporaryDirectory() as d:
        cfg = ModelConfig(model_cache_dir=d, synthetic_model=True, model_spec=a.spec, acoustic_dtype=a.dtype, nfe_step=a.nfe, max_batch_chunks=a.batch)
        eng = TTSEngine(cfg)
        for v in SERVE_VOICES:                                   # warm-up: every voice enters the HBM voice bank, kernels are configured
            eng.config.speed = 0.9
            eng.synthesize(SERVE_SENTENCES[0], **v)
        # ---- serial: one request at a time through TTSEngine.synthesize (config.speed set per call, as the REST layer does)
        torch.cuda.synchronize()
        lat, tot = [], 0
        t0 = time.perf_counter()
        for r in reqs[: a.serial_requests]:
            ts = time.perf_counter()
            eng.config.speed = r["speed"]
            w, _ = eng.synthesize(r["text"], **r["voice"])
            lat.append(time.perf_counter() - ts)
            tot += w.size
        el = time.perf_counter() - t0
        out["serial"] = {"requests": len(lat), "audio_s_per_s": round(tot / 24000.0 / el, 2), "requests_per_s": round(len(lat) / el, 2),
                         "latency_p50_ms": round(pct(lat, 0.5) * 1e3, 1), "latency_p95_ms": round(pct(lat, 0.95) * 1e3, 1)}
        eng.config.speed = 0.9
        # ---- the front end, pipelined (overlap) and on one thread (the round-3 loop)
        for name, overlap in (("frontend_overlapped", True), ("frontend_single_thread", False)):
            fe = BatchingFrontend(eng, max_wait_ms=a.max_wait_ms, max_requests=a.batch, overlap=overlap)
            lat, sizes, lock, nxt = [], [], threading.Lock(), [0]

            def client():
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(reqs):
                        return
                    r = reqs[i]
                    ts = time.perf_counter()
                    w, _ = fe.submit(r["text"], speed=r["speed"], serial=r["serial"], **r["voice"]).result(timeout=600)
                    with lock:
                        lat.append(time.perf_counter() - ts)
                        sizes.append(w.size)
            ths = [threading.Thread(target=client) for _ in range(a.clients)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            el = time.perf_counter() - t0
            st = fe.stats()
            fe.close()
            out[name] = {"requests": len(lat), "audio_s_per_s": round(sum(sizes) / 24000.0 / el, 2), "requests_per_s": round(len(lat) / el, 2),
                         "latency_p50_ms": round(pct(lat, 0.5) * 1e3, 1), "latency_p95_ms": round(pct(lat, 0.95) * 1e3, 1),
                         "batches": st["batches"], "requests_per_batch": round(st["requests_per_batch"], 2), "chunks_per_batch": round(st["chunks_per_batch"], 2),
                         "frames_per_batch": round(st["frames_per_batch"], 1), "gpu_busy_frac": round(st["gpu_busy_s"] / el, 3)}
        eng.cleanup()
    best = out["frontend_overlapped"]
    print(json.dumps({"metric": "served audio-seconds/sec through the batching front end (SURVEY 8f N2)", "value": best["audio_s_per_s"], "unit": "audio-seconds/sec",
                      "n_gpus": 1, "steps": a.requests, "warmup": len(SERVE_VOICES), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                      "data": "synthetic voice pack + random-init weights; seeded request list",
                      "config": {"workload": f"serve: {a.requests} requests (1-3 sentences, 5 voices, 2 speeds), {a.clients} closed-loop client threads, "
                                             f"max_requests={a.batch}, max_wait_ms={a.max_wait_ms}", "spec": a.spec}, **out}), flush=True)


def cpu_baseline(spec, weights, nfe_step):
    """Oracle (kind 'port') on the host cores, bounded sample, rank 0 only."""
    from oracle.vv_oracle import Oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))        # the GPU box grants a 16-core CPU share per GPU; more threads only thrash
    torch.set_num_threads(threads)
    orc = Oracle(spec, weights, nfe_step=nfe_step)
    g = torch.Generator().manual_seed(SEED)
    audio = synth_reference_clip(0)
    ids = torch.randint(1, spec.vocab_size, (TEXT_TOKENS,), generator=g, dtype=torch.int32)
    N = REF_SAMPLES // spec.hop_length + 1 + GEN_FRAMES
    noise = torch.randn(N, spec.n_mel, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        pre = orc.preprocess(audio, ids, N, noise)
        t1 = time.perf_counter()
        n_timed = min(3, nfe_step - 1)              # ~12 s of CPU work at the full model size
        x = pre["noise"]
        for st_ in range(n_timed):
            x = orc.transformer_step(x, pre, st_)
        t2 = time.perf_counter()
        orc.decode(x, pre["ref_signal_len"])
        t3 = time.perf_counter()
    steps = nfe_step - 1
    est = (t1 - t0) + (t2 - t1) / n_timed * steps + (t3 - t2)
    audio_s = GEN_FRAMES * spec.hop_length / spec.sample_rate
    return {"value": round(audio_s / est, 5), "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
            "sample": f"1 utterance (N=1600): preprocess {t1 - t0:.2f}s + {n_timed} of {steps} Euler steps (both CFG branches) "
                      f"{t2 - t1:.2f}s scaled x{steps}/{n_timed} + vocoder {t3 - t2:.2f}s; torch CPU fp32 oracle",
            "measured_s": round(t3 - t0, 3), "estimated_full_s": round(est, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--nfe", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="time the K steps WITH the PCIe legs (pinned host inputs -> H2D ... PCM -> D2H, SURVEY 8d's wall-clock "
                    "definition; rounds 1-3 reported this as `value`).  Default: inputs and PCM resident in HBM; the PCIe-inclusive rate is then "
                    "measured on --pcie-steps extra steps and reported beside it")
    ap.add_argument("--pcie-steps", type=int, default=2, help="extra steps for the PCIe-inclusive figure (0 = skip)")
    ap.add_argument("--graph-steps", action="store_true", help="replay all Euler steps + the decode of a batch from ONE captured hipGraph "
                    "(runtime.GraphedSteps; the single-utterance latency experiment: --batch 1 --graph-steps)")
    ap.add_argument("--spec", default="full", choices=["full", "small", "tiny"])
    ap.add_argument("--requests", type=int, default=96, help="--workload serve: requests issued through the front end")
    ap.add_argument("--serial-requests", type=int, default=24, help="--workload serve: requests of the same list issued one at a time")
    ap.add_argument("--clients", type=int, default=16, help="--workload serve: closed-loop client threads")
    ap.add_argument("--max-wait-ms", type=float, default=5.0, help="--workload serve: the front end's collect window")
    ap.add_argument("--workload", default="batch32", choices=["batch32", "mixed256", "longform", "serve"],
                    help="batch32 = the headline metric (BASELINE configs[2]); mixed256 = configs[3] (32 ragged units per GPU); "
                         "longform = configs[4] (4k-char text through TTSEngine, 8 chunks in flight, hipGraph vocoder); "
                         "serve = the batching front end under 16 client threads next to serial TTSEngine.synthesize calls")
    a = ap.parse_args()
    if a.workload == "longform":
        return longform(a)
    if a.workload == "serve":
        return serve(a)

    backend = os.environ.get("VV_BENCH_DIST_BACKEND", "nccl")      # "gloo" = single-GPU rehearsal of the N>1 path (ranks share cuda:0)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU, as a CHILD process tree --
        # before anything in this process has touched the GPU (device_count() does not initialise it) and never by exec -- pass the
        # children's output through and exit with their return code.  Without this a bare --gpus 8 would run one rank and report n_gpus 1.
        import socket
        import subprocess
        if backend == "nccl" and torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the synthesis hot path has no CPU fallback")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # backend "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    spec = {"full": ModelSpec.full, "small": ModelSpec.small, "tiny": ModelSpec.tiny}[a.spec]()
    adt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    weights = make_synthetic_weights(spec, SEED) if rank == 0 else None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
    t0 = time.perf_counter()
    flat, _table = sharding.broadcast_weights(spec, adt, weights, device)      # C1: the only collective; RCCL over xGMI
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else None           # includes rank 0's pack + H2D
    eng = HipSynth(spec, None, device=str(device), acoustic_dtype=a.dtype, nfe_step=a.nfe, flat_weights=flat)
    lanes_opt = 0                                    # vv_set_option "lanes": 0 = auto (two lanes from 65,536 packed rows), 1 = one, 2 = two
    voc_x3 = a.dtype == "bf16"                       # the bf16 context's default (vv_set_option "voc_x3" -1): vocoder products as 3-way bf16 splits
    for opt in [o for o in os.environ.get("VV_BENCH_OPTIONS", "").split(",") if o]:      # A/B switches of the C ABI, e.g. rope_rows=0
        k, _, v = opt.partition("=")
        if k == "rope_theta":                            # 0 = read the rope tables instead of computing the angles in the QKV epilogue
            eng.set_rope_theta(float(v))
            continue
        eng.set_option(k, int(v))
        if k == "lanes":
            lanes_opt = int(v)
        if k == "voc_x3" and int(v) >= 0:
            voc_x3 = int(v) == 1
    if os.environ.get("VV_BENCH_DUMP_MAPS"):          # diagnostics for profiler-side crashes: the loaded images, so a raw stack can be symbolised
        with open("/proc/self/maps") as src, open(os.environ["VV_BENCH_DUMP_MAPS"], "w") as dst:
            dst.write(src.read())
    if a.workload == "mixed256":
        batches, audio_s_rank, nb, fill = make_mixed_inputs(spec, a.batch, rank, world, device)
    else:
        d, N = make_inputs(spec, a.batch, rank, device)
        batches, audio_s_rank, nb, fill = [(d, N, GEN_FRAMES)], a.batch * GEN_FRAMES * spec.hop_length / spec.sample_rate, a.batch, 1.0

    pcm_host = [None] * len(batches)          # pinned landing buffers for the D2H leg, allocated by the first (warm-up) step
    graphs = [None] * len(batches)            # --graph-steps: one captured (steps + decode) graph per batch shape

    def step(pcie=None):
        pcie = a.pcie if pcie is None else pcie
        """host inputs -> H2D -> three stages -> D2H of the PCM; returns when the PCM is on the host."""
        outs = []
        for i, (d, N, t_gen) in enumerate(batches):
            if pcie:
                audio, ids, noise = (d["host"][k].to(device, non_blocking=True) for k in ("audio", "ids", "noise"))
            else:
                audio, ids, noise = d["audio"], d["ids"], d["noise"]
            if a.graph_steps:
                pre = eng.preprocess(audio, d["audio_len"], ids, d["text_len"], d["seq_len"], N, seq_len_host=d["seq_len_host"])
                if graphs[i] is None:
                    graphs[i] = eng.capture_steps(d["seq_len"].numel(), N, d["seq_len_host"], t_gen)
                o = graphs[i](noise, pre) + (pre,)
            else:
                o = eng.synthesize_batch(audio, d["audio_len"], ids, d["text_len"], d["seq_len"], N, noise, t_gen, gen_frames=d.get("gen_frames"),
                                         seq_len_host=d.get("seq_len_host"))
            if pcie:
                if pcm_host[i] is None:
                    pcm_host[i] = (torch.empty(o[1].shape, dtype=o[1].dtype).pin_memory(), torch.empty(o[2].shape, dtype=o[2].dtype).pin_memory())
                pcm_host[i][0].copy_(o[1], non_blocking=True)
                pcm_host[i][1].copy_(o[2], non_blocking=True)
            outs.append(o)
        torch.cuda.synchronize()               # the step ends when the int16 PCM is on the host
        return outs

    for _ in range(a.warmup):
        step()
    if a.warmup and a.pcie_steps and not a.pcie:
        step(pcie=True)                                   # allocates the pinned landing buffers outside every timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ts = time.perf_counter()
        out = step()
        step_s.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_busy_s = sum(step_s)                              # this rank's own step time, without the waits at the barriers
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert abs(sum(float(o[2].sum().item()) for o in out) / spec.sample_rate - audio_s_rank) < 1e-3
    pcie_inclusive = None
    if a.pcie:                                            # what landed on the host is the PCM the device produced
        assert all(torch.equal(h[1], o[2].cpu()) and int(h[0].abs().max()) > 0 for h, o in zip(pcm_host, out))
    elif a.pcie_steps > 0 and dist is None:
        # the same step with its PCIe legs (pinned host int16 clips / ids / noise -> H2D, int16 PCM -> D2H to pinned memory): reported
        # beside `value`, never as `value` (the contract: inputs resident in HBM when the timed region starts)
        step(pcie=True)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(a.pcie_steps):
            outp = step(pcie=True)
        torch.cuda.synchronize()
        elp = time.perf_counter() - tp
        assert all(torch.equal(h[1], o[2].cpu()) and int(h[0].abs().max()) > 0 for h, o in zip(pcm_host, outp))
        pcie_inclusive = {"value": round(audio_s_rank * a.pcie_steps / elp, 3), "unit": "audio-seconds/sec", "steps": a.pcie_steps,
                          "ms_per_step": round(elp / a.pcie_steps * 1e3, 2),
                          "what": "the same step timed from pinned host inputs (H2D) to int16 PCM in pinned host memory (D2H): SURVEY 8d's wall-clock definition"}
    if dist is not None:                                  # total audio over all ranks (ragged shards differ)
        ta = torch.tensor([audio_s_rank], dtype=torch.float64, device=device)
        dist.all_reduce(ta, op=dist.ReduceOp.SUM)
        audio_s_all = float(ta.item())
    else:
        audio_s_all = audio_s_rank

    devices = [f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(local)}"]
    rows_rank = int(sum(int(b[0]["seq_len"].sum()) for b in batches))
    per_rank = [{"rank": rank, "ms_per_step": round(rank_busy_s / a.steps * 1e3, 2), "units": nb, "rows": rows_rank, "audio_s": round(audio_s_rank, 3)}]
    if dist is not None:                                  # which card every rank ran on (two ranks on one card = a rehearsal, and says so)
        got = [None] * world
        dist.all_gather_object(got, (devices[0], per_rank[0]))
        devices, per_rank = [g_[0] for g_ in got], [g_[1] for g_ in got]
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel-class timing with HIP events on the launch stream (one extra, untimed pass)
    # The timed steps run the Euler loop as two half batches on two streams (option "lanes": one lane's kernel tails are filled by the
    # other's kernels).  A kernel's own rate is measured with the chip to itself: the profiled pass runs ONE lane, so the class times
    # below are each kernel alone and their sum exceeds the two-lane step time by what the overlap recovers.
    eng.prof_enable(True)
    was_graph, a.graph_steps = a.graph_steps, False       # events cannot be recorded inside a replayed graph: the profiled pass runs eagerly
    eng.set_option("lanes", 1)
    torch.cuda.synchronize()
    t_cls = time.perf_counter()
    step()
    class_pass_ms = (time.perf_counter() - t_cls) * 1e3
    eng.set_option("lanes", lanes_opt)
    a.graph_steps = was_graph
    prof = eng.prof_collect()
    eng.prof_enable(False)
    gm = prof["gemm"]
    peak = MFMA_BF16_PEAK_TFLOPS if a.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
    # HBM-side bytes per launch: PMC counters cannot be read from inside the process, so the figure comes from the committed
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py).  Passes over THIS command die inside the profiler
    # (profiles/r02/gemm_notes.md, stack + maps committed), so the file is taken over tools/gemm_ab.py at the block's four shapes; the
    # file names the program it was taken over ("how_short"), and that string is what traffic_source reports.
    traffic, traffic_src = None, None
    here = os.path.dirname(os.path.abspath(__file__))
    for rel in ("profiles/r05/gemm_pmc_traffic.json", "profiles/r04/gemm_pmc_traffic.json", "profiles/r03/gemm_pmc_traffic.json", "profiles/r02/gemm_pmc_traffic.json", "profiles/r01/gemm_pmc_traffic.json"):
        tf = os.path.join(here, rel)
        if a.dtype == "bf16" and a.workload == "batch32" and a.batch == 32 and a.spec == "full" and os.path.exists(tf):
            with open(tf) as fh:
                tj = json.load(fh)
            traffic, traffic_src = tj["traffic_bytes_per_launch"], rel + " (" + tj.get("how_short", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction") + ")"
            break
    kname = "gemm_pp_kernel (K6, persistent 256x256x64 ping-pong, bf16 MFMA 16x16x32)" if a.dtype == "bf16" else "gemm_kernel (K6, f32 MFMA 32x32x2)"
    roofline = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches": gm["launches"], "avg_launch_ms": round(gm["ms"] / max(gm["launches"], 1), 4)}
    if a.dtype == "bf16":
        # peak is the nominal 2.4 GHz figure; measured with the stamp build on one box (not in this run): the chip holds this loop at 1.8-1.9 GHz on
        # random operands and at 2.23-2.35 GHz on all-zero operands (same cycles, 19 % less time)
        roofline["clock_note"] = "power-limited clock: profiles/r03/gemm_notes.md (power probe)"
    def _rf(name, bound, peak):
        v = prof[name]
        if not v["launches"] or v["ms"] <= 0:
            return None
        ach = (v["flops"] / 1e12 if bound == "mfma" else v["bytes"] / 1e9) / (v["ms"] * 1e-3)
        return {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                "frac": round(ach / peak, 4), "launches": v["launches"]}
    other_rooflines = {"attention (K8, bf16 MFMA)": _rf("attention", "mfma", peak),
                       "layernorm+residual (K4)": _rf("norm", "hbm", HBM_PEAK_GBS),
                       "pos-conv (K3, bf16 MFMA)": _rf("posconv", "mfma", peak),
                       # x3: six bf16 piece products per fp32 product, so the matrix roof for ALGORITHMIC flops is the bf16 peak / 6
                       ("vocoder convs (K11x: fp32 products as six bf16 MFMA terms; roof = bf16 peak / 6)" if voc_x3 else "vocoder convs (K11/K12, f32 MFMA)"):
                           _rf("voc_conv", "mfma", round(MFMA_BF16_PEAK_TFLOPS / 6, 1) if voc_x3 else MFMA_F32_PEAK_TFLOPS),
                       "vocoder convs, HBM side (algorithmic bytes per launch)": _rf("voc_conv", "hbm", HBM_PEAK_GBS),
                       "vocoder conv_post+tanh+int16 (K13)": _rf("voc_post", "hbm", HBM_PEAK_GBS)}
    # ---- the vocoder convs by stage: one entry per upsampler / MRF stack against the roof SURVEY 8(a) names for it
    # (K11 stages 0-1 and K12: matrix; K11 stages 2-3: HBM), with the other side beside it
    voc_peak = round(MFMA_BF16_PEAK_TFLOPS / 6, 1) if voc_x3 else MFMA_F32_PEAK_TFLOPS
    voc_stages = {}
    for name, bound in (("voc_pre", "hbm"), ("voc_up0", "mfma"), ("voc_up1", "mfma"), ("voc_up2", "hbm"), ("voc_up3", "hbm"),
                        ("voc_mrf0", "mfma"), ("voc_mrf1", "mfma"), ("voc_mrf2", "mfma"), ("voc_mrf3", "mfma")):
        v = prof.get(name)
        if not v or not v["launches"] or v["ms"] <= 0:
            continue
        tf, gb = v["flops"] / 1e12 / (v["ms"] * 1e-3), v["bytes"] / 1e9 / (v["ms"] * 1e-3)
        voc_stages[name] = {"bound": bound, "ms": round(v["ms"], 3), "launches": v["launches"], "TFLOP/s": round(tf, 2), "frac_mfma": round(tf / voc_peak, 4),
                            "GB/s": round(gb, 1), "frac_hbm": round(gb / HBM_PEAK_GBS, 4), "flop_per_byte": round(v["flops"] / max(v["bytes"], 1.0), 1)}
        if bound == "hbm" and name != "voc_pre":
            other_rooflines[f"vocoder upsample stage {name[-1]} (K11, HBM-bound by SURVEY 8a)"] = {
                "bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4), "launches": v["launches"]}
    # ---- K5 GroupNorm: not on the synthesis path of this architecture (the survey lists it "as used"); one micro-measurement so that it has a number
    if a.spec == "full":
        gB, gC, gT, gG = a.batch, 512, GEN_FRAMES, 32
        gx = torch.randn(gB, gC, gT, device=device)
        gy, gg, gb_ = torch.empty_like(gx), torch.ones(gC, device=device), torch.zeros(gC, device=device)
        st_ = torch.cuda.current_stream(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(25):
            if it == 5:
                e0.record(st_)
            eng._check(eng.lib.vv_groupnorm(eng.ctx, gx.data_ptr(), gy.data_ptr(), gg.data_ptr(), gb_.data_ptr(), gB, gC, gT, gG, 1e-5, 0, st_.cuda_stream))
        e1.record(st_)
        torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / 20
        g_gbs = 8.0 * gx.numel() / 1e9 / (g_ms * 1e-3)
        other_rooflines["groupnorm (K5; off the synthesis path, micro-measured)"] = {
            "bound": "hbm", "achieved": round(g_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g_gbs / HBM_PEAK_GBS, 4), "launches": 20,
            "shape": f"[{gB}, {gC}, {gT}] f32, {gG} groups; algorithmic bytes = one read + one write"}
    classes = {}
    for k, v in prof.items():
        if v["launches"] and not (k.startswith("voc_") and k not in ("voc_conv", "voc_post")):      # the stage classes repeat voc_conv: `vocoder_stages`
            classes[k] = {"ms": round(v["ms"], 2), "launches": v["launches"],
                          "TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                          "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
    total_audio = audio_s_all * a.steps
    res = {
        "metric": "audio-seconds/sec (24 kHz) at batch 32, 256-token utterances; RTF",
        "value": round(total_audio / elapsed, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 2), "median_ms_per_step": round(sorted(step_s)[len(step_s) // 2] * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic (seeded utterances and random-init weights; no checkpoint offline)",
        "rtf": round(elapsed / total_audio, 6),
        "timed_region": ("host (pinned) int16 clips / ids / noise -> H2D -> preprocess + Euler steps + vocoder -> int16 PCM D2H to pinned host memory (--pcie; SURVEY 8d)"
                         if a.pcie else "inputs (int16 clips, ids, noise) and int16 PCM resident in HBM: preprocess + Euler steps + vocoder; the PCIe-inclusive "
                                        "rate of the same step is in `pcie_inclusive`"),
        "hbm_target_note": "north_star's '>= 40 % of HBM roofline on the vocoder kernel': met by the kernels SURVEY 8(a) puts on the HBM roof that are "
                           "memory-bound in practice -- K13 conv_post+tanh+int16 (0.70-0.75) and the x2 up-sampler of stage 3 (0.41-0.44, round 5); stage 2 "
                           "(arithmetic intensity 64 flop/B, above the ~52 ridge of the six-term bf16 form) and every MRF conv (43-450 flop/B) are reported "
                           "against their matrix roof with the HBM side beside it: `vocoder_stages` (SURVEY 7 / 8d)",
        "vocoder_arith": ("fp32 in / out / accumulate; every product a*w taken as the six bf16 piece products >= 2^-16 |a w| of exact 3-way splits a = h+m+l, "
                          "w = h+m+l (v_mfma_f32_32x32x16_bf16, vv_vocoder_x3.hip): fp32 fidelity -- waveform max |err| vs the float64 oracle 7.6e-7 against "
                          "7.4e-7 for v_mfma_f32_32x32x2_f32 (tests/test_fullsize_gpu.py)") if voc_x3 else "fp32 (v_mfma_f32_32x32x2_f32)",
        "config": {"workload": (f"batch={a.batch} per GPU, 256-token utterances (N=1600 frames, 11.061 s generated each), "
                                if a.workload == "batch32" else
                                f"mixed256: {nb} ragged units on this rank of {a.batch * world} (64-512 tokens, 3-9 s reference clips) in "
                                f"{len(batches)} length-bucketed batches {[b[0]['seq_len'].numel() for b in batches]} (row fill {fill:.3f}), ")
                               + f"{a.dtype} acoustic + fp32 vocoder, nfe_step={a.nfe} ({a.nfe - 1} Euler steps x 2 CFG branches)",
                   "spec": a.spec, "global_batch": world * a.batch, "parallelism": f"dp{world} (independent utterances, weight broadcast only)"},
        "roofline": roofline, "other_rooflines": other_rooflines, "kernel_classes": classes, "vocoder_stages": voc_stages,
        "value_definition": "whole-job audio-seconds per wall second with the inputs resident in HBM when the timed region starts (the bench contract); "
                            "SURVEY 8(d)'s host-to-host wording of the same metric is `pcie_inclusive` (rounds 1-3 reported that one as `value`: "
                            "about 0.25 % lower)",
    }
    res["lanes"] = {"option": lanes_opt, "class_pass_ms": round(class_pass_ms, 2),
                    "what": "timed steps: the Euler steps run as two lanes (half batches, or the two CFG branches of a single item) on two HIP streams, "
                            "bit-identical to one lane (tests/test_e2e_gpu.py::test_two_lanes_equal_one_lane_bit_for_bit).  kernel_classes / rooflines: ONE extra "
                            "untimed pass on one lane with HIP events around every launch (class_pass_ms: its wall time, which the class ms add up to) -- each "
                            "kernel with the chip to itself; the two-lane step is shorter than that sum by the kernel tails the other lane fills"}
    if pcie_inclusive is not None:
        res["pcie_inclusive"] = pcie_inclusive
    res["devices"] = devices
    if a.graph_steps:
        res["graph_steps"] = "all Euler steps + decode replayed from one captured hipGraph per batch shape (kernel-class timings from an eager pass)"
    if world > 1:
        # what the driver's scaling run needs to be read: every rank's own time and shard (weak scaling: 32 units per rank; for mixed256 the
        # LPT shard sizes differ), the imbalance of the shard plan (max / mean of the ranks' frame rows and of their step times)
        ms_all, rows_all = [p["ms_per_step"] for p in per_rank], [p["rows"] for p in per_rank]
        res["per_rank"] = per_rank
        res["shard_imbalance"] = {"rows_max_over_mean": round(max(rows_all) / (sum(rows_all) / world), 4),
                                  "ms_max_over_mean": round(max(ms_all) / (sum(ms_all) / world), 4)}
        res["env"] = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if bcast_ms is not None:
        res["weight_pack_and_broadcast_ms"] = round(bcast_ms, 2)
        res["dist_backend"] = "nccl (RCCL)" if backend == "nccl" else backend + " (rehearsal: ranks may share a card)"
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(spec, weights, a.nfe)
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
