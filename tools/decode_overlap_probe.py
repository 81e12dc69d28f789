"""Would the vocoder stage gain from running as TWO half batches on two HIP streams (as the Euler steps do)?  Diagnostic, no product change:
one context decodes B = 32 items on one stream; two contexts decode 16 items each from two threads on two streams; alternated."""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
spec = bench.ModelSpec.full()
w = bench.make_synthetic_weights(spec, bench.SEED)
flat, _ = bench.sharding.broadcast_weights(spec, torch.bfloat16, w, dev)
engs = [bench.HipSynth(spec, None, device="cuda:0", acoustic_dtype="bf16", nfe_step=32, flat_weights=flat) for _ in range(3)]
B, N, GEN = 32, 1600, bench.GEN_FRAMES
g = torch.Generator().manual_seed(1)
x = torch.randn(B, N, spec.n_mel, generator=g).to(dev)
pre = {"ref_signal_len": torch.full((B,), N - GEN, dtype=torch.int32, device=dev), "seq_len": torch.full((B,), N, dtype=torch.int32, device=dev)}
half = lambda lo, hi: (x[lo:hi].contiguous(), {k: v[lo:hi].contiguous() for k, v in pre.items()})
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def whole():
    p, n = engs[0].decode(x, pre, GEN)
    torch.cuda.synchronize()
    return p


def overlapped():
    outs = [None, None]

    def run(i):
        xi, pi = half(i * 16, i * 16 + 16)
        with torch.cuda.stream(streams[i]):
            outs[i] = engs[1 + i].decode(xi, pi, GEN)[0]
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return torch.cat(outs)


ref = whole(); got = overlapped()
print("equal PCM:", bool(torch.equal(ref, got)))
for rnd in range(3):
    for name, fn in (("whole", whole), ("two streams", overlapped)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        print(f"round {rnd} {name:12s} {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per decode of 32 items", flush=True)
