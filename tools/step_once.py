#!/usr/bin/env python3
"""One headline step and nothing else, for counter passes over the REAL step (VERDICT r4 #6d):

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 <repo>/tools/step_once.py [--batch 32]

bench.py under `rocprofv3 --pmc` died inside librocprofiler-sdk in round 2 (profiles/r02/pmc_fetch_sigsegv_stack.txt); this script
removes everything that run had around the step: no torch.distributed import, no launcher, no pinned-memory legs, no HIP-event
profiling pass, no CPU oracle.  It builds HipSynth on the seeded synthetic weights, runs one untimed warm-up step and ONE
synthesize_batch, prints a line and exits.  tools/pmc_traffic.py aggregates the per-dispatch CSV by kernel name."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights  # noqa: E402
from vietvoice_tts_amd.runtime import HipSynth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--nfe", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=1, help="1 = every kernel with the chip to itself (what the per-kernel counters should describe)")
    a = ap.parse_args()
    SEED, REF, TOK, GEN = 9527, 144000, 256, 1037
    spec = ModelSpec.full()
    eng = HipSynth(spec, make_synthetic_weights(spec, SEED), acoustic_dtype="bf16", nfe_step=a.nfe)
    eng.set_option("lanes", a.lanes)
    g = torch.Generator().manual_seed(SEED)
    B, dev = a.batch, "cuda:0"
    N = REF // spec.hop_length + 1 + GEN
    audio = (torch.randn(B, REF, generator=g) * 4000).clamp(-29000, 29000).to(torch.int16).to(dev)
    ids = torch.randint(1, spec.vocab_size, (B, TOK), generator=g, dtype=torch.int32).to(dev)
    noise = torch.randn(B, N, spec.n_mel, generator=g).to(dev)
    i32 = lambda v: torch.full((B,), v, dtype=torch.int32, device=dev)
    for it in range(a.warmup + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _x, pcm, pcm_len, _pre = eng.synthesize_batch(audio, i32(REF), ids, i32(TOK), i32(N), N, noise, GEN, seq_len_host=[N] * B)
        torch.cuda.synchronize()
        print(f"step {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, {int(pcm_len.sum())} samples", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
