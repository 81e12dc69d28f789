"""Request-batching front end: coalesce concurrent ``synthesize`` calls into GPU batches.

The reference REST layer pushes each request onto a worker thread against ONE engine that is not thread safe and
mutates ``engine.config.speed`` around the call (api/tts_engine.py:64-91).  Here every request carries its own speed,
all chunks of all waiting requests are flattened into ragged GPU batches (different reference clips per item are
fine: per-item lengths live on the device), and each request gets its own cross-faded PCM back through a Future.
Noise is drawn from a per-request generator seeded with (config.random_seed, request serial), so a request's audio
does not depend on which other requests happened to share its batch.
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from typing import List, Optional, Tuple

import numpy as np


class BatchingFrontend:
    def __init__(self, engine, max_wait_ms: float = 5.0, max_requests: int = 16):
        self.engine = engine
        self.max_wait = max_wait_ms / 1e3
        self.max_requests = max_requests
        self._q: "queue.Queue" = queue.Queue()
        self._serial = 0
        self._serial_lock = threading.Lock()
        self._stop = False
        self.batches_run = 0
        self.requests_done = 0
        self._worker = threading.Thread(target=self._loop, name="vvtts-batcher", daemon=True)
        self._worker.start()

    def submit(self, text: str, speed: Optional[float] = None, serial: Optional[int] = None, **voice) -> Future:
        """voice: gender / group / area / emotion / sample_iteration / reference_audio / reference_text.
        ``serial`` fixes the request's noise stream (default: arrival counter)."""
        fut: Future = Future()
        with self._serial_lock:
            if serial is None:
                serial = self._serial
            self._serial = max(self._serial, serial) + 1
        self._q.put((serial, text, speed, voice, fut))
        return fut

    def synthesize(self, text: str, speed: Optional[float] = None, **voice) -> Tuple[np.ndarray, float]:
        return self.submit(text, speed, **voice).result()

    def close(self):
        self._stop = True
        self._q.put(None)
        self._worker.join(timeout=60)

    # ------------------------------------------------------------------ worker
    def _collect(self) -> List[tuple]:
        first = self._q.get()
        if first is None:
            return []
        reqs = [first]
        deadline = time.monotonic() + self.max_wait
        while len(reqs) < self.max_requests:
            left = deadline - time.monotonic()
            if left <= 0:
                break
            try:
                nxt = self._q.get(timeout=left)
            except queue.Empty:
                break
            if nxt is None:
                self._stop = True
                break
            reqs.append(nxt)
        return reqs

    def _loop(self):
        import torch
        eng = self.engine
        while not self._stop:
            reqs = self._collect()
            if not reqs:
                continue
            t0 = time.time()
            plans, flat, blocks = [], [], []
            n_mel = eng.model_session_manager.spec.n_mel
            for serial, text, speed, voice, fut in reqs:
                try:
                    ref_audio, ref_text = eng.model_session_manager.select_sample(
                        voice.get("gender"), voice.get("group"), voice.get("area"), voice.get("emotion"), voice.get("sample_iteration"),
                        voice.get("reference_audio"), voice.get("reference_text"))
                except Exception as e:          # propagate unwrapped, as TTSEngine.synthesize does (tts_engine.py:217)
                    fut.set_exception(e)
                    continue
                try:
                    inputs = eng._prepare_inputs(ref_audio, ref_text, text, speed=speed)
                except Exception as e:
                    fut.set_exception(RuntimeError(f"Speech synthesis failed: {e}"))
                    continue
                gen = torch.Generator().manual_seed(eng.config.random_seed * 1000003 + serial)
                plans.append((fut, len(inputs)))
                flat.extend(inputs)
                blocks.extend(torch.randn((int(i[2][0]), n_mel), generator=gen, dtype=torch.float32) for i in inputs)
            if not flat:
                continue
            try:
                with eng._lock:
                    if eng.model_session_manager.engine is not None:
                        waves = eng._synthesize_device(flat, noise_blocks=blocks)
                    else:
                        waves = eng._synthesize_sessions(flat)      # CPU plumbing tests: the oracle sessions draw their own noise
                self.batches_run += 1
                pos = 0
                for fut, n in plans:
                    final = eng.audio_processor.concatenate_with_crossfade_improved(waves[pos: pos + n], eng.config.cross_fade_duration,
                                                                                    eng.config.sample_rate)
                    pos += n
                    self.requests_done += 1
                    fut.set_result((final, time.time() - t0))
            except Exception as e:
                for fut, _n in plans:
                    if not fut.done():
                        fut.set_exception(RuntimeError(f"Speech synthesis failed: {e}"))
