"""Request-batching front end: coalesce concurrent ``synthesize`` calls into GPU batches.

The reference REST layer pushes each request onto a worker thread against ONE engine that is not thread safe and
mutates ``engine.config.speed`` around the call (api/tts_engine.py:64-91).  Here every request carries its own speed,
all chunks of all waiting requests are flattened into ragged GPU batches (different reference clips per item are
fine: per-item lengths live on the device), and each request gets its own cross-faded PCM back through a Future.
Noise is drawn from a per-request generator seeded with (config.random_seed, request serial), and every kernel of the
path is row- or sequence-local with one arithmetic whatever the launch size (round 4: the split-K tail is off, the two
GEMM kernels share their epilogue arithmetic), so a request's audio does not depend on which other requests happened to
share its batch -- bit for bit (tests/test_engine_gpu.py::test_batching_frontend_batch_composition_invariance*).

Round 4: the loop is a three-stage pipeline instead of collect -> GPU -> cross-fade -> collect on one thread:

  preparer   collects requests, runs the host side of each (``select_sample``, ``_prepare_inputs``: text cleaning, chunk plan,
             voice-bank lookup; the per-request noise draw) and hands a prepared batch to the GPU stage.  While the GPU stage is
             busy it KEEPS COLLECTING into the waiting batch (up to ``max_requests``), so the batch that runs next holds
             everything that arrived during the previous one: under load the collect window is the GPU time of the batch in
             front, not ``max_wait_ms``.
  gpu        one ragged batch after the other through ``TTSEngine._synthesize_device`` (serialised by the engine lock).
  finisher   per-request cross-fade (host numpy, audio_processor.py:122-192) and Future completion, off the GPU stage's path.

``overlap=False`` keeps the three stages on one thread in the old order (the A/B baseline and the equality test).
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from typing import List, Optional, Tuple

import numpy as np


class _Batch:
    __slots__ = ("plans", "flat", "blocks", "t0", "n_req")

    def __init__(self):
        self.plans, self.flat, self.blocks, self.t0, self.n_req = [], [], [], time.time(), 0


class BatchingFrontend:
    def __init__(self, engine, max_wait_ms: float = 5.0, max_requests: int = 16, overlap: bool = True):
        self.engine = engine
        self.max_wait = max_wait_ms / 1e3
        self.max_requests = max_requests
        self.overlap = bool(overlap)
        self._q: "queue.Queue" = queue.Queue()
        self._ready: "queue.Queue" = queue.Queue(maxsize=1)      # prepared batches waiting for the GPU stage
        self._done: "queue.Queue" = queue.Queue()                # synthesised batches waiting for their cross-fades
        self._serial = 0
        self._serial_lock = threading.Lock()
        self._closed = False           # close() has been called: submit() refuses (set under _serial_lock)
        self._stage_down = False       # the collecting stage has ended (normally or not): nothing will take a request off the queue any more
        self._closing = False          # the close() sentinel has been taken off the request queue
        self.batches_run = 0
        self.requests_done = 0
        self.chunks_run = 0
        self.frames_run = 0
        self.gpu_busy_s = 0.0
        if self.overlap:
            self._threads = [threading.Thread(target=self._prep_loop, name="vvtts-batcher-prep", daemon=True),
                             threading.Thread(target=self._gpu_loop, name="vvtts-batcher-gpu", daemon=True),
                             threading.Thread(target=self._finish_loop, name="vvtts-batcher-finish", daemon=True)]
        else:
            self._threads = [threading.Thread(target=self._serial_loop, name="vvtts-batcher", daemon=True)]
        for t in self._threads:
            t.start()

    def submit(self, text: str, speed: Optional[float] = None, serial: Optional[int] = None, **voice) -> Future:
        """voice: gender / group / area / emotion / sample_iteration / reference_audio / reference_text.
        ``serial`` fixes the request's noise stream (default: arrival counter)."""
        fut: Future = Future()
        with self._serial_lock:           # close() raises `_closed` under the same lock: a request is either queued in front of the
            if self._closed or self._stage_down:      # sentinel (and served) or refused here -- never orphaned behind it
                fut.set_exception(RuntimeError("Speech synthesis failed: the batching front end is closed"))
                return fut
            if serial is None:
                serial = self._serial
            self._serial = max(self._serial, serial) + 1
            self._q.put((serial, text, speed, voice, fut))
        return fut

    def synthesize(self, text: str, speed: Optional[float] = None, **voice) -> Tuple[np.ndarray, float]:
        return self.submit(text, speed, **voice).result()

    def close(self):
        """Drain and stop: requests submitted before the call are still served (the sentinel queues up behind them); a request that
        could not be (a stage died, the join timed out) gets an exception instead of a Future that never completes."""
        with self._serial_lock:
            if self._closed:
                return
            self._closed = True
            self._q.put(None)
        for t in self._threads:
            t.join(timeout=120)
        while True:
            try:
                req = self._q.get_nowait()
            except queue.Empty:
                break
            if req is not None and not req[4].done():
                req[4].set_exception(RuntimeError("Speech synthesis failed: the batching front end is closed"))

    def stats(self) -> dict:
        b = max(self.batches_run, 1)
        return {"batches": self.batches_run, "requests": self.requests_done, "requests_per_batch": self.requests_done / b,
                "chunks_per_batch": self.chunks_run / b, "frames_per_batch": self.frames_run / b, "gpu_busy_s": self.gpu_busy_s}

    # ------------------------------------------------------------------ stage 1: collect + host preparation
    def _prepare_one(self, req, batch: _Batch) -> None:
        import torch
        eng = self.engine
        serial, text, speed, voice, fut = req
        try:
            ref_audio, ref_text = eng.model_session_manager.select_sample(
                voice.get("gender"), voice.get("group"), voice.get("area"), voice.get("emotion"), voice.get("sample_iteration"),
                voice.get("reference_audio"), voice.get("reference_text"))
        except Exception as e:          # propagate unwrapped, as TTSEngine.synthesize does (tts_engine.py:217)
            fut.set_exception(e)
            return
        try:
            inputs = eng._prepare_inputs(ref_audio, ref_text, text, speed=speed)
        except Exception as e:
            fut.set_exception(RuntimeError(f"Speech synthesis failed: {e}"))
            return
        try:
            n_mel = eng.model_session_manager.spec.n_mel
            gen = torch.Generator().manual_seed(eng.config.random_seed * 1000003 + serial)
            blocks = [torch.randn((int(i[2][0]), n_mel), generator=gen, dtype=torch.float32) for i in inputs]
        except Exception as e:          # noqa: BLE001  (nothing of this request has entered the batch yet)
            fut.set_exception(RuntimeError(f"Speech synthesis failed: {e}"))
            return
        batch.plans.append((fut, len(inputs)))
        batch.flat.extend(inputs)
        batch.blocks.extend(blocks)
        batch.n_req += 1

    def _collect(self) -> Optional[_Batch]:
        """Block for the first request, then take what arrives within ``max_wait`` -- and whatever is ALREADY queued when the window
        closes (the window bounds the wait for arrivals, not the time spent on requests that are there) -- up to ``max_requests``;
        the host preparation of the collected requests follows.  None = the close() sentinel arrived with nothing collected."""
        if self._closing:
            return None
        first = self._q.get()
        if first is None:
            self._closing = True
            return None
        reqs = [first]
        deadline = time.monotonic() + self.max_wait
        while len(reqs) < self.max_requests:
            left = deadline - time.monotonic()
            try:
                nxt = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if nxt is None:
                self._closing = True
                break
            reqs.append(nxt)
        batch = _Batch()
        for k, r in enumerate(reqs):
            try:
                self._prepare_one(r, batch)
            except BaseException as e:    # noqa: BLE001  (_prepare_one reports its own failures; this is the last line of defence)
                if not r[4].done():
                    r[4].set_exception(RuntimeError(f"Speech synthesis failed: {e!r}"))
                if not isinstance(e, Exception):          # the stage is going down: nothing taken off the queue may be left pending
                    for later in reqs[k + 1:]:
                        later[4].set_exception(RuntimeError(f"Speech synthesis failed: {e!r}"))
                    self._fail(batch, e)
                    raise
        return batch

    @staticmethod
    def _fail(batch: Optional[_Batch], err) -> None:
        if batch is not None:
            for fut, _n in batch.plans:
                if not fut.done():
                    fut.set_exception(RuntimeError(f"Speech synthesis failed: {err}"))

    def _prep_loop(self):
        self._in_hand = None
        try:
            self._prep_body()
        except BaseException as e:        # noqa: BLE001  (the batch being handed over dies with the stage, loudly)
            self._closing = True
            self._fail(self._in_hand, e)
            self._drain_requests(e)
            if not isinstance(e, Exception):
                raise
        finally:
            with self._serial_lock:        # from here on submit() refuses; whatever raced in before is failed, not stranded
                self._stage_down = True
            self._drain_requests("the batching front end has stopped")
            self._ready.put(None)          # whatever happened here, the stages behind must see the end of the stream

    def _prep_body(self):
        while True:
            try:
                batch = self._collect()
            except Exception as e:        # noqa: BLE001  (an unexpected failure while collecting: stop taking requests)
                self._closing = True
                self._drain_requests(e)
                return
            if batch is None:
                break
            self._in_hand = batch
            # hand over; while the GPU stage still has a batch waiting in front, keep filling this one (a collected batch is always
            # delivered, also when close() arrives meanwhile)
            while batch.flat:
                try:
                    self._ready.put(batch, timeout=0.002)
                    self._in_hand = None
                    break
                except queue.Full:
                    pass
                if batch.n_req < self.max_requests and not self._closing:
                    try:
                        nxt = self._q.get(timeout=0.002)
                    except queue.Empty:
                        continue
                    if nxt is None:
                        self._closing = True
                    else:
                        self._prepare_one(nxt, batch)

    def _drain_requests(self, err) -> None:
        while True:
            try:
                req = self._q.get_nowait()
            except queue.Empty:
                return
            if req is not None and not req[4].done():
                req[4].set_exception(RuntimeError(f"Speech synthesis failed: {err}"))

    # ------------------------------------------------------------------ stage 2: the GPU
    def _run_batch(self, batch: _Batch):
        eng = self.engine
        t0 = time.perf_counter()
        with eng._lock:
            if eng.model_session_manager.engine is not None:
                waves = eng._synthesize_device(batch.flat, noise_blocks=batch.blocks)
            else:
                waves = eng._synthesize_sessions(batch.flat)      # CPU plumbing tests: the oracle sessions draw their own noise
        self.gpu_busy_s += time.perf_counter() - t0
        self.batches_run += 1
        self.chunks_run += len(batch.flat)
        self.frames_run += sum(int(i[2][0]) for i in batch.flat)
        return waves

    def _gpu_loop(self):
        try:
            while True:
                batch = self._ready.get()
                if batch is None:
                    break
                try:
                    self._done.put((batch, self._run_batch(batch), None))
                except BaseException as e:        # noqa: BLE001
                    self._done.put((batch, None, e))
                    if not isinstance(e, Exception):
                        raise
        finally:
            self._done.put(None)

    # ------------------------------------------------------------------ stage 3: cross-fade + completion
    def _finish(self, batch: _Batch, waves, err) -> None:
        eng = self.engine
        if err is not None:
            for fut, _n in batch.plans:
                if not fut.done():
                    fut.set_exception(RuntimeError(f"Speech synthesis failed: {err}"))
            return
        pos = 0
        for fut, n in batch.plans:
            try:
                final = eng.audio_processor.concatenate_with_crossfade_improved(waves[pos: pos + n], eng.config.cross_fade_duration,
                                                                                eng.config.sample_rate)
                fut.set_result((final, time.time() - batch.t0))
            except Exception as e:        # noqa: BLE001
                fut.set_exception(RuntimeError(f"Speech synthesis failed: {e}"))
            pos += n
            self.requests_done += 1

    def _finish_loop(self):
        while True:
            item = self._done.get()
            if item is None:
                break
            try:
                self._finish(*item)
            except Exception as e:        # noqa: BLE001  (never leave a Future of the batch in hand pending)
                self._fail(item[0], e)

    # ------------------------------------------------------------------ overlap=False: the three stages in order on one thread
    def _serial_loop(self):
        batch = None
        try:
            while True:
                batch = self._collect()
                if batch is None:
                    break
                if not batch.flat:
                    continue
                try:
                    waves, err = self._run_batch(batch), None
                except Exception as e:    # noqa: BLE001
                    waves, err = None, e
                self._finish(batch, waves, err)
        except BaseException as e:        # noqa: BLE001  (same promise as the pipelined stages: no Future left pending)
            self._fail(batch, e)
            self._drain_requests(e)
            if not isinstance(e, Exception):
                raise
        finally:
            with self._serial_lock:
                self._stage_down = True
            self._drain_requests("the batching front end has stopped")
