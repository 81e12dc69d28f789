"""TTSEngine -- same class surface as the reference engine (vietvoicetts/core/tts_engine.py:17-267):
``__init__(config)``, ``synthesize(text, gender, group, area, emotion, sample_iteration,
output_path, reference_audio, reference_text) -> (int16 PCM, seconds)``, context manager,
``cleanup``, ``validate_configuration`` -- and the same host arithmetic in ``_prepare_inputs``
(duration model, chunk plan, frames = samples // hop + 1; :43-131, pinned by golden vectors).

What changes is where the work runs.  The reference walks the chunks one by one and pays
1 + 31 + 1 ``session.run`` host round trips per chunk (:225-238, 157-172).  Here all chunks of a text
(they are independent until the final cross-fade, :244-246) go to the GPU as ONE ragged batch whose
state stays in HBM for all Euler steps; only int16 PCM comes back.  ``_run_preprocess`` /
``_run_transformer_steps`` / ``_run_decode`` remain for reference-style callers and drive the
session objects exactly as the reference does.

Error conventions follow the reference: select_sample errors propagate unwrapped (:217), anything in
the per-chunk work becomes RuntimeError("Speech synthesis failed: ...") (:256-257).  Calls are
serialised by a lock: the REST layer enters from several worker threads (api/tts_engine.py:79-87).
"""
from __future__ import annotations

import logging
import threading
import time
from typing import List, Optional, Tuple

import numpy as np

from .audio_processor import AudioProcessor
from .model import ModelSessionManager
from .model_config import ModelConfig
from .text_processor import TextProcessor

logger = logging.getLogger("vietvoicetts")


class TTSEngine:
    def __init__(self, config: Optional[ModelConfig] = None, session_factory=None):
        self.config = config or ModelConfig()
        self.model_session_manager = ModelSessionManager(self.config, session_factory=session_factory)
        self.model_session_manager.load_models()
        if not self.model_session_manager.vocab_path:
            raise RuntimeError("Vocabulary file not found in model tar archive")
        self.text_processor = TextProcessor(self.model_session_manager.vocab_path)
        self.audio_processor = AudioProcessor()
        self.sample_cache = {}
        self.voice_bank = None                    # N3: device-resident reference clips (HIP engine only)
        if self.model_session_manager.engine is not None:
            from ..voice_bank import VoiceBank
            self.voice_bank = VoiceBank(self.model_session_manager.engine, self.config.sample_rate)
        self._lock = threading.Lock()
        self._decode_graphs = None               # DecodeGraphCache, created with the first captured decode (use_hip_graph)
        self._last_plan = []

    def cleanup(self) -> None:
        if self._decode_graphs is not None:       # captured graphs point into the context about to be destroyed, and pin HBM
            self._decode_graphs.clear()
            self._decode_graphs = None
        if self.model_session_manager:
            self.model_session_manager.cleanup()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.cleanup()

    # ------------------------------------------------------------------ host arithmetic
    def _chunk_seconds(self, chunk: str, rate: float, speed: float) -> float:
        n = self.text_processor.calculate_text_length(chunk, self.config.pause_punctuation)
        return max(n / rate / speed, self.config.min_target_duration)

    def _prepare_inputs(self, reference_audio_path_or_bytes, reference_text: str, target_text: str,
                        speed: Optional[float] = None) -> List[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        cfg = self.config
        speed = cfg.speed if speed is None else speed
        if getattr(self, "voice_bank", None) is not None:   # decoded / resampled / normalised once on the GPU, then cached in HBM
            audio = self.voice_bank.get(reference_audio_path_or_bytes).pcm_host.reshape(1, 1, -1)
        else:
            audio = self.audio_processor.load_audio(reference_audio_path_or_bytes, cfg.sample_rate).reshape(1, 1, -1)
        reference_text = self.text_processor.clean_text(reference_text)
        target_text = self.text_processor.clean_text(target_text)

        n_samples = audio.shape[-1]
        # The reference admits any clip (core/audio_processor.py:15-26, core/tts_engine.py:46-56) and leaves a too-short one to the
        # preprocess graph.  Its mel front end is a centred STFT, which reflects n_fft / 2 samples at both ends: defined only for
        # clips of more than n_fft / 2 samples.  Refused here, before anything is launched, in the wording of the reference's
        # other reference-audio error (:73); `synthesize` wraps it like every error of the per-chunk work (:256-257).
        spec = getattr(getattr(self, "model_session_manager", None), "spec", None)
        min_samples = (spec.n_fft // 2 + 1) if spec is not None else 1
        if n_samples < min_samples:
            raise ValueError(f"Reference audio is too short ({n_samples} samples, {n_samples / cfg.sample_rate:.3f}s): "
                             f"at least {min_samples} samples ({min_samples / cfg.sample_rate:.3f}s) are needed")
        ref_frames = n_samples // cfg.hop_length + 1
        ref_seconds = n_samples / cfg.sample_rate
        ref_units = self.text_processor.calculate_text_length(reference_text, cfg.pause_punctuation)
        rate = ref_units / ref_seconds if ref_seconds > 0 else 100          # text units per second of the voice
        total = ref_seconds + self._chunk_seconds(target_text, rate, speed)

        if total <= cfg.max_chunk_duration:
            chunks = [target_text]
        else:
            budget = cfg.max_chunk_duration - ref_seconds - 1.0             # 1 s safety margin
            if budget <= 0:
                raise ValueError(f"Reference audio duration ({ref_seconds:.1f}s) exceeds max chunk duration ({cfg.max_chunk_duration}s)")
            chunks = []
            for piece in self.text_processor.chunk_text(target_text, max_chars=int(rate * budget * speed)):
                secs = self._chunk_seconds(piece, rate, speed)
                if ref_seconds + secs <= cfg.max_chunk_duration:
                    chunks.append(piece)
                else:                                                         # still too long: split again, 10 % tighter
                    chunks.extend(self.text_processor.chunk_text(piece, max_chars=int(len(piece) * budget / secs * 0.9)))
            logger.info("Long text (estimated %.1fs) split into %d chunks", total, len(chunks))

        prepared = []
        for piece in chunks:
            secs = self._chunk_seconds(piece, rate, speed)
            frames = ref_frames + int(secs * cfg.sample_rate) // cfg.hop_length + 1
            ids = self.text_processor.text_to_indices([list(reference_text + piece)])
            prepared.append((audio, ids, np.array([frames], dtype=np.int64), np.array([0], dtype=np.int32)))
        return prepared

    # ------------------------------------------------------------------ reference-style stage drivers
    def _run_preprocess(self, audio: np.ndarray, text_ids: np.ndarray, max_duration: np.ndarray):
        m = self.model_session_manager
        names = m.input_names["preprocess"]
        return m.sessions["preprocess"].run(m.output_names["preprocess"], {names[0]: audio, names[1]: text_ids, names[2]: max_duration})

    def _run_transformer_steps(self, noise, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, cat_mel_text, cat_mel_text_drop, time_step):
        m = self.model_session_manager
        names, outs, sess = m.input_names["transformer"], m.output_names["transformer"], m.sessions["transformer"]
        for _ in range(0, self.config.nfe_step - 1, self.config.fuse_nfe):
            noise, time_step = sess.run(outs, dict(zip(names, (noise, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k,
                                                               cat_mel_text, cat_mel_text_drop, time_step))))
        return noise, time_step

    def _run_decode(self, noise: np.ndarray, ref_signal_len: np.ndarray) -> np.ndarray:
        m = self.model_session_manager
        names = m.input_names["decode"]
        return m.sessions["decode"].run(m.output_names["decode"], {names[0]: noise, names[1]: ref_signal_len})[0]

    def _synthesize_sessions(self, inputs_list) -> List[np.ndarray]:
        waves = []
        for audio, text_ids, max_duration, time_step in inputs_list:
            pre = self._run_preprocess(audio, text_ids, max_duration)
            noise, _ = self._run_transformer_steps(*pre[:7], time_step)
            waves.append(self._run_decode(noise, pre[7]))
        return waves

    # ------------------------------------------------------------------ device-resident batched path
    def _synthesize_device(self, inputs_list, noise_blocks=None) -> List[np.ndarray]:
        """inputs_list items are (audio (1,1,S_i), text_ids (1,T_i), max_duration (1,), time_step); the reference
        clips may differ per item (cross-request batches).  One ragged GPU batch per ``max_batch_chunks`` items.
        noise_blocks: optional pre-drawn (N_i, n_mel) fp32 tensors, one per item (the batching front end draws them from
        per-request generators); default = the manager's seeded stream, in item order, like the session path."""
        import torch
        m = self.model_session_manager
        eng, spec = m.engine, m.spec
        dev = eng.device
        eng.set_nfe(self.config.nfe_step)
        hop = self.config.hop_length
        from ..sharding import plan_batches
        n_items = len(inputs_list)
        seq_all = [int(g[2][0]) for g in inputs_list]
        if noise_blocks is None:      # the same seeded stream the session path draws from: one (N_i, n_mel) block per chunk, in item order
            noise_blocks = [torch.randn((n, spec.n_mel), generator=m.noise_gen, dtype=torch.float32) for n in seq_all]
        waves: List[Optional[np.ndarray]] = [None] * n_items
        # the device packs ragged rows, so padding is free for the acoustic stages; sorting by length still puts similar
        # lengths into the same batch when a text has more chunks than max_batch_chunks (vocoder planes are padded)
        groups = []
        row_cap = eng.max_rows_per_call()          # the packed qkv buffer of one call stays below 2 GiB
        for idx in plan_batches(seq_all, max(1, int(self.config.max_batch_chunks)), pad_frac=1.0):
            cur, rows = [], 0
            for i in idx:
                if cur and rows + seq_all[i] > row_cap:
                    groups.append(cur)
                    cur, rows = [], 0
                cur.append(i)
                rows += seq_all[i]
            groups.append(cur)
        for idx in groups:
            group = [inputs_list[i] for i in idx]
            B = len(group)
            lens_a = np.array([g[0].shape[-1] for g in group], dtype=np.int32)
            lens_t = np.array([g[1].shape[1] for g in group], dtype=np.int32)
            # the audio plane is at least n_fft wide (vv_preprocess's contract): a lone clip of n_fft/2 + 1 ... n_fft - 1 samples is
            # admitted by _prepare_inputs, its zeros past audio_len are never read (the mel front end reflects inside audio_len)
            S, T = max(int(lens_a.max()), int(spec.n_fft)), int(lens_t.max())
            ids = np.zeros((B, T), dtype=np.int32)
            for i, g in enumerate(group):
                ids[i, : lens_t[i]] = g[1][0]
            banked = [self.voice_bank.entry_for_host(g[0]) if self.voice_bank is not None else None for g in group]
            if all(e is not None for e in banked):               # clips already live in HBM: assemble the batch device-side
                audio = torch.zeros((B, S), dtype=torch.int16, device=dev)
                for i, e in enumerate(banked):
                    audio[i, : lens_a[i]] = e.pcm_dev
            else:
                audio_np = np.zeros((B, S), dtype=np.int16)
                for i, g in enumerate(group):
                    audio_np[i, : lens_a[i]] = g[0].reshape(-1)
                audio = torch.from_numpy(audio_np).to(dev)
            seq = np.array([seq_all[i] for i in idx], dtype=np.int32)
            ref_frames = lens_a // hop + 1
            N = int(seq.max())
            t_gen = int((seq - ref_frames).max())
            if self.config.use_hip_graph:
                # fixed buckets (frames to multiples of 128, generated frames to multiples of 64) so that one captured vocoder graph
                # serves every chunk group and nearby reference-clip lengths; the decode masks every item by its own lengths
                from ..runtime import DecodeGraphCache
                Nb = DecodeGraphCache.bucket(N, 0)[0]
                N, t_gen = DecodeGraphCache.bucket(Nb, Nb - int(ref_frames.min()))      # every chunk group of a text lands on one key
            noise = torch.zeros((B, N, spec.n_mel), dtype=torch.float32)
            for i, j in enumerate(idx):
                noise[i, : seq[i]] = noise_blocks[j]
            t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
            if self.config.use_hip_graph:
                pre = eng.preprocess(audio, t32(lens_a), t32(ids), t32(lens_t), t32(seq), N, seq_len_host=seq, audio_len_host=lens_a)
                x = noise.to(dev)
                eng.transformer_steps(x, pre, 0, eng.n_steps)
                if self._decode_graphs is None:
                    self._decode_graphs = DecodeGraphCache(eng, self.config.decode_graph_cache_entries, self.config.decode_graph_cache_bytes)
                pcm, pcm_len = self._decode_graphs.get(B, N, t_gen)(x, pre["ref_signal_len"], pre["seq_len"])
            else:
                _x, pcm, pcm_len, _pre = eng.synthesize_batch(audio, t32(lens_a), t32(ids), t32(lens_t), t32(seq), N, noise.to(dev), t_gen,
                                                              gen_frames=[int(v) for v in (seq - ref_frames)], seq_len_host=seq,
                                                              audio_len_host=lens_a)
            pcm, pcm_len = pcm.cpu().numpy(), pcm_len.cpu().numpy()
            for i, j in enumerate(idx):
                waves[j] = pcm[i, : pcm_len[i]].reshape(1, 1, -1)
        return waves

    # ------------------------------------------------------------------ public API
    def synthesize(self, text: str, gender: Optional[str] = None, group: Optional[str] = None, area: Optional[str] = None,
                   emotion: Optional[str] = None, sample_iteration: Optional[int] = None, output_path: Optional[str] = None,
                   reference_audio: Optional[str] = None, reference_text: Optional[str] = None) -> Tuple[np.ndarray, float]:
        start = time.time()
        speed = self.config.speed      # read once: the REST layer mutates config.speed around the call (api/tts_engine.py:68-91)
        ref_audio, ref_text = self.model_session_manager.select_sample(gender, group, area, emotion, sample_iteration,
                                                                       reference_audio, reference_text)
        try:
            with self._lock:
                inputs_list = self._prepare_inputs(ref_audio, ref_text, text, speed=speed)
                self._last_plan = [int(i[2][0]) for i in inputs_list]
                if self.model_session_manager.engine is not None:
                    waves = self._synthesize_device(inputs_list)
                else:
                    waves = self._synthesize_sessions(inputs_list)
            final_wave = self.audio_processor.concatenate_with_crossfade_improved(waves, self.config.cross_fade_duration,
                                                                                  self.config.sample_rate)
            generation_time = time.time() - start
            if output_path:
                self.audio_processor.save_audio(final_wave, output_path, self.config.sample_rate)
                logger.info("Audio saved to: %s", output_path)
            return final_wave, generation_time
        except Exception as e:
            raise RuntimeError(f"Speech synthesis failed: {str(e)}")

    def synthesize_stream(self, text: str, gender: Optional[str] = None, group: Optional[str] = None, area: Optional[str] = None,
                          emotion: Optional[str] = None, sample_iteration: Optional[int] = None,
                          reference_audio: Optional[str] = None, reference_text: Optional[str] = None, chunks_per_step: int = 1):
        """Generator of int16 PCM blocks: audio is emitted as soon as a group of ``chunks_per_step``
        chunks is synthesised instead of after the whole text (the reference buffers everything, api/app.py:59-65).
        Overlap-save: the improved cross-fade only rewrites the last ``cross_fade_duration`` of what has been joined
        so far (audio_processor.py:122-192), so everything before that tail is final and can be yielded.  The joiner
        (``CrossfadeStream``) keeps only that tail: each raw chunk is clip-repaired once, emitted samples are never
        revisited, and the concatenation of all yielded blocks equals ``synthesize(text)`` sample for sample."""
        speed = self.config.speed
        ref_audio, ref_text = self.model_session_manager.select_sample(gender, group, area, emotion, sample_iteration,
                                                                       reference_audio, reference_text)
        try:
            with self._lock:
                inputs_list = self._prepare_inputs(ref_audio, ref_text, text, speed=speed)
            self._last_plan = [int(i[2][0]) for i in inputs_list]
            from .audio_processor import CrossfadeStream
            joiner = CrossfadeStream(len(inputs_list), self.config.cross_fade_duration, self.config.sample_rate)
            step = max(1, int(chunks_per_step))
            for lo in range(0, len(inputs_list), step):
                with self._lock:
                    if self.model_session_manager.engine is not None:
                        waves = self._synthesize_device(inputs_list[lo: lo + step])
                    else:
                        waves = self._synthesize_sessions(inputs_list[lo: lo + step])
                blocks = [joiner.push(w) for w in waves]
                block = np.concatenate(blocks) if len(blocks) > 1 else blocks[0]
                if block.size:
                    yield block
        except Exception as e:
            raise RuntimeError(f"Speech synthesis failed: {str(e)}")

    def validate_configuration(self, reference_audio: Optional[str] = None) -> bool:
        if reference_audio is None:
            return True          # built-in voice samples are used
        return self.config.validate_with_reference_audio(reference_audio)
