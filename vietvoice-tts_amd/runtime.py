"""ctypes binding of libvvtts_hip.so (include/vvtts.h) and the device-resident synthesis driver.

PyTorch is used only as plumbing here: it owns HBM tensors, the HIP stream and (multi-GPU) the
RCCL broadcast.  Every arithmetic step of the hot path runs in the hand-written gfx950 kernels
behind the C ABI.  There is NO CPU fallback: if the library or a GPU is missing this module
raises, loudly (the oracle under oracle/ is test infrastructure and is never imported here).
"""
from __future__ import annotations

import ctypes as C
import os
import contextlib
import threading
from typing import Dict, Optional, Tuple

import torch

from . import pack
from .model_spec import ModelSpec, time_grid

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VVTTS_LIB") or os.path.join(_HERE, "libvvtts_hip.so")   # VVTTS_LIB: A/B builds in tools/

VV_F32, VV_BF16 = 0, 1
PROF_CLASSES = ["gemm", "attention", "norm", "posconv", "elementwise", "voc_conv", "voc_post", "mel", "text",
                "voc_pre", "voc_up0", "voc_up1", "voc_up2", "voc_up3", "voc_mrf0", "voc_mrf1", "voc_mrf2", "voc_mrf3"]   # voc_conv once more, by stage


class HipUnavailable(RuntimeError):
    pass


class vv_model_cfg(C.Structure):
    _fields_ = [
        ("n_mel", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop_length", C.c_int32),
        ("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("ff_mult", C.c_int32),
        ("text_dim", C.c_int32), ("text_layers", C.c_int32), ("text_conv_k", C.c_int32), ("text_ff_mult", C.c_int32),
        ("vocab_rows", C.c_int32), ("pos_conv_k", C.c_int32), ("pos_conv_groups", C.c_int32), ("time_freq_dim", C.c_int32),
        ("cfg_strength", C.c_float),
        ("voc_pre_ch", C.c_int32), ("voc_pre_k", C.c_int32), ("voc_post_k", C.c_int32),
        ("voc_n_up", C.c_int32), ("voc_up_rates", C.c_int32 * 8), ("voc_up_kernels", C.c_int32 * 8),
        ("voc_n_res", C.c_int32), ("voc_res_kernels", C.c_int32 * 4),
        ("voc_n_dil", C.c_int32), ("voc_res_dilations", C.c_int32 * 4),
        ("voc_lrelu", C.c_float), ("max_pos", C.c_int32),
    ]


class vv_gemm_args(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_dtype", C.c_int32), ("mode", C.c_int32), ("act", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int32), ("W", C.c_void_p), ("ldw", C.c_int32), ("C", C.c_void_p), ("ldc", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("bias", C.c_void_p), ("gate", C.c_void_p), ("cos_q", C.c_void_p), ("sin_q", C.c_void_p),
                ("cos_k", C.c_void_p), ("sin_k", C.c_void_p),
                ("n_store", C.c_int32), ("seq_n", C.c_int32), ("rope_dim", C.c_int32),
                ("rope_cs_q", C.c_void_p), ("rope_cs_k", C.c_void_p), ("tile", C.c_int32), ("rope_pos", C.c_void_p), ("rope_by_row", C.c_int32),
                ("tail_parts", C.c_int32), ("tail_row0", C.c_int32), ("C_tail", C.c_void_p), ("rope_theta", C.c_float), ("rope_skip_q", C.c_int32), ("chip_share", C.c_int32)]


class vv_attn_args(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("qkv", C.c_void_p), ("ld_qkv", C.c_int32), ("out", C.c_void_p), ("ld_out", C.c_int32),
                ("n_seq", C.c_int32), ("seq_n", C.c_int32), ("heads", C.c_int32), ("dim", C.c_int32), ("kv_len", C.c_void_p),
                ("row_start", C.c_void_p), ("total_rows", C.c_int32), ("q_scale", C.c_float), ("rope_cs_q", C.c_void_p)]


class vv_ln_args(C.Structure):
    _fields_ = [("out_dtype", C.c_int32), ("x", C.c_void_p), ("ldx", C.c_int32), ("y", C.c_void_p), ("ldy", C.c_int32),
                ("R", C.c_int32), ("D", C.c_int32), ("w", C.c_void_p), ("b", C.c_void_p), ("add_one", C.c_int32), ("eps", C.c_float),
                ("delta", C.c_void_p), ("delta_dtype", C.c_int32), ("ld_delta", C.c_int32),
                ("delta2", C.c_void_p), ("keep_x", C.c_int32), ("tail_row0", C.c_int32), ("delta_tail_parts", C.c_int32),
                ("delta2_tail_parts", C.c_int32), ("delta_tail", C.c_void_p), ("delta2_tail", C.c_void_p)]


class vv_posconv_args(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_dtype", C.c_int32), ("in_", C.c_void_p), ("ld_in", C.c_int32), ("W", C.c_void_p),
                ("bias", C.c_void_p), ("out", C.c_void_p), ("ld_out", C.c_int32), ("resid", C.c_void_p), ("ld_resid", C.c_int32),
                ("n_seq", C.c_int32), ("seq_n", C.c_int32), ("groups", C.c_int32), ("KW", C.c_int32), ("B", C.c_int32),
                ("seq_len", C.c_void_p), ("row_start", C.c_void_p)]


class vv_conv_args(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("resid", C.c_void_p),
                ("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("T_in", C.c_int32), ("T_out", C.c_int32),
                ("KW", C.c_int32), ("dil", C.c_int32), ("transposed", C.c_int32), ("up", C.c_int32),
                ("rows_total", C.c_int32), ("rows_pad", C.c_int32), ("accumulate", C.c_int32),
                ("pre_slope", C.c_float), ("out_scale", C.c_float), ("len_in", C.c_void_p), ("W_x3", C.c_void_p), ("wg_rows", C.c_int32)]


class vv_mrf_args(C.Structure):
    _fields_ = [("y", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("KW", C.c_int32), ("dil", C.c_int32), ("rows_pad", C.c_int32),
                ("accumulate", C.c_int32), ("slope", C.c_float), ("out_scale", C.c_float), ("len_in", C.c_void_p)]


EXPORTS = {
    # name: (restype, argtypes)
    "vv_version": (C.c_char_p, []),
    "vv_last_error": (C.c_char_p, [C.c_void_p]),
    "vv_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(vv_model_cfg), C.c_int]),
    "vv_destroy": (None, [C.c_void_p]),
    "vv_bind_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]),
    "vv_finalize_weights": (C.c_int, [C.c_void_p]),
    "vv_set_time_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vv_preprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vv_preprocess_h": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vv_transformer_steps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vv_transformer_steps_h": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vv_transformer_ws_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]),
    "vv_transformer_steps_into": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "vv_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "vv_decode_ws_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "vv_decode_into": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "vv_ws_generation": (C.c_uint64, [C.c_void_p]),
    "vv_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vv_set_rope_theta": (C.c_int, [C.c_void_p, C.c_float]),
    "vv_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "vv_prof_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vv_gemm": (C.c_int, [C.c_void_p, C.POINTER(vv_gemm_args), C.c_void_p]),
    "vv_attention": (C.c_int, [C.c_void_p, C.POINTER(vv_attn_args), C.c_void_p]),
    "vv_layernorm": (C.c_int, [C.c_void_p, C.POINTER(vv_ln_args), C.c_void_p]),
    "vv_posconv": (C.c_int, [C.c_void_p, C.POINTER(vv_posconv_args), C.c_void_p]),
    "vv_conv1d": (C.c_int, [C.c_void_p, C.POINTER(vv_conv_args), C.c_void_p]),
    "vv_mrf_resblock": (C.c_int, [C.c_void_p, C.POINTER(vv_mrf_args), C.c_void_p]),
    "vv_conv_post": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "vv_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vv_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                               C.c_int, C.c_void_p]),
    "vv_rope_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vv_conv_split_bytes": (C.c_uint64, [C.c_int32, C.c_int32, C.c_int32]),
    "vv_conv_split_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vv_gemm_tail_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vv_rope_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vv_cfg_euler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "vv_resample_poly": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vv_ingest_pcm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vv_normalize_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "vv_normalize_clips": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library(path: Optional[str] = None):
    """dlopen the in-tree library and type every export of include/vvtts.h.  No compute happens."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise HipUnavailable(f"{p} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                                 "the HIP synthesis path has no CPU fallback")
        lib = C.CDLL(p)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)          # AttributeError if an export is missing
            fn.restype, fn.argtypes = res, args
        if path is None:
            _lib = lib
        return lib


def cfg_from_spec(spec: ModelSpec) -> vv_model_cfg:
    c = vv_model_cfg()
    c.n_mel, c.n_fft, c.win_length, c.hop_length = spec.n_mel, spec.n_fft, spec.win_length, spec.hop_length
    c.dim, c.depth, c.heads, c.head_dim, c.ff_mult = spec.dim, spec.depth, spec.heads, spec.head_dim, spec.ff_mult
    c.text_dim, c.text_layers, c.text_conv_k, c.text_ff_mult = spec.text_dim, spec.text_layers, spec.text_conv_k, spec.text_ff_mult
    c.vocab_rows = spec.vocab_size + 1
    c.pos_conv_k, c.pos_conv_groups, c.time_freq_dim = spec.pos_conv_k, spec.pos_conv_groups, spec.time_freq_dim
    c.cfg_strength = spec.cfg_strength
    c.voc_pre_ch, c.voc_pre_k, c.voc_post_k = spec.voc_pre_ch, spec.voc_pre_k, spec.voc_post_k
    c.voc_n_up = len(spec.voc_up_rates)
    for i, (r, k) in enumerate(zip(spec.voc_up_rates, spec.voc_up_kernels)):
        c.voc_up_rates[i], c.voc_up_kernels[i] = r, k
    c.voc_n_res = len(spec.voc_res_kernels)
    for i, k in enumerate(spec.voc_res_kernels):
        c.voc_res_kernels[i] = k
    c.voc_n_dil = len(spec.voc_res_dilations)
    for i, d in enumerate(spec.voc_res_dilations):
        c.voc_res_dilations[i] = d
    c.voc_lrelu = spec.voc_lrelu
    c.max_pos = pack.MAX_POS
    return c


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dt(s) -> Tuple[int, torch.dtype]:
    if s in ("bf16", "bfloat16", torch.bfloat16, VV_BF16):
        return VV_BF16, torch.bfloat16
    if s in ("fp32", "f32", "float32", torch.float32, VV_F32):
        return VV_F32, torch.float32
    raise ValueError(f"acoustic dtype must be bf16 or fp32, got {s!r}")


class HipSynth:
    """One GPU's synthesis engine: weights resident in HBM, three device-resident stages.

    flat_weights: optional pre-filled flat uint8 device buffer (e.g. received by RCCL broadcast);
    otherwise ``weights`` (fp32 CPU dict) is packed and uploaded here.
    """

    def __init__(self, spec: ModelSpec, weights: Optional[Dict[str, torch.Tensor]] = None, device: str = "cuda:0",
                 acoustic_dtype="bf16", nfe_step: int = 32, flat_weights: Optional[torch.Tensor] = None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise HipUnavailable("no HIP device is visible to torch; the synthesis hot path runs only on the GPU")
        self.spec = spec
        self.device = torch.device(device)
        self.dt_code, self.dt_torch = _dt(acoustic_dtype)
        self._lock = threading.RLock()       # one stream of calls per context (reference: api/tts_engine.py:64-67); re-entrant: reading_rope_tables
        self.ctx = C.c_void_p()
        cfg = cfg_from_spec(spec)
        idx = self.device.index if self.device.index is not None else 0
        rc = self.lib.vv_create(C.byref(self.ctx), idx, C.byref(cfg), self.dt_code)
        if rc != 0:
            raise HipUnavailable(f"vv_create failed ({rc}): {self.lib.vv_last_error(None).decode()}")
        table, total = pack.plan(spec, self.dt_torch)
        if flat_weights is None:
            if weights is None:
                raise ValueError("either weights or flat_weights is required")
            cpu = torch.zeros(total, dtype=torch.uint8)
            pack.fill(spec, self.dt_torch, weights, cpu)
            flat_weights = cpu.to(self.device)
        assert flat_weights.dtype == torch.uint8 and flat_weights.numel() >= total and flat_weights.is_cuda
        self.flat = flat_weights
        base = self.flat.data_ptr()
        for name, off, nb in table:
            self._check(self.lib.vv_bind_weight(self.ctx, name.encode(), base + off, nb))
        self._check(self.lib.vv_finalize_weights(self.ctx))
        cq, sq, ck, sk = pack.rope_tables(spec)
        self.rope = tuple(t.to(self.device) for t in (cq, sq, ck, sk))
        # the tables this engine hands to the transformer stage are the standard ones of spec.rope_theta: the bf16 model may compute the
        # angles in the QKV epilogue instead of reading them (vv_set_rope_theta; the fp32 model reads the tables either way)
        self.grid_generation = 0         # bumped by whatever a captured Euler-step graph has baked in (time grid, rope mode, options)
        self.set_rope_theta(float(spec.rope_theta))
        self.nfe_step = None
        self.set_nfe(nfe_step)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"vvtts HIP call failed ({rc}): {self.lib.vv_last_error(self.ctx).decode()}")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.vv_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_nfe(self, nfe_step: int):
        if nfe_step == self.nfe_step:
            return
        if nfe_step < 2:
            raise ValueError("nfe_step must be >= 2")
        t, dt = time_grid(nfe_step, self.spec.sway_coef)
        sinus = pack.time_sinus_table(self.spec, t).contiguous()
        dtc = dt.contiguous()
        with self._lock, torch.cuda.device(self.device):          # under the engine lock: no step call or graph replay runs across the change
            self._check(self.lib.vv_set_time_grid(self.ctx, sinus.data_ptr(), dtc.data_ptr(), int(t.numel()), self._stream()))
            self.nfe_step = nfe_step
            self.n_steps = int(t.numel())
            self.grid_generation += 1    # vv_set_time_grid frees and reallocates the tables a captured step graph points into

    # ------------------------------------------------------------------ stages
    def preprocess(self, audio: torch.Tensor, audio_len: torch.Tensor, text_ids: torch.Tensor, text_len: torch.Tensor,
                   seq_len: torch.Tensor, N: int, max_audio_len: Optional[int] = None, seq_len_host=None,
                   audio_len_host=None) -> Dict[str, torch.Tensor]:
        """audio int16 [B,S], text_ids int32 [B,T], *_len int32 [B] -- all on the device.  audio_len_host (optional, the same
        values as audio_len): lets the call refuse a clip too short for the centred STFT (< n_fft/2 + 1 samples) before launching."""
        B = audio.shape[0]
        for t, d in ((audio, torch.int16), (audio_len, torch.int32), (text_ids, torch.int32), (text_len, torch.int32), (seq_len, torch.int32)):
            assert t.is_cuda and t.dtype == d and t.is_contiguous(), "preprocess inputs must be contiguous device tensors"
        s = self.spec
        cat = torch.empty((B, N, s.cond_dim), dtype=torch.float32, device=self.device)
        cat_drop = torch.empty_like(cat)
        ref_len = torch.empty((B,), dtype=torch.int32, device=self.device)
        mal = int(max_audio_len if max_audio_len is not None else audio.shape[1])
        with self._lock, torch.cuda.device(self.device):
            if audio_len_host is not None:
                host = (C.c_int32 * B)(*[int(v) for v in audio_len_host])
                self._check(self.lib.vv_preprocess_h(self.ctx, B, N, audio.data_ptr(), audio.shape[1], mal, audio_len.data_ptr(), host,
                                                     text_ids.data_ptr(), text_ids.shape[1], text_len.data_ptr(), seq_len.data_ptr(),
                                                     cat.data_ptr(), cat_drop.data_ptr(), ref_len.data_ptr(), self._stream()))
            else:
                self._check(self.lib.vv_preprocess(self.ctx, B, N, audio.data_ptr(), audio.shape[1], mal, audio_len.data_ptr(),
                                                   text_ids.data_ptr(), text_ids.shape[1], text_len.data_ptr(), seq_len.data_ptr(),
                                                   cat.data_ptr(), cat_drop.data_ptr(), ref_len.data_ptr(), self._stream()))
        return {"cat_mel_text": cat, "cat_mel_text_drop": cat_drop, "ref_signal_len": ref_len, "seq_len": seq_len,
                "seq_len_host": None if seq_len_host is None else [int(v) for v in seq_len_host],
                "rope_cos_q": self.rope[0][:N], "rope_sin_q": self.rope[1][:N], "rope_cos_k": self.rope[2][:N],
                "rope_sin_k": self.rope[3][:N]}

    def max_rows_per_call(self) -> int:
        """Upper bound on sum(seq_len) of one transformer_steps call: the packed qkv buffer (2 branches x rows x 3D) must stay
        below 2 GiB (32-bit byte offsets in the kernels)."""
        es = 2 if self.dt_torch == torch.bfloat16 else 4
        return ((1 << 31) - 1) // (2 * 3 * self.spec.dim * es)

    def transformer_steps(self, x: torch.Tensor, pre: Dict[str, torch.Tensor], step0: int, n_steps: int, seq_len_host=None) -> torch.Tensor:
        """x fp32 [B,N,n_mel] updated in place on the device.  seq_len_host (optional list / array of the B lengths, the same
        values as pre["seq_len"]): the call then needs no read-back and no stream synchronisation (vv_transformer_steps_h)."""
        B, N, M = x.shape
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and M == self.spec.n_mel
        if seq_len_host is None:
            seq_len_host = pre.get("seq_len_host")
        with self._lock, torch.cuda.device(self.device):
            tail = (pre["cat_mel_text"].data_ptr(), pre["cat_mel_text_drop"].data_ptr(), pre["rope_cos_q"].data_ptr(), pre["rope_sin_q"].data_ptr(),
                    pre["rope_cos_k"].data_ptr(), pre["rope_sin_k"].data_ptr(), step0, n_steps, self._stream())
            if seq_len_host is not None:
                host = (C.c_int32 * B)(*[int(v) for v in seq_len_host])
                assert len(host) == B
                self._check(self.lib.vv_transformer_steps_h(self.ctx, B, N, pre["seq_len"].data_ptr(), host, x.data_ptr(), *tail))
            else:
                self._check(self.lib.vv_transformer_steps(self.ctx, B, N, pre["seq_len"].data_ptr(), x.data_ptr(), *tail))
        return x

    def decode(self, x: torch.Tensor, pre: Dict[str, torch.Tensor], t_gen_max: int, want_wave: bool = False):
        B, N, _ = x.shape
        hop = self.spec.hop_length
        pcm = torch.zeros((B, t_gen_max * hop), dtype=torch.int16, device=self.device)
        pcm_len = torch.empty((B,), dtype=torch.int32, device=self.device)
        wave = torch.zeros((B, t_gen_max * hop), dtype=torch.float32, device=self.device) if want_wave else None
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.vv_decode(self.ctx, B, N, x.data_ptr(), pre["ref_signal_len"].data_ptr(), pre["seq_len"].data_ptr(),
                                           t_gen_max, pcm.data_ptr(), pcm.shape[1], pcm_len.data_ptr(), _ptr(wave), self._stream()))
        return (pcm, pcm_len, wave) if want_wave else (pcm, pcm_len)

    def decode_bucketed(self, x: torch.Tensor, pre: Dict[str, torch.Tensor], gen_frames, pad_frac: float = 0.10, min_units: int = 4):
        """The vocoder works on padded [B][C][T_max] planes (the acoustic stages pack ragged rows, the conv stack does not),
        so a ragged batch is decoded in length buckets.  gen_frames: host list of generated frames per item."""
        from .sharding import plan_batches
        B = x.shape[0]
        hop = self.spec.hop_length
        t_max = int(max(gen_frames))
        groups = plan_batches([int(f) for f in gen_frames], B, pad_frac=pad_frac, min_units=min_units)
        if len(groups) == 1:
            return self.decode(x, pre, t_max)
        pcm = torch.zeros((B, t_max * hop), dtype=torch.int16, device=self.device)
        pcm_len = torch.empty((B,), dtype=torch.int32, device=self.device)
        for grp in groups:
            idx = torch.tensor(grp, dtype=torch.int64, device=self.device)
            sub = {"ref_signal_len": pre["ref_signal_len"].index_select(0, idx).contiguous(), "seq_len": pre["seq_len"].index_select(0, idx).contiguous()}
            t_g = int(max(gen_frames[i] for i in grp))
            p, n = self.decode(x.index_select(0, idx).contiguous(), sub, t_g)
            pcm[idx, : t_g * hop] = p
            pcm_len[idx] = n
        return pcm, pcm_len

    def synthesize_batch(self, audio, audio_len, text_ids, text_len, seq_len, N: int, noise: torch.Tensor, t_gen_max: int,
                         n_steps: Optional[int] = None, max_audio_len: Optional[int] = None, gen_frames=None, seq_len_host=None,
                         audio_len_host=None):
        """Whole hot path for a batch, state resident in HBM: preprocess -> ODE steps -> vocoder.
        gen_frames (host list, optional): per-item generated frames; lets the vocoder run in length buckets on ragged batches.
        seq_len_host (optional): the lengths on the host too -- the Euler-step call then runs without any stream synchronisation."""
        pre = self.preprocess(audio, audio_len, text_ids, text_len, seq_len, N, max_audio_len, seq_len_host=seq_len_host,
                              audio_len_host=audio_len_host)
        x = noise.clone()
        self.transformer_steps(x, pre, 0, self.n_steps if n_steps is None else n_steps)
        if gen_frames is not None and len(gen_frames) == x.shape[0]:
            pcm, pcm_len = self.decode_bucketed(x, pre, gen_frames)
            if pcm.shape[1] < t_gen_max * self.spec.hop_length:
                pcm = torch.nn.functional.pad(pcm, (0, t_gen_max * self.spec.hop_length - pcm.shape[1]))
        else:
            pcm, pcm_len = self.decode(x, pre, t_gen_max)
        return x, pcm, pcm_len, pre

    # ------------------------------------------------------------------ hipGraph-captured vocoder step (config 5)
    def capture_decode(self, B: int, N: int, t_gen_max: int) -> "GraphedDecode":
        """Capture the decode stage (frame slice -> vocoder -> int16) for a fixed shape into a hipGraph.
        Long-form synthesis replays it per chunk group: ~80 launches become one graph launch."""
        return GraphedDecode(self, B, N, t_gen_max)

    def capture_steps(self, B: int, N: int, seq_len_host, t_gen_max: int) -> "GraphedSteps":
        """Capture all Euler steps + the decode of one batch shape into ONE hipGraph (single-utterance latency path)."""
        return GraphedSteps(self, B, N, seq_len_host, t_gen_max)

    # ------------------------------------------------------------------ reference-clip ingest (N3)
    def resample_poly(self, x: torch.Tensor, taps: torch.Tensor, up: int, down: int, skip: int, n_out: int) -> torch.Tensor:
        """x f32 [n_in] and taps f64 [n_taps] on the device -> f32 [n_out]."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and taps.is_cuda and taps.dtype == torch.float64
        y = torch.empty((n_out,), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.vv_resample_poly(self.ctx, x.data_ptr(), x.numel(), taps.data_ptr(), taps.numel(), up, down, skip,
                                                  y.data_ptr(), n_out, self._stream()))
        return y

    def ingest_pcm(self, pcm: torch.Tensor, desc, total_out: int) -> torch.Tensor:
        """pcm uint8 [bytes] on the device (the clips' interleaved PCM back to back) and desc = HOST rows of 8 ints per clip
        (include/vvtts.h: byte offset, width, channels, n_frames, src / g, dst / g, out offset, n_out) -> f32 [total_out]: mono samples at
        the destination rate (audioop.tomono + audioop.ratecv arithmetic).  The rows are validated here, on the host, before anything
        is launched: the kernel indexes the byte buffer by them."""
        assert pcm.is_cuda and pcm.dtype == torch.uint8 and pcm.is_contiguous()
        rows = [[int(v) for v in r] for r in desc]
        if not rows:
            raise ValueError("ingest_pcm: no clips")
        for r in rows:
            if len(r) != 8:
                raise ValueError("ingest_pcm: a descriptor row has 8 entries")
            off, width, ch, n_frames, I, O, out_off, n_out = r
            want = n_frames if I == O else ((n_frames - 1) * O // I + 1 if n_frames > 0 else 0)
            if (width not in (1, 2, 4) or ch < 1 or n_frames < 1 or I < 1 or O < 1 or off < 0 or off % width or n_out != want or n_out < 1
                    or off + n_frames * ch * width > pcm.numel() or out_off < 0 or out_off + n_out > total_out):
                raise ValueError(f"ingest_pcm: descriptor row {r} does not fit the buffers ({pcm.numel()} PCM bytes, {total_out} output samples)")
        y = torch.empty((total_out,), dtype=torch.float32, device=self.device)
        d = torch.tensor(rows, dtype=torch.int64).to(self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.vv_ingest_pcm(self.ctx, pcm.data_ptr(), d.data_ptr(), len(rows), max(r[7] for r in rows), y.data_ptr(), self._stream()))
        return y

    def normalize_clips(self, x: torch.Tensor, offsets: torch.Tensor, max_len: int = 0) -> torch.Tensor:
        """x f32 [total] = clips back to back, offsets int64 [n+1] (device) -> int16 [total] (DC removed, peak 29491);
        max_len = the longest clip when the caller knows it (else it is read back from the offsets)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and offsets.is_cuda and offsets.dtype == torch.int64
        n = offsets.numel() - 1
        if not max_len:
            oh = offsets.cpu()
            if int(oh[0]) != 0 or int(oh[-1]) != x.numel() or bool((oh[1:] < oh[:-1]).any()):
                raise ValueError("normalize_clips: offsets must run from 0 to x.numel(), non-decreasing")
            max_len = int((oh[1:] - oh[:-1]).max())
        out = torch.empty((x.numel(),), dtype=torch.int16, device=self.device)
        scratch = torch.empty((int(self.lib.vv_normalize_scratch_bytes(n, x.numel())),), dtype=torch.uint8, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.vv_normalize_clips(self.ctx, x.data_ptr(), offsets.data_ptr(), n, int(max_len), scratch.data_ptr(),
                                                    out.data_ptr(), self._stream()))
        return out

    def set_rope_theta(self, theta: float):
        """theta > 1: the rope tables of this engine are the standard ones of that base, the bf16 QKV epilogue computes the angles;
        0: the tables are read (vv_set_rope_theta)."""
        with self._lock:
            self._check(self.lib.vv_set_rope_theta(self.ctx, float(theta)))
            self._rope_theta = float(theta)
            self.grid_generation += 1

    @contextlib.contextmanager
    def reading_rope_tables(self):
        """For ONE caller's calls: the rope tables handed to transformer_steps are read, not recomputed from theta (a session that
        is fed non-standard tables).  Holds the engine lock, so no other call and no graph replay sees the switched mode; the
        previous mode is restored on exit (captured graphs stay valid: none can run in between)."""
        with self._lock:
            prev = self._rope_theta
            self._check(self.lib.vv_set_rope_theta(self.ctx, 0.0))
            try:
                yield
            finally:
                self._check(self.lib.vv_set_rope_theta(self.ctx, prev))

    def set_option(self, name: str, value: int):
        """Context switches of the C ABI (vv_set_option), e.g. ``fuse_mrf`` 0/1."""
        with self._lock:
            self._check(self.lib.vv_set_option(self.ctx, name.encode(), int(value)))
            self.grid_generation += 1

    # ------------------------------------------------------------------ profiling
    def prof_enable(self, on: bool):
        self._check(self.lib.vv_prof_enable(self.ctx, 1 if on else 0))

    def prof_collect(self) -> Dict[str, Dict[str, float]]:
        n = len(PROF_CLASSES)
        la, ms, fl, by = (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        self._check(self.lib.vv_prof_collect(self.ctx, la, ms, fl, by))
        return {PROF_CLASSES[i]: {"launches": int(la[i]), "ms": float(ms[i]), "flops": float(fl[i]), "bytes": float(by[i])}
                for i in range(n)}


class GraphedDecode:
    """hipGraph replay of the decode stage at a fixed (B, N, t_gen_max).  Static input/output buffers AND the
    stage's whole workspace are owned here (``vv_decode_into``): the captured launches point only into memory
    that lives as long as this object, so later calls that grow the context arena cannot invalidate the graph.
    ``__call__`` copies the state in and replays."""

    def __init__(self, eng: HipSynth, B: int, N: int, t_gen_max: int, ws: Optional[torch.Tensor] = None):
        """ws: optional caller-owned uint8 workspace of at least ``vv_decode_ws_bytes`` bytes (a ``DecodeGraphCache`` shares one
        block between all graphs of an engine: replays are serialised by the engine lock on one stream)."""
        self.eng, self.B, self.N, self.t_gen_max = eng, B, N, t_gen_max
        dev, s = eng.device, eng.spec
        self.x = torch.zeros((B, N, s.n_mel), dtype=torch.float32, device=dev)
        self.ref_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.seq_len = torch.full((B,), N, dtype=torch.int32, device=dev)
        self.pcm = torch.zeros((B, t_gen_max * s.hop_length), dtype=torch.int16, device=dev)
        self.pcm_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        nb = C.c_uint64()
        eng._check(eng.lib.vv_decode_ws_bytes(eng.ctx, B, t_gen_max, C.byref(nb)))
        if ws is not None:
            assert ws.is_cuda and ws.dtype == torch.uint8 and ws.numel() >= int(nb.value)
        self.ws = ws if ws is not None else torch.empty((int(nb.value),), dtype=torch.uint8, device=dev)   # torch allocations are 512-byte aligned
        self.ws_need = int(nb.value)
        self._launch()                                   # warm-up: sets kernel attributes before capture
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch()

    def _launch(self):
        e = self.eng
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.vv_decode_into(e.ctx, self.B, self.N, self.x.data_ptr(), self.ref_len.data_ptr(), self.seq_len.data_ptr(),
                                          self.t_gen_max, self.pcm.data_ptr(), self.pcm.shape[1], self.pcm_len.data_ptr(), None,
                                          self.ws.data_ptr(), self.ws.numel(), e._stream()))

    def __call__(self, x: torch.Tensor, ref_len: torch.Tensor, seq_len: torch.Tensor):
        with self.eng._lock:                             # one stream of work per context, replay included
            self.x.copy_(x)
            self.ref_len.copy_(ref_len)
            self.seq_len.copy_(seq_len)
            self.graph.replay()
        return self.pcm, self.pcm_len


    def io_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.x, self.ref_len, self.seq_len, self.pcm, self.pcm_len))


class GraphedSteps:
    """ONE hipGraph for the whole ODE integration + decode of a fixed batch shape (B, N, the per-item lengths, t_gen_max): all
    ``n_steps`` Euler steps (``vv_transformer_steps_into``: ~170 launches per step at the full model) and the decode stage
    (``vv_decode_into``) are captured once and replayed as a single graph launch -- the single-utterance latency path, where the ~5,300
    launch-sized kernels of an utterance are bound by launch cadence, not by the kernels.  The row count of every launch is part
    of the captured shapes, so a graph serves exactly one tuple of lengths; static I/O buffers and ONE workspace block (the steps and
    the decode run one after the other on one stream and share it) are owned here.  ``__call__`` copies the inputs in and replays.
    No reference counterpart: the reference pays a host round trip per step (core/tts_engine.py:157-172)."""

    def __init__(self, eng: HipSynth, B: int, N: int, seq_len_host, t_gen_max: int, n_steps: Optional[int] = None):
        self.eng, self.B, self.N, self.t_gen_max = eng, int(B), int(N), int(t_gen_max)
        self.seq_host = [int(v) for v in seq_len_host]
        assert len(self.seq_host) == self.B
        self.n_steps = eng.n_steps if n_steps is None else int(n_steps)
        dev, s = eng.device, eng.spec
        self.x = torch.zeros((B, N, s.n_mel), dtype=torch.float32, device=dev)
        self.cat = torch.zeros((B, N, s.cond_dim), dtype=torch.float32, device=dev)
        self.cat_drop = torch.zeros_like(self.cat)
        self.seq_len = torch.tensor(self.seq_host, dtype=torch.int32, device=dev)
        self.ref_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.pcm = torch.zeros((B, t_gen_max * s.hop_length), dtype=torch.int16, device=dev)
        self.pcm_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._host = (C.c_int32 * B)(*self.seq_host)
        nb_t, nb_d = C.c_uint64(), C.c_uint64()
        eng._check(eng.lib.vv_transformer_ws_bytes(eng.ctx, B, N, self._host, C.byref(nb_t)))
        eng._check(eng.lib.vv_decode_ws_bytes(eng.ctx, B, t_gen_max, C.byref(nb_d)))
        self.ws = torch.empty((max(int(nb_t.value), int(nb_d.value)),), dtype=torch.uint8, device=dev)
        self._launch()                                   # warm-up: sets kernel attributes before capture
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch()
        # the captured launches hold device pointers into the context's time-grid tables and its step count, rope mode and lane
        # plan: anything that changes those (HipSynth.set_nfe / set_rope_theta / set_option) makes this graph stale
        self.generation = eng.grid_generation

    def stale(self) -> bool:
        return self.generation != self.eng.grid_generation

    def _launch(self):
        e = self.eng
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.vv_transformer_steps_into(e.ctx, self.B, self.N, self.seq_len.data_ptr(), self._host, self.x.data_ptr(),
                                                     self.cat.data_ptr(), self.cat_drop.data_ptr(), e.rope[0].data_ptr(), e.rope[1].data_ptr(),
                                                     e.rope[2].data_ptr(), e.rope[3].data_ptr(), 0, self.n_steps, self.ws.data_ptr(), self.ws.numel(),
                                                     e._stream()))
            e._check(e.lib.vv_decode_into(e.ctx, self.B, self.N, self.x.data_ptr(), self.ref_len.data_ptr(), self.seq_len.data_ptr(),
                                          self.t_gen_max, self.pcm.data_ptr(), self.pcm.shape[1], self.pcm_len.data_ptr(), None,
                                          self.ws.data_ptr(), self.ws.numel(), e._stream()))

    def __call__(self, noise: torch.Tensor, pre: Dict[str, torch.Tensor]):
        """noise [B,N,n_mel] and ``pre`` (HipSynth.preprocess of the same batch) -> (x, pcm, pcm_len): views of the static buffers,
        valid until the next call."""
        with self.eng._lock:                             # the setters take the same lock: the check and the replay see one state
            if self.stale():
                raise RuntimeError("captured Euler-step graph is stale: the engine's time grid, rope mode or an option changed after the "
                                   "capture (set_nfe / set_rope_theta / set_option); capture again with HipSynth.capture_steps")
            self.x.copy_(noise)
            self.cat.copy_(pre["cat_mel_text"])
            self.cat_drop.copy_(pre["cat_mel_text_drop"])
            self.ref_len.copy_(pre["ref_signal_len"])
            self.graph.replay()
        return self.x, self.pcm, self.pcm_len

    def pinned_bytes(self) -> int:
        return self.ws.numel() + sum(t.numel() * t.element_size() for t in (self.x, self.cat, self.cat_drop, self.pcm))


class DecodeGraphCache:
    """Bounded cache of captured decode graphs for one engine, keyed by (B, N, t_gen_max).

    A service with varied voices sees a new generated-frame count per reference clip; unbounded, every key would pin its own
    workspace (5 x B x max(C*T) x 4 B: ~170 MB per item at full size) for the life of the engine.  Here
      * the workspace is ONE block shared by every graph of the cache (replays are serialised by the engine lock on one stream);
        it is replaced by a larger one only when a new key needs more -- graphs captured on the old block keep it alive until
        they are evicted;
      * entries are evicted least-recently-used beyond ``max_entries`` or when the pinned bytes (distinct workspace blocks +
        per-graph I/O buffers) exceed ``max_bytes``;
      * callers bucket the key (``bucket``): N to multiples of 128 frames, t_gen_max to multiples of 64, so that nearby clips and
        chunk lengths share a graph (the decode masks every item by its own lengths).
    """

    def __init__(self, eng: HipSynth, max_entries: int = 8, max_bytes: int = 16 << 30):
        from collections import OrderedDict
        self.eng, self.max_entries, self.max_bytes = eng, max(1, int(max_entries)), int(max_bytes)
        self._graphs: "OrderedDict[Tuple[int, int, int], GraphedDecode]" = OrderedDict()
        self._ws: Optional[torch.Tensor] = None
        self.hits = self.misses = self.evictions = 0

    @staticmethod
    def bucket(N: int, t_gen: int) -> Tuple[int, int]:
        Nb = (int(N) + 127) // 128 * 128
        return Nb, min(Nb, (int(t_gen) + 63) // 64 * 64)

    def pinned_bytes(self) -> int:
        blocks = {g.ws.data_ptr(): g.ws.numel() for g in self._graphs.values()}
        if self._ws is not None:
            blocks[self._ws.data_ptr()] = self._ws.numel()
        return sum(blocks.values()) + sum(g.io_bytes() for g in self._graphs.values())

    def __len__(self):
        return len(self._graphs)

    def __contains__(self, key):
        return key in self._graphs

    def get(self, B: int, N: int, t_gen_max: int) -> GraphedDecode:
        key = (int(B), int(N), int(t_gen_max))
        g = self._graphs.get(key)
        if g is not None:
            self.hits += 1
            self._graphs.move_to_end(key)
            return g
        self.misses += 1
        nb = C.c_uint64()
        self.eng._check(self.eng.lib.vv_decode_ws_bytes(self.eng.ctx, key[0], key[2], C.byref(nb)))
        if self._ws is None or self._ws.numel() < int(nb.value):
            self._ws = torch.empty((int(nb.value),), dtype=torch.uint8, device=self.eng.device)
        g = GraphedDecode(self.eng, *key, ws=self._ws)
        self._graphs[key] = g
        while len(self._graphs) > 1 and (len(self._graphs) > self.max_entries or self.pinned_bytes() > self.max_bytes):
            self._graphs.popitem(last=False)          # least recently used; its graph, I/O buffers (and old block) are released
            self.evictions += 1
        return g

    def clear(self):
        self._graphs.clear()
        self._ws = None
