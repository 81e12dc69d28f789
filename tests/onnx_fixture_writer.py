"""Fixture writer for the ONNX importer tests -- TEST INFRASTRUCTURE (moved out of the product package in round 5).

Writes this build's weights as ONNX ModelProto bytes with torch-exporter-style naming (Linear weights renamed to
``onnx::MatMul_<n>`` and stored transposed, biases keeping their module path, q / k / v as separate Linears, ...), i.e. the
inverse of vietvoice_tts_amd.onnx_import's NAME_RULES, so that import(export(w)) == w can be asserted and an archive in the
reference's tar layout (core/model.py:73-110) can be produced for the engine tests.  Because it is the importer's own inverse,
a wrong NAME_RULES guess is invisible to these round trips: the real archive is not available offline (PARITY UNPINNED).
"""
from __future__ import annotations

import io
import struct
import tarfile
from typing import Dict, List, Optional, Tuple

import numpy as np

from vietvoice_tts_amd.onnx_import import (NAME_RULES, OnnxNode, OnnxValue, _BF16, _DTYPES, _QKV, _RES, exporter_names)  # noqa: F401


# ------------------------------------------------------------------ minimal writer (fixtures / export)
def _enc_varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _enc_varint(fno << 3 | 2) + _enc_varint(len(payload)) + payload


def _vi(fno: int, x: int) -> bytes:
    return _enc_varint(fno << 3) + _enc_varint(x)


_ONNX_TYPE = {np.dtype(v): k for k, v in _DTYPES.items()}


def encode_tensor(name: str, arr: np.ndarray, how: str = "raw") -> bytes:
    """how: raw (raw_data), typed (float_data / int64_data / int32_data packed), bf16 (f32 rounded to bf16 raw_data)."""
    arr = np.asarray(arr)
    out = b"".join(_vi(1, d) for d in arr.shape)
    if how == "bf16":
        u = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32)
        bits = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2")
        return out + _vi(2, _BF16) + _ld(8, name.encode()) + _ld(9, bits.tobytes())
    out += _vi(2, _ONNX_TYPE[arr.dtype]) + _ld(8, name.encode())
    if how == "raw":
        return out + _ld(9, np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<")).tobytes())
    flat = arr.reshape(-1)
    if arr.dtype == np.float32:
        return out + _ld(4, flat.astype("<f4").tobytes())
    if arr.dtype == np.float64:
        return out + _ld(10, flat.astype("<f8").tobytes())
    if arr.dtype == np.int64:
        return out + _ld(7, b"".join(_enc_varint(int(x)) for x in flat))
    if arr.dtype == np.float16:
        return out + _ld(5, b"".join(_enc_varint(int(x)) for x in flat.view(np.uint16)))
    return out + _ld(5, b"".join(_enc_varint(int(x)) for x in flat))


def _enc_attr(name: str, v) -> bytes:
    out = _ld(1, name.encode())
    if isinstance(v, float):
        return out + _enc_varint(2 << 3 | 5) + struct.pack("<f", v) + _vi(20, 1)
    if isinstance(v, int):
        return out + _vi(3, v) + _vi(20, 2)
    if isinstance(v, bytes):
        return out + _ld(4, v) + _vi(20, 3)
    if isinstance(v, np.ndarray):
        return out + _ld(5, encode_tensor("", v)) + _vi(20, 4)
    if v and isinstance(v[0], float):
        return out + _ld(7, np.asarray(v, "<f4").tobytes()) + _vi(20, 6)
    return out + b"".join(_vi(8, int(x)) for x in v) + _vi(20, 7)          # unpacked repeated ints, as protobuf2-era writers emit


def encode_node(n: OnnxNode) -> bytes:
    return (b"".join(_ld(1, s.encode()) for s in n.inputs) + b"".join(_ld(2, s.encode()) for s in n.outputs) + _ld(3, n.name.encode())
            + _ld(4, n.op_type.encode()) + b"".join(_ld(5, _enc_attr(k, v)) for k, v in n.attrs.items()))


def _enc_value(v: OnnxValue) -> bytes:
    dims = b"".join(_ld(1, _ld(2, d.encode()) if isinstance(d, str) else _vi(1, int(d))) for d in v.shape)
    return _ld(1, v.name.encode()) + _ld(2, _ld(1, _vi(1, v.elem_type) + _ld(2, dims)))


def encode_model(nodes: List[OnnxNode], initializers: List[Tuple[str, np.ndarray, str]], inputs: List[OnnxValue],
                 outputs: List[OnnxValue], graph_name: str = "main_graph", opset: int = 17, producer: str = "pytorch") -> bytes:
    g = (b"".join(_ld(1, encode_node(n)) for n in nodes) + _ld(2, graph_name.encode())
         + b"".join(_ld(5, encode_tensor(k, a, how)) for k, a, how in initializers)
         + b"".join(_ld(11, _enc_value(v)) for v in inputs) + b"".join(_ld(12, _enc_value(v)) for v in outputs))
    return _vi(1, 8) + _ld(2, producer.encode()) + _ld(7, g) + _ld(8, _ld(1, b"") + _vi(2, opset))




# ------------------------------------------------------------------ export (inverse mapping; fixture maker)
def export_archive_members(spec, weights, storage: str = "raw") -> Dict[str, bytes]:
    """This build's weights as three ONNX files with exporter-style structure: Linear = MatMul(x, onnx::MatMul_n [in,out]) +
    Add(bias name kept); Conv / ConvTranspose keep parameter names and carry kernel_shape / strides / dilations / group
    attributes.  The node lists are a structural skeleton (enough to recover names and constants), not a runnable graph."""
    n_k = len(spec.voc_res_kernels)
    files = {"preprocess": ([], []), "transformer": ([], []), "decode": ([], [])}
    anon = [0]

    def where(name):
        return "decode" if name.startswith("voc.") else ("preprocess" if name.startswith("text.") else "transformer")

    def linear(dst, ename, w, b):
        nodes, inits = files[dst]
        anon[0] += 1
        wn = f"onnx::MatMul_{1000 + anon[0]}"
        stem = ename[: -len(".weight")]
        inits.append((wn, np.ascontiguousarray(w.T), storage))
        inits.append((stem + ".bias", b, storage))
        nodes.append(OnnxNode("MatMul", f"/{stem}/MatMul", [f"/{stem}/in", wn], [f"/{stem}/MatMul_output_0"]))
        nodes.append(OnnxNode("Add", f"/{stem}/Add", [stem + ".bias", f"/{stem}/MatMul_output_0"], [f"/{stem}/Add_output_0"]))

    w = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in weights.items()}
    done = set()
    for name in w:
        if name in done or name.endswith(".bias"):
            continue
        dst = where(name)
        nodes, inits = files[dst]
        if _QKV.fullmatch(name):
            bias = name[: -len("weight")] + "bias"
            for part, ename, bname in zip(np.split(w[name], 3, axis=0), exporter_names(name, n_k), exporter_names(bias, n_k)):
                idx = "qkv".index(ename.split(".to_")[1][0])
                linear(dst, ename, part, np.split(w[bias], 3)[idx])
            done |= {name, bias}
            continue
        ename = exporter_names(name, n_k)[0]
        arr = w[name]
        if name.endswith(".grn.gamma") or name.endswith(".grn.beta"):
            inits.append((ename, arr.reshape(1, 1, -1), storage))
        elif name == "text.embed.weight":
            inits.append((ename, arr, storage))
            nodes.append(OnnxNode("Gather", "/text_embed/Gather", [ename, "text_ids"], ["/text_embed/Gather_output_0"]))
        elif arr.ndim == 2:
            linear(dst, ename, arr, w[name[: -len("weight")] + "bias"])
            done.add(name[: -len("weight")] + "bias")
        elif arr.ndim == 3:
            bias = name[: -len("weight")] + "bias"
            bname = exporter_names(bias, n_k)[0]
            inits.append((ename, arr, storage))
            inits.append((bname, w[bias], storage))
            done.add(bias)
            attrs: Dict[str, object] = {"kernel_shape": [int(arr.shape[2])]}
            op = "Conv"
            m = _RES.fullmatch(name)
            if name.startswith("voc.up."):
                op = "ConvTranspose"
                i = int(name.split(".")[2])
                attrs.update(strides=[int(spec.voc_up_rates[i])], pads=[(arr.shape[2] - spec.voc_up_rates[i]) // 2] * 2, group=1, dilations=[1])
            elif m:
                d = spec.voc_res_dilations[int(m[3])] if m[4] == "1" else 1
                attrs.update(dilations=[int(d)], strides=[1], pads=[d * (arr.shape[2] - 1) // 2] * 2, group=1)
            elif name.startswith("input.pos_conv"):
                attrs.update(group=int(spec.pos_conv_groups), dilations=[1], strides=[1], pads=[arr.shape[2] // 2] * 2)
            elif ".dwconv." in name:
                attrs.update(group=int(arr.shape[0]), dilations=[1], strides=[1], pads=[arr.shape[2] // 2] * 2)
            else:
                attrs.update(group=1, dilations=[1], strides=[1], pads=[arr.shape[2] // 2] * 2)
            stem = ename[: -len(".weight")]
            nodes.append(OnnxNode(op, f"/{stem}/{op}", [f"/{stem}/in", ename, bname], [f"/{stem}/{op}_output_0"], attrs))
        else:                                                     # 1-D named parameters (LayerNorm weight)
            inits.append((ename, arr, storage))
        done.add(name)
    for name in w:                                                # biases of non-linear / non-conv parameters (LayerNorm bias)
        if name not in done:
            files[where(name)][1].append((exporter_names(name, n_k)[0], w[name], storage))
    io_ = {"preprocess": ([OnnxValue("audio", 5, (1, 1, "audio_len")), OnnxValue("text_ids", 6, (1, "text_len")), OnnxValue("max_duration", 7, (1,))],
                          [OnnxValue("noise", 1, (1, "max_duration", spec.n_mel))]),
           "transformer": ([OnnxValue("noise", 1, (1, "max_duration", spec.n_mel)), OnnxValue("time_step", 6, (1,))],
                           [OnnxValue("denoised", 1, (1, "max_duration", spec.n_mel)), OnnxValue("time_step_out", 6, (1,))]),
           "decode": ([OnnxValue("denoised", 1, (1, "max_duration", spec.n_mel)), OnnxValue("ref_signal_len", 7, (1,))],
                      [OnnxValue("generated_signal", 5, (1, 1, "out_len"))])}
    return {k + ".onnx": encode_model(files[k][0], files[k][1], io_[k][0], io_[k][1], graph_name="main_graph") for k in files}


def write_onnx_archive(path: str, spec, weights, extra_members: Optional[Dict[str, bytes]] = None, storage: str = "raw") -> None:
    members = dict(export_archive_members(spec, weights, storage))
    members.update(extra_members or {})
    with tarfile.open(path, "w") as tar:
        for name, data in members.items():
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
