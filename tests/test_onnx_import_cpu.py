"""-m "not gpu": N1, the ONNX initializer importer.  PARITY UNPINNED against the real archive (it cannot be fetched,
SURVEY 8c): the wire-format reader is checked against hand-assembled protobuf bytes and its own writer, the name recovery and
the architecture inference on graphs written with exporter-style structure (anonymous transposed MatMul weights + named
biases, Conv / ConvTranspose attributes), and the engine plumbing on an archive in the reference's layout."""
import io
import json
import tarfile

import numpy as np
import pytest
import torch

from vietvoice_tts_amd import onnx_import as oi
from tests import onnx_fixture_writer as ow
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights


def test_wire_format_known_bytes():
    # hand-assembled TensorProto: dims [2,3] (field 1, unpacked varints), data_type 1, name "w" (field 8), float_data packed (field 4)
    vals = np.arange(6, dtype="<f4")
    raw = bytes([0x08, 2, 0x08, 3, 0x10, 1, 0x42, 1, ord("w"), 0x22, 24]) + vals.tobytes()
    name, arr = oi._tensor(memoryview(raw))
    assert name == "w" and arr.dtype == np.float32 and arr.shape == (2, 3) and np.array_equal(arr.reshape(-1), vals)
    # varints: 300 = 0xAC 0x02; -1 as int64 = ten bytes
    assert oi._varint(memoryview(bytes([0xAC, 0x02])), 0) == (300, 2)
    assert ow._enc_varint(300) == bytes([0xAC, 0x02]) and len(ow._enc_varint(-1)) == 10
    assert oi._sint(oi._varint(memoryview(ow._enc_varint(-7)), 0)[0]) == -7
    with pytest.raises(ValueError):
        list(oi._fields(memoryview(bytes([0x0A, 0x7F, 1, 2]))))            # length runs past the buffer


@pytest.mark.parametrize("how,dtype", [("raw", np.float32), ("typed", np.float32), ("typed", np.int64), ("raw", np.int64), ("typed", np.float16),
                                       ("typed", np.int32), ("raw", np.float64), ("typed", np.float64), ("raw", np.uint8)])
def test_tensor_storage_round_trips(how, dtype):
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((3, 4, 5)) * 50).astype(dtype)
    if np.issubdtype(dtype, np.signedinteger):
        a[0, 0, 0] = -12345
    name, b = oi._tensor(memoryview(ow.encode_tensor("some.name", a, how)))
    assert name == "some.name" and b.dtype == a.dtype and np.array_equal(a, b)
    s_name, s = oi._tensor(memoryview(ow.encode_tensor("scalar", np.array(3.5, np.float32), how if dtype == np.float32 else "raw")))
    assert s.shape == () and float(s) == 3.5


def test_bf16_and_external_data():
    a = np.random.default_rng(2).standard_normal(64).astype(np.float32)
    _, b = oi._tensor(memoryview(ow.encode_tensor("x", a, "bf16")))
    assert b.dtype == np.float32 and np.array_equal(b, torch.from_numpy(a).bfloat16().float().numpy())
    ext = ow._vi(1, 64) + ow._vi(2, 1) + ow._ld(8, b"big") + ow._vi(14, 1)
    with pytest.raises(oi.UnsupportedGraph, match="external"):
        oi._tensor(memoryview(ext))


def test_model_round_trip_nodes_attrs_values():
    nodes = [oi.OnnxNode("MatMul", "/l/MatMul", ["x", "onnx::MatMul_7"], ["/l/MatMul_output_0"]),
             oi.OnnxNode("Add", "/l/Add", ["l.bias", "/l/MatMul_output_0"], ["y"]),
             oi.OnnxNode("Conv", "/c/Conv", ["y", "c.weight", "c.bias"], ["z"],
                         {"dilations": [3], "group": 1, "kernel_shape": [7], "pads": [9, 9], "strides": [1], "alpha": 0.5, "mode": b"linear",
                          "value": np.arange(4, dtype=np.int64)})]
    w = np.arange(12, dtype=np.float32).reshape(4, 3)                      # stored [in, out]
    inits = [("onnx::MatMul_7", w, "raw"), ("l.bias", np.ones(3, np.float32), "typed"), ("c.weight", np.zeros((2, 3, 7), np.float32), "raw"),
             ("c.bias", np.zeros(2, np.float32), "raw")]
    data = ow.encode_model(nodes, inits, [oi.OnnxValue("x", 1, (1, "n", 4))], [oi.OnnxValue("z", 1, (1, 2, "n"))])
    m = oi.parse_model(data)
    assert m.ir_version == 8 and m.opset == {"": 17} and m.producer == "pytorch" and m.graph_name == "main_graph"
    assert [n.op_type for n in m.nodes] == ["MatMul", "Add", "Conv"] and m.nodes[2].inputs == ["y", "c.weight", "c.bias"]
    a = m.nodes[2].attrs
    assert a["dilations"] == [3] and a["pads"] == [9, 9] and a["group"] == 1 and a["alpha"] == 0.5 and a["mode"] == b"linear"
    assert np.array_equal(a["value"], np.arange(4))
    assert m.inputs[0].shape == (1, "n", 4) and m.outputs[0].name == "z" and m.outputs[0].elem_type == 1
    named = oi.recover_names(m)
    assert set(named) == {"l.weight", "l.bias", "c.weight", "c.bias"} and np.array_equal(named["l.weight"], w.T)
    assert oi.summarize(m)["ops"] == {"Add": 1, "Conv": 1, "MatMul": 1}
    with pytest.raises(ValueError):
        oi.parse_model(b"\x08\x08")                                       # a ModelProto without a graph


@pytest.mark.parametrize("preset,storage", [("tiny", "raw"), ("small", "typed")])
def test_import_recovers_spec_and_weights(preset, storage):
    spec = getattr(ModelSpec, preset)()
    w = make_synthetic_weights(spec, 77)
    files = ow.export_archive_members(spec, w, storage)
    models = {k: oi.parse_model(v) for k, v in files.items()}
    assert any(k.startswith("onnx::MatMul_") for k in models["transformer.onnx"].initializers)        # weights really are anonymous
    spec2, w2 = oi.import_graphs(models["preprocess.onnx"], models["transformer.onnx"], models["decode.onnx"], base=spec)
    assert spec2 == spec
    assert set(w2) == set(w)
    for k in w:
        assert w2[k].shape == w[k].shape and torch.equal(w2[k], w[k].float()), k
    # inference from the graphs alone (base = the full preset): everything the shapes determine must come out right
    spec3, _ = oi.import_graphs(models["preprocess.onnx"], models["transformer.onnx"], models["decode.onnx"])
    for f in ("n_mel", "dim", "depth", "heads", "ff_mult", "text_dim", "text_layers", "text_conv_k", "vocab_size", "pos_conv_k", "pos_conv_groups",
              "time_freq_dim", "voc_pre_ch", "voc_up_rates", "voc_up_kernels", "voc_res_kernels", "voc_res_dilations", "voc_pre_k", "voc_post_k"):
        assert getattr(spec3, f) == getattr(spec, f), f


def test_unsupported_graphs_are_named():
    spec = ModelSpec.tiny()
    files = ow.export_archive_members(spec, make_synthetic_weights(spec, 1))
    pre, tr, dec = (oi.parse_model(files[k + ".onnx"]) for k in ("preprocess", "transformer", "decode"))
    dec_no_up = oi.OnnxModel(dec.ir_version, dec.opset, dec.producer, dec.graph_name, [n for n in dec.nodes if n.op_type != "ConvTranspose"],
                             dec.initializers, dec.inputs, dec.outputs)
    with pytest.raises(oi.UnsupportedGraph, match="no ConvTranspose"):
        oi.import_graphs(pre, tr, dec_no_up)
    dec.initializers.pop("conv_pre.bias")
    with pytest.raises(oi.UnsupportedGraph, match="conv_pre.bias"):
        oi.import_graphs(pre, tr, dec)


def test_engine_loads_reference_layout_archive(tmp_path):
    """An archive in the reference's layout (three .onnx + vocab.txt + audio_metadata.json + cleaned_audios/, no model_spec.json)
    loads through ModelSessionManager and synthesises on injected oracle sessions exactly like the synthetic pack does."""
    from oracle.vv_oracle import Oracle, OracleSession
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    from vietvoice_tts_amd.model_pack import write_synthetic_pack
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(), d2.mkdir()
    write_synthetic_pack(str(d1 / "model-bin.pt"), "tiny", seed=9527)
    spec = ModelSpec.tiny()
    members = {}
    with tarfile.open(d1 / "model-bin.pt") as tar:
        for n in tar.getnames():
            if n != "model_spec.json":
                members["pack/" + n if n.endswith("vocab.txt") else n] = tar.extractfile(n).read()       # nested path: matched by suffix
    ow.write_onnx_archive(str(d2 / "model-bin.pt"), spec, make_synthetic_weights(spec, 9527), {k: v for k, v in members.items()})
    seen = {}

    def factory(sp, weights, cfg):
        seen["spec"], seen["w"] = sp, weights
        orc = Oracle(sp, weights, nfe_step=cfg.nfe_step)
        return {k: OracleSession(orc, k, seed=cfg.random_seed) for k in ("preprocess", "transformer", "decode")}

    outs = []
    for d in (d1, d2):
        e = TTSEngine(ModelConfig(model_cache_dir=str(d), nfe_step=3, model_spec="tiny"), session_factory=factory)
        assert seen["spec"].dim == spec.dim and seen["spec"].depth == spec.depth and seen["spec"].voc_up_rates == spec.voc_up_rates
        outs.append(e.synthesize("Xin chào.")[0])
        e.cleanup()
    assert np.array_equal(outs[0], outs[1])


def test_reader_on_hand_assembled_protobuf():
    """The reader against bytes that do NOT come from tests/onnx_fixture_writer.py: a ModelProto assembled here byte by byte from the
    public onnx.proto3 field numbers (ModelProto.graph = 7, GraphProto.node = 1 / name = 2 / initializer = 5 / input = 11 / output = 12,
    NodeProto.input = 1 / output = 2 / name = 3 / op_type = 4 / attribute = 5, AttributeProto.name = 1 / i = 3 / ints = 8 / type = 20,
    TensorProto.dims = 1 / data_type = 2 / float_data = 4 / int64_data = 7 / name = 8 / raw_data = 9; wire types 0 = varint,
    2 = length-delimited, 5 = fixed32) -- so a field number that writer and reader both have wrong would show here."""
    import struct
    f32 = lambda *v: struct.pack("<%df" % len(v), *v)
    # TensorProto "w": dims [2, 3], FLOAT (1), raw_data
    t_w = bytes([0x08, 2, 0x08, 3, 0x10, 1, 0x42, 1]) + b"w" + bytes([0x4A, 24]) + f32(1, 2, 3, 4, 5, 6)
    # TensorProto "b": dims packed [3] (field 1, wire type 2), FLOAT, float_data packed (field 4)
    t_b = bytes([0x0A, 1, 3, 0x10, 1, 0x42, 1]) + b"b" + bytes([0x22, 12]) + f32(0.5, -1.5, 2.25)
    # TensorProto "idx": dims [2], INT64 (7), int64_data packed (field 7): 300 = AC 02, -1 = ten bytes
    t_i = bytes([0x08, 2, 0x10, 7, 0x42, 3]) + b"idx" + bytes([0x3A, 12, 0xAC, 0x02]) + bytes([0xFF] * 9 + [0x01])
    # AttributeProto group = 4 (INT: type 2) and kernel_shape = [7] (INTS: type 7, unpacked field 8)
    a_g = bytes([0x0A, 5]) + b"group" + bytes([0x18, 4, 0xA0, 0x01, 2])
    a_k = bytes([0x0A, 12]) + b"kernel_shape" + bytes([0x40, 7, 0xA0, 0x01, 7])
    node = (bytes([0x0A, 1]) + b"x" + bytes([0x0A, 1]) + b"w" + bytes([0x0A, 1]) + b"b" + bytes([0x12, 1]) + b"y" + bytes([0x1A, 5]) + b"/c/Cv"
            + bytes([0x22, 4]) + b"Conv" + bytes([0x2A, len(a_g)]) + a_g + bytes([0x2A, len(a_k)]) + a_k)
    graph = (bytes([0x0A, len(node)]) + node + bytes([0x12, 1]) + b"g" + bytes([0x2A, len(t_w)]) + t_w + bytes([0x2A, len(t_b)]) + t_b
             + bytes([0x2A, len(t_i)]) + t_i)
    assert 128 <= len(graph) < 16384
    model = bytes([0x08, 8]) + bytes([0x3A, 0x80 | (len(graph) & 0x7F), len(graph) >> 7]) + graph     # ir_version = 8; graph, two-byte length varint
    m = oi.parse_model(model)
    assert [n.op_type for n in m.nodes] == ["Conv"] and m.nodes[0].name == "/c/Cv"
    assert m.nodes[0].inputs == ["x", "w", "b"] and m.nodes[0].outputs == ["y"]
    assert m.nodes[0].attrs["group"] == 4 and list(m.nodes[0].attrs["kernel_shape"]) == [7]
    assert m.initializers["w"].dtype == np.float32 and m.initializers["w"].tolist() == [[1, 2, 3], [4, 5, 6]]
    assert m.initializers["b"].tolist() == [0.5, -1.5, 2.25]
    assert m.initializers["idx"].dtype == np.int64 and m.initializers["idx"].tolist() == [300, -1]
