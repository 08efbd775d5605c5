"""CPU: the SequenceExample wire codec of e2e_asr_amd/tfrecord.py against an INDEPENDENT implementation -- Google's protobuf
runtime, with the message types of tensorflow/core/example/{example,feature}.proto declared on the fly (no TensorFlow).
Records written by our writer must parse with protobuf to the same values, and records serialised by protobuf (packed and
unpacked repeated fields) must parse with our reader -- the record layout of speech_dataset.py:15-45."""
import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory   # noqa: E402

from e2e_asr_amd import tfrecord   # noqa: E402


def _messages():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "asr_example_test.proto"; fd.package = "asrtest"; fd.syntax = "proto3"
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add(); m.name = name; return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        f = m.field.add(); f.name = name; f.number = num; f.type = typ; f.label = label
        if type_name: f.type_name = ".asrtest." + type_name
        if packed is not None: f.options.packed = packed
        if oneof is not None: f.oneof_index = oneof
        return f
    m = msg("BytesList"); field(m, "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    m = msg("FloatList"); field(m, "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    m = msg("Int64List"); field(m, "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    m = msg("FloatListU"); field(m, "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=False)      # unpacked variants:
    m = msg("Int64ListU"); field(m, "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=False)      # same wire numbers
    m = msg("Feature"); m.oneof_decl.add().name = "kind"
    field(m, "bytes_list", 1, T.TYPE_MESSAGE, type_name="BytesList", oneof=0)
    field(m, "float_list", 2, T.TYPE_MESSAGE, type_name="FloatList", oneof=0)
    field(m, "int64_list", 3, T.TYPE_MESSAGE, type_name="Int64List", oneof=0)
    m = msg("FeatureU"); m.oneof_decl.add().name = "kind"
    field(m, "bytes_list", 1, T.TYPE_MESSAGE, type_name="BytesList", oneof=0)
    field(m, "float_list", 2, T.TYPE_MESSAGE, type_name="FloatListU", oneof=0)
    field(m, "int64_list", 3, T.TYPE_MESSAGE, type_name="Int64ListU", oneof=0)
    for suffix in ("", "U"):
        fe = msg("FeaturesEntry" + suffix); fe.options.map_entry = True
        field(fe, "key", 1, T.TYPE_STRING); field(fe, "value", 2, T.TYPE_MESSAGE, type_name="Feature" + suffix)
        m = msg("Features" + suffix); field(m, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="FeaturesEntry" + suffix)
        m = msg("FeatureList" + suffix); field(m, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="Feature" + suffix)
        fe = msg("FeatureListsEntry" + suffix); fe.options.map_entry = True
        field(fe, "key", 1, T.TYPE_STRING); field(fe, "value", 2, T.TYPE_MESSAGE, type_name="FeatureList" + suffix)
        m = msg("FeatureLists" + suffix); field(m, "feature_list", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="FeatureListsEntry" + suffix)
        m = msg("SequenceExample" + suffix)
        field(m, "context", 1, T.TYPE_MESSAGE, type_name="Features" + suffix)
        field(m, "feature_lists", 2, T.TYPE_MESSAGE, type_name="FeatureLists" + suffix)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("asrtest." + n))
    return get("SequenceExample"), get("SequenceExampleU")


def _entries(mp):
    """A protobuf map field as a plain dict {key: value message}."""
    return dict(mp.items())


def _utt(rng, T=7, F=5):
    return dict(segment=b"sw02001-A_000098-001156", logmel=rng.standard_normal((T, F)).astype(np.float32),
                cint=rng.integers(0, 1000, 9).astype(np.int64), pint=rng.integers(0, 50, 12).astype(np.int64))


def test_our_writer_parses_with_protobuf():
    SE, _ = _messages()
    rng = np.random.default_rng(1)
    u = _utt(rng)
    rec = tfrecord.make_sequence_example(
        dict(segment=u["segment"], logmel_len=np.int64(7), cint_len=np.int64(9), pint_len=np.int64(12)),
        dict(logmel=[row for row in u["logmel"]], cint=[np.int64(v) for v in u["cint"]], pint=[np.int64(v) for v in u["pint"]]))
    m = SE(); m.ParseFromString(bytes(rec))
    ctx = _entries(m.context.feature)
    assert ctx["segment"].bytes_list.value[0] == u["segment"]
    assert list(ctx["logmel_len"].int64_list.value) == [7] and list(ctx["cint_len"].int64_list.value) == [9]
    fl = _entries(m.feature_lists.feature_list)
    got = np.array([list(f.float_list.value) for f in fl["logmel"].feature], np.float32)
    np.testing.assert_array_equal(got, u["logmel"])
    assert [f.int64_list.value[0] for f in fl["cint"].feature] == list(u["cint"])
    assert [f.int64_list.value[0] for f in fl["pint"].feature] == list(u["pint"])


@pytest.mark.parametrize("packed", [True, False])
def test_protobuf_serialisation_parses_with_our_reader(packed):
    SE, SEU = _messages()
    M = SE if packed else SEU
    rng = np.random.default_rng(2)
    u = _utt(rng, T=11, F=80)
    m = M()

    def ctx(key):
        return m.context.feature[key]
    ctx("segment").bytes_list.value.append(u["segment"])
    ctx("logmel_len").int64_list.value.append(11)
    ctx("cint_len").int64_list.value.append(9)
    ctx("pint_len").int64_list.value.append(-3 % (1 << 63))          # a large varint
    def flist(key):
        return m.feature_lists.feature_list[key]
    fl = flist("logmel")
    for row in u["logmel"]:
        fl.feature.add().float_list.value.extend([float(x) for x in row])
    fl = flist("cint")
    for v in u["cint"]:
        fl.feature.add().int64_list.value.append(int(v))
    fl = flist("pint")
    for v in u["pint"]:
        fl.feature.add().int64_list.value.append(int(v))
    context, lists = tfrecord.parse_sequence_example(m.SerializeToString())
    assert bytes(context["segment"][0]) == u["segment"] and int(context["logmel_len"][0]) == 11
    assert int(context["pint_len"][0]) == -3 % (1 << 63)
    np.testing.assert_array_equal(np.stack([np.asarray(f, np.float32) for f in lists["logmel"]]), u["logmel"])
    assert [int(f[0]) for f in lists["cint"]] == list(u["cint"]) and [int(f[0]) for f in lists["pint"]] == list(u["pint"])
