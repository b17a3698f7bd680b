"""TensorFlow V2 checkpoint bundles: the product reader/writer (rcgan_amd/tf_bundle.py, CRC-32C through the C ABI)
against the independent restatement in oracle/tf_bundle_ref.py, known answers of the checksum, the table layout's
structural invariants, corruption detection, and the Saver / latest_checkpoint / load_checkpoint round trip."""
import os
import struct

import numpy as np
import pytest

import rcgan_amd  # noqa: F401
from oracle import tf_bundle_ref as R
from rcgan_amd import host, tf_bundle as P


def _tensors(seed=0, big=False):
    rs = np.random.RandomState(seed)
    t = {
        "Generator.Input/Generator.Input.W": rs.randn(128, 512 if not big else 4096).astype(np.float32),
        "Generator.Input/Generator.Input.W/Adam": rs.randn(128, 512 if not big else 4096).astype(np.float32),
        "Generator.Input/Generator.Input.b": np.zeros(512, np.float32),
        "Discriminator.1.Conv1/Discriminator.1.Conv1.Filters": rs.randn(3, 3, 3, 128).astype(np.float32),
        "Discriminator.1.Conv1/u": rs.randn(1, 128).astype(np.float32),
        "confusion_logits": rs.randn(10, 10).astype(np.float32),
        "beta1_power": np.float32(0.0),
        "beta2_power": np.float32(0.9 ** 7),
        "_opt/Generator/step": np.array([6], np.int64),
        "empty": np.zeros((0, 4), np.float32),
        "flags": np.array([True, False, True]),
        "labels": np.arange(12, dtype=np.int32).reshape(3, 4),
    }
    return t


def _same(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), k


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors + the classic check value
    assert P.crc32c(b"123456789") == 0xE3069283
    assert P.crc32c(bytes(32)) == 0x8A9136AA
    assert P.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert P.crc32c(bytes(range(32))) == 0x46DD794E
    assert P.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    data = np.random.RandomState(1).bytes(1000)
    assert P.crc32c(data) == R.crc32c_bitwise(data)
    assert P.crc32c(data[300:], P.crc32c(data[:300])) == P.crc32c(data)          # continuation
    # leveldb's crc32c_test: Mask(crc("foo")) differs from crc, unmask inverts it
    c = P.crc32c(b"foo")
    assert P.mask_crc(c) != c and P.unmask_crc(P.mask_crc(c)) == c and P.mask_crc(c) == R.masked(c)


def test_product_writer_oracle_reader(tmp_path):
    t = _tensors()
    P.write_bundle(str(tmp_path / "model.ckpt-7"), t)
    _same(R.read_bundle(str(tmp_path / "model.ckpt-7")), t)
    _same(P.read_bundle(str(tmp_path / "model.ckpt-7")), t)


def test_oracle_writer_product_reader(tmp_path):
    t = _tensors(3)
    R.write_bundle(str(tmp_path / "b"), t)
    _same(P.read_bundle(str(tmp_path / "b")), t)


def test_layout_invariants(tmp_path):
    t = _tensors()
    prefix = str(tmp_path / "m")
    P.write_bundle(prefix, t)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    assert len(raw[-48:]) == 48
    entries = P.read_table(prefix + ".index")
    assert entries[0] == (b"", bytes.fromhex("08011a020801"))                    # num_shards=1, version{producer=1}
    keys = [k for k, _ in entries]
    assert keys == sorted(keys)
    # data file = the tensors back to back in key order, no padding
    off = 0
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    for k, v in entries[1:]:
        e = P.parse_proto(v)
        assert e.get(4, [0])[0] == off and e.get(3, [0])[0] == 0
        a = np.asarray(t[k.decode()])
        assert e[5][0] == a.nbytes and data[off:off + a.nbytes] == a.tobytes()
        off += a.nbytes
    assert off == len(data)


def test_many_entries_span_several_table_blocks(tmp_path, monkeypatch):
    monkeypatch.setattr(P, "BLOCK_SIZE", 512)
    rs = np.random.RandomState(5)
    t = {"scope_%03d/var_with_a_long_common_prefix/w" % i: rs.randn(3, i % 5 + 1).astype(np.float32) for i in range(200)}
    prefix = str(tmp_path / "many")
    P.write_bundle(prefix, t)
    raw = open(prefix + ".index", "rb").read()
    assert len(raw) > 4 * 512
    _same(R.read_bundle(prefix), t)
    _same(P.read_bundle(prefix), t)


def test_corruption_is_detected(tmp_path):
    t = _tensors()
    prefix = str(tmp_path / "c")
    P.write_bundle(prefix, t)
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    d[100] ^= 0x01
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError, match="checksum"):
        P.read_bundle(prefix)
    P.write_bundle(prefix, t)
    i = bytearray(open(prefix + ".index", "rb").read())
    i[10] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(i))
    with pytest.raises(ValueError, match="checksum"):
        P.read_bundle(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError, match="magic|index"):
        P.read_bundle(prefix)


def test_saver_round_trip_and_retention(tmp_path):
    d = str(tmp_path / "ck")
    s = host.Saver(max_to_keep=2)
    for step in (1, 2, 3):
        t = _tensors(step)
        s.save(t, d, "model.ckpt", step)
    assert not P.exists(os.path.join(d, "model.ckpt-1")) and P.exists(os.path.join(d, "model.ckpt-2"))
    text = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert text[0] == 'model_checkpoint_path: "model.ckpt-3"'
    assert text[1:] == ['all_model_checkpoint_paths: "model.ckpt-2"', 'all_model_checkpoint_paths: "model.ckpt-3"']
    latest = host.latest_checkpoint(d)
    assert latest == os.path.join(d, "model.ckpt-3")
    _same(host.load_checkpoint(latest), _tensors(3))
    _same(R.read_bundle(latest), _tensors(3))


def test_npz_checkpoints_of_earlier_runs_still_load(tmp_path):
    t = {"a/b": np.arange(4, dtype=np.float32)}
    np.savez(str(tmp_path / "old-5.npz"), **{k.replace("/", "|"): v for k, v in t.items()})
    open(str(tmp_path / "checkpoint"), "w").write('model_checkpoint_path: "old-5"\n')
    p = host.latest_checkpoint(str(tmp_path))
    _same(host.load_checkpoint(p), t)


def test_adam_power_tensors_round_trip():
    pw = host.adam_power_tensors([(0, 0.0, 0.9), (41, 0.5, 0.999), (7, 0.0, 0.9)])
    assert pw["beta1_power"] == 0.0 and np.isclose(pw["beta2_power"], 0.9)
    assert np.isclose(pw["beta2_power_1"], 0.999 ** 42) and np.isclose(pw["beta1_power_1"], 0.5 ** 42)
    assert host.steps_from_beta_power(float(pw["beta2_power"]), 0.9) == 0
    assert host.steps_from_beta_power(float(pw["beta2_power_1"]), 0.999) == 41
    assert host.steps_from_beta_power(float(pw["beta2_power_2"]), 0.9) == 7
    # a bundle written by TensorFlow after > ~830 steps holds an underflowed fp32 beta2_power: "many steps", never 0
    old = host.adam_power_tensors([(5000, 0.0, 0.9)])
    assert float(old["beta2_power"]) == 0.0 and host.steps_from_beta_power(float(old["beta2_power"]), 0.9) >= 100000
    assert host.steps_from_beta_power(float(np.float32(0.9) ** np.float32(850)), 0.9) >= 100000      # subnormal
