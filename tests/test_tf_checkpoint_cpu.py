"""TF checkpoint (tensor bundle) reader/writer: round trip, table-format structure checks, CRC known answers, scope mapping.
Parity against a checkpoint written by real TensorFlow is unpinned (none exists offline) -- see the module header."""
import os
import struct

import numpy as np
import pytest

from stabnet_amd import tf_checkpoint as C


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert C.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert C.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert C.crc32c(bytes(range(32))) == 0x46DD794E
    assert C.crc32c(b"123456789") == 0xE3069283
    assert C.crc32c(b"6789", C.crc32c(b"12345")) == 0xE3069283          # incremental


def test_bundle_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    names = ["stable_net/resnet/resnet_v2_50/block%d/unit_%d/bottleneck_v2/conv%d/weights" % (b, u, c)
             for b in range(1, 5) for u in range(1, 4) for c in range(1, 4)]
    var = {n: rng.standard_normal((1, 1, 8, 4)).astype(np.float32) for n in names}      # enough keys for several table blocks
    var["stable_net/resnet/fc/fc_weights"] = rng.standard_normal((512, 50)).astype(np.float32)
    var["stable_net/resnet/fc/fc_weights/Adam"] = np.zeros((512, 50), np.float32)
    var["global_step"] = np.array(80000, np.int64)
    var["beta1_power"] = np.array(0.9 ** 5, np.float32)
    prefix = str(tmp_path / "model-80000")
    C.write_bundle(prefix, var, block_size=512)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    got = C.read_bundle(prefix, verify_crc_below=1 << 30)
    assert set(got) == set(var)
    for k in var:
        assert got[k].dtype == var[k].dtype and got[k].shape == var[k].shape and np.array_equal(got[k], var[k]), k
    params, extras = C.load_stabnet_variables(prefix)
    assert "fc/fc_weights" in params and "resnet_v2_50/block1/unit_1/bottleneck_v2/conv1/weights" in params
    assert "stable_net/resnet/fc/fc_weights/Adam" in extras and int(extras["global_step"]) == 80000
    assert not any(k.endswith("/Adam") for k in params)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    C.write_bundle(prefix, {"a": np.arange(6, dtype=np.float32).reshape(2, 3)})
    data = prefix + ".data-00000-of-00001"
    b = bytearray(open(data, "rb").read()); b[5] ^= 0x40
    open(data, "wb").write(bytes(b))
    with pytest.raises(ValueError, match="crc mismatch"):
        C.read_bundle(prefix)
    idx = prefix + ".index"
    b = bytearray(open(idx, "rb").read()); b[-1] ^= 1
    open(idx, "wb").write(bytes(b))
    with pytest.raises(ValueError, match="bad magic"):
        C.read_bundle(prefix)


def test_checkpoint_feeds_the_plan_layout(tmp_path):
    """A full synthetic StabNet parameter set written as a TF checkpoint under the reference's scope comes back as the dict
    NetPlan.pack() consumes (names, HWIO / [in,out] layouts untouched)."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    cfg = Config(height=64, width=64)
    P = synthetic.make_params(cfg, seed=0, theta_scale=0.2)
    small = {k: v for k, v in P.items() if v.size <= 4096}                               # keep the CPU test quick
    prefix = str(tmp_path / "model-1")
    C.write_bundle(prefix, {C.STABNET_SCOPE + k: v for k, v in small.items()})
    params, extras = C.load_stabnet_variables(prefix)
    assert not extras and set(params) == set(small)
    assert all(np.array_equal(params[k], small[k]) for k in small)


def test_imagenet_init_formats(tmp_path):
    """The reference warm-starts from `data_video/resnet_v2_50.ckpt` (train_bundle_nobm.py:184,208), which slim ships as a V1
    single-file checkpoint: a bare table file without `.index`.  Both formats initialise the backbone; conv1 (13 input channels
    instead of 3) and the classifier are excluded."""
    rng = np.random.default_rng(1)
    var = {"resnet_v2_50/block1/unit_1/bottleneck_v2/conv1/weights": rng.standard_normal((1, 1, 64, 64)).astype(np.float32),
           "resnet_v2_50/conv1/weights": rng.standard_normal((7, 7, 3, 64)).astype(np.float32),
           "resnet_v2_50/logits/weights": rng.standard_normal((1, 1, 2048, 1001)).astype(np.float32)}
    v2 = str(tmp_path / "v2" / "resnet_v2_50.ckpt")
    os.makedirs(os.path.dirname(v2))
    C.write_bundle(v2, var)
    assert C.checkpoint_format(v2) == "v2"
    got, note = C.try_load_imagenet_resnet(v2)
    assert set(got) == {"resnet_v2_50/block1/unit_1/bottleneck_v2/conv1/weights"} and "V2" in note   # conv1 + logits excluded
    v1 = str(tmp_path / "v1" / "resnet_v2_50.ckpt")
    C.write_v1(v1, var)
    assert C.checkpoint_format(v1) == "v1"
    got1, note = C.try_load_imagenet_resnet(v1)
    assert set(got1) == set(got) and "V1" in note
    assert all(np.array_equal(got1[k], var[k]) for k in got1)
    junk = str(tmp_path / "junk.ckpt")
    open(junk, "wb").write(b"not a checkpoint" * 10)
    assert C.checkpoint_format(junk) == "unknown" and C.try_load_imagenet_resnet(junk)[0] is None
    assert C.checkpoint_format(str(tmp_path / "absent")) == "none"
    assert "not found" in C.try_load_imagenet_resnet(str(tmp_path / "absent"))[1]


def test_v1_checkpoint_layouts(tmp_path):
    """V1 single-file checkpoints (tensor_slice_writer.cc): values in the TensorProto's typed repeated field, packed (what
    proto3 writes) or one element per tag, all four dtypes, scalars, a partitioned variable assembled from its slices, and
    Snappy-compressed table blocks."""
    rng = np.random.default_rng(2)
    var = {"a/weights": rng.standard_normal((3, 3, 5, 7)).astype(np.float32), "a/step": np.array(12345678901, np.int64),
           "a/d": rng.standard_normal((4, 2)), "a/i": rng.integers(-1000, 1000, (9,), dtype=np.int32),
           "part/w": rng.standard_normal((10, 6)).astype(np.float32), "z/empty_dim": np.zeros((0, 4), np.float32)}
    for packed in (True, False):
        path = str(tmp_path / ("v1_%d.ckpt" % packed))
        C.write_v1(path, var, packed=packed, slices_of={"part/w": 3})
        got = C.read_v1(path)
        assert set(got) == set(var)
        for k in var:
            assert got[k].dtype == var[k].dtype and got[k].shape == var[k].shape and np.array_equal(got[k], var[k]), k
        assert C.read_checkpoint(path).keys() == got.keys()
    # the Snappy block decoder (format_description.txt): one literal; literal + overlapping copy; a copy from before the start
    assert C._snappy_decompress(bytes([11, 0x28]) + b"ab" + b"cdefghi" + b"jk") == b"abcdefghijk"[:11]
    # Snappy: literal "abcd" + copy(offset 4, length 8) -> "abcdabcdabcd"
    assert C._snappy_decompress(bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])) == b"abcdabcdabcd"
    with pytest.raises(ValueError):
        C._snappy_decompress(bytes([4, 0x01 | (0 << 2), 9]))                 # copy from before the start


def test_imagenet_init_is_strict():
    """`tf.train.Saver(vtr).restore` (train_bundle_nobm.py:185-191,208) fails on a missing or mis-shaped variable; so does the warm
    start here: a truncated / foreign checkpoint must not turn into training from the seeded initialiser."""
    from stabnet_amd import synthetic
    from stabnet_amd.config import Config
    init = synthetic.make_params(Config(height=64, width=64), seed=0, theta_scale=0.2)
    want = C.imagenet_expected_names(init)
    assert want and all(k.startswith("resnet_v2_50/") for k in want)
    assert not any(k.startswith("resnet_v2_50/conv1/") or k.startswith("fc/") for k in want)
    assert "resnet_v2_50/postnorm/gamma" in want and "resnet_v2_50/block4/unit_3/bottleneck_v2/conv3/biases" in want
    pre = {k: np.full_like(init[k], 0.5) for k in want}
    pre["resnet_v2_50/logits/weights"] = np.zeros((1, 1, 2048, 1001), np.float32)        # extras in the file are ignored
    fresh = {k: v.copy() for k, v in init.items()}
    assert C.apply_imagenet_init(fresh, pre) == len(want)
    assert all(np.all(fresh[k] == 0.5) for k in want)
    assert np.array_equal(fresh["resnet_v2_50/conv1/weights"], init["resnet_v2_50/conv1/weights"])    # stem and head untouched
    assert np.array_equal(fresh["fc/fc_weights"], init["fc/fc_weights"])
    short = dict(pre)
    del short[want[3]]
    with pytest.raises(ValueError, match="missing"):
        C.apply_imagenet_init({k: v.copy() for k, v in init.items()}, short)
    bad = dict(pre)
    bad[want[5]] = np.zeros(tuple(init[want[5]].shape) + (2,), np.float32)
    with pytest.raises(ValueError, match="wrong shape"):
        C.apply_imagenet_init({k: v.copy() for k, v in init.items()}, bad)
