"""checkpoint / array formats (reference src/utils.py:20-75).  flax is absent and the reference's
checkpoint/ directory is not shipped, so the msgpack reader is pinned by (a) a byte-level fixture written here with
plain msgpack according to the documented Flax wire format and (b) a save -> load round trip."""
import os

import msgpack
import numpy as np
import pytest
import torch

from lip_amd.checkpoint import (load_array_checkpoint, load_checkpoint, msgpack_restore, msgpack_serialize,
                                save_array_checkpoint, save_checkpoint)
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import SimpleClassifier, create_state


def test_array_checkpoint_roundtrip(tmp_path):
    Z = torch.randn(5, 2)
    save_array_checkpoint(Z, tmp_path, "Z", 150)
    assert os.path.exists(tmp_path / "Z_150.npy")
    assert torch.equal(load_array_checkpoint(tmp_path, "Z", 150), Z)
    with pytest.raises(FileNotFoundError):
        load_array_checkpoint(tmp_path, "Z", 7)


def test_flax_wire_format_fixture():
    """Bytes produced exactly as flax.serialization._ndarray_to_bytes documents them: ExtType(1, packb((shape,
    dtype.name, tobytes)))."""
    k = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.array([1.5, -2.0, 0.25], dtype=np.float32)
    ext = lambda a: msgpack.ExtType(1, msgpack.packb((a.shape, a.dtype.name, a.tobytes()), use_bin_type=True))
    blob = msgpack.packb({"step": 3, "params": {"Dense_0": {"kernel": ext(k), "bias": ext(b)}}}, use_bin_type=True)
    tree = msgpack_restore(blob)
    assert tree["step"] == 3
    assert np.array_equal(tree["params"]["Dense_0"]["kernel"], k) and tree["params"]["Dense_0"]["kernel"].dtype == np.float32
    assert np.array_equal(tree["params"]["Dense_0"]["bias"], b)
    assert msgpack_restore(msgpack_serialize({"a": {"w": k}}))["a"]["w"].tolist() == k.tolist()


@pytest.mark.parametrize("which", ["mlp", "resnet"])
def test_trainstate_checkpoint_roundtrip(tmp_path, which):
    net = SimpleClassifier(16, 2, 2) if which == "mlp" else ResNet1M(4, input_shape=(8, 8, 3), widths=(4, 8, 12), blocks_per_stage=1)
    src = create_state(net, 11)
    save_checkpoint(src, tmp_path, "map_xor", 40)
    save_checkpoint(create_state(net, 12), tmp_path, "map_xor", 7)          # an older step must be ignored
    dst = load_checkpoint(tmp_path, "map_xor", target=create_state(net, 99))
    x = torch.rand((3,) + tuple(net.input_shape_raw))
    a = src.apply_fn({"params": src.params["params"]}, x)
    b = dst.apply_fn({"params": dst.params["params"]}, x)
    assert torch.allclose(a, b, atol=1e-6)
    from lip_amd.utils import flatten_nn_params
    assert torch.equal(flatten_nn_params(src.params)[0], flatten_nn_params(dst.params)[0])
    # the scale-experiments convention: checkpoint 'params' without the top-level 'params' key
    raw = {"step": 1, "params": src.params["params"], "batch_stats": src.batch_stats}
    with open(tmp_path / "bare_1", "wb") as f:
        f.write(msgpack_serialize(raw))
    dst2 = load_checkpoint(tmp_path, "bare", target=create_state(net, 5))
    assert torch.equal(flatten_nn_params(src.params)[0], flatten_nn_params(dst2.params)[0])
    with pytest.raises(FileNotFoundError):
        load_checkpoint(tmp_path, "nope", target=src)
    with pytest.raises(ValueError):
        load_checkpoint(tmp_path, "map_xor", target=create_state(SimpleClassifier(8, 2, 2), 1))


def test_chunked_array_wire_format(monkeypatch):
    """Arrays above the chunk limit travel as {'__msgpack_chunked_array__', 'shape': {'0': d0, ...}, 'chunks': {...}}
    (flax writes the shape tuple through its tuple-to-dict helper); a small limit exercises the path, and a
    hand-built message with the dict-form shape — and one with the list form — both read back."""
    import msgpack
    import numpy as np
    from lip_amd import checkpoint as ck
    monkeypatch.setattr(ck, "_MAX_CHUNK", 64)
    a = np.arange(60, dtype=np.float32).reshape(3, 4, 5)
    enc = ck.msgpack_serialize({"w": a})
    raw = msgpack.unpackb(enc, raw=False, strict_map_key=False, ext_hook=lambda c, d: msgpack.ExtType(c, d))
    assert raw["w"]["__msgpack_chunked_array__"] is True and raw["w"]["shape"] == {"0": 3, "1": 4, "2": 5}
    assert len(raw["w"]["chunks"]) == 4
    back = ck.msgpack_restore(enc)
    assert back["w"].shape == (3, 4, 5) and np.array_equal(back["w"], a)
    # hand-built: two chunks, dict-form and list-form shapes
    def nd(x):
        return msgpack.ExtType(1, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes()), use_bin_type=True))
    flat = a.reshape(-1)
    for shape in ({"0": 3, "1": 4, "2": 5}, [3, 4, 5]):
        msg = msgpack.packb({"w": {"__msgpack_chunked_array__": True, "shape": shape,
                                   "chunks": {"0": nd(flat[:32]), "1": nd(flat[32:])}}}, use_bin_type=True)
        assert np.array_equal(ck.msgpack_restore(msg)["w"], a)


def test_same_numel_wrong_layout_is_refused():
    import numpy as np
    from lip_amd import checkpoint as ck
    with pytest.raises(ValueError):
        ck._check_like({"k": np.zeros((3, 3, 4, 8), np.float32)}, {"k": np.zeros((3, 3, 8, 4), np.float32)}, "params")
    ck._check_like({"s": np.zeros((), np.float32)}, {"s": np.zeros((1,), np.float32)}, "params")
