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
