"""Scale workloads — the nets of ``src/scalemodels.py`` as layer programs.

``LargeClassifier`` (``:52-67``): flatten -> numl x [Dense(numh[j]) -> tanh] -> Dense(numc).
``ResNet1M`` (``:115-157``) with ``BasicBlock`` (``:70-112``): 3x3 conv(32)+BN+ReLU, then
3 blocks @32, 3 @64 (first stride 2), 3 @128 (first stride 2), global mean pool, Dense.
Convs are bias-free 'SAME'; BatchNorm runs in eval mode (running statistics), as the
GGN code calls ``apply_fn(..., train=False, mutable=False)`` (``src/ggn.py:52,123``).
Flax auto-names: ``Conv_0, BatchNorm_0, BasicBlock_0..8/{Conv_0,BatchNorm_0,Conv_1,
BatchNorm_1[,Conv_2,BatchNorm_2]}, Dense_0``.
"""
from __future__ import annotations

from typing import Sequence

from .netspec import NetSpec
from .utils import TrainState  # noqa: F401  (reference exports TrainState here, :161)


def LargeClassifier(input_shape: Sequence[int], numh: Sequence[int], numl: int, numc: int) -> NetSpec:
    net = NetSpec(tuple(input_shape))
    t = net.flatten(0)
    for j in range(numl):
        t = net.dense(t, f"Dense_{j}", int(numh[j]), act="tanh")
    net.dense(t, f"Dense_{numl}", numc)
    net.model_type = "classifier"
    return net


def LeNet5(num_classes: int = 10, input_shape: Sequence[int] = (28, 28, 1)) -> NetSpec:
    """``src/scalemodels.py:11-49``: zero-pad 2 + 5x5 VALID conv(6) (== 5x5 conv with padding 2) -> ReLU -> 2x2 avg
    pool -> 5x5 VALID conv(16) -> ReLU -> 2x2 avg pool -> flatten(400) -> Dense(120) -> ReLU -> Dense(84) -> ReLU ->
    Dense(10); convolutions carry a bias (Flax default).  61 706 parameters."""
    net = NetSpec(tuple(input_shape))
    x = net.conv(0, "Conv_0", 6, 5, 1, padding=2, act="relu", use_bias=True)
    x = net.avgpool(x, 2, 2)
    x = net.conv(x, "Conv_1", 16, 5, 1, padding="VALID", act="relu", use_bias=True)
    x = net.avgpool(x, 2, 2)
    x = net.dense(x, "Dense_0", 120, act="relu")
    x = net.dense(x, "Dense_1", 84, act="relu")
    net.dense(x, "Dense_2", num_classes)
    net.model_type = "classifier"
    return net


def _basic_block(net: NetSpec, x: int, scope: str, channels: int, stride: int) -> int:
    sc = (scope,)
    h = net.conv(x, "Conv_0", channels, 3, stride, bn="BatchNorm_0", act="relu", scope=sc)
    need_proj = (stride != 1) or (net.tensors[x][2] != channels)
    if need_proj:
        # 1x1 projection of the block input (Flax names it Conv_2/BatchNorm_2, :100-108)
        r = net.conv(x, "Conv_2", channels, 1, stride, bn="BatchNorm_2", act="none", scope=sc)
        return net.conv(h, "Conv_1", channels, 3, 1, bn="BatchNorm_1", res=r, act="relu", scope=sc)
    return net.conv(h, "Conv_1", channels, 3, 1, bn="BatchNorm_1", res=x, act="relu", scope=sc)


def ResNet1M(num_classes: int, input_shape: Sequence[int] = (32, 32, 3),
             widths: Sequence[int] = (32, 64, 128), blocks_per_stage: int = 3) -> NetSpec:
    ishape = tuple(input_shape)
    # grayscale input is replicated to 3 channels (``src/scalemodels.py:127-128``)
    gray = (len(ishape) == 2) or (len(ishape) == 3 and ishape[2] == 1)
    net = NetSpec(ishape, tile_channels=3 if gray else 1)
    x = net.conv(0, "Conv_0", widths[0], 3, 1, bn="BatchNorm_0", act="relu")
    b = 0
    for si, ch in enumerate(widths):
        for j in range(blocks_per_stage):
            stride = 2 if (si > 0 and j == 0) else 1
            x = _basic_block(net, x, f"BasicBlock_{b}", ch, stride)
            b += 1
    x = net.meanpool(x)
    net.dense(x, "Dense_0", num_classes)
    net.model_type = "classifier"
    return net


def _bottleneck(net: NetSpec, x: int, scope: str, planes: int, stride: int, expansion: int = 4) -> int:
    """torchvision-style bottleneck: 1x1 -> 3x3 (stride) -> 1x1 (x expansion), BN after each, projection shortcut
    when the shape changes; symmetric padding 1 on the 3x3 (torch convention)."""
    sc = (scope,)
    out_ch = planes * expansion
    h = net.conv(x, "Conv_0", planes, 1, 1, padding=0, bn="BatchNorm_0", act="relu", scope=sc)
    h = net.conv(h, "Conv_1", planes, 3, stride, padding=1, bn="BatchNorm_1", act="relu", scope=sc)
    if stride != 1 or net.tensors[x][2] != out_ch:
        r = net.conv(x, "Conv_3", out_ch, 1, stride, padding=0, bn="BatchNorm_3", act="none", scope=sc)
    else:
        r = x
    return net.conv(h, "Conv_2", out_ch, 1, 1, padding=0, bn="BatchNorm_2", res=r, act="relu", scope=sc)


def ResNet50(num_classes: int = 1000, input_shape: Sequence[int] = (224, 224, 3), stem: int = 64,
             widths: Sequence[int] = (64, 128, 256, 512), blocks: Sequence[int] = (3, 4, 6, 3)) -> NetSpec:
    """BASELINE.json configs[4]: the torchvision ResNet-50 architecture defined locally (~25.6 M parameters at the
    defaults; random init, no download): 7x7/2 stem + BN + ReLU, 3x3/2 max pool, bottleneck stages, global mean
    pool, Dense.  Not in the reference; smaller ``widths`` / ``blocks`` give parity-test sizes."""
    net = NetSpec(tuple(input_shape))
    x = net.conv(0, "Conv_0", stem, 7, 2, padding=3, bn="BatchNorm_0", act="relu")
    x = net.maxpool(x, 3, 2, padding=1)
    b = 0
    for si, (planes, nb) in enumerate(zip(widths, blocks)):
        for j in range(nb):
            x = _bottleneck(net, x, f"Bottleneck_{b}", planes, 2 if (si > 0 and j == 0) else 1)
            b += 1
    x = net.meanpool(x)
    net.dense(x, "Dense_0", num_classes)
    net.model_type = "classifier"
    return net


def get_model(model_cfg) -> NetSpec:
    """Reference ``src/scalemodels.py:166-186``; ``input_shape`` may be given for ResNet1 (the reference infers it
    from the data: grayscale inputs are tiled to 3 channels)."""
    name = model_cfg["name"]
    if name == "LeNet5":
        return LeNet5()
    if name == "large_classifier":
        return LargeClassifier(tuple(model_cfg["input_shape"]), model_cfg["num_h"],
                               model_cfg["num_l"], model_cfg.get("num_c"))
    if name == "classifier":
        from .toymodels import SimpleClassifier
        return SimpleClassifier(model_cfg["num_h"], model_cfg["num_l"], model_cfg.get("num_c"))
    if name == "ResNet1":
        return ResNet1M(model_cfg.get("num_c"), input_shape=tuple(model_cfg.get("input_shape", (32, 32, 3))))
    raise ValueError(f"Unknown model name: {name}")
