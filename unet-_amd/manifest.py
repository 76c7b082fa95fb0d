"""Key/shape manifests of the two reference modules' ``state_dict()`` — what ``load_state_dict(strict=True)``
checks against (packing.py) and what the synthetic weight generator fills (synthetic.py).

NestedUNet: src/models/unetpp.py:13-26 (ConvBlock), :66-91 (members); SimpleUNet: src/models/simple_unet.py:30-92.
Pinned by tests/golden/state_dict_manifest.json, which oracle/make_golden.py dumps from the reference classes."""
from __future__ import annotations

NB_FILTER = (32, 64, 128, 256, 512)  # src/models/unetpp.py:49

# (name, in_channels, out_channels) in the order the forward uses them (unetpp.py:104-116)
def conv_blocks(in_channels: int = 3):
    f = NB_FILTER
    return [
        ("conv0_0", in_channels, f[0]),
        ("conv1_0", f[0], f[1]),
        ("conv2_0", f[1], f[2]),
        ("conv3_0", f[2], f[3]),
        ("conv4_0", f[3], f[4]),
        ("conv3_1", f[3] + f[4], f[3]),
        ("conv2_2", f[2] + f[3], f[2]),
        ("conv1_3", f[1] + f[2], f[1]),
        ("conv0_4", f[0] + f[1], f[0]),
    ]


def state_dict_manifest(num_classes: int, in_channels: int = 3, deep_supervision: bool = True):
    """Ordered (key, shape, dtype) list identical to the reference module's state_dict()."""
    ordered = []
    for name, ci, co in conv_blocks(in_channels):
        for j, cin in ((1, ci), (2, co)):       # module order: conv1, bn1, conv2, bn2
            ordered.append((f"{name}.conv{j}.weight", (co, cin, 3, 3), "float32"))
            ordered.append((f"{name}.conv{j}.bias", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.weight", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.bias", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.running_mean", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.running_var", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.num_batches_tracked", (), "int64"))
    ordered.append(("final.weight", (num_classes, NB_FILTER[0], 1, 1), "float32"))
    ordered.append(("final.bias", (num_classes,), "float32"))
    if deep_supervision:
        for nm, c in (("ds3_1", NB_FILTER[3]), ("ds2_2", NB_FILTER[2]), ("ds1_3", NB_FILTER[1])):
            ordered.append((f"{nm}.weight", (num_classes, c, 1, 1), "float32"))
            ordered.append((f"{nm}.bias", (num_classes,), "float32"))
    return ordered


SIMPLE_WIDTHS = (64, 128, 256, 512)  # src/models/simple_unet.py:32-57


def simple_unet_manifest(num_classes: int = 7, num_channels: int = 3):
    """Ordered (key, shape, dtype) list identical to SimpleUNet.state_dict() (src/models/simple_unet.py:30-92):
    enc{1..4}.{0,2}, up3, up2, up1 (ConvTranspose2d: weight [Cin, Cout, 2, 2]), dec{3,2,1}.{0,2}, final."""
    w = SIMPLE_WIDTHS
    out = []
    for l in range(4):
        ci = num_channels if l == 0 else w[l - 1]
        out += [(f"enc{l+1}.0.weight", (w[l], ci, 3, 3), "float32"), (f"enc{l+1}.0.bias", (w[l],), "float32"),
                (f"enc{l+1}.2.weight", (w[l], w[l], 3, 3), "float32"), (f"enc{l+1}.2.bias", (w[l],), "float32")]
    for l in (2, 1, 0):
        out += [(f"up{l+1}.weight", (w[l + 1], w[l], 2, 2), "float32"), (f"up{l+1}.bias", (w[l],), "float32")]
    for l in (2, 1, 0):
        out += [(f"dec{l+1}.0.weight", (w[l], 2 * w[l], 3, 3), "float32"), (f"dec{l+1}.0.bias", (w[l],), "float32"),
                (f"dec{l+1}.2.weight", (w[l], w[l], 3, 3), "float32"), (f"dec{l+1}.2.bias", (w[l],), "float32")]
    out += [("final.weight", (num_classes, w[0], 1, 1), "float32"), ("final.bias", (num_classes,), "float32")]
    return out
