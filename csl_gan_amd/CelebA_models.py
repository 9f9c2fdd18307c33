"""CelebA DCResNet sizes (reference CelebA_models.py:10-24) plus the 128x128 extension that
BASELINE.json config 5 asks for (not in the reference: im_size choices are [64, 48], options.py:124)."""
from .DCResNet_models import DCResNetDiscriminator, DCResNetGenerator


def _g(channels, first):
    class _G(DCResNetGenerator):
        def __init__(self, z_dim=128, channels=channels, first_filter_size=first, **kwargs):
            super().__init__(z_dim=z_dim, channels=list(channels), first_filter_size=first_filter_size, out_ch=3, **kwargs)
    return _G


def _d(channels, last):
    class _D(DCResNetDiscriminator):
        def __init__(self, channels=channels, last_filter_size=last, **kwargs):
            super().__init__(channels=list(channels), last_filter_size=last_filter_size, **kwargs)
    return _D


CelebA_DCRN_G64 = _g((512, 512, 256, 128, 64), 4)
CelebA_DCRN_D64 = _d((3, 64, 128, 256, 512), 4)
CelebA_DCRN_G48 = _g((512, 512, 256, 128), 6)
CelebA_DCRN_D48 = _d((3, 128, 256, 512), 6)
# extension: same discriminator depth, 8x8 final map; generator gets one more up block
CelebA_DCRN_G128 = _g((512, 512, 256, 128, 64, 64), 4)
CelebA_DCRN_D128 = _d((3, 64, 128, 256, 512), 8)
for _n, _c in list(globals().items()):
    if _n.startswith("CelebA_DCRN_"):
        _c.__name__ = _c.__qualname__ = _n
