"""simplestereo_amd -- the SimpleStereo passive-matching hot path on AMD MI355X.

Drop-in for the slice of ``simplestereo`` that feeds and runs
``ss.passive.StereoASW`` / ``ss.passive.StereoGSW``:

    import simplestereo_amd as ss
    rig = ss.RectifiedStereoRig.fromFile("rigRect.json")
    imgL, imgR = rig.rectifyImages(imgL, imgR)
    disparity = ss.passive.StereoASW(winSize=35, maxDisparity=64).compute(imgL, imgR)

The matchers run as hand-written HIP kernels (gfx950) behind the C ABI of
``libssamd.so`` (see ``include/ssamd.h``); nothing here falls back to the CPU.
"""
from . import passive
from ._rigs import StereoRig, RectifiedStereoRig
from . import strips
from . import points

__version__ = "0.6.0"
__all__ = ["passive", "StereoRig", "RectifiedStereoRig", "strips", "points"]
