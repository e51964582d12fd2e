"""
neural-imaging_amd - MI355X-native (gfx950, HIP) implementation of the pkorus/neural-imaging channel hot path
    RAW -> UNet ISP -> photo manipulations -> differentiable JPEG / TwitterDCN -> FAN classifier, forward + backward,
behind the reference's own operator surface (TFModel.process / training_step, ManipulationClassification).

The directory name contains a hyphen, so import it with importlib:

    import importlib; nimg = importlib.import_module('neural-imaging_amd')
    # or, after the first import, `import neural_imaging_amd` (alias registered below)
    from neural_imaging_amd.models import pipelines, jpeg, forensics
    from neural_imaging_amd.workflows import manipulation_classification

Arithmetic happens only in libnimg.so (hand-written HIP kernels, C ABI in include/nimg.h).  torch supplies device
memory, streams and torch.distributed (RCCL); there is no CPU fallback.
"""
import sys as _sys

_sys.modules.setdefault('neural_imaging_amd', _sys.modules[__name__])

__version__ = '0.1.0'
