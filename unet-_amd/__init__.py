"""MI355X-native UNet++ (NestedUNet) segmentation inference engine.

Drop-in behind the reference's ``src/models/unetpp.py::NestedUNet.forward`` and the model-call lines of
``infer_two_stage_burr.py`` (292-304).  Python host code -> ctypes -> C ABI (include/unetpp.h) -> HIP
kernels for gfx950.  PyTorch tensors are I/O buffers only.
"""
__version__ = "0.1.0"
