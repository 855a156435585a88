"""MI355X-native BASD loss path (Grassmannian layer selector + attention-weighted
Procrustes loss) behind the reference's ``src.losses`` API.

Sub-modules:
  ``_lib``     ctypes binding of the C-ABI in ``include/basd_hip.h`` (raises if the
               HIP library is missing -- there is no CPU fallback)
  ``ops``      torch.autograd Functions / host orchestration over the C-ABI
  ``losses``   ``BASDLoss`` / ``GrassmannianLayerSelector`` / free functions
  ``synth``    seeded synthetic feature stacks (benchmark + tests)
"""
__version__ = "0.1.0"
