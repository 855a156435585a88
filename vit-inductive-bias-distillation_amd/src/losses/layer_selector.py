"""Drop-in for the reference's ``src/losses/layer_selector.py``:
``from src.losses.layer_selector import marchenko_pastur_rank`` as in reference ``src/models/teacher.py:6``."""
from basd_amd.losses import (  # noqa: F401
    GrassmannianLayerSelector,
    _grassmann_subspace,
    marchenko_pastur_rank,
)
