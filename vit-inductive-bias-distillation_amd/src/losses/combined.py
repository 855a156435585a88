"""Drop-in for the reference's ``src/losses/combined.py`` (same module path and names):
``from src.losses.combined import BASDLoss`` as in reference ``src/training/trainer.py:12``."""
from basd_amd.losses import BASDLoss, _align_token_count  # noqa: F401
from src.losses.layer_selector import GrassmannianLayerSelector  # noqa: F401
from src.losses.relational import geometric_relational_loss  # noqa: F401
