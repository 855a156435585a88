"""Drop-in for the reference's ``src/losses/relational.py``."""
from basd_amd.losses import geometric_relational_loss  # noqa: F401
