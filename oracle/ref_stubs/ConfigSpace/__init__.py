"""Empty stub (kge/util/configspace_converter.py:1-2)."""
from . import hyperparameters  # noqa: F401
