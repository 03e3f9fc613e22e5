"""dwave.plugins.torch.models -> image_generation_amd.plugin (/root/reference/src/model_wrapper.py:25-28)."""
from image_generation_amd.plugin import DiscreteVariationalAutoencoder, GraphRestrictedBoltzmannMachine  # noqa: F401
