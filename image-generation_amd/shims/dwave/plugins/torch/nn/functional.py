"""dwave.plugins.torch.nn.functional -> image_generation_amd.plugin (/root/reference/src/model_wrapper.py:29)."""
from image_generation_amd.plugin import maximum_mean_discrepancy_loss  # noqa: F401
