"""dwave.plugins.torch.nn.modules.kernels -> image_generation_amd.plugin (/root/reference/src/model_wrapper.py:30)."""
from image_generation_amd.plugin import GaussianKernel  # noqa: F401
