"""MI355X-native Oniris denoiser step.  Importing the package loads liboniris_hip.so and fails
loudly when it is missing -- there is no CPU or PyTorch fallback in the product path."""
from . import _lib  # noqa: F401  (raises ImportError if the HIP library is not built)

__all__ = ["_lib"]
