"""rdfc_gan_amd -- MI355X-native train-step hot path of RDFC-GAN.

Python here is host plumbing (module surface, autograd wiring, device memory);
all arithmetic runs in librdfc_hip.so (include/rdfc_hip.h)."""
from .functional import compute_dtype, set_compute_dtype  # noqa: F401
