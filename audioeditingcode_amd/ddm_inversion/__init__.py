"""Mirror of the reference's code/ddm_inversion package (same function names and signatures)."""
from .inversion_utils import inversion_forward_process, inversion_reverse_process  # noqa: F401
from .ddim_inversion import ddim_inversion, text2image_ldm_stable  # noqa: F401
