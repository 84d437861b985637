"""sbtv — host-side mirror of the reference's operator API over libsbtv.so (MI355X / gfx950).

Names follow the reference (SALSA_v2, chambolle_prox_TV_stop, TVnorm,
A_wrapper, Gaussian_psf, ...).  Everything computes on the GPU through the
C-ABI in include/sbtv.h; there is no CPU implementation in this package.
"""
from ._lib import (Context, Group, SbtvError, default_context, load_library, to_device, to_host, LIB_PATH,
                   SBTV_DEVICE_PTRS, SBTV_HOST_PTRS, switches)
from .tv import chambolle_prox_TV_stop, TVnorm
from .operators import (BlurOperator, A_wrapper, Gaussian_psf, psf_gaussian, psf_moffat, psf_laplace, psf_family,
                        rfft2_packed, unpack_half_spectrum, conv2c, diffh, diffv)
from .salsa import SALSA_v2
from .admm import csalsa, CSALSA_v2, CoRAL, CoRAL_v2
from .diagnostics import ssim, save_results, load_results, plot_traces, save_image
from .metrics import PSNR, MSE
from .fista import my_fista, my_deblur_fista, Psi_TV
from .sapg import (SAPG_algorithm_Guassian, SAPG_algorithm_moffat, SAPG_algorithm_laplace, max_eigenval,
                   demo_setup, myula)

__all__ = [
    "my_fista", "my_deblur_fista", "Psi_TV", "SAPG_algorithm_Guassian", "SAPG_algorithm_moffat",
    "SAPG_algorithm_laplace", "max_eigenval", "demo_setup", "myula",
    "Context", "Group", "SbtvError", "switches", "default_context", "load_library", "to_device", "to_host", "LIB_PATH",
    "chambolle_prox_TV_stop", "TVnorm", "BlurOperator", "A_wrapper", "Gaussian_psf", "psf_gaussian",
    "psf_moffat", "psf_laplace", "psf_family", "rfft2_packed", "unpack_half_spectrum", "SALSA_v2", "PSNR", "MSE",
    "csalsa", "CSALSA_v2", "CoRAL", "CoRAL_v2", "ssim", "save_results", "load_results", "plot_traces", "save_image", "conv2c", "diffh", "diffv",
]
