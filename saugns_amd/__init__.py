"""saugns_amd -- MI355X-native backend for the saugns audio generator hot path.

The product is ``libsaugns_amd.so`` (HIP kernels + C++ host control plane behind
the reference's C API); this package builds it in-tree and binds it with ctypes.
"""
from .api import (Batch, Generator, Program, SNDFILE_AU, SNDFILE_RAW, SNDFILE_WAV,  # noqa: F401
                  get_piluts, last_error, lib, render_file, set_piluts)
from .build import build  # noqa: F401
