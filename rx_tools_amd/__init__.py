"""rx_tools_amd -- Python-side binding of librxgpu.so (the MI355X rx_tools DSP path).

The product is the C-ABI shared library built from rx_tools_amd/csrc (see include/rxgpu.h);
this package is only the ctypes plumbing that tests and bench.py drive it with.  There is
no CPU implementation here: without librxgpu.so and a HIP device every call fails.
"""
from ._lib import lib, RxGpuError, check          # noqa: F401
from .fm import FmParams, FmCarry, FmStream, ChanParams, Channeliser        # noqa: F401
from .power import PowerParams, PowerPlan, PowerScan, plan_range, sine_table, window_coefs  # noqa: F401
from .sdr import sdr_convert, sdr_convert_host, wav_header, SDR_CONVERSIONS      # noqa: F401
from . import synth  # noqa: F401
