"""MI355X-native Spherical-DYffusion sampling path (forecaster/interpolator SFNO loop on the 180x360 grid).

The directory name carries a hyphen (it is the name the build contract asks for), so import it through the
`sdy_amd` alias module at the repository root: `import sdy_amd`.

Everything numerical runs in `libsdy_amd.so` (hand-written HIP for gfx950 behind the C ABI of
`include/sdy_amd.h`); importing this package without the built library raises ImportError -- there is no
CPU or PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when libsdy_amd.so is missing)
from ._lib import LIB_PATH, SdyError, lib  # noqa: F401
from .dyffusion import DYffusion  # noqa: F401
from .experiment import InterpolationExperiment, MultiHorizonForecastingDYffusion  # noqa: F401
from .sfno import SphericalFourierNeuralOperatorNet  # noqa: F401
from .sht import InverseRealSHT, RealSHT  # noqa: F401
from .stepper import MultiStepStepper, Prescriber, SteppedData  # noqa: F401
from .loop import NullAggregator, NullDataWriter, WindowStitcher, run_inference  # noqa: F401
from . import checkpoint, ensemble, interface, metrics, normalizer, ops, synthetic  # noqa: F401

__version__ = "0.1.0"
