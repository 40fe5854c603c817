"""mimikit_amd -- the auto-regressive generate path of ktonal/mimikit, built natively for the
AMD Instinct MI355X (gfx950): WaveNet / SampleRNN / Seq2Seq ``generate_step`` loops driven by
``GenerateLoopV2`` plus the mu-law and framed-STFT feature functionals, behind mimikit's own
``ARM`` network protocol, ``IOSpec`` config API and ``state_dict`` layout.

All compute of that path lives in ``libmmk_hip.so`` (hand-written HIP kernels + C ABI declared
in ``include/mmk.h``); there is no CPU or eager fallback.  See DESIGN.md.
"""
__version__ = "0.1.0"

from .config import *
from .utils import *
from .features import *
from .modules import *
from .io_spec import *
from .networks import *
from .loops import *
from .models import *
from .checkpoint import *
