from .activations import *
from .misc import *
from .resamplers import *
from .targets import *
from .io import *
