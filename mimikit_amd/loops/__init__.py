from .generate import *
from .samplers import *
from .callbacks import *
from .generate_chunks import *
