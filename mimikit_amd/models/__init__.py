from .ensemble_generator import *
