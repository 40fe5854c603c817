from .generate import *
