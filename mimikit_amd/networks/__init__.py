from .arm import *
from .mlp import *
from .wavenet_v2 import *
from .sample_rnn_v2 import *
from .s2s_lstm_v2 import *
