"""tensorflow.keras.backend stand-in (image_data_format / floatx / ndim)."""
import numpy as np

_FMT = "channels_last"


def image_data_format():
    return _FMT


def set_image_data_format(fmt):
    global _FMT
    _FMT = fmt


def floatx():
    return "float32"


def ndim(x):
    return np.ndim(x)
