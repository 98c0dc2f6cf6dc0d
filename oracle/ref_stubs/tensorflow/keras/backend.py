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


def permute_dimensions(x, pattern):
    return np.transpose(x, pattern)


def arange(start, stop=None, step=1, dtype=None):
    return np.arange(start, stop, step, dtype=np.float64 if dtype in (None, "float32", "float64") else dtype)


def reshape(x, shape):
    return np.reshape(x, shape)


def conv2d(x, kernel, data_format="channels_last"):
    """'valid' cross-correlation (what tf.nn.conv2d computes), NHWC x HWIO."""
    assert data_format == "channels_last"
    x = np.asarray(x, dtype=np.float64)
    kh, kw, ci, co = kernel.shape
    assert x.shape[3] == ci, "conv2d: input has %d channels, kernel expects %d" % (x.shape[3], ci)
    b, h, w, _ = x.shape
    out = np.zeros((b, h - kh + 1, w - kw + 1, co))
    for i in range(kh):
        for j in range(kw):
            out += np.einsum("bhwc,co->bhwo", x[:, i:i + h - kh + 1, j:j + w - kw + 1, :], kernel[i, j])
    return out
