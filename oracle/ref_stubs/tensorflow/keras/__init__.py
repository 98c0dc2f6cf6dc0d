"""Keras stand-in: just enough of Layer / Sequential / backend for Kapre's hot-path modules."""
import types

import numpy as np

from . import backend  # noqa: F401


class _Layer:
    _count = {}

    def __init__(self, name=None, input_shape=None, dtype=None, trainable=True, **kwargs):
        if kwargs:
            raise TypeError("unexpected keyword arguments %r" % sorted(kwargs))
        cls = type(self).__name__.lower()
        if name is None:
            n = _Layer._count.get(cls, 0)
            _Layer._count[cls] = n + 1
            name = cls if n == 0 else "%s_%d" % (cls, n)
        self.name = name
        self.trainable = trainable
        self._dtype = dtype or "float32"

    def __call__(self, x, training=None):
        return self.call(x)

    def call(self, x):
        return x

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self._dtype}


class Sequential(_Layer):
    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self.layers = list(layers or [])

    def add(self, layer):
        self.layers.append(layer)

    def call(self, x):
        for layer in self.layers:
            x = layer(x)
        return x

    def predict(self, x):
        return np.asarray(self(x))


class Model(_Layer):
    pass


layers = types.ModuleType("tensorflow.keras.layers")
layers.Layer = _Layer

utils = types.ModuleType("tensorflow.keras.utils")


def _register_keras_serializable(package="Custom", name=None):
    def deco(cls):
        cls._keras_registered_name = "%s>%s" % (package, name or cls.__name__)
        return cls
    return deco


utils.register_keras_serializable = _register_keras_serializable

import sys as _sys  # noqa: E402

_sys.modules[__name__ + ".layers"] = layers
_sys.modules[__name__ + ".utils"] = utils
