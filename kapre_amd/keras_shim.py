"""The slice of the Keras Layer / Sequential protocol that Kapre's layers and tests rely on.

The reference's "operator API" is the Keras layer protocol:
``Layer.__init__(**kwargs)`` swallowing ``name`` / ``input_shape`` / ``dtype``,
``__call__(x, training=None) -> call(x)``, ``get_config()`` / ``from_config()``,
``@register_keras_serializable(package='Kapre')`` and composition in ``keras.Sequential``
(``add``, ``.layers``, ``.name``, ``__call__``, ``predict(np) -> np``, ``get_config``).
TensorFlow/Keras are not available on the target image, so this module re-provides exactly that
protocol in plain Python; tensors are torch tensors (device memory container only).

``Sequential`` additionally performs peephole fusion of Kapre layer chains into single HIP
launches (see kapre_amd.time_frequency.fuse_and_run); ``.layers`` still exposes the individual
layers, which remain individually callable, as the reference documents (composed.py:1-13).
"""
from __future__ import annotations

import collections
from typing import Any, Dict, List, Optional

import numpy as np

_NAME_COUNTS: Dict[str, int] = collections.defaultdict(int)
_REGISTRY: Dict[str, type] = {}


def _to_snake_case(name: str) -> str:
    out = []
    for i, ch in enumerate(name):
        if ch.isupper() and i > 0 and (not name[i - 1].isupper() or
                                       (i + 1 < len(name) and name[i + 1].islower())):
            out.append('_')
        out.append(ch.lower())
    return ''.join(out)


def _unique_name(prefix: str) -> str:
    n = _NAME_COUNTS[prefix]
    _NAME_COUNTS[prefix] += 1
    return prefix if n == 0 else '%s_%d' % (prefix, n)


def register_keras_serializable(package: str = 'Custom', name: Optional[str] = None):
    """Same decorator signature as tf.keras.utils.register_keras_serializable."""

    def decorator(cls):
        registered = '%s>%s' % (package, name or cls.__name__)
        cls._keras_registered_name = registered
        _REGISTRY[registered] = cls
        _REGISTRY[cls.__name__] = cls
        return cls

    return decorator


def get_registered_object(name: str):
    return _REGISTRY.get(name)


class Layer:
    """Minimal keras.layers.Layer: naming, config round trip, __call__ -> call."""

    def __init__(self, name: Optional[str] = None, input_shape=None, dtype=None,
                 trainable: bool = True, batch_input_shape=None, **kwargs):
        if kwargs:
            raise TypeError('Unrecognized keyword arguments passed to %s: %s'
                            % (type(self).__name__, sorted(kwargs)))
        self.name = name if name is not None else _unique_name(_to_snake_case(type(self).__name__))
        self.trainable = trainable
        self._dtype = dtype or 'float32'
        self._input_shape_arg = tuple(input_shape) if input_shape is not None else None
        if batch_input_shape is not None:
            self._input_shape_arg = tuple(batch_input_shape[1:])

    @property
    def dtype(self):
        return self._dtype

    @property
    def _f64(self) -> bool:
        """Keras casts a layer's inputs to the layer dtype (autocast): a layer built with dtype='float64'
        computes in float64 / complex128, every other layer in float32 / complex64."""
        return str(self._dtype) in ('float64', 'double', "<class 'numpy.float64'>", 'torch.float64')

    @property
    def weights(self):
        return []

    def count_params(self) -> int:
        return 0          # all Kapre hot-path layers are parameter free (docs/quickstart.rst:76-78)

    def build(self, input_shape):
        pass

    def call(self, x):
        return x

    def __call__(self, x, training=None, **kwargs):
        return self.call(x)

    def get_config(self) -> Dict[str, Any]:
        return {'name': self.name, 'trainable': self.trainable, 'dtype': self._dtype}

    @classmethod
    def from_config(cls, config: Dict[str, Any]):
        return cls(**config)


class InputLayer(Layer):
    """keras.Input(shape=...) placeholder: records the per-sample shape, passes data through."""

    def __init__(self, shape=None, dtype=None, name=None, batch_size=None, **kwargs):
        super().__init__(name=name, dtype=dtype)
        self.shape = tuple(shape) if shape is not None else None

    def get_config(self):
        config = super().get_config()
        config.update({'shape': self.shape})
        return config


def Input(shape=None, dtype=None, name=None, batch_size=None, **kwargs):
    return InputLayer(shape=shape, dtype=dtype, name=name, batch_size=batch_size)


class Sequential(Layer):
    """keras.Sequential over Kapre layers, with kernel fusion at call time."""

    def __init__(self, layers: Optional[List[Layer]] = None, name: Optional[str] = None,
                 trainable: bool = True):
        super().__init__(name=name if name is not None else _unique_name('sequential'),
                         trainable=trainable)
        self._layers: List[Layer] = []
        for layer in layers or []:
            self.add(layer)

    @property
    def layers(self) -> List[Layer]:
        return [l for l in self._layers if not isinstance(l, InputLayer)]

    def add(self, layer) -> None:
        if not isinstance(layer, Layer):
            raise TypeError('The added layer must be an instance of class Layer. Received: %r'
                            % (layer,))
        self._layers.append(layer)

    def pop(self):
        return self._layers.pop()

    def _flat_layers(self) -> List[Layer]:
        flat: List[Layer] = []
        for layer in self._layers:
            if isinstance(layer, InputLayer):
                continue
            if isinstance(layer, Sequential):
                flat.extend(layer._flat_layers())
            else:
                flat.append(layer)
        return flat

    def call(self, x):
        from .time_frequency import fuse_and_run
        return fuse_and_run(self._flat_layers(), x)

    def predict(self, x, batch_size=None, verbose=0, **kwargs) -> np.ndarray:
        """numpy in -> numpy out (the way the reference's tests drive models)."""
        import torch

        y = self(x)
        if isinstance(y, torch.Tensor):
            return y.detach().cpu().numpy()
        return np.asarray(y)

    def get_config(self):
        return {
            'name': self.name,
            'layers': [{'class_name': type(l).__name__,
                        'registered_name': getattr(type(l), '_keras_registered_name', None),
                        'config': l.get_config()} for l in self._layers],
        }

    @classmethod
    def from_config(cls, config):
        layers = []
        for spec in config['layers']:
            klass = (get_registered_object(spec.get('registered_name') or '')
                     or get_registered_object(spec['class_name']))
            if klass is None and spec['class_name'] == 'InputLayer':
                klass = InputLayer
            if klass is None and spec['class_name'] == 'Sequential':
                klass = Sequential
            if klass is None:
                raise ValueError('Unknown layer class %r' % spec['class_name'])
            layers.append(klass.from_config(spec['config']))
        return cls(layers, name=config.get('name'))

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        for l in self.layers:
            print_fn('  %-32s %s' % (l.name, type(l).__name__))
        print_fn('Total params: 0')


Model = Sequential
