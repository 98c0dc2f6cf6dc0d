"""kapre_amd -- Kapre's time-frequency hot path on AMD Instinct MI355X (gfx950).

Import surface mirrors /root/reference/kapre/__init__.py for the components in scope
(``import kapre_amd as kapre`` is the intended drop-in): the layers run hand-written HIP kernels
through a C ABI (include/kapre_hip.h); there is no CPU fallback.
"""
__version__ = '0.4.0+mi355x.1'
VERSION = __version__

from . import backend
from . import composed

from .keras_shim import Layer, Sequential, Model, Input, register_keras_serializable

from .time_frequency import (
    STFT,
    InverseSTFT,
    Magnitude,
    Phase,
    MagnitudeToDecibel,
    ApplyFilterbank,
    Delta,
)

from . import signal
from .signal import Frame, Energy, LogmelToMFCC

from .composed import (
    get_stft_magnitude_layer,
    get_melspectrogram_layer,
    get_log_frequency_spectrogram_layer,
    get_perfectly_reconstructing_stft_istft,
    get_stft_mag_phase,
)



def check_device(device=None):
    """Report what only a kernel can see: waits for ``device`` (default: the current one), then reads
    and clears the library's status word.  Raises ``RuntimeError`` naming the kernels when a bounded in-kernel wait ran out or a
    packed filterbank changed under a cached band plan since the last check -- the outputs of the affected launches are wrong --
    and returns ``None`` otherwise.  ``Sequential.predict`` does this after its copy; callers that hand torch tensors to the
    layers call it once per batch (or less often, if a late error is acceptable): the failed launch itself cannot report, the
    NEXT forward call would."""
    import torch
    from . import _ffi
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    _ffi.device_status(synchronize=False, raise_on_error=True)


__all__ = [
    '__version__',
    'VERSION',
    'check_device',
    'STFT',
    'InverseSTFT',
    'Magnitude',
    'Phase',
    'MagnitudeToDecibel',
    'ApplyFilterbank',
    'Delta',
    'Frame',
    'Energy',
    'LogmelToMFCC',
    'get_stft_magnitude_layer',
    'get_melspectrogram_layer',
    'get_log_frequency_spectrogram_layer',
    'get_perfectly_reconstructing_stft_istft',
    'get_stft_mag_phase',
    'Layer',
    'Sequential',
    'Model',
    'Input',
]
