"""encodermap_amd -- MI355X-native training engine for EncoderMap.

Drop-in for the training hot path of AG-Peter/encodermap (``em.Parameters`` / ``em.EncoderMap(...)`` /
``train()`` / ``encode()`` / ``generate()``): Python host code, hand-written HIP kernels behind a C ABI
(``libemx.so``, ``include/emx.h``), one RCCL all-reduce per step for data parallelism.  ``import encodermap_amd as em``.

Importing the package needs neither a GPU nor the built library (parameters, data generators and checkpoint
helpers are plain Python); constructing a model does, and fails loudly without them -- there is no CPU path.
"""
from .parameters import ADCParameters, Parameters
from .misc import create_n_cube, random_on_cube_edges
from .autoencoder import Autoencoder, DihedralEncoderMap, EncoderMap
from .adc import AngleDihedralCartesianEncoderMap
from .features import featurize_backbone
from . import backmapping, callbacks
from .backmapping import AllAtomTopology, TorsionTable, allatom_backmap
from .callbacks import ADCClashMetric, ADCRMSDMetric, CheckpointSaver, EarlyStop, EncoderMapBaseCallback, History, ProgressBar

__all__ = [
    "Parameters", "ADCParameters", "Autoencoder", "EncoderMap", "DihedralEncoderMap", "AngleDihedralCartesianEncoderMap",
    "create_n_cube", "random_on_cube_edges", "featurize_backbone", "EncoderMapBaseCallback", "ProgressBar", "CheckpointSaver",
    "EarlyStop", "History", "ADCClashMetric", "ADCRMSDMetric", "callbacks", "backmapping", "AllAtomTopology", "TorsionTable", "allatom_backmap",
]
__version__ = "0.1.0"
