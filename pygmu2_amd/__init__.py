"""
pygmu2_amd -- MI355X-native render path for pygmu2-style audio graphs.

Same public names as the reference package for the accelerated hot path:
ProcessingElement / SourcePE / Snippet / Extent / Renderer / NullRenderer and the PEs
SinePE, BlitSawPE, SuperSawPE, BiquadPE, LadderPE, CombPE, MixPE, GainPE, ConvolvePE,
AdsrGatedPE, AdsrTriggeredPE, PeriodicGate, PeriodicTrigger, ConstantPE, ArrayPE,
DiracPE, IdentityPE, CachePE, CropPE, SVFilterPE, EnvelopePE, TransformPE, DelayPE, PiecewisePE,
TriggerRestartPE, ReverbPE, WavWriterPE, WavReaderPE (+ render_to_file).  Snippet payloads live in HBM; all DSP runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/pygmu_hip.h.
"""

from .config import (ErrorMode, get_error_mode, get_sample_rate, handle_error, set_error_mode,
                     set_sample_rate)
from .extent import ExtendMode, Extent
from .snippet import Snippet
from .processing_element import ProcessingElement
from .source_pe import SourcePE
from .renderer import PEProfile, ProfileReport, Renderer
from .null_renderer import NullRenderer
from .gate_signal import GateSignal
from .trigger_signal import TriggerSignal
from .constant_pe import ConstantPE
from .identity_pe import IdentityPE
from .dirac_pe import DiracPE
from .array_pe import ArrayPE
from .cache_pe import CachePE
from .crop_pe import CropPE
from .sine_pe import SinePE
from .gain_pe import GainPE
from .mix_pe import MixPE
from .biquad_pe import BiquadMode, BiquadPE
from .blit_saw_pe import BlitSawPE
from .super_saw_pe import SuperSawPE
from .ladder_pe import LadderMode, LadderPE
from .comb_pe import CombPE
from .periodic_gate import PeriodicGate
from .periodic_trigger import PeriodicTrigger
from .adsr_pe import AdsrGatedPE, AdsrTriggeredPE
from .convolve_pe import ConvolvePE
from .svfilter_pe import SVFilterPE
from .envelope_pe import DetectionMode, EnvelopePE
from .transform_pe import TransformPE
from . import transforms
from .delay_pe import DelayPE, InterpolationMode
from .piecewise_pe import PiecewisePE, TransitionType
from .trigger_restart_pe import TriggerRestartPE
from .reverb_pe import ReverbPE
from .spatial_pe import (SpatialAdapter, SpatialConstantPower, SpatialHRTF, SpatialLinear, SpatialMethod,
                         SpatialPE)
from .loop_pe import LoopPE
from .window_pe import WindowMode, WindowPE
from .dynamics_pe import DynamicsMode, DynamicsPE, db_to_ratio, ratio_to_db
from .compressor_pe import CompressorPE, ExpanderPE, LimiterPE
from .wav_writer_pe import WavWriterPE
from .wav_reader_pe import WavReaderPE
from .utils import render_to_file
from . import device, diagnostics

__all__ = [
    "ErrorMode", "get_error_mode", "get_sample_rate", "handle_error", "set_error_mode", "set_sample_rate",
    "ExtendMode", "Extent", "Snippet", "ProcessingElement", "SourcePE", "PEProfile", "ProfileReport",
    "Renderer", "NullRenderer", "GateSignal", "TriggerSignal", "ConstantPE", "IdentityPE", "DiracPE",
    "ArrayPE", "CachePE", "CropPE", "SinePE", "GainPE", "MixPE", "BiquadMode", "BiquadPE", "BlitSawPE",
    "SuperSawPE", "LadderMode", "LadderPE", "CombPE", "PeriodicGate", "PeriodicTrigger", "AdsrGatedPE",
    "AdsrTriggeredPE", "ConvolvePE", "SVFilterPE", "DetectionMode", "EnvelopePE", "TransformPE",
    "transforms", "DelayPE", "InterpolationMode", "PiecewisePE", "TransitionType", "TriggerRestartPE",
    "ReverbPE", "SpatialPE", "SpatialMethod", "SpatialAdapter", "SpatialLinear", "SpatialConstantPower",
    "SpatialHRTF", "LoopPE", "WindowMode", "WindowPE", "DynamicsMode", "DynamicsPE", "CompressorPE", "LimiterPE",
    "ExpanderPE", "db_to_ratio", "ratio_to_db", "WavWriterPE", "WavReaderPE", "render_to_file", "device", "diagnostics",
]
