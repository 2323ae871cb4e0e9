"""MI355X-native implementation of AVCER's inference hot path.

The compute lives in avcer_amd/csrc/libavcer_hip.so (C ABI: include/avcer_hip.h); this package is the host mirror of the
reference's call surface.  Importing the package does not load the library; `Engine()` does, and raises without it.
"""
__all__ = ["engine", "models", "pipeline", "video_pipeline", "audio_pipeline", "fusion", "io_formats", "dist", "packing",
           "synth", "build"]
__version__ = "0.1.0"
