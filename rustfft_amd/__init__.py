"""rustfft_amd — MI355X-native engine behind RustFFT's `Fft<T>::process()` hot path.

Host-side mirror of the reference's planner/trait surface (src/plan.rs:72-126, src/lib.rs:140-278) over the
C ABI of include/mi355fft.h.  The compute path is the HIP library only (no CPU fallback).
"""
from .planner import (ALGO_AUTO, ALGO_BLUESTEIN, ALGO_MIXED_RADIX, ALGO_RADER, Fft, FftDirection, FftPanic, FftPlanner,  # noqa: F401
                      FftMulti, FftPlannerHip, FftPlannerHipMulti, Recipe, device_count, device_cpulist)

__all__ = ["ALGO_AUTO", "ALGO_BLUESTEIN", "ALGO_MIXED_RADIX", "ALGO_RADER", "Fft", "FftDirection", "FftPanic", "FftPlanner", "FftMulti", "FftPlannerHip", "FftPlannerHipMulti", "Recipe", "device_count", "device_cpulist"]
