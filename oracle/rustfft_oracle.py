"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/librustfft_oracle.so, the C++ restatement of RustFFT's scalar CPU path
(oracle/rustfft_scalar.hpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and only as the checker / reported baseline.  The product (rustfft_amd) never does.

The Python classes mirror the reference's trait surface (src/lib.rs:184-278): `len()`,
`fft_direction()`, `process`, `process_with_scratch`, `process_outofplace_with_scratch`,
`process_immutable_with_scratch`, `get_*_scratch_len`.  A reference panic becomes `OraclePanic`
carrying the reference's message (src/common.rs:13-104).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librustfft_oracle.so")

FORWARD, INVERSE = 0, 1


class OraclePanic(RuntimeError):
    pass


def build(force=False):
    """Compile the restatement with g++ (see oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("rustfft_oracle_capi.cpp", "rustfft_scalar.hpp", "Makefile")]
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.rfo_last_error.restype = ctypes.c_char_p
        L.rfo_free.argtypes = [vp]
        for name, args in {
            "rfo_plan": [ci, sz, ci], "rfo_new_dft": [ci, sz, ci], "rfo_new_butterfly": [ci, sz, ci],
            "rfo_new_radix4": [ci, sz, ci], "rfo_new_radix4_with_base": [ctypes.c_uint, vp],
            "rfo_new_radixn": [ctypes.c_char_p, sz, vp], "rfo_new_mixed_radix": [vp, vp, ci],
            "rfo_new_good_thomas": [vp, vp, ci], "rfo_new_raders": [vp], "rfo_new_bluesteins": [sz, vp],
        }.items():
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = args
        L.rfo_len.restype = sz
        L.rfo_len.argtypes = [vp]
        L.rfo_direction.argtypes = [vp]
        L.rfo_name.restype = ctypes.c_char_p
        L.rfo_name.argtypes = [vp]
        L.rfo_scratch_len.restype = sz
        L.rfo_scratch_len.argtypes = [vp, ci]
        L.rfo_process.argtypes = [vp, vp, sz]
        L.rfo_process_with_scratch.argtypes = [vp, vp, sz, vp, sz]
        L.rfo_process_outofplace_with_scratch.argtypes = [vp, vp, sz, vp, sz, vp, sz]
        L.rfo_process_immutable_with_scratch.argtypes = [vp, vp, sz, vp, sz, vp, sz]
        L.rfo_time_batch.restype = ctypes.c_double
        L.rfo_time_batch.argtypes = [vp, vp, sz, ci, ci]
        L.rfo_recipe.argtypes = [sz, ctypes.c_char_p, sz]
        L.rfo_same_instance.argtypes = [vp, vp]
        ull = ctypes.c_ulonglong
        L.rfo_modular_exponent.restype = ull
        L.rfo_modular_exponent.argtypes = [ull, ull, ull]
        L.rfo_primitive_root.restype = ull
        L.rfo_primitive_root.argtypes = [ull]
        L.rfo_distinct_prime_factors.restype = sz
        L.rfo_distinct_prime_factors.argtypes = [ull, ctypes.POINTER(ull), sz]
        L.rfo_prime_factors.restype = sz
        L.rfo_prime_factors.argtypes = [sz, ctypes.POINTER(ull), sz]
        L.rfo_partition_factors.argtypes = [sz, ctypes.POINTER(ull), ctypes.POINTER(ull)]
        L.rfo_reverse_bits.restype = sz
        L.rfo_reverse_bits.argtypes = [sz, sz, ctypes.c_uint]
        L.rfo_compute_twiddle.argtypes = [ci, sz, sz, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


def _prec_of(dtype):
    dtype = np.dtype(dtype)
    if dtype in (np.dtype(np.complex64), np.dtype(np.float32)):
        return 32
    if dtype in (np.dtype(np.complex128), np.dtype(np.float64)):
        return 64
    raise TypeError("precision must be complex64/float32 or complex128/float64")


def _cdtype(prec):
    return np.complex64 if prec == 32 else np.complex128


def _check(rc):
    if rc != 0:
        raise OraclePanic(lib().rfo_last_error().decode())


class Fft:
    """Mirror of `dyn Fft<T>` over the oracle."""

    def __init__(self, handle, prec, keep=()):
        if not handle:
            raise OraclePanic(lib().rfo_last_error().decode())
        self._h = ctypes.c_void_p(handle)
        self.prec = prec
        self.dtype = _cdtype(prec)
        self._keep = keep

    def __del__(self):
        try:
            lib().rfo_free(self._h)
        except Exception:
            pass

    def len(self):
        return lib().rfo_len(self._h)

    def fft_direction(self):
        return lib().rfo_direction(self._h)

    def name(self):
        return lib().rfo_name(self._h).decode()

    def get_inplace_scratch_len(self):
        return lib().rfo_scratch_len(self._h, 0)

    def get_outofplace_scratch_len(self):
        return lib().rfo_scratch_len(self._h, 1)

    def get_immutable_scratch_len(self):
        return lib().rfo_scratch_len(self._h, 2)

    def _p(self, a, writable=True):
        assert a.dtype == self.dtype and a.flags.c_contiguous, (a.dtype, self.dtype)
        if writable:
            assert a.flags.writeable
        return a.ctypes.data_as(ctypes.c_void_p)

    def process(self, buffer):
        _check(lib().rfo_process(self._h, self._p(buffer), buffer.size))

    def process_with_scratch(self, buffer, scratch):
        _check(lib().rfo_process_with_scratch(self._h, self._p(buffer), buffer.size, self._p(scratch), scratch.size))

    def process_outofplace_with_scratch(self, input, output, scratch):
        _check(lib().rfo_process_outofplace_with_scratch(self._h, self._p(input), input.size, self._p(output), output.size,
                                                         self._p(scratch), scratch.size))

    def process_immutable_with_scratch(self, input, output, scratch):
        _check(lib().rfo_process_immutable_with_scratch(self._h, self._p(input, False), input.size, self._p(output),
                                                        output.size, self._p(scratch), scratch.size))

    def time_batch(self, buffer, batch, reps=1, threads=1):
        return lib().rfo_time_batch(self._h, self._p(buffer), batch, reps, threads)

    # convenience for tests: transform a copy, return it
    def transform(self, x):
        y = np.ascontiguousarray(x, dtype=self.dtype).copy().reshape(-1)
        self.process(y)
        return y.reshape(np.shape(x))


def plan(dtype, length, direction=FORWARD):
    """FftPlannerScalar::plan_fft (src/plan.rs:289-295)."""
    p = _prec_of(dtype)
    return Fft(lib().rfo_plan(p, length, direction), p)


def dft(dtype, length, direction=FORWARD):
    p = _prec_of(dtype)
    return Fft(lib().rfo_new_dft(p, length, direction), p)


def butterfly(dtype, length, direction=FORWARD):
    p = _prec_of(dtype)
    return Fft(lib().rfo_new_butterfly(p, length, direction), p)


def radix4(dtype, length, direction=FORWARD):
    p = _prec_of(dtype)
    return Fft(lib().rfo_new_radix4(p, length, direction), p)


def radix4_with_base(k, base):
    return Fft(lib().rfo_new_radix4_with_base(k, base._h), base.prec, keep=(base,))


def radixn(factors, base):
    fs = bytes(factors)
    return Fft(lib().rfo_new_radixn(fs, len(fs), base._h), base.prec, keep=(base,))


def mixed_radix(width_fft, height_fft, small=False):
    return Fft(lib().rfo_new_mixed_radix(width_fft._h, height_fft._h, int(small)), width_fft.prec, keep=(width_fft, height_fft))


def good_thomas(width_fft, height_fft, small=False):
    return Fft(lib().rfo_new_good_thomas(width_fft._h, height_fft._h, int(small)), width_fft.prec, keep=(width_fft, height_fft))


def raders(inner):
    return Fft(lib().rfo_new_raders(inner._h), inner.prec, keep=(inner,))


def bluesteins(length, inner):
    return Fft(lib().rfo_new_bluesteins(length, inner._h), inner.prec, keep=(inner,))


def recipe(length):
    buf = ctypes.create_string_buffer(4096)
    _check(lib().rfo_recipe(length, buf, 4096))
    return buf.value.decode()


def same_instance(a, b):
    return bool(lib().rfo_same_instance(a._h, b._h))


def modular_exponent(b, e, m):
    return lib().rfo_modular_exponent(b, e, m)


def primitive_root(p):
    r = lib().rfo_primitive_root(p)
    return r if r else None


def distinct_prime_factors(n):
    out = (ctypes.c_ulonglong * 64)()
    k = lib().rfo_distinct_prime_factors(n, out, 64)
    return [int(out[i]) for i in range(k)]


def prime_factors(n):
    out = (ctypes.c_ulonglong * 256)()
    k = lib().rfo_prime_factors(n, out, 256)
    v = [int(out[i]) for i in range(k)]
    d = {"n": v[0], "power_two": v[1], "power_three": v[2], "total": v[3], "distinct": v[4], "other": []}
    for i in range(v[5]):
        d["other"].append((v[6 + 2 * i], v[7 + 2 * i]))
    return d


def partition_factors(n):
    l, r = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    _check(lib().rfo_partition_factors(n, ctypes.byref(l), ctypes.byref(r)))
    return int(l.value), int(r.value)


def reverse_bits(value, d, digits):
    return lib().rfo_reverse_bits(value, d, digits)


def compute_twiddle(dtype, index, fft_len, direction=FORWARD):
    re, im = ctypes.c_double(), ctypes.c_double()
    lib().rfo_compute_twiddle(_prec_of(dtype), index, fft_len, direction, ctypes.byref(re), ctypes.byref(im))
    return complex(re.value, im.value)


def time_batch_native(dtype, length, direction, buffer, batch, reps=1, threads=1):
    """The same restatement compiled `-O3 -march=native` ON THE MACHINE THAT RUNS IT (SURVEY section 8(d) asks for this
    figure next to the -O2 one), timed the way Fft.time_batch times the default build.  The library goes to the
    temporary directory: it is host-specific and never travels.  Returns seconds, or None when g++ is unavailable."""
    import tempfile

    out = os.path.join(tempfile.gettempdir(), "librustfft_oracle_native.so")
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(_HERE, "rustfft_scalar.hpp")):
            subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-pthread",
                                   "-o", out, os.path.join(_HERE, "rustfft_oracle_capi.cpp")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = ctypes.CDLL(out)
    except Exception:
        return None
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.rfo_plan.restype = vp
    L.rfo_plan.argtypes = [ci, sz, ci]
    L.rfo_free.argtypes = [vp]
    L.rfo_time_batch.restype = ctypes.c_double
    L.rfo_time_batch.argtypes = [vp, vp, sz, ci, ci]
    h = L.rfo_plan(_prec_of(dtype), length, direction)
    if not h:
        return None
    try:
        return float(L.rfo_time_batch(vp(h), buffer.ctypes.data_as(vp), batch, reps, threads))
    finally:
        L.rfo_free(vp(h))
