"""CPU-side checks of the drop-in boundary: the HIP library builds, loads, exports every symbol that
include/mi355fft.h declares, and FAILS LOUDLY (no CPU fallback) when no gfx950 device is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    import __graft_entry__ as g

    g.build()
    from rustfft_amd import _native

    return _native.load()


def test_header_symbols_are_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355fft_[a-z_0-9]+)\s*\(", hdr)))
    from rustfft_amd import _native

    assert sorted(_native.EXPORTS) == declared, "rustfft_amd/_native.py EXPORTS must list exactly the header's functions"
    for name in declared:
        assert hasattr(hiplib, name), f"libmi355fft.so does not export {name}"
    assert b"gfx950" in hiplib.mi355fft_version()


def test_no_gpu_means_loud_failure(hiplib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; this test pins the no-device behaviour")
    assert hiplib.mi355fft_device_count() == 0
    h = ctypes.c_void_p()
    rc = hiplib.mi355fft_plan_create(1024, 0, 32, ctypes.byref(h))
    assert rc == 1 and not h.value  # MI355FFT_ERR_NO_DEVICE
    assert b"gfx950" in hiplib.mi355fft_last_error()
    import rustfft_amd

    with pytest.raises(rustfft_amd.FftPanic):
        rustfft_amd.FftPlanner(np.complex64)


def test_product_does_not_reference_the_oracle():
    """The shipped package must never import, link or call oracle/ or the emulator."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rustfft_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "rustfft_oracle" not in text and "librustfft_oracle" not in text, f
                assert "libmi355fft_emu" not in text, f


def test_strerror_covers_reference_panics(hiplib):
    msgs = [hiplib.mi355fft_strerror(i).decode() for i in range(0, 10)]
    assert "Provided FFT buffer was too small" in msgs[2]
    assert "must be a multiple of FFT length" in msgs[3]
    assert "Not enough scratch space" in msgs[4]
    assert "must have the same length" in msgs[5]


def test_cpp_host_mirror_fails_loudly_without_gpu(hiplib):
    """rustfft_amd/host/mi355fft.hpp (FftPlanner<T> / Fft<T> in C++17 over the C ABI) compiles against include/mi355fft.h and,
    with no gfx950 device, its planner constructor throws FftPanic(NO_DEVICE) -- no CPU fallback behind the mirror either."""
    import subprocess

    import torch

    from helpers import build_cpp_mirror_check

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the -m gpu run of the same program")
    exe = build_cpp_mirror_check()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert "no gfx950" in r.stdout


def test_header_is_plain_c_and_layouts_match(tmp_path):
    """include/mi355fft.h must be consumable from plain C (what bindgen / cgo / a C caller sees), and the struct layouts the
    bindings mirror -- ctypes here, #[repr(C)] in shim/rustfft-mi355 and INTEGRATION.md -- must be the compiler's: sizes and
    field offsets of mi355fft_plan_options and mi355fft_recipe_node from a C program against rustfft_amd/_native.py."""
    import subprocess

    from rustfft_amd import _native

    fields = {"mi355fft_plan_options": [f for f, _ in _native.PlanOptions._fields_],
              "mi355fft_recipe_node": [f for f, _ in _native.RecipeNode._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mi355fft.h"', "int main(void) {"]
    for st, fs in fields.items():
        src.append(f'    printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            src.append(f'    printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src += ["    return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for st, cls in (("mi355fft_plan_options", _native.PlanOptions), ("mi355fft_recipe_node", _native.RecipeNode)):
        assert int(got[st]) == ctypes.sizeof(cls), (st, got[st], ctypes.sizeof(cls))
        for f, _ in cls._fields_:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)
    # the same header through a C++ compiler (the host mirror's view)
    cpp = tmp_path / "layout.cpp"
    cpp.write_text('#include "mi355fft.h"\nint main() { return sizeof(mi355fft_plan_options) == 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(cpp), "-o", str(tmp_path / "layout_cpp.o")])
