"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol declared in
include/repet_hip.h, the ctypes structs match the header, and the host-side shim derives the integer
parameters exactly as the reference does. No compute call is made here (no GPU in this tier)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import repet
from repet import _native
from oracle import repet_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "repet_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(repet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.lib()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in repet_hip.h but not exported"
    assert set(names) == set(_native.EXPORTED_SYMBOLS)
    assert lib.repet_abi_version() == _native.ABI_VERSION == 4


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "repet_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(repet_params), sizeof(repet_timing),'
                   'offsetof(repet_params, seg_len_samples), offsetof(repet_params, sim_threshold),'
                   'offsetof(repet_timing, stage_bytes), sizeof(repet_settings), offsetof(repet_settings, filter_order));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(_native.Params), ctypes.sizeof(_native.Timing),
                     _native.Params.seg_len_samples.offset, _native.Params.sim_threshold.offset,
                     _native.Timing.stage_bytes.offset, ctypes.sizeof(_native.Settings),
                     _native.Settings.filter_order.offset]


@pytest.mark.parametrize("fs,expect", [
    (44100, dict(window_length=2048, step_length=1024, period_lo=43, period_hi=431, cutoff_bins=5,
                 sim_distance_frames=43, buffer_frames=431, seg_len_frames=431, seg_step_frames=215,
                 seg_len_samples=441000, seg_step_samples=220500)),
    (48000, dict(window_length=2048, period_lo=47, period_hi=469, cutoff_bins=4, seg_len_frames=469,
                 seg_step_frames=234)),
    (8000, dict(window_length=512, step_length=256, cutoff_bins=6, buffer_frames=312)),
    (16000, dict(window_length=1024, cutoff_bins=6)),
])
def test_derived_sizes(fs, expect):
    p = repet.derive_params(fs)
    for k, v in expect.items():
        assert getattr(p, k) == v, k
    # and they agree with the oracle's own derivation
    op = orc.Params()
    w, _, h = orc.stft_geometry(fs)
    assert (p.window_length, p.step_length) == (w, h)
    assert [p.period_lo, p.period_hi] == list(orc.period_range_frames(op, fs, h))
    assert p.cutoff_bins == orc.cutoff_bins(op, fs, w)


def test_c_side_derivation_matches_python(monkeypatch):
    """repet_derive_params (for hosts without Python globals) against derive_params, half-to-even cases included."""
    lib = _native.lib()
    d = _native.Settings()
    lib.repet_default_settings(ctypes.byref(d))
    assert (d.cutoff_frequency, list(d.period_range), d.segment_length, d.segment_step) == (100, [1, 10], 10, 5)
    assert (d.filter_order, d.similarity_threshold, d.similarity_distance, d.similarity_number, d.buffer_length) == (5, 0, 1, 100, 10)
    rs = np.random.RandomState(5)
    rates = [4000, 8000, 11025, 16000, 22050, 32000, 44100, 48000, 51200, 88200, 96000, 12345, 6400.5]
    for trial in range(60):
        fs = rates[trial % len(rates)]
        s = _native.Settings()
        lib.repet_default_settings(ctypes.byref(s))
        if trial >= len(rates):                     # random settings; x.5 products exercise the rounding rule
            s.cutoff_frequency = float(rs.choice([0, 62.5, 100, 250.5, 1000]))
            s.period_range[0], s.period_range[1] = float(rs.choice([0.5, 1, 1.5])), float(rs.choice([3, 7.25, 10]))
            s.segment_length, s.segment_step = float(rs.choice([5, 8, 10, 12.5])), float(rs.choice([1.25, 2.5, 5]))
            s.filter_order, s.similarity_number = int(rs.randint(1, 12)), int(rs.randint(1, 300))
            s.similarity_threshold, s.similarity_distance = float(rs.rand()), float(rs.choice([0, 0.5, 1, 2.5]))
            s.buffer_length = float(rs.choice([2, 5, 10, 12.5]))
        for name in ("cutoff_frequency", "segment_length", "segment_step", "filter_order", "similarity_threshold",
                     "similarity_distance", "similarity_number", "buffer_length"):
            monkeypatch.setattr(repet, name, getattr(s, name))
        monkeypatch.setattr(repet, "period_range", list(s.period_range))
        want = repet.derive_params(fs)
        got = _native.Params()
        assert lib.repet_derive_params(ctypes.byref(s), fs, ctypes.byref(got)) == 0
        for field, _ in _native.Params._fields_:
            assert getattr(got, field) == getattr(want, field), (fs, field)
    got = _native.Params()
    assert lib.repet_derive_params(None, 8000.0, ctypes.byref(got)) == 0 and got.buffer_frames == 312   # round(312.5)
    assert lib.repet_derive_params(None, 0.0, ctypes.byref(got)) == -1


def test_c_client_example_compiles_against_the_header(tmp_path):
    exe = tmp_path / "c_client"
    lib_dir = os.path.dirname(_native.LIB_PATH)
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_client.c"),
                           "-o", str(exe), "-L", lib_dir, "-lrepet_hip", f"-Wl,-rpath,{lib_dir}"])
    assert exe.exists()


def test_device_code_has_no_packed_fp32_or_scratch(tmp_path):
    """The library is built without packed-fp32 VALU ops (csrc/Makefile NO_PK, DESIGN.md "Contexts and concurrency":
    FFT kernels built with them were damaged by co-resident kernels of other contexts) and no kernel may touch scratch
    memory (a spill in a hot loop costs a full memory wait; a select between a global address and the address of a local
    turns the loads into flat ones)."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    copy = tmp_path / "librepet_hip.so"
    shutil.copy(_native.LIB_PATH, copy)
    subprocess.check_call([objdump, "--offloading", str(copy)], cwd=tmp_path, stdout=subprocess.DEVNULL)
    bundles = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f)
    assert bundles, "no device code objects found in librepet_hip.so"
    spilling = set()
    for b in bundles:
        text = subprocess.check_output([objdump, "-d", str(tmp_path / b)], text=True)
        assert not re.search(r"\bv_pk_(add|mul|fma)_f32\b|\bv_pk_mov_b32\b", text), f"packed fp32 op in {b}"
        assert not re.search(r"\bscratch_(load|store)", text), f"scratch access in {b}"
        # the kernel descriptors say it directly: no private segment, no spilled registers (a buffer access through
        # s[0:3] is NOT evidence of scratch: kernels build their own buffer resources there)
        notes = subprocess.check_output([objdump.replace("llvm-objdump", "llvm-readelf"), "--notes", str(tmp_path / b)], text=True)
        kernel = None
        for line in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", line)
            if m:
                kernel = m.group(1)
            m = re.search(r"\.(private_segment_fixed_size|vgpr_spill_count):\s+(\d+)", line)
            if m and int(m.group(2)) != 0:
                spilling.add((kernel, m.group(1), int(m.group(2))))
    assert not spilling, spilling


def test_parameters_are_read_at_call_time():
    saved = repet.period_range, repet.similarity_number
    try:
        repet.period_range = [2, 4]
        repet.similarity_number = 7
        p = repet.derive_params(44100)
        assert (p.period_lo, p.period_hi, p.sim_number) == (86, 172, 7)
    finally:
        repet.period_range, repet.similarity_number = saved


def test_defaults_match_reference_globals():
    assert (repet.cutoff_frequency, repet.period_range, repet.segment_length, repet.segment_step) == (100, [1, 10], 10, 5)
    assert (repet.filter_order, repet.similarity_threshold, repet.similarity_distance) == (5, 0, 1)
    assert (repet.similarity_number, repet.buffer_length) == (100, 10)
    for name in ("original", "extended", "adaptive", "sim", "simonline", "wavread", "wavwrite", "specshow",
                 "_stft", "_istft"):
        assert callable(getattr(repet, name))


@pytest.mark.parametrize("algo", ["original", "extended", "adaptive", "sim", "simonline"])
def test_one_dimensional_input_raises_value_error(algo):
    with pytest.raises(ValueError):
        getattr(repet, algo)(np.zeros(44100), 44100)


def test_frame_count_matches_oracle():
    lib = _native.lib()
    for n in (0, 1, 511, 512, 513, 1023, 1024, 1025, 44100, 1014301):
        for w in (512, 2048):
            h = w // 2
            assert lib.repet_frame_count(n, w, h, 1) == orc.centred_frame_count(n, w, h)
            if n >= w:
                assert lib.repet_frame_count(n, w, h, 0) == orc.online_frame_count(n, w, h)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "repet-python_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "repet_oracle" not in text and "import oracle" not in text, f


def test_wav_roundtrip(tmp_path):
    x = (np.random.RandomState(0).randn(1000, 2) * 1000).astype(np.int16)
    path = str(tmp_path / "a.wav")
    repet.wavwrite(x, 8000, path)
    y, fs = repet.wavread(path)
    assert fs == 8000 and np.allclose(y, x / 32768.0)


def test_no_compiled_binary_is_tracked():
    """Executables and shared objects are build products (they travel to the GPU box with the snapshot, not with the history):
    none of the files git tracks may be an ELF image."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        files = subprocess.run(["git", "ls-files"], cwd=root, capture_output=True, text=True, check=True).stdout.split("\n")
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout")
    elf = []
    for name in files:
        path = os.path.join(root, name)
        if name and os.path.isfile(path):
            with open(path, "rb") as fh:
                if fh.read(4) == b"\x7fELF":
                    elf.append(name)
    assert not elf, elf


@pytest.mark.parametrize("n,seed", [(1, 1), (63, 2), (4095, 3), (4096, 4), (70001, 5), (1 << 20, 6)])
def test_host_conversions_against_scalar_loops(n, seed):
    """The conversions of a staged upload / download (hostio.hip: float64 -> fp32 samples + fp32 remainders with the
    PCM-exact fast blocks, fp32 -> float64; non-temporal AVX-512 / AVX2 lines where the CPU has them, REPET_HOST_NT) against
    scalar loops: NaN, infinities, denormals, PCM-exact runs, every misalignment of a part's first element, nothing written
    outside the part. Runs without a GPU."""
    assert _native.lib().repet_host_conversion_selftest(n, seed) == 0
    assert _native.lib().repet_host_conversion_selftest(0, seed) == -1
