"""WAVE files either side of the path (wavread / wavwrite, repet.py:914-946): the library's header parser
and the module's reader / writer against scipy.io.wavfile on files written in the test (CPU), and the device decode /
encode path against the host one (GPU)."""
import os
import struct

import numpy as np
import pytest
import scipy.io.wavfile

import repet
from repet import _native
from repet_synth import synth


def _reference_wavread(path):
    """What repet.py:914-931 computes: SciPy's array over 2^(8 * itemsize - 1)."""
    fs, x = scipy.io.wavfile.read(path)
    return x / pow(2, x.itemsize * 8 - 1), fs


def _pcm24_file(path, samples24, fs, extensible=False, junk=False):
    """A packed 24-bit PCM file (SciPy cannot write one): samples24 is (N, C) int in [-2^23, 2^23)."""
    n, c = samples24.shape
    raw = (samples24.astype("<i4").reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3]).tobytes()
    if extensible:
        sub = struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHH", 0xFFFE, c, fs, fs * 3 * c, 3 * c, 24) + struct.pack("<HHI", 22, 24, 3 if c == 2 else 0) + sub
    else:
        fmt = struct.pack("<HHIIHH", 1, c, fs, fs * 3 * c, 3 * c, 24)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if junk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde\x00"              # an odd-sized chunk and its pad byte
    chunks += b"data" + struct.pack("<I", len(raw)) + raw
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def _test_files(tmp_path):
    fs = 16000
    x = synth(5.0, fs, 2, 41)
    files = {}
    for name, arr in [("i16", (x * 32767).astype(np.int16)), ("i32", (x * 2147483000).astype(np.int32)),
                      ("u8", ((x * 0.5 + 0.5) * 255).astype(np.uint8)), ("f32", x.astype(np.float32)), ("f64", x),
                      ("i16mono", (x[:, 0] * 32767).astype(np.int16))]:
        files[name] = str(tmp_path / f"{name}.wav")
        scipy.io.wavfile.write(files[name], fs, arr)
    s24 = np.round(x * 8388607).astype(np.int64)
    for name, kw in [("i24", {}), ("i24ext", {"extensible": True}), ("i24junk", {"junk": True})]:
        files[name] = str(tmp_path / f"{name}.wav")
        _pcm24_file(files[name], s24, fs, **kw)
    return files, fs


def test_wavread_equals_scipy_plus_the_references_division(tmp_path):
    files, fs = _test_files(tmp_path)
    for name, path in files.items():
        want, want_fs = _reference_wavread(path)
        got, got_fs = repet.wavread(path)
        assert got_fs == want_fs == fs, name
        assert got.dtype == want.dtype and got.shape == want.shape, name        # float64, except float32 files (NumPy keeps float32 / int)
        assert np.array_equal(got, want), name
    assert repet.wavread(files["u8"])[0].min() >= 0.0                       # the reference's 8-bit quirk: [0, 2)
    assert np.abs(repet.wavread(files["f32"])[0]).max() < 1e-9              # ... and its float quirk: divided by 2^31


def test_wav_header_parser(tmp_path):
    files, fs = _test_files(tmp_path)
    lib = _native.lib()
    expect = {"i16": (1, 2, 16, 2), "i32": (1, 2, 32, 4), "u8": (1, 2, 8, 1), "f32": (3, 2, 32, 4), "f64": (3, 2, 64, 8),
              "i16mono": (1, 1, 16, 2), "i24": (1, 2, 24, 3), "i24ext": (1, 2, 24, 3), "i24junk": (1, 2, 24, 3)}
    for name, path in files.items():
        image = np.fromfile(path, dtype=np.uint8)
        info = _native.WavInfo()
        assert lib.repet_wav_parse(_native.ptr(image), image.size, info) == 0, name
        assert (info.format, info.n_channels, info.bits_per_sample, info.bytes_per_sample) == expect[name], name
        assert info.sampling_frequency == fs and info.n_samples == 5 * fs
        assert bytes(image[info.data_offset - 8:info.data_offset - 4]) == b"data"
    for bad in (b"", b"RIFF\x00\x00\x00\x00WAVE", b"RIFX" + b"\x00" * 40, b"OggS" + b"\x00" * 40,
                open(files["i16"], "rb").read()[:30]):
        image = np.frombuffer(bad + b"\x00", dtype=np.uint8)
        assert lib.repet_wav_parse(_native.ptr(image), len(bad), _native.WavInfo()) == _native.ERR_BAD_ARG
    # a format the parser declines (A-law) still reads through SciPy's own error path, like the reference
    alaw = bytearray(open(files["u8"], "rb").read())
    alaw[20:22] = struct.pack("<H", 6)
    p = tmp_path / "alaw.wav"
    p.write_bytes(bytes(alaw))
    with pytest.raises(ValueError):
        repet.wavread(str(p))


def test_wavwrite_is_byte_identical_to_scipy(tmp_path):
    fs = 22050
    x = synth(1.5, fs, 2, 3)
    for name, arr in [("f64", x), ("f32", x.astype(np.float32)), ("i16", (x * 32767).astype(np.int16)),
                      ("i32", (x * 2e9).astype(np.int32)), ("u8", ((x + 1) * 127).astype(np.uint8)), ("mono", x[:, 0].copy()),
                      ("strided", x[::2])]:
        a, b = str(tmp_path / f"a_{name}.wav"), str(tmp_path / f"b_{name}.wav")
        scipy.io.wavfile.write(a, fs, arr)
        repet.wavwrite(arr, fs, b)
        assert open(a, "rb").read() == open(b, "rb").read(), name
    with pytest.raises(ValueError):
        repet.wavwrite(x.astype(np.float16), fs, str(tmp_path / "f16.wav"))     # SciPy's error, like the reference


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["i16", "i24", "i24ext", "i32", "u8", "f32", "f64", "i16mono"])
def test_device_decode_equals_wavread(tmp_path, name):
    """repet_ctx_upload_wav: raw PCM bytes over PCIe, decode + normalisation on the device, bit-identical to uploading
    what wavread returns; and the result written from the device equals wavwrite of the downloaded arrays."""
    files, fs = _test_files(tmp_path)
    x, _ = repet.wavread(files[name])
    if x.ndim == 1:
        x = x[:, np.newaxis]
    if name in ("f32", "f64"):
        x = x * 2.0 ** 30                                     # the float quirk leaves 1e-10-sized samples: same bits, useful scale
    ctx = repet.Context(0)
    assert ctx.upload_wav(files[name]) == fs
    p = repet.derive_params(fs)
    ctx.execute("original", p)
    got = ctx.download()
    fg = ctx.foreground()
    host = repet.Context(0)
    host.upload(repet.wavread(files[name])[0].reshape(got.shape))
    host.execute("original", p)
    assert np.array_equal(got, host.download()), name
    for which, want in (("background", got), ("foreground", fg)):
        for dtype in (np.float64, np.float32):
            out, ref = str(tmp_path / f"{which}_{np.dtype(dtype).name}.wav"), str(tmp_path / "ref.wav")
            ctx.write_wav(out, which, dtype)
            repet.wavwrite(want.astype(dtype), fs, ref)
            assert open(out, "rb").read() == open(ref, "rb").read(), (name, which, dtype)
    ctx.close()
    host.close()


@pytest.mark.gpu
def test_separate_file_is_the_readme_workflow(tmp_path):
    """README.md:62-72: wavread -> repet.<algo> -> wavwrite of background and foreground, kept on the device."""
    fs = 44100
    x = synth(12.0, fs, 2, 8)
    src = str(tmp_path / "mixture.wav")
    scipy.io.wavfile.write(src, fs, (x * 32767).astype(np.int16))
    for algo in ("sim", "adaptive"):
        bg, fgf = str(tmp_path / "bg.wav"), str(tmp_path / "fg.wav")
        assert repet.separate_file(algo, src, bg, fgf) == fs
        audio, _ = repet.wavread(src)
        want = getattr(repet, algo)(audio, fs)
        _, got_bg = scipy.io.wavfile.read(bg)
        _, got_fg = scipy.io.wavfile.read(fgf)
        assert got_bg.dtype == np.float64 and np.array_equal(got_bg, want)
        assert np.max(np.abs(got_fg - (audio - want))) < 1e-7
