"""Host audio decoding (csrc/host_audio.cpp): native FLAC next to WAV -- what the reference gets from libsndfile
(sonar/inference_pipelines/speech.py:292-308).  Vectors: the decoding example of RFC 9639 (appendix D.1, both CRCs
verified by the decoder) and streams written by tests/flac_writer.py that force every subframe type, residual coding,
stereo decorrelation, block-size code and sample size the decoder has a branch for."""
import ctypes as C
import random

import numpy as np
import pytest

from tests import flac_writer as FW


def _decode(data: bytes):
    from sonar_amd import _lib

    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    ch, rate, frames = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    _lib.check(lib.smi_host_audio_info(buf, len(data), C.byref(ch), C.byref(rate), C.byref(frames)))
    out = np.zeros((frames.value, ch.value), dtype=np.float32)
    if frames.value:
        _lib.check(lib.smi_host_audio_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), frames.value, ch.value))
    return out, rate.value


RFC9639_D1 = bytes.fromhex(
    "664c6143" "80000022" "10001000" "00000f00" "000f0ac4" "42f00000" "00013e84" "b41807dc" "69030758" "6a3dad1a" "2e0f"
    "fff86918" "0000bf03" "58fd0312" "8baa9a")


def test_rfc9639_example_stream():
    out, rate = _decode(RFC9639_D1)
    assert rate == 44100 and out.shape == (1, 2)
    assert (out[0] * 32768).tolist() == [25588.0, 10416.0]     # verbatim subframes with 2 and 4 wasted bits


def _signal(n, bps, seed, nch=2):
    rng = random.Random(seed)
    amp = (1 << (bps - 1)) - 1
    chans = []
    for c in range(nch):
        ph, s = rng.random() * 6.28, []
        for i in range(n):
            v = 0.55 * amp * np.sin(ph + i * (0.02 + 0.013 * c)) + 0.02 * amp * rng.uniform(-1, 1)
            s.append(int(max(-amp - 1, min(amp, round(v)))))
        chans.append(s)
    if nch == 2:                                   # correlated pair, as real stereo is
        chans[1] = [int(max(-amp - 1, min(amp, a + (b >> 4)))) for a, b in zip(chans[0], chans[1])]
    return chans


def _expect(chans, bps):
    return (np.array(chans, dtype=np.float64).T / float(1 << (bps - 1))).astype(np.float32)


@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24])
def test_flac_subframe_types_and_residual_codings(bps):
    blocks = [192, 576, 256, 1024, 100, 4096, 300, 7]
    n = sum(blocks)
    chans = _signal(n, bps, seed=bps)
    for i in range(192, 192 + 576):                # a constant run and a run with wasted bits
        chans[0][i] = chans[0][192]
    for i in range(768, 1024):
        chans[1][i] &= ~7
    lpc = dict(kind="lpc", coefs=[1450, -820, 150], prec=12, shift=10, porder=2)
    frames = [
        dict(n=192, sub=[dict(kind="verbatim"), dict(kind="fixed", order=0, porder=0)]),
        dict(n=576, sub=[dict(kind="constant"), dict(kind="fixed", order=1, porder=2, rice2=True)]),
        dict(n=256, stereo="ls", sub=[dict(kind="fixed", order=2, porder=3), dict(kind="fixed", order=3, porder=1, escape=True)]),
        dict(n=1024, stereo="ms", sub=[dict(kind="fixed", order=4, porder=4, rice2=True, escape=True), lpc]),
        dict(n=100, stereo="sr", sub=[dict(kind="lpc", coefs=[16000, -7000], prec=15, shift=13), dict(kind="fixed", order=2)]),
        dict(n=4096, sub=[dict(kind="lpc", coefs=[900, -300, 120, -60, 30, -15, 8, -4, 2, -1, 1, 1], prec=11, shift=9, porder=5),
                          dict(kind="fixed", order=2, porder=6)]),
        dict(n=300, explicit_bs=True, sub=[dict(kind="nowaste"), dict(kind="verbatim")]),
        dict(n=7, sub=[dict(kind="fixed", order=4), dict(kind="lpc", coefs=[255, 120, -60, 30, -500], prec=10, shift=9)]),
    ]
    for kw in (dict(), dict(total_known=False), dict(variable=True, id3=True), dict(explicit_codes=False)):
        data = FW.encode(chans, bps, 16000, frames, **kw)
        out, rate = _decode(data)
        assert rate == 16000
        np.testing.assert_array_equal(out, _expect(chans, bps))


def test_flac_mono_32_bit_and_many_channels():
    chans = _signal(600, 32, seed=5, nch=1)
    frames = [dict(n=512, sub=[dict(kind="fixed", order=2, porder=1, rice2=True)]), dict(n=88, sub=[dict(kind="verbatim")])]
    out, _ = _decode(FW.encode(chans, 32, 48000, frames))
    np.testing.assert_array_equal(out, _expect(chans, 32))
    chans = _signal(1152, 16, seed=6, nch=1) + _signal(1152, 16, seed=7, nch=2) + _signal(1152, 16, seed=8, nch=2)
    frames = [dict(n=1152, sub=[dict(kind="fixed", order=c % 5, porder=c % 3) for c in range(5)])]
    out, _ = _decode(FW.encode(chans, 16, 44100, frames))
    np.testing.assert_array_equal(out, _expect(chans, 16))


def test_flac_long_frame_numbers_and_sample_rate_field():
    chans = _signal(2 * 256, 16, seed=9, nch=1)
    frames = [dict(n=256, number_offset=70000, sr_hz16=True, sub=[dict(kind="fixed", order=1)]),
              dict(n=256, number_offset=70000, sub=[dict(kind="fixed", order=2)])]
    out, rate = _decode(FW.encode(chans, 16, 16000, frames))
    assert rate == 16000
    np.testing.assert_array_equal(out, _expect(chans, 16))


def test_flac_errors_are_reported():
    from sonar_amd import _lib

    chans = _signal(512, 16, seed=1)
    frames = [dict(n=512, sub=[dict(kind="fixed", order=2, porder=2)] * 2)]
    good = FW.encode(chans, 16, 16000, frames)
    _decode(good)
    bad = bytearray(good)
    bad[-40] ^= 0x10                                # a flipped residual bit: the frame CRC-16 catches it
    with pytest.raises(_lib.SmiError, match="CRC|residual|sync|subframe"):
        _decode(bytes(bad))
    with pytest.raises(_lib.SmiError, match="stream ends|truncated|CRC"):
        _decode(good[:-100])
    with pytest.raises(_lib.SmiError, match="Ogg"):
        _decode(b"OggS" + b"\x00" * 64)
    with pytest.raises(_lib.SmiError, match="RIFF"):
        _decode(b"\x00" * 64)
    # an ID3v2 tag in front of something that is not FLAC (an MP3 file) is not claimed as FLAC
    with pytest.raises(_lib.SmiError, match="MPEG"):
        _decode(b"ID3\x04\x00\x00\x00\x00\x00\x10" + b"\x00" * 16 + b"\xff\xfb\x90\x00" + b"\x00" * 64)
    # STREAMINFO's sample count sizes the caller's buffer: a count the stream cannot hold is refused up front
    # (2^36 - 1 samples claimed by a 600-byte file would otherwise be a terabyte-sized allocation)
    huge = bytearray(good)
    huge[4 + 4 + 13] |= 0x0f
    huge[4 + 4 + 14: 4 + 4 + 18] = b"\xff\xff\xff\xff"
    with pytest.raises(_lib.SmiError, match="STREAMINFO claims"):
        _decode(bytes(huge))
    # a crafted residual that overflows the predictor is arithmetic garbage, not undefined behaviour: reported
    wild = bytearray(good)
    for i in range(60, len(wild) - 4):
        wild[i] = 0x00
    with pytest.raises(_lib.SmiError):
        _decode(bytes(wild))


def test_pipeline_reads_flac_files(tmp_path):
    from sonar_amd.inference_pipelines.speech import read_wav

    chans = _signal(16000, 16, seed=3, nch=1)
    frames = [dict(n=4096, sub=[dict(kind="fixed", order=2, porder=4)])] * 3 + [dict(n=16000 - 3 * 4096, sub=[dict(kind="fixed", order=2)])]
    p = tmp_path / "clip.flac"
    p.write_bytes(FW.encode(chans, 16, 16000, frames))
    wav = read_wav(p)
    assert wav.shape == (1, 16000)
    np.testing.assert_array_equal(wav.numpy()[0], _expect(chans, 16)[:, 0])
    q = tmp_path / "clip44.flac"
    q.write_bytes(FW.encode(chans, 16, 44100, frames))
    with pytest.raises(ValueError, match="16 kHz"):
        read_wav(q)
