"""CPU: the RIFF/WAVE container code (pygmu2_amd/wav_io.py) against the standard library's `wave`
reader, and the oracle's restatement of libsndfile's PCM_16 conversion on known values."""

import os
import struct
import wave

import numpy as np
import pytest

from oracle import pe_oracle as O
from pygmu2_amd import wav_io


def test_pcm16_conversion_known_answers():
    # libsndfile f2s_clip_array (what python-soundfile selects): x * 32768, saturate, lrintf half-to-even
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 0.25, 1.5 / 32768, 2.5 / 32768, 1.00004, -1.00004, 3.0e-5,
                  0.99997, 32766.5 / 32768, 3.0, -3.0, 1e30, -1e30, np.nan], dtype=np.float32)
    got = O.float_to_pcm16(x)
    #                0  clip   -full   exact   exact  exact 1.5->2 2.5->2 clip   clip    0.98->1
    assert got.tolist() == [0, 32767, -32768, 16384, -16384, 8192, 2, 2, 32767, -32768, 1,
                            32767, 32766, 32767, -32768, 32767, -32768, 0]
    #                       32766.017->clip? no: 0.99997*32768 = 32767.02 -> >= 32767 -> 32767; 32766.5 -> 32766 (even)
    back = O.pcm16_to_float(np.array([0, 32767, -32768, 1], dtype=np.int16))
    assert back.dtype == np.float32
    assert back.tolist() == [0.0, 32767 / 32768, -1.0, 1 / 32768]


def test_pcm16_file_is_readable_by_the_standard_library(tmp_path):
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=(1001, 2), dtype=np.int16)          # odd frame count
    path = str(tmp_path / "a.wav")
    w = wav_io.WavFileWriter(path, 44100, 2, "PCM_16")
    w.write(pcm[:400])
    w.write(pcm[400:])
    w.close()
    with wave.open(path, "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 2, 44100, 1001)
        raw = np.frombuffer(f.readframes(1001), dtype="<i2").reshape(-1, 2)
    assert np.array_equal(raw, pcm)
    info = wav_io.read_info(path)
    assert (info.frames, info.channels, info.sample_rate, info.subtype) == (1001, 2, 44100, "PCM_16")
    assert np.array_equal(wav_io.read_frames(path, info, 0, 1001), pcm)
    assert np.array_equal(wav_io.read_frames(path, info, 990, 2000), pcm[990:])
    assert wav_io.read_frames(path, info, 5000, 6000).shape == (0, 2)


def test_float_file_round_trip_and_header_fields(tmp_path):
    x = np.random.default_rng(1).standard_normal((77, 1)).astype(np.float32)    # odd byte count -> pad byte
    path = str(tmp_path / "f.wav")
    w = wav_io.WavFileWriter(path, 48000, 1, "FLOAT")
    w.write(x)
    w.close()
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8 and raw[8:12] == b"WAVE"
    assert b"fact" in raw[:64]
    info = wav_io.read_info(path)
    assert (info.frames, info.channels, info.sample_rate, info.subtype) == (77, 1, 48000, "FLOAT")
    assert np.array_equal(wav_io.read_frames(path, info, 0, 77), x)


def test_reader_skips_unknown_chunks_and_rejects_other_formats(tmp_path):
    pcm = np.arange(10, dtype="<i2").reshape(-1, 1)
    body = (b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 16000, 2, 16) + b"LIST" + struct.pack("<I", 3) +
            b"abc\0" + b"data" + struct.pack("<I", 20) + pcm.tobytes())
    path = str(tmp_path / "l.wav")
    open(path, "wb").write(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)
    info = wav_io.read_info(path)
    assert info.frames == 10 and np.array_equal(wav_io.read_frames(path, info, 0, 10), pcm)
    bad = str(tmp_path / "b.wav")
    body = b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 8000, 1, 8) + b"data" + struct.pack("<I", 4) + b"\0" * 4
    open(bad, "wb").write(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)
    with pytest.raises(ValueError, match="unsupported sample format"):
        wav_io.read_info(bad)
    with pytest.raises(ValueError, match="not a RIFF"):
        open(bad, "wb").write(b"nope")
        wav_io.read_info(bad)
    with pytest.raises(ValueError, match="unsupported WAV subtype"):
        wav_io.WavFileWriter(str(tmp_path / "c.wav"), 8000, 1, "PCM_24")


KEMAR_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kemar")


@pytest.mark.parametrize("name", sorted(os.listdir(KEMAR_DIR)))
def test_reader_on_the_reference_trees_own_recordings(name):
    """The KEMAR impulse responses of the reference tree (src/pygmu2/assets/kemar: 22 of its 371 files, every 19th by
    name, kept as data fixtures): the reader against the standard library's decoding of the same file, whole and in an
    inner window (the s / 32768 conversion is the device's: tests/test_gpu_wav.py), and the file name against the grid
    rule that picks it (spatial_pe.py:293-520)."""
    import wave
    from pygmu2_amd import wav_io
    from pygmu2_amd.spatial_pe import kemar_entries
    path = os.path.join(KEMAR_DIR, name)
    with wave.open(path, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
        rate = w.getframerate()
    info = wav_io.read_info(path)
    assert (info.frames, info.channels, info.sample_rate) == (pcm.shape[0], pcm.shape[1], rate)
    got = wav_io.read_frames(path, info, 0, info.frames)          # the stored samples; s / 32768 happens on the device
    assert info.subtype == "PCM_16" and got.dtype == np.int16 and np.array_equal(got, pcm)
    a, b = info.frames // 3, info.frames - 7
    assert np.array_equal(wav_io.read_frames(path, info, a, b), pcm[a:b])
    assert any(entry[2] == name for entry in kemar_entries())
