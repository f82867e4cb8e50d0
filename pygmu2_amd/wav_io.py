"""
Minimal RIFF/WAVE reader and writer for the two sample formats the render path meets:
PCM_16 (the reference's default subtype, wav_writer_pe.py:67) and 32-bit IEEE FLOAT.

The reference goes through python-soundfile/libsndfile, which this image does not have; the
container format is the public RIFF/WAVE layout ("RIFF" size "WAVE", "fmt " chunk with
format tag 1 = PCM or 3 = IEEE float, "data" chunk, little endian), the sample conversion
libsndfile's (see pgx_f32_to_pcm16 in include/pygmu_hip.h).  Files written here open in the
standard library's `wave` module and in libsndfile; files with a WAVE_FORMAT_EXTENSIBLE header
(tag 0xFFFE) are read through their sub-format.
"""

from __future__ import annotations

import os
import struct

import numpy as np

SUBTYPES = {"PCM_16": (1, 16), "FLOAT": (3, 32)}


class WavInfo:
    __slots__ = ("frames", "channels", "sample_rate", "format_tag", "bits", "data_offset")

    def __init__(self, frames, channels, sample_rate, format_tag, bits, data_offset):
        self.frames, self.channels, self.sample_rate = frames, channels, sample_rate
        self.format_tag, self.bits, self.data_offset = format_tag, bits, data_offset

    @property
    def subtype(self) -> str:
        return "FLOAT" if self.format_tag == 3 else f"PCM_{self.bits}"


def read_info(path: str) -> WavInfo:
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError(f"{path}: no data chunk")
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            if cid == b"fmt ":
                body = f.read(size)
                tag, ch, rate, _bps, align, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and size >= 26:                 # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                    tag = struct.unpack("<H", body[24:26])[0]
                fmt = (tag, ch, rate, align, bits)
                if size & 1:
                    f.seek(1, os.SEEK_CUR)
            elif cid == b"data":
                if fmt is None:
                    raise ValueError(f"{path}: data chunk before fmt chunk")
                tag, ch, rate, align, bits = fmt
                if (tag, bits) not in ((1, 16), (3, 32)):
                    raise ValueError(f"{path}: unsupported sample format (tag {tag}, {bits} bits); "
                                     "PCM_16 and FLOAT are supported")
                offset = f.tell()
                avail = os.path.getsize(path) - offset
                size = min(size, avail) if size not in (0, 0xFFFFFFFF) else avail
                return WavInfo(size // (ch * bits // 8), ch, rate, tag, bits, offset)
            else:
                f.seek(size + (size & 1), os.SEEK_CUR)


def read_frames(path: str, info: WavInfo, start: int, stop: int) -> np.ndarray:
    """Raw samples [start, stop) as stored: int16 (PCM_16) or float32 (FLOAT), shape (frames, channels)."""
    start, stop = max(0, start), min(info.frames, stop)
    dtype = np.dtype("<f4") if info.format_tag == 3 else np.dtype("<i2")
    count = max(0, stop - start) * info.channels
    with open(path, "rb") as f:
        f.seek(info.data_offset + start * info.channels * dtype.itemsize)
        data = np.fromfile(f, dtype=dtype, count=count)
    return data.reshape(-1, info.channels)


class WavFileWriter:
    """Streams sample blocks into a WAV file; the sizes in the header are patched on close."""

    def __init__(self, path: str, sample_rate: int, channels: int, subtype: str = "PCM_16"):
        if subtype not in SUBTYPES:
            raise ValueError(f"unsupported WAV subtype {subtype!r}; supported: {sorted(SUBTYPES)}")
        self.tag, self.bits = SUBTYPES[subtype]
        self.channels = int(channels)
        self.frames = 0
        self._f = open(path, "wb")
        align = self.channels * self.bits // 8
        self._f.write(b"RIFF" + struct.pack("<I", 0) + b"WAVE")
        self._f.write(b"fmt " + struct.pack("<IHHIIHH", 16, self.tag, self.channels, int(sample_rate),
                                            int(sample_rate) * align, align, self.bits))
        if self.tag == 3:                                  # non-PCM formats carry a fact chunk
            self._f.write(b"fact" + struct.pack("<II", 4, 0))
        self._f.write(b"data" + struct.pack("<I", 0))
        self._data_start = self._f.tell()

    def write(self, samples: np.ndarray) -> None:
        """samples: (frames, channels) int16 for PCM_16 or float32 for FLOAT, already converted."""
        want = np.dtype("<f4") if self.tag == 3 else np.dtype("<i2")
        a = np.ascontiguousarray(samples, dtype=want)
        self._f.write(a.tobytes())
        self.frames += a.shape[0]

    def close(self) -> None:
        if self._f is None:
            return
        nbytes = self.frames * self.channels * self.bits // 8
        if nbytes & 1:
            self._f.write(b"\0")
        end = self._f.tell()
        self._f.seek(4)
        self._f.write(struct.pack("<I", end - 8))
        if self.tag == 3:
            self._f.seek(self._data_start - 12)
            self._f.write(struct.pack("<I", self.frames))
        self._f.seek(self._data_start - 4)
        self._f.write(struct.pack("<I", nbytes))
        self._f.close()
        self._f = None
