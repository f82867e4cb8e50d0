"""GPU: WAV data formats either side of the render path -- device PCM16 conversion against the
oracle's restatement of libsndfile's rule, WavWriterPE -> WavReaderPE round trips, render_to_file."""

import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pcm16_kernels_match_oracle():
    from oracle import pe_oracle as O
    from pygmu2_amd import device
    lib = device.ensure_init()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 200_000), (np.arange(-40000, 40000) + 0.5) / 32767.0,
                        (np.arange(-40000, 40000) + 0.5) / 32768.0,
                        np.array([0.0, -0.0, 1.0, -1.0, 1e-9, 3.0, -3.0, 1e30, -1e30, np.inf, -np.inf, np.nan])]).astype(np.float32)
    xd = device.DeviceBuffer.from_host(x)
    out = device.DeviceBuffer(x.shape, np.int16)
    device.check(lib.pgx_f32_to_pcm16(out.ptr, xd.ptr, x.size))
    got = out.to_host()
    assert np.array_equal(got, O.float_to_pcm16(x))          # overflowing samples saturate, they do not wrap
    assert got[x > 1.0].min() == 32767 and got[x < -1.0].max() == -32768
    back = device.DeviceBuffer(x.shape, np.float32)
    device.check(lib.pgx_pcm16_to_f32(back.ptr, out.ptr, x.size))
    assert np.array_equal(back.to_host(), O.pcm16_to_float(got))


@pytest.mark.parametrize("subtype", ["PCM_16", "FLOAT"])
def test_writer_reader_round_trip(tmp_path, subtype):
    from oracle import pe_oracle as O
    import pygmu2_amd as pg
    pg.set_sample_rate(44100)
    path = str(tmp_path / f"{subtype}.wav")
    src = pg.GainPE(pg.SinePE(frequency=440.0, channels=2), gain=0.8)
    w = pg.WavWriterPE(src, path, subtype=subtype)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(w)
    r.start()
    blocks = [w.render(s, n).data for s, n in ((0, 1024), (1024, 17), (1041, 5000))]
    assert w.frames_written == 6041
    r.stop()
    rendered = np.concatenate(blocks)
    reader = pg.WavReaderPE(path)
    assert reader.channel_count() == 2 and reader.file_sample_rate == 44100
    assert (reader.extent().start, reader.extent().end) == (0, 6041)
    got = reader.render(-10, 6100).data
    assert np.all(got[:10] == 0) and np.all(got[10 + 6041:] == 0)
    body = got[10:10 + 6041]
    if subtype == "FLOAT":
        assert np.array_equal(body, rendered)
    else:
        assert np.array_equal(body, O.pcm16_to_float(O.float_to_pcm16(rendered)))
        with wave.open(path, "rb") as f:
            raw = np.frombuffer(f.readframes(6041), dtype="<i2").reshape(-1, 2)
        assert np.array_equal(raw, O.float_to_pcm16(rendered))


def test_render_to_file_writes_the_whole_extent(tmp_path):
    import pygmu2_amd as pg
    pg.set_sample_rate(48000)
    path = str(tmp_path / "out.wav")
    piece = pg.CropPE(pg.SinePE(frequency=330.0), 0, 24000)
    pg.render_to_file(piece, path)
    with wave.open(path, "rb") as f:
        assert (f.getnchannels(), f.getframerate(), f.getnframes()) == (1, 48000, 24000)
    with pytest.raises(RuntimeError, match="infinite extent"):
        pg.render_to_file(pg.SinePE(frequency=330.0), path)
