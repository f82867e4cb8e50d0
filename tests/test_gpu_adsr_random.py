"""GPU: randomized bit-exact check of the wave-parallel ADSR kernels against the CPU oracle's
literal per-sample loop (oracle/seq_kernels.c), through the batched C-ABI entry points.

Covers what the closed-form "runs" must get right: accumulation rounding in every binade,
ties, clamps that land exactly on a step count (attack_time * sr integral), gate edges on
every chunk offset, arbitrary (non 0/1) gate values, sustain levels 0 and 1, re-triggers in
every phase, and state carried across odd-sized blocks."""

import ctypes as C

import numpy as np
import pytest

from oracle import pe_oracle as O
from pygmu2_amd import device

pytestmark = pytest.mark.gpu

SR = 48000.0


def _params(rng, k, triggered):
    rec = np.zeros(k, dtype=device.ADSR_PARAMS)
    for i in range(k):
        kind = rng.integers(0, 4)
        if kind == 0:      # times that are exact multiples of the sample period
            at, dt, rt = (rng.integers(1, 2000, 3) / SR)
        elif kind == 1:    # power-of-two step counts: exact additions, exact clamp hits
            at, dt, rt = (2.0 ** rng.integers(3, 12, 3)) / SR
        else:
            at, dt, rt = 10 ** rng.uniform(-4, -0.5, 3)
        sl = [0.0, 1.0, 0.5, 0.7, float(rng.uniform(0.01, 0.99))][rng.integers(0, 5)]
        a, d, r = O.adsr_slopes(at, dt, sl, rt, SR)
        rec[i] = (a, d, r, sl, int(round(float(rng.uniform(0, 0.05)) * SR)) if triggered else 0)
    return rec


def _gates(rng, k, n):
    g = np.zeros((k, n), np.float32)
    for i in range(k):
        pos, level = 0, 0.0
        while pos < n:
            run = int(rng.integers(1, [40, 700, 6000][rng.integers(0, 3)]))
            g[i, pos:pos + run] = level
            pos += run
            level = 1.0 - level if rng.random() < 0.9 else [0.5, 2.0, -1.0][rng.integers(0, 3)]
    return g


def _triggers(rng, k, n):
    t = np.zeros((k, n), np.float32)
    for i in range(k):
        m = int(rng.integers(1, 40))
        t[i, rng.integers(0, n, m)] = rng.choice([1.0, 2.0, 0.5], m)
        t[i, rng.integers(0, n, 3)] = -1.0
    return t


def _gates_sparse(rng, k, n):
    """Gates that stay put for hundreds to tens of thousands of samples: stretches long enough for the envelope to
    settle (independent starts of k_adsr_walk_par) next to ones that are not, a few crowded spots, odd levels."""
    g = np.zeros((k, n), np.float32)
    for i in range(k):
        pos, level = 0, float(rng.integers(0, 2))
        top = [900, 5000, 14000, 30000][rng.integers(0, 4)]
        while pos < n:
            run = int(rng.integers(1, 30)) if rng.random() < 0.04 else int(rng.integers(200, top))
            g[i, pos:pos + run] = level
            pos += run
            level = 1.0 - level if rng.random() < 0.95 else [0.5, 2.0, -1.0][rng.integers(0, 3)]
    return g


def _gates_fast(rng, k, n):
    """Gates of 3 .. 12 Hz with jitter: on- and off-times too short for the level to be pinned at sustain / zero, long
    enough (mostly) for the attack to reach 1.0 -- the stretches k_adsr_walk_par starts speculatively at the attack
    completions -- with a few short blips in between."""
    g = np.zeros((k, n), np.float32)
    for i in range(k):
        pos, level = 0, float(rng.integers(0, 2))
        half = int(rng.integers(2000, 8000))
        while pos < n:
            run = int(rng.integers(1, 60)) if rng.random() < 0.03 else int(half * rng.uniform(0.7, 1.3))
            g[i, pos:pos + run] = level
            pos += run
            level = 1.0 - level
    return g


@pytest.mark.parametrize("case", ["gated", "triggered", "gated_small_bank_sparse_gates", "gated_small_bank_fast_gates"])
def test_adsr_batch_bit_exact_random(case):
    lib = device.ensure_init()
    triggered = case == "triggered"
    rng = np.random.default_rng(1234 + int(triggered))
    if case == "gated_small_bank_sparse_gates":
        # up to 128 envelopes and 65 536 frames a gated envelope is walked by eight waves from its independent starts
        rng = np.random.default_rng(99)
        k, n = 96, 150000
        rec = _params(rng, k, False)
        ctl = _gates_sparse(rng, k, n)
        blocks = [48000, 48000, 1, 4097, 30000, 65536 - 48000]
    elif case == "gated_small_bank_fast_gates":
        rng = np.random.default_rng(7)
        k, n = 96, 200000
        rec = _params(rng, k, False)
        for i in range(0, k, 2):                # half of them C5's envelope (10 / 100 / 200 ms, sustain 0.7)
            rec[i] = (*O.adsr_slopes(0.01, 0.1, 0.7, 0.2, SR), 0.7, 0)
        ctl = _gates_fast(rng, k, n)
        blocks = [48000, 48000, 48000, 5000, 33333]
    else:
        k, n = 192, 20000
        rec = _params(rng, k, triggered)
        ctl = _triggers(rng, k, n) if triggered else _gates(rng, k, n)
        blocks = [1, 63, 64, 65, 1000, 4097, 129, 7000]
    blocks.append(n - sum(blocks))
    assert blocks[-1] > 0

    # oracle: literal loop per envelope, state carried across the same blocks
    want = np.zeros((k, n), np.float32)
    for i in range(k):
        st = np.zeros(3)
        pos = 0
        for b in blocks:
            out = np.zeros(b, np.float32)
            seg = np.ascontiguousarray(ctl[i, pos:pos + b])
            if triggered:
                O._c().orc_adsr_triggered(O._p(seg, C.c_float), O._p(out, C.c_float), C.c_int64(pos + 5),
                                          C.c_int64(b), C.c_double(rec[i]["attack_dvdt"]),
                                          C.c_double(rec[i]["decay_dvdt"]), C.c_double(rec[i]["release_dvdt"]),
                                          C.c_double(rec[i]["sustain_level"]),
                                          C.c_int64(int(rec[i]["sustain_samples"])), O._p(st))
            else:
                O._c().orc_adsr_gated(O._p(seg, C.c_float), O._p(out, C.c_float), C.c_int64(b),
                                      C.c_double(rec[i]["attack_dvdt"]), C.c_double(rec[i]["decay_dvdt"]),
                                      C.c_double(rec[i]["release_dvdt"]), C.c_double(rec[i]["sustain_level"]),
                                      O._p(st))
            want[i, pos:pos + b] = out
            pos += b

    params = device.DeviceBuffer.from_host(rec)
    state = device.DeviceBuffer((k, 3), np.float64, zero=True)
    ws = device.DeviceBuffer((lib.pgx_adsr_workspace_bytes(k, max(blocks)),), np.uint8)
    got = np.zeros((k, n), np.float32)
    pos = 0
    for b in blocks:
        cin = device.DeviceBuffer.from_host(np.ascontiguousarray(ctl[:, pos:pos + b]))
        cout = device.DeviceBuffer((k, b), np.float32)
        if triggered:
            device.check(lib.pgx_adsr_triggered(cout.ptr, b, cin.ptr, b, k, pos + 5, b, params.ptr, state.ptr, ws.ptr))
        else:
            device.check(lib.pgx_adsr_gated(cout.ptr, b, cin.ptr, b, k, b, params.ptr, state.ptr, ws.ptr))
        got[:, pos:pos + b] = cout.to_host()
        pos += b
    bad = np.argwhere(got != want)
    assert bad.size == 0, (f"{len(bad)} samples differ; first at envelope {bad[0][0]} sample {bad[0][1]}: "
                           f"got {got[tuple(bad[0])]!r} want {want[tuple(bad[0])]!r}, params {rec[bad[0][0]]}")


@pytest.mark.parametrize("k", [7, 96, 300])
def test_fused_periodic_gate_matches_rendered_gate(k):
    """pgx_adsr_gated_periodic[_to] -- the gate evaluated inside the edge search, one bitmap byte per wave, no clearing
    launch -- against pgx_periodic_gate + pgx_adsr_gated (bit-exact against the oracle's loop, above), bit for bit:
    slow gates (one look per 64 chunks), gates with a phase shorter than 65 samples (every chunk expanded), duty
    cycles near 0 and 1, odd block lengths, states carried, and the variant that writes its states elsewhere."""
    lib = device.ensure_init()
    rng = np.random.default_rng(500 + k)
    rec = _params(rng, k, False)
    gates = np.zeros(k, dtype=device.GATE_PARAMS)
    for i in range(k):
        kind = i % 4
        freq = [rng.uniform(0.3, 12.0), rng.uniform(200.0, 5000.0), rng.uniform(20.0, 200.0), 48000.0 / 130.0][kind]
        duty = [0.5, 0.1, 0.9, 0.01, 0.999, float(rng.uniform(0.02, 0.98))][int(rng.integers(0, 6))]
        gates[i] = (freq / SR, float(rng.uniform(0.0, 1.0)), duty)
    blocks = [4096, 777, 48000, 65, 1, 20000, 65536, 100000]
    gp, params = device.DeviceBuffer.from_host(gates), device.DeviceBuffer.from_host(rec)
    st_ref = device.DeviceBuffer((k, 3), np.float64, zero=True)
    st_fused = device.DeviceBuffer((k, 3), np.float64, zero=True)
    st_a = device.DeviceBuffer((k, 3), np.float64, zero=True)
    st_b = device.DeviceBuffer((k, 3), np.float64, zero=True)
    ws = [device.DeviceBuffer((lib.pgx_adsr_workspace_bytes(k, max(blocks)),), np.uint8) for _ in range(3)]
    pos = 1000
    for b in blocks:
        g = device.DeviceBuffer((k, b), np.float32)
        device.check(lib.pgx_periodic_gate(g.ptr, b, k, pos, b, gp.ptr))
        want = device.DeviceBuffer((k, b), np.float32)
        device.check(lib.pgx_adsr_gated(want.ptr, b, g.ptr, b, k, b, params.ptr, st_ref.ptr, ws[0].ptr))
        fused = device.DeviceBuffer((k, b), np.float32)
        device.check(lib.pgx_adsr_gated_periodic(fused.ptr, b, k, pos, b, gp.ptr, params.ptr, st_fused.ptr, ws[1].ptr, 0))
        to = device.DeviceBuffer((k, b), np.float32)
        device.check(lib.pgx_adsr_gated_periodic_to(to.ptr, b, k, pos, b, gp.ptr, params.ptr, st_a.ptr, st_b.ptr,
                                                    ws[2].ptr, 0))
        st_a, st_b = st_b, st_a
        w = want.to_host()
        for name, got in (("fused", fused.to_host()), ("fused, states elsewhere", to.to_host())):
            bad = np.argwhere(got != w)
            assert bad.size == 0, (f"{name}, block of {b} at {pos}: {len(bad)} samples differ; first at envelope "
                                   f"{bad[0][0]} sample {bad[0][1]}: {got[tuple(bad[0])]!r} != {w[tuple(bad[0])]!r}, "
                                   f"gate {gates[bad[0][0]]}")
        assert np.array_equal(st_fused.to_host(), st_ref.to_host()) and np.array_equal(st_a.to_host(), st_ref.to_host())
        pos += b


@pytest.mark.parametrize("gate_hz,times", [(2.0, (0.01, 0.1, 0.7, 0.2)), (7.0, (0.01, 0.02, 0.7, 0.03)),
                                           (5.3, (0.3, 0.05, 0.4, 0.6)),       # attacks that never complete: no pins
                                           (0.11, (0.01, 0.1, 0.7, 0.2)),      # a gate cycle of nine seconds: chunks without an edge
                                           (311.7, (0.0005, 0.001, 0.5, 0.001))])
def test_lone_envelope_over_many_chunks_at_once(gate_hz, times):
    """adsr_run_chunks (round 4): a lone envelope over a window of many 65 536-frame chunks -- a look-ahead window of a
    stand-alone AdsrGatedPE -- is walked with all chunks as one batch, in rounds (every chunk from the carried state,
    then from its left neighbour's exit), settled when two rounds' exits agree bit for bit; what has not settled after
    three rounds takes the chunk-after-chunk loop.  Fused PeriodicGate and a gate stream from memory, against the
    oracle's literal loop: bit-exact, states included; a second block continues from the carried state."""
    lib = device.ensure_init()
    at, dt, sl, rt = times
    a, d, r = O.adsr_slopes(at, dt, sl, rt, SR)
    rec = np.zeros(1, dtype=device.ADSR_PARAMS)
    rec[0] = (a, d, r, sl, 0)
    gates = np.zeros(1, dtype=device.GATE_PARAMS)
    gates[0] = (gate_hz / SR, 0.25, 0.5)
    gp, params = device.DeviceBuffer.from_host(gates), device.DeviceBuffer.from_host(rec)
    blocks = [700_001, 65_536 * 3, 1_000_000]
    st_f = device.DeviceBuffer((1, 3), np.float64, zero=True)
    st_m = device.DeviceBuffer((1, 3), np.float64, zero=True)
    ostate = O.adsr_state()
    pos = 123
    for b in blocks:
        ws = device.DeviceBuffer((lib.pgx_adsr_workspace_bytes(1, b),), np.uint8)
        g = device.DeviceBuffer((1, b), np.float32)
        device.check(lib.pgx_periodic_gate(g.ptr, b, 1, pos, b, gp.ptr))
        want = O.adsr_gated(ostate, g.to_host()[0], at, dt, sl, rt, SR)
        fused = device.DeviceBuffer((1, b), np.float32)
        device.check(lib.pgx_adsr_gated_periodic(fused.ptr, b, 1, pos, b, gp.ptr, params.ptr, st_f.ptr, ws.ptr, 0))
        mem = device.DeviceBuffer((1, b), np.float32)
        device.check(lib.pgx_adsr_gated(mem.ptr, b, g.ptr, b, 1, b, params.ptr, st_m.ptr, ws.ptr))
        for name, got in (("fused gate", fused.to_host()[0]), ("gate from memory", mem.to_host()[0])):
            bad = np.flatnonzero(got != np.asarray(want).reshape(-1))
            assert bad.size == 0, (name, b, pos, len(bad), int(bad[0]), float(got[bad[0]]))
        assert np.array_equal(st_f.to_host(), st_m.to_host())
        pos += b
