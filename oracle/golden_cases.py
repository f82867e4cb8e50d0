"""
oracle/golden_cases.py -- TEST INFRASTRUCTURE ONLY.

Declarative list of parity cases for the render hot path.  A case is

    {"name": str, "sr": int, "graph": SPEC, "blocks": [[start, n], ...]}

and a SPEC is a nested dict {"pe": "<ClassName>", <ctor kwargs>}, where a kwarg may
itself be a SPEC (PE-valued parameter), {"rng": seed, "n": N, "ch": C, "scale": s,
"decay": tau} (deterministic pseudo-random float32 array -> ArrayPE input) or
{"values": [...]} (literal array).  Enum-valued kwargs are plain strings.

The same SPEC is interpreted three ways:
  * oracle/gen_golden.py      -> the reference's own PE classes (golden outputs)
  * oracle/graph_eval.py      -> the CPU oracle functions
  * tests/spec_build.py       -> pygmu2_amd PE classes (the HIP product path)

Blocks are rendered in order on one started graph, so contiguous blocks exercise the
carried state and gaps exercise the reset-on-discontinuity rules.
Case shapes follow SURVEY.md section 8(c) "golden vectors to capture".
"""

from __future__ import annotations


def materialize_array(spec):
    """Turn an array SPEC ({"values": ...} or {"rng": ...}) into a float32 ndarray."""
    import numpy as np
    if "values" in spec:
        return np.asarray(spec["values"], dtype=np.float32)
    rng = np.random.default_rng(int(spec["rng"]))
    n, ch = int(spec["n"]), int(spec.get("ch", 1))
    a = rng.standard_normal((n, ch)) * float(spec.get("scale", 1.0))
    if "decay" in spec:
        a = a * np.exp(-np.arange(n) / float(spec["decay"])).reshape(-1, 1)
    return a.astype(np.float32)


def S(pe, **kw):
    d = {"pe": pe}
    d.update(kw)
    return d


def blocks_contig(start, sizes):
    out, pos = [], start
    for n in sizes:
        out.append([pos, n])
        pos += n
    return out


ODD = [17, 23, 19, 41, 7, 93]          # tests/test_convolve_pe.py:153 chunk sizes


def cases():
    C = []

    def add(name, sr, graph, blocks, keep=None):
        # keep: indices of blocks whose output is stored in the fixture (default: all);
        # every block is still rendered, in order, so state is carried through.
        C.append({"name": name, "sr": sr, "graph": graph, "blocks": blocks,
                  "keep": list(range(len(blocks))) if keep is None else list(keep)})

    # ---------------------------------------------------------------- sources (bit-exact)
    add("constant_stereo", 44100, S("ConstantPE", value=0.25, channels=2), [[-3, 9]])
    add("identity_neg", 44100, S("IdentityPE", channels=2), [[-5, 12], [16777210, 12]])
    add("dirac_window", 44100, S("DiracPE", channels=2), [[-2, 5], [1, 4], [0, 1]])
    add("array_zero", 44100, S("ArrayPE", data={"values": [0.0, 0.5, 1.0, 0.5, -1.0]}),
        [[-2, 10], [2, 2], [7, 3]])
    add("array_hold_both", 44100,
        S("ArrayPE", data={"values": [[1.0, -1.0], [0.5, 0.25], [2.0, 3.0]]}, extend_mode="hold_both"),
        [[-3, 9], [5, 4], [-6, 2]])
    add("crop_zero", 44100, S("CropPE", source=S("IdentityPE"), start=3, duration=4),
        [[0, 10], [4, 2], [8, 3], [-4, 3]])
    add("crop_hold", 44100,
        S("CropPE", source=S("IdentityPE", channels=2), start=3, duration=4, extend_mode="hold_both"),
        [[0, 10], [8, 3], [-4, 3]])
    add("crop_open_end", 44100, S("CropPE", source=S("IdentityPE"), start=5, duration=None),
        [[0, 10], [20, 3]])

    # ---------------------------------------------------------------- SinePE pure
    add("sine_kat", 44100, S("SinePE", frequency=440.0), [[0, 64]])
    add("sine_late_stereo", 44100,
        S("SinePE", frequency=440.0, amplitude=0.8, phase=0.3, channels=2),
        [[441000, 2048], [-1000, 512], [44100 * 600, 1024]])
    add("sine_lowfreq", 48000, S("SinePE", frequency=0.37, amplitude=2.0), [[0, 4096]])

    # ---------------------------------------------------------------- SinePE stateful
    fm = S("MixPE", inputs=[S("ConstantPE", value=440.0), S("SinePE", frequency=5.0, amplitude=50.0)])
    add("sine_fm", 44100, S("SinePE", frequency=fm, amplitude=0.5),
        blocks_contig(0, [1024, 1024, 17, 4096]))
    add("sine_am_pm", 44100,
        S("SinePE", frequency=220.0, amplitude=S("SinePE", frequency=3.0, amplitude=0.5),
          phase=S("SinePE", frequency=110.0, amplitude=2.0), channels=2),
        blocks_contig(0, [512, 2048]))
    add("sine_fm_constphase_quirk", 44100, S("SinePE", frequency=fm, phase=0.5),
        blocks_contig(0, [256, 256, 256]))

    # ---------------------------------------------------------------- Gain / Mix
    add("gain_const", 44100, S("GainPE", source=S("SinePE", frequency=440.0, channels=2), gain=0.5),
        [[0, 1024], [1024, 1024]])
    add("gain_const_third", 44100, S("GainPE", source=S("IdentityPE"), gain=1.0 / 3.0), [[0, 4096]])
    add("gain_pe_mono_on_stereo", 44100,
        S("GainPE", source=S("SinePE", frequency=440.0, channels=2), gain=S("SinePE", frequency=5.0)),
        [[0, 4096]])
    add("gain_pe_stereo", 44100,
        S("GainPE", source=S("SinePE", frequency=440.0, channels=2),
          gain=S("ArrayPE", data={"rng": 3, "n": 600, "ch": 2, "scale": 1.0})),
        [[0, 512], [512, 512]])
    add("mix3", 44100,
        S("MixPE", inputs=[S("SinePE", frequency=440.0, amplitude=0.3),
                           S("SinePE", frequency=550.0, amplitude=0.3),
                           S("SinePE", frequency=660.0, amplitude=0.3)]),
        [[0, 4096]])
    add("mix_partial_extents", 44100,
        S("MixPE", inputs=[S("CropPE", source=S("SinePE", frequency=440.0), start=0, duration=300),
                           S("CropPE", source=S("SinePE", frequency=550.0), start=200, duration=300),
                           S("ArrayPE", data={"rng": 5, "n": 100, "ch": 1, "scale": 0.5})]),
        [[0, 256], [256, 256], [512, 64], [1000, 16]])
    add("c1_hello_sine", 44100,
        S("CropPE", start=0, duration=8 * 44100,
          source=S("GainPE", gain=0.3,
                   source=S("MixPE", inputs=[S("SinePE", frequency=440.0),
                                             S("SinePE", frequency=550.0),
                                             S("SinePE", frequency=660.0)]))),
        [[0, 1024], [1024, 1024], [8 * 44100 - 512, 1024]])
    add("c1_sine_gain", 44100,
        S("GainPE", source=S("SinePE", frequency=440.0, amplitude=1.0, phase=0.0, channels=2), gain=0.5),
        blocks_contig(0, [1024] * 4) + [[430 * 1024, 672]])

    # ---------------------------------------------------------------- Biquad constant
    noise2 = {"rng": 11, "n": 6000, "ch": 2, "scale": 0.5}
    for mode in ("lowpass", "highpass", "bandpass", "notch", "allpass",
                 "peaking", "lowshelf", "highshelf"):
        add(f"biquad_const_{mode}", 44100,
            S("BiquadPE", source=S("ArrayPE", data=noise2), frequency=1500.0, q=1.3,
              mode=mode, gain_db=4.5),
            blocks_contig(0, [1024, 1024, 17, 23, 19, 41, 7, 93, 2000]))
    add("c2_biquad_sine", 44100,
        S("BiquadPE", source=S("SinePE", frequency=440.0), frequency=1000.0, q=0.707, mode="lowpass"),
        [[0, 16384]])
    add("c2_biquad_sine_stream", 44100,
        S("BiquadPE", source=S("SinePE", frequency=440.0), frequency=1000.0, q=0.707, mode="lowpass"),
        blocks_contig(0, [1024] * 8))
    add("biquad_highq", 48000,
        S("BiquadPE", source=S("DiracPE"), frequency=300.0, q=100.0, mode="bandpass"),
        blocks_contig(0, [8192, 8192]))
    add("biquad_clamps", 48000,
        S("BiquadPE", source=S("ArrayPE", data={"rng": 12, "n": 3000, "ch": 1, "scale": 0.5}),
          frequency=90000.0, q=0.0001, mode="lowpass"),
        blocks_contig(0, [3000]))

    # ---------------------------------------------------------------- Biquad varying
    add("biquad_var_freq", 44100,
        S("BiquadPE", source=S("SinePE", frequency=440.0),
          frequency=S("SinePE", frequency=5.0, amplitude=500.0), q=0.707, mode="lowpass"),
        blocks_contig(0, [4096, 1024, 17, 2, 2000]))
    add("biquad_var_q", 44100,
        S("BiquadPE", source=S("SinePE", frequency=440.0, channels=2), frequency=1000.0,
          q=S("SinePE", frequency=2.0, amplitude=1.0), mode="bandpass"),
        blocks_contig(0, [4096, 4096]))
    fsweep = S("MixPE", inputs=[S("ConstantPE", value=1200.0), S("SinePE", frequency=3.0, amplitude=900.0)])
    for mode in ("highpass", "peaking", "lowshelf", "highshelf", "notch", "allpass"):
        add(f"biquad_var_{mode}", 48000,
            S("BiquadPE", source=S("ArrayPE", data=noise2), frequency=fsweep,
              q=S("MixPE", inputs=[S("ConstantPE", value=2.0), S("SinePE", frequency=1.0, amplitude=1.5)]),
              mode=mode, gain_db=-6.0),
            blocks_contig(0, [3000, 3000]))

    # ---------------------------------------------------------------- BlitSaw
    add("blitsaw_kat", 44100, S("BlitSawPE", frequency=440.0), blocks_contig(0, [4096]))
    add("blitsaw_chunked", 44100, S("BlitSawPE", frequency=440.0),
        blocks_contig(0, [1024, 1024] + ODD + [1848]))
    add("blitsaw_m20_stereo", 48000,
        S("BlitSawPE", frequency=110.0, amplitude=0.7, initial_phase=0.25, m=20, leak=0.995, channels=2),
        blocks_contig(100, [2048, 2048]))
    add("blitsaw_gap_reset", 44100, S("BlitSawPE", frequency=220.0, initial_phase=1.75),
        [[0, 1024], [1024, 1024], [5000, 1024], [6024, 512]])
    add("blitsaw_fm", 44100,
        S("BlitSawPE", frequency=S("MixPE", inputs=[S("ConstantPE", value=200.0),
                                                     S("SinePE", frequency=4.0, amplitude=120.0)]),
          amplitude=S("MixPE", inputs=[S("ConstantPE", value=0.6), S("SinePE", frequency=2.0, amplitude=0.3)])),
        blocks_contig(0, [4096, 4096]))
    add("blitsaw_m_pe", 44100,
        S("BlitSawPE", frequency=330.0,
          m=S("MixPE", inputs=[S("ConstantPE", value=12.0), S("SinePE", frequency=2.0, amplitude=10.0)])),
        blocks_contig(0, [4096]))
    add("blitsaw_low_high", 48000, S("BlitSawPE", frequency=27.5), blocks_contig(0, [8192]))
    add("blitsaw_high", 48000, S("BlitSawPE", frequency=9000.0), blocks_contig(0, [2048]))
    add("blitsaw_late", 48000, S("BlitSawPE", frequency=440.0), blocks_contig(0, [30000, 30000, 4096]),
        keep=[2])

    # ---------------------------------------------------------------- SuperSaw
    add("supersaw_kat", 44100, S("SuperSawPE", frequency=440.0, voices=7, seed=1234),
        blocks_contig(0, [4096]))
    for mm in ("equal", "linear", "center_heavy"):
        add(f"supersaw_{mm}_6v", 48000,
            S("SuperSawPE", frequency=110.0, amplitude=0.8, voices=6, detune_cents=35.0,
              mix_mode=mm, channels=2, seed=7),
            blocks_contig(0, [1024, 1024, 93]))
    add("supersaw_nophase_3v", 44100,
        S("SuperSawPE", frequency=220.0, voices=3, randomize_phase=False),
        blocks_contig(0, [2048]))
    add("supersaw_1v", 44100, S("SuperSawPE", frequency=220.0, voices=1, seed=3), blocks_contig(0, [1024]))
    add("supersaw_freq_pe", 44100,
        S("SuperSawPE", frequency=S("MixPE", inputs=[S("ConstantPE", value=300.0),
                                                      S("SinePE", frequency=3.0, amplitude=40.0)]),
          amplitude=S("SinePE", frequency=1.0, amplitude=0.9), voices=5, seed=99),
        blocks_contig(0, [2048, 2048]))

    # ---------------------------------------------------------------- Ladder
    saw = S("SuperSawPE", frequency=110.0, voices=7, detune_cents=20.0, seed=0)
    for mode in ("lp24", "lp12", "bp24", "bp12", "hp24", "hp12"):
        add(f"ladder_{mode}", 48000,
            S("LadderPE", source=S("ArrayPE", data=noise2), frequency=1200.0, resonance=0.3,
              mode=mode, drive=1.0, oversample=2),
            blocks_contig(0, [1024, 1024, 17, 983]))
    add("ladder_dc_kat", 44100,
        S("LadderPE", source=S("ConstantPE", value=1.0), frequency=1000.0, resonance=0.0),
        blocks_contig(0, [4096]))
    add("ladder_res09_drive", 48000,
        S("LadderPE", source=saw, frequency=800.0, resonance=0.9, mode="lp24", drive=2.5, oversample=4),
        blocks_contig(0, [2048, 2048]))
    add("ladder_os1_pbg", 48000,
        S("LadderPE", source=S("ArrayPE", data=noise2), frequency=3000.0, resonance=0.6,
          mode="bp12", drive=0.5, passband_gain=0.2, oversample=1),
        blocks_contig(0, [3000]))
    add("ladder_silence_decay", 48000,
        S("LadderPE", source=S("ArrayPE", data={"rng": 21, "n": 600, "ch": 2, "scale": 0.8}),
          frequency=500.0, resonance=0.5),
        blocks_contig(0, [512, 512, 512]))
    add("ladder_mod", 48000,
        S("LadderPE", source=saw,
          frequency=S("MixPE", inputs=[S("ConstantPE", value=1500.0), S("SinePE", frequency=2.0, amplitude=1400.0)]),
          resonance=S("MixPE", inputs=[S("ConstantPE", value=0.5), S("SinePE", frequency=0.5, amplitude=0.7)]),
          drive=S("MixPE", inputs=[S("ConstantPE", value=2.0), S("SinePE", frequency=1.0, amplitude=2.5)]),
          mode="lp24"),
        blocks_contig(0, [4096, 4096]))
    add("c4_voice", 48000,
        S("MixPE", inputs=[
            S("LadderPE", source=S("SuperSawPE", frequency=55.0 * 2 ** (i / 12.0), voices=7,
                                    detune_cents=20.0, seed=i),
              frequency=1200.0, resonance=0.3, mode="lp24", drive=1.0, oversample=2)
            for i in range(3)]),
        blocks_contig(0, [4096, 4096]))

    # ---------------------------------------------------------------- Comb
    add("comb_kat", 44100,
        S("CombPE", source=S("ConstantPE", value=0.25, channels=2), frequency=440.0, feedback=0.7),
        blocks_contig(0, [512]))
    add("comb_sine_chunked", 48000,
        S("CombPE", source=S("SinePE", frequency=330.0), frequency=220.0, feedback=-0.9),
        blocks_contig(0, [1024, 1024] + ODD + [2000]))
    add("comb_sweep", 48000,
        S("CombPE", source=S("ArrayPE", data=noise2),
          frequency=S("MixPE", inputs=[S("ConstantPE", value=400.0), S("SinePE", frequency=3.0, amplitude=380.0)]),
          feedback=S("SinePE", frequency=1.0, amplitude=1.5), min_frequency=30.0, smoothing_samples=200),
        blocks_contig(0, [3000, 3000]))
    add("comb_high_freq", 48000,
        S("CombPE", source=S("ArrayPE", data=noise2), frequency=30000.0, feedback=0.99, smoothing_samples=1),
        blocks_contig(0, [2048]))
    add("comb_step", 48000,
        S("CombPE", source=S("ArrayPE", data=noise2),
          frequency=S("ArrayPE", data={"values": [100.0] * 1000 + [1000.0] * 3000}, extend_mode="hold_last"),
          feedback=0.8, smoothing_samples=480),
        blocks_contig(0, [2500, 2500]))

    # ---------------------------------------------------------------- LoopPE (bit-exact) / WindowPE / dynamics
    add("loop_sine_region", 44100,                       # benchmark_pes.py:310
        S("LoopPE", source=S("CropPE", source=S("SinePE", frequency=440.0), start=0, duration=4410)),
        [[0, 10000], [10000, 4410], [-3000, 5000], [44100 * 50 + 7, 4096]])
    add("loop_count_crossfade", 48000,
        S("LoopPE", source=S("ArrayPE", data={"rng": 21, "n": 3000, "ch": 2, "scale": 0.5}),
          loop_start=200, loop_end=2600, count=4, crossfade_seconds=0.01),
        [[0, 5000], [5000, 4000], [9000, 2000], [9600, 100], [-100, 300]])
    add("loop_crossfade_clamped", 1000,
        S("LoopPE", source=S("IdentityPE"), loop_start=5, loop_end=16, crossfade_seconds=1.0),
        [[0, 40], [-7, 30]])
    win_src = S("ArrayPE", data={"rng": 22, "n": 9000, "ch": 2, "scale": 0.6}, extend_mode="zero")
    for mode in ("max", "min", "mean", "rms"):
        add(f"window_{mode}", 44100, S("WindowPE", source=win_src, window=0.01, mode=mode),
            [[0, 4000], [4000, 5200], [-500, 700], [8800, 600]])
    add("window_mean_signed", 44100, S("WindowPE", source=win_src, window=0.003, mode="mean", rectify=False),
        [[100, 3000]])
    add("window_max_default_sine", 44100, S("WindowPE", source=S("SinePE", frequency=440.0)),      # benchmark_pes.py:349
        [[0, 8192], [8192, 1000]])
    add("window_tiny", 44100, S("WindowPE", source=win_src, window=0.0, mode="max"), [[0, 300]])
    dyn_src = S("GainPE", source=S("SinePE", frequency=220.0, channels=2),
                gain=S("PiecewisePE", points=[[0, 0.02], [6000, 1.0], [12000, 0.05], [20000, 0.6]],
                       transition_type="linear", extend_mode="hold_both"))
    dyn_env = S("EnvelopePE", source=dyn_src, attack=0.002, release=0.02)
    for mode, extra in (("compress", {}), ("compress", {"knee": 6.0}), ("limit", {}), ("limit", {"knee": 4.0}),
                        ("expand", {"ratio": 2.0}), ("expand", {"ratio": 2.0, "knee": 8.0}),
                        ("gate", {"threshold": -25.0}), ("gate", {"threshold": -25.0, "knee": 10.0})):
        name = f"dynamics_{mode}" + ("_soft" if "knee" in extra else "")
        kw = dict(threshold=-18.0, ratio=4.0, mode=mode)
        kw.update(extra)
        add(name, 44100, S("DynamicsPE", source=dyn_src, envelope=dyn_env, **kw), blocks_contig(0, [8000, 8000, 6000]))
    add("dynamics_bench", 44100,                         # benchmark_pes.py:315-321
        S("DynamicsPE", source=S("SinePE", frequency=440.0), envelope=S("EnvelopePE", source=S("SinePE", frequency=440.0)),
          mode="compress", threshold=-10.0, ratio=4.0), blocks_contig(0, [8192, 8192]))
    add("dynamics_unlinked_makeup", 44100,
        S("DynamicsPE", source=dyn_src, envelope=dyn_env, threshold=-15.0, ratio=8.0, makeup_gain=3.0,
          stereo_link=False), blocks_contig(0, [9000]))
    add("dynamics_mono_env_sidechain", 44100,
        S("DynamicsPE", source=dyn_src,
          envelope=S("EnvelopePE", source=S("SinePE", frequency=3.0, amplitude=0.9), attack=0.001, release=0.05),
          threshold=-12.0, ratio=6.0, knee=3.0), blocks_contig(0, [9000, 9000]))
    add("compressor_default", 44100, S("CompressorPE", source=dyn_src), blocks_contig(0, [8192, 8192, 4000]))
    add("compressor_bench", 44100, S("CompressorPE", source=S("SinePE", frequency=440.0)),          # benchmark_pes.py:325
        blocks_contig(0, [8192, 8192]))
    add("compressor_peak_lookahead", 44100,
        S("CompressorPE", source=dyn_src, threshold=-24.0, ratio=3.0, attack=0.005, release=0.05, knee=0.0,
          makeup_gain=2.0, lookahead=0.003, detection="peak"), blocks_contig(0, [8000, 8000]))
    add("limiter_default", 44100, S("LimiterPE", source=dyn_src), blocks_contig(0, [8192, 8192, 4000]))
    add("expander_default", 44100, S("ExpanderPE", source=dyn_src, threshold=-20.0), blocks_contig(0, [8192, 8192, 4000]))
    add("expander_soft", 44100, S("ExpanderPE", source=dyn_src, threshold=-20.0, knee=6.0, gate_range=-40.0),
        blocks_contig(0, [8192, 8192]))

    # ---------------------------------------------------------------- gates / ADSR (bit-exact)
    add("periodic_gate", 48000, S("PeriodicGate", frequency=2.0, duty_cycle=0.5),
        [[0, 48000 // 4], [48000 * 100, 4096], [-5000, 4096]])
    add("periodic_gate_odd", 44100, S("PeriodicGate", frequency=7.3, duty_cycle=0.31, phase=0.4),
        [[0, 20000], [44100 * 1000 + 17, 4096]])
    # PE-driven parameters: FunctionGenPE's stateful path (running phase, restarted on a seek)
    add("periodic_gate_fm", 48000,
        S("PeriodicGate", frequency=S("MixPE", inputs=[S("ConstantPE", value=5.3),
                                                       S("SinePE", frequency=0.7, amplitude=2.0)]),
          duty_cycle=0.37),
        blocks_contig(0, [3000, 5000, 1, 4096]) + [[100, 3000], [3100, 2900], [-700, 1500]])
    add("periodic_gate_pwm", 44100,
        S("PeriodicGate", frequency=6.1, duty_cycle=S("SinePE", frequency=1.3, amplitude=0.8),
          phase=S("ConstantPE", value=0.21)),
        blocks_contig(5000, [4096, 4096, 7]) + [[0, 9000]])
    add("adsr_gate_fm", 48000,
        S("AdsrGatedPE", gate=S("PeriodicGate", frequency=S("ArrayPE", data={"values": [3.0] * 9000 + [11.0] * 9000},
                                                            extend_mode="hold_last"), duty_cycle=0.6),
          attack_time=0.02, decay_time=0.03, sustain_level=0.6, release_time=0.05),
        blocks_contig(0, [6000, 6000, 6000, 6000]))
    add("periodic_trigger", 44100, S("PeriodicTrigger", hz=7.0, phase=0.25, amplitude=2),
        [[0, 20000], [-9000, 9000], [44100 * 3000, 8000]])
    g1 = {"values": [1.0] * 500 + [0.0] * 500}
    add("adsr_full_cycle", 1000,
        S("AdsrGatedPE", gate=S("ArrayPE", data=g1), attack_time=0.010, decay_time=0.020,
          sustain_level=0.5, release_time=0.030),
        blocks_contig(0, [1000]))
    add("adsr_early_release_chunked", 1000,
        S("AdsrGatedPE", gate=S("ArrayPE", data={"values": [1.0] * 5 + [0.0] * 100 + [1.0] * 12 + [0.0] * 3 + [1.0] * 200 + [0.0] * 80}),
          attack_time=0.010, decay_time=0.020, sustain_level=0.5, release_time=0.030),
        blocks_contig(0, [7, 93, 19, 41, 240]))
    add("adsr_bad_gate_values", 1000,
        S("AdsrGatedPE", gate=S("ArrayPE", data={"values": [0.0, 0.5, 1.0, 1.0, 0.5, 0.0, 1.0, 1.0, 0.0, 0.0] * 8}),
          attack_time=0.004, decay_time=0.003, sustain_level=0.25, release_time=0.002),
        blocks_contig(0, [80]))
    add("adsr_periodic_gate", 48000,
        S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.03, duty_cycle=0.5),
          attack_time=0.01, decay_time=0.1, sustain_level=0.7, release_time=0.2),
        blocks_contig(0, [16384, 16384, 16384]))
    add("adsr_sustain_edges", 48000,
        S("AdsrGatedPE", gate=S("PeriodicGate", frequency=5.0, duty_cycle=0.9),
          attack_time=0.02, decay_time=0.02, sustain_level=1.0, release_time=0.005),
        blocks_contig(0, [20000]))
    add("adsr_triggered", 1000,
        S("AdsrTriggeredPE", trigger=S("PeriodicTrigger", hz=2.0), attack_time=0.010, decay_time=0.020,
          sustain_time=0.1, sustain_level=0.5, release_time=0.030),
        blocks_contig(0, [300, 300, 17, 883]))
    add("adsr_triggered_retrigger", 1000,
        S("AdsrTriggeredPE", trigger=S("PeriodicTrigger", hz=25.0, phase=0.5), attack_time=0.030,
          decay_time=0.020, sustain_time=0.005, sustain_level=0.6, release_time=0.050),
        blocks_contig(-100, [500, 500]))

    # ---------------------------------------------------------------- C5 voice graph
    def c5_voice(i):
        return S("GainPE",
                 source=S("BiquadPE", source=S("BlitSawPE", frequency=27.5 * 2 ** (i / 48.0)),
                          frequency=2000.0, q=0.707),
                 gain=S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.0 + 0.01 * i, duty_cycle=0.5),
                        attack_time=0.01, decay_time=0.1, sustain_level=0.7, release_time=0.2))
    add("c5_one_voice", 48000, c5_voice(100), blocks_contig(0, [8192, 8192]))
    add("c5_mix4", 48000, S("MixPE", inputs=[c5_voice(i) for i in (0, 171, 342, 511)]),
        blocks_contig(0, [8192, 8192]))

    # ---------------------------------------------------------------- SVFilter / Envelope / Transform
    for mode in ("lowpass", "highpass", "bandpass", "notch", "peaking", "lowshelf", "highshelf"):
        add(f"svf_const_{mode}", 44100,
            S("SVFilterPE", source=S("ArrayPE", data=noise2), frequency=1500.0, q=1.3, mode=mode, gain_db=4.5),
            blocks_contig(0, [1024, 1024, 17, 23, 19, 41, 7, 93, 2000]))
    add("svf_var_freq", 44100,
        S("SVFilterPE", source=S("SinePE", frequency=440.0),
          frequency=S("SinePE", frequency=5.0, amplitude=500.0), q=0.707, mode="lowpass"),
        blocks_contig(0, [4096, 1024, 17, 2, 2000]))
    add("svf_var_peaking_q", 48000,
        S("SVFilterPE", source=S("ArrayPE", data=noise2), frequency=fsweep,
          q=S("MixPE", inputs=[S("ConstantPE", value=2.0), S("SinePE", frequency=1.0, amplitude=1.5)]),
          mode="peaking", gain_db=-6.0),
        blocks_contig(0, [3000, 3000]))
    add("svf_var_highshelf", 48000,
        S("SVFilterPE", source=S("ArrayPE", data=noise2), frequency=fsweep, q=0.9, mode="highshelf", gain_db=9.0),
        blocks_contig(0, [3000, 3000]))
    burst = {"rng": 31, "n": 9000, "ch": 2, "scale": 0.6, "decay": 1500.0}
    add("envelope_peak", 44100,
        S("EnvelopePE", source=S("ArrayPE", data=burst), attack=0.005, release=0.05, mode="peak"),
        blocks_contig(0, [1024, 1024, 17, 4000, 4000]))
    add("envelope_rms_lookahead", 44100,
        S("EnvelopePE", source=S("ArrayPE", data=burst), attack=0.01, release=0.1, lookahead=0.004, mode="rms"),
        blocks_contig(0, [2048, 2048, 93, 3000]))
    add("envelope_equal_times", 48000,
        S("EnvelopePE", source=S("SinePE", frequency=220.0, amplitude=0.8), attack=0.02, release=0.02),
        blocks_contig(0, [4096, 4096]))
    add("envelope_instant_attack", 48000,
        S("EnvelopePE", source=S("ArrayPE", data=burst), attack=0.0, release=0.03),
        blocks_contig(0, [3000, 3000]))
    env_to_freq = [["clip", 0.0, 1.0], ["sqrt"], ["affine", 2900.0, 100.0]]
    add("transform_chain", 44100,
        S("TransformPE", source=S("ArrayPE", data={"rng": 32, "n": 4000, "ch": 2, "scale": 0.7}), ops=env_to_freq),
        [[0, 4000]])
    add("transform_tanh_abs", 44100,
        S("TransformPE", source=S("SinePE", frequency=330.0, amplitude=3.0), ops=[["tanh"], ["abs"], ["one_minus"], ["square"]]),
        [[0, 4096]])

    def autowah(filter_kind):
        src = S("SinePE", frequency=220.0, amplitude=0.8)
        env = S("EnvelopePE", source=src, attack=0.005, release=0.05, mode="peak")
        ctl = S("TransformPE", source=env, ops=env_to_freq)
        return S("GainPE", source=S(filter_kind, source=src, frequency=ctl, q=10.0, mode="lowpass"), gain=1.0)
    add("autowah_biquad", 44100, autowah("BiquadPE"), blocks_contig(0, [1024] * 8))
    add("autowah_svf", 44100, autowah("SVFilterPE"), blocks_contig(0, [1024] * 8))

    # ---------------------------------------------------------------- Delay / Piecewise / TriggerRestart / Reverb
    tone = {"rng": 40, "n": 3000, "ch": 2, "scale": 0.5}
    add("delay_int", 44100, S("DelayPE", source=S("ArrayPE", data=tone), delay=100),
        [[-50, 200], [150, 1000], [2900, 400]])
    add("delay_int_negative", 44100, S("DelayPE", source=S("ArrayPE", data=tone), delay=-37),
        [[-100, 300], [2800, 300]])
    add("delay_float_linear", 44100, S("DelayPE", source=S("ArrayPE", data=tone), delay=10.5, interpolation="linear"),
        [[-20, 100], [80, 1024], [2990, 60]])
    add("delay_float_cubic", 44100, S("DelayPE", source=S("ArrayPE", data=tone), delay=3.25, interpolation="cubic"),
        [[-20, 100], [80, 1024], [2990, 60]])
    vib = S("MixPE", inputs=[S("ConstantPE", value=100.0), S("SinePE", frequency=5.0, amplitude=50.0)])
    add("delay_pe_vibrato_linear", 44100, S("DelayPE", source=S("SinePE", frequency=440.0), delay=vib),
        blocks_contig(0, [4096, 1024, 17]))
    add("delay_pe_vibrato_cubic", 44100,
        S("DelayPE", source=S("ArrayPE", data={"rng": 41, "n": 6000, "ch": 1, "scale": 0.5}), delay=vib,
          interpolation="cubic"), blocks_contig(0, [4096, 2048, 500]))
    pts = [[0, 0.0], [100, 1.0], [400, 0.25], [401, 0.9], [1000, 0.5]]
    for tt in ("step", "linear", "exponential", "sigmoid", "constant_power"):
        add(f"piecewise_{tt}", 44100, S("PiecewisePE", points=pts, transition_type=tt, extend_mode="zero"),
            [[-50, 200], [150, 1000]])
    add("piecewise_hold_both_stereo", 44100,
        S("PiecewisePE", points=[[10, 2.0], [500, -1.0], [200, 0.0]], transition_type="exponential",
          extend_mode="hold_both", channels=2), [[-100, 300], [200, 500], [700, 100]])
    add("piecewise_single_point", 44100,
        S("PiecewisePE", points=[[5, 0.75]], extend_mode="hold_last"), [[0, 20], [20, 10]])
    add("piecewise_single_point_zero", 44100, S("PiecewisePE", points=[[5, 0.75]]), [[0, 20]])
    add("trigger_restart_sine", 44100,
        S("TriggerRestartPE", trigger=S("PeriodicTrigger", hz=30.0, phase=0.25), src=S("SinePE", frequency=700.0)),
        blocks_contig(0, [1024, 1024, 17, 3000]))
    add("trigger_restart_blitsaw", 48000,
        S("TriggerRestartPE", trigger=S("PeriodicTrigger", hz=20.0),
          src=S("BlitSawPE", frequency=330.0)), blocks_contig(0, [4096, 100, 4000]))
    room = {"rng": 42, "n": 300, "ch": 1, "scale": 0.2, "decay": 60.0}
    add("reverb_mix_03", 10000,
        S("ReverbPE", source=S("ArrayPE", data={"rng": 43, "n": 2000, "ch": 2, "scale": 0.5}),
          ir=S("ArrayPE", data=room), mix=0.3, fft_size=1024), blocks_contig(0, [700, 700, 700, 400]))
    add("reverb_unnormalised", 10000,
        S("ReverbPE", source=S("ArrayPE", data={"rng": 43, "n": 2000, "ch": 2, "scale": 0.5}),
          ir=S("ArrayPE", data=room), mix=1.0, normalize_ir=False, fft_size=1024), blocks_contig(0, [1000, 1400]))
    add("reverb_mix_pe", 10000,
        S("ReverbPE", source=S("ArrayPE", data={"rng": 43, "n": 2000, "ch": 1, "scale": 0.5}),
          ir=S("ArrayPE", data=room), mix=S("MixPE", inputs=[S("ConstantPE", value=0.5),
                                                              S("SinePE", frequency=3.0, amplitude=0.4)]),
          fft_size=1024), blocks_contig(0, [1000, 1400]))

    # ---- semantics a random-graph differential test tripped over (tests/test_gpu_fuzz.py), pinned by the reference
    add("reverb_held_source_negative_blocks", 10000,       # MixPE inside ReverbPE skips blocks that miss both extents
        S("ReverbPE", source=S("ArrayPE", data={"rng": 44, "n": 1500, "ch": 1, "scale": 0.5}, extend_mode="hold_both"),
          ir=S("ArrayPE", data=room), mix=0.4, fft_size=1024), [[-445, 64], [-381, 1000], [619, 1200]])
    add("trigger_restart_comb_keeps_state", 22050,         # CombPE has no _reset_state hook: restarts do not clear it
        S("TriggerRestartPE", trigger=S("PeriodicTrigger", hz=60.0),
          src=S("CombPE", source=S("SinePE", frequency=300.0, amplitude=0.5), frequency=1500.0, feedback=0.8)),
        blocks_contig(-58, [1024, 17, 900]))
    add("trigger_restart_fm_sine_keeps_phase", 22050,      # nor has SinePE: its phase runs on across restarts
        S("TriggerRestartPE", trigger=S("PeriodicTrigger", hz=45.0),
          src=S("SinePE", frequency=S("MixPE", inputs=[S("ConstantPE", value=400.0),
                                                        S("SinePE", frequency=3.0, amplitude=100.0)]))),
        blocks_contig(0, [1024, 1024]))
    add("trigger_restart_reverb_memo", 22050,              # ReverbPE's CachePE memo survives restarts (equal segments!)
        S("TriggerRestartPE", trigger=S("PeriodicTrigger", hz=63.0),
          src=S("ReverbPE", source=S("BlitSawPE", frequency=500.0), ir=S("ArrayPE", data={"values": [0.5, 0.3, -0.2]}),
                mix=0.6)), [[-244, 3000], [2756, 700]])
    add("mix_transform_of_bounded_source", 44100,          # TransformPE passes its source's extent on to MixPE's skip rule
        S("MixPE", inputs=[S("TransformPE", source=S("ArrayPE", data={"rng": 45, "n": 900, "ch": 1, "scale": 0.5}),
                             ops=[["abs"], ["affine", 0.8, 0.1], ["sqrt"]]),
                           S("ConstantPE", value=0.25)]), [[-182, 17], [-165, 64], [-101, 257], [800, 300], [1100, 50]])

    # ---------------------------------------------------------------- SpatialPE
    def chans(c, seed):
        return S("ArrayPE", data={"rng": seed, "n": 1500, "ch": c, "scale": 0.5})
    for src_c, out_c in ((1, 2), (2, 1), (2, 4), (4, 2), (3, 5), (5, 2), (1, 4), (2, 2)):
        add(f"spatial_adapter_{src_c}_to_{out_c}", 44100,
            S("SpatialPE", source=chans(src_c, 50 + src_c), method="adapter", channels=out_c), [[-10, 800], [790, 800]])
    sweep = S("SinePE", frequency=0.7, amplitude=120.0)           # beyond +-90: exercises the clip
    add("spatial_linear_scalar", 44100, S("SpatialPE", source=chans(2, 60), method="linear", azimuth=-35.0),
        [[0, 1500]])
    add("spatial_linear_pe", 44100, S("SpatialPE", source=chans(1, 61), method="linear", azimuth=sweep),
        blocks_contig(0, [1000, 500]))
    add("spatial_constant_power_scalar", 44100,
        S("SpatialPE", source=chans(2, 62), method="constant_power", azimuth=50.0), [[0, 1500]])
    add("spatial_constant_power_pe", 44100,
        S("SpatialPE", source=chans(3, 63), method="constant_power", azimuth=sweep), blocks_contig(0, [1000, 500]))
    add("spatial_hrtf_right", 44100, S("SpatialPE", source=chans(1, 64), method="hrtf", azimuth=45.0),
        blocks_contig(0, [700, 100, 17, 683]))
    add("spatial_hrtf_left_stereo_src", 44100,
        S("SpatialPE", source=chans(2, 65), method="hrtf", azimuth=-30.0, elevation=2.0),
        [[0, 600], [600, 600], [100, 300]])                       # last block not contiguous: tail cleared

    # ---------------------------------------------------------------- Convolve
    add("conv_kat", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"values": [1.0, 2.0, 3.0, 4.0]}),
          fir=S("ArrayPE", data={"values": [1.0, 0.5, -1.0]}), fft_size=16),
        [[0, 6]])
    add("conv_identity", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 0, "n": 256, "ch": 1, "scale": 1.0}),
          fir=S("ArrayPE", data={"values": [1.0]}), fft_size=64),
        [[0, 256]])
    add("conv_stereo_src_mono_fir", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 1, "n": 200, "ch": 2, "scale": 1.0}),
          fir=S("ArrayPE", data={"rng": 2, "n": 33, "ch": 1, "scale": 1.0}), fft_size=128),
        [[0, 232]])
    add("conv_mono_src_stereo_fir", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 3, "n": 200, "ch": 1, "scale": 1.0}),
          fir=S("ArrayPE", data={"rng": 4, "n": 33, "ch": 2, "scale": 1.0}), fft_size=128),
        [[0, 232]])
    add("conv_stereo_stereo_chunked", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 5, "n": 200, "ch": 2, "scale": 1.0}),
          fir=S("ArrayPE", data={"rng": 6, "n": 33, "ch": 2, "scale": 1.0}), fft_size=64),
        blocks_contig(0, ODD + [32]))
    add("conv_default_fft", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 7, "n": 3000, "ch": 1, "scale": 1.0}),
          fir=S("ArrayPE", data={"rng": 8, "n": 129, "ch": 1, "scale": 1.0})),
        blocks_contig(0, [1000, 2128]))
    add("conv_gap_clears_history", 10000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 9, "n": 600, "ch": 2, "scale": 1.0}),
          fir=S("ArrayPE", data={"rng": 10, "n": 65, "ch": 1, "scale": 1.0}), fft_size=256),
        [[0, 100], [100, 100], [300, 100], [400, 264], [-50, 120]])
    add("c3_reduced", 48000,
        S("ConvolvePE", src=S("ArrayPE", data={"rng": 0, "n": 12000, "ch": 2, "scale": 0.1}),
          fir=S("ArrayPE", data={"rng": 1, "n": 4096, "ch": 1, "scale": 1.0, "decay": 500.0}),
          fft_size=8192),
        blocks_contig(0, [4097, 4097, 3806, 4095]))
    return C
