"""CPU: `bench.py --gpus N` creates N ranks by itself (no launcher), every rank joins the rendezvous, the
shards cover the workload, and the parent fails when a rank does.  --dry-run stops short of the device."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("n,workload,voices", [(2, "supersaw", 512), (3, "c5", 512), (2, "c4", 64)])
def test_gpus_n_spawns_n_ranks(n, workload, voices):
    p = _run(["--gpus", str(n), "--workload", workload, "--dry-run"])
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # exactly one JSON line: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["n_ranks_seen"] == n
    assert d["config"]["voices_on_all_ranks"] == voices
    assert d["config"]["voices_on_rank0"] == len(range(0, voices, n))


def test_single_rank_needs_no_rendezvous():
    p = _run(["--dry-run", "--workload", "c5"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1


def test_parent_reports_a_failed_rank():
    # without --dry-run the ranks need the device: here (no GPU) every rank fails loudly, and so must the parent
    from pygmu2_amd import device
    if device.device_available():
        pytest.skip("needs a host without a GPU")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"], timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_two_ranks_without_a_communicator_fail_unless_told_otherwise():
    """Two real ranks.  On a box with one GPU both land on device 0 and RCCL refuses the communicator: the run must
    FAIL (rank 0's line says n_ranks_seen 0 and why) -- a multi-GPU number without RCCL is not a measurement.  With
    --allow-no-rccl the ranks agree to carry on with store barriers, measure the replica workload, report
    n_ranks_seen = 0 / collective "store" and say that the sharded mixes were not run.  On a box with two GPUs the
    communicator exists and both runs are ordinary.  Either way stdout is exactly one JSON line (RCCL's greeting
    banner goes to stderr)."""
    args = ["--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu"]
    p = _run(args, env={"PGX_BENCH_RCCL_TIMEOUT": "60"}, timeout=400)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    if p.returncode != 0:
        assert d["n_ranks_seen"] == 0 and d["value"] is None and "RCCL" in d["error"]
    else:
        assert d["collective"] == "rccl" and d["n_ranks_seen"] == 2
        assert d["voice_mix"]["n_ranks"] == 2 and d["voice_mix"]["value"] > 0
    p = _run(args + ["--allow-no-rccl"], env={"PGX_BENCH_RCCL_TIMEOUT": "60"}, timeout=400)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["collective"] in ("rccl", "store")
    if d["collective"] == "store":
        assert d["n_ranks_seen"] == 0 and d["n_ranks_present"] == 2
        assert "error" in d["voice_mix"] and "error" in d["supersaw_mix"]
    else:
        assert d["n_ranks_seen"] == 2 and d["voice_mix"]["n_ranks"] == 2 and d["voice_mix"]["value"] > 0


def test_dry_run_names_the_flat_keys_of_a_measured_line():
    """the driver's record keeps only scalar `config` entries.  The dry run lists the keys a measured
    line carries per sharded mix; flat_keys() makes them from a line's nested objects."""
    p = _run(["--gpus", "2", "--workload", "supersaw", "--dry-run"])
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    keys = d["config"]["flat_mix_keys"].split(",")
    for want in ("supersaw_msamples_s", "supersaw_ms_per_block", "c5_allreduce_wait_ms", "c5_collectives_in_timed_region",
                 "c4_render_ms_max_over_ranks", "c4_sharded_vs_unsharded_max_err_over_peak", "c5_msamples_s",
                 "supersaw_agreement_checks"):
        assert want in keys, want


def test_flat_keys_are_scalars_and_cover_the_north_star_numbers():
    sys.path.insert(0, ROOT)
    import bench
    mix = {"value": 100.0, "ms_per_block": 0.5, "render_ms": 0.4, "render_ms_max_over_ranks": 0.45,
           "allreduce_wait_ms": 0.05, "collectives_in_timed_region": 6, "sharded_vs_unsharded_max_err_over_peak": 3e-8,
           "agreement_checks": 9, "voices_on_this_rank": 64, "oscillator_msamples_s": 7.0,
           "cpu_baseline": {"value": 0.006}, "roofline": {"frac": 1e-4}, "roofline_fp64": {"frac": 0.5}}
    result = {"value": 1.0, "supersaw_mix": dict(mix), "voice_mix": dict(mix), "ladder_mix": dict(mix),
              "value_with_d2h": {"pipelined": {"value": 10000.0, "over_cpu": 150.0}, "sync": {"value": 8000.0}},
              "cases": {"c3_convolve_64k_taps": {"value": 5000.0, "cpu_oracle_msamples_s": 22.0,
                                                 "roofline": {"frac": 0.01, "avg_launch_ms": 0.0156, "traffic": 29e6,
                                                              "algorithmic_bytes_per_launch": 1.536e6}},
                        "c4_res06_supersaw_ladder_mix_64": dict(mix)},
              "north_star_pes": {"rows": {"LadderPE (lp24, 800 Hz, res 0.6, drive 1.5)":
                                          {"sync": 400.0, "pipelined": 460.0, "cpu": 8.0, "pipelined_over_cpu": 57.0}}},
              "roofline_fp64": {"frac": 0.6}}
    flat = bench.flat_keys(result)
    assert all(isinstance(v, (int, float)) or v is None for v in flat.values()), flat
    for key in ("supersaw_msamples_s", "supersaw_ms_per_block", "c5_allreduce_wait_ms", "c4_collectives_in_timed_region",
                "c4_sharded_vs_unsharded_max_err_over_peak", "c2_with_d2h_pipelined", "c3_96000_call_us",
                "c3_96000_traffic_over_algorithmic", "ladder_res06_pipelined", "ladder_res06_cpu", "c4_res06_msamples_s",
                "c2_roofline_fp64_frac", "supersaw_roofline_fp64_frac", "c5_cpu_msamples_s"):
        assert key in flat, key
