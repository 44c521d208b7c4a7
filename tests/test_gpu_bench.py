"""bench.py end to end on the GPU box: the single-process line and a 2-rank rehearsal of both multi-GPU modes
(gloo backend, both ranks on the one GPU here; the driver's scaling runs use RCCL with one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--N", "4096", "--M", "64", "--P", "2", "--partials", "3", "--steps", "2", "--warmup", "1", "--no-cpu"]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["metric"] == "ELBO-steps/sec" and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["value"] > 0
    assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"]
    # roofline.frac is the dominant kernel's own flops over its own duration: reproducible with one division
    r = d["roofline"]
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-9 * r["achieved"]
    assert abs(d["mfma_frac_step"]["achieved"]
               - d["mfma_frac_step"]["algorithmic_flops_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12) < 1e-6
    assert d["config"]["workload"].startswith("pdgp ELBO step")


@pytest.mark.parametrize("shard,scaling", [("window", "weak"), ("pitch", "strong"), ("gp", "strong")])
def test_bench_two_rank_rehearsal(shard, scaling):
    port = 29700 + (os.getpid() % 200) + {"window": 0, "pitch": 1, "gp": 2}[shard]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--shard", shard] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["parallelism"].startswith({"window": "window-per-gpu", "pitch": "pitch-sharded", "gp": "gp-sharded"}[shard])


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's form of the command) must start two ranks
    and say so; the extra pitch-sharded strong-scaling line rides on the same JSON."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + SMALL,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak" and d["value"] > 0
    assert d["pitch_sharded"]["scaling"] == "strong" and d["pitch_sharded"]["value"] > 0
    assert d["gp_sharded"]["scaling"] == "strong" and d["gp_sharded"]["value"] > 0 and d["gp_sharded"]["ceiling_x"] == 2.0


@pytest.mark.parametrize("shard", ["window", "pitch", "gp"])
def test_bench_one_rank_through_rccl(shard):
    """the launcher form with ONE rank and the nccl backend: RCCL communicator set-up and the per-step all-reduce (the
    device-resident scalar ELBO in window mode; the 3N+1 exchange vector between gp_pdgp_elbo_begin / _end in pitch mode,
    ordered against the library's streams) run on this one-GPU box; the 2 / 4 / 8-rank runs are the driver's"""
    port = 29950 + (os.getpid() % 40) + {"window": 0, "pitch": 1, "gp": 2}[shard]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--shard", shard] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["backend"] == "nccl" and d["rccl_ranks"] == 1 and d["value"] > 0


def test_one_call_sharded_steps_match_the_two_stage_path():
    """with an RCCL communicator (here: one rank, no torch.distributed group needed — rank 0 draws the
    unique id itself) a sharded evaluation is ONE library call that issues the collective between its two stages on the
    handle's stream (gp_pdgp_elbo_pitch_sharded / gp_pdgp_elbo_gp_sharded / gp_sgpr_bound_grad_sharded), with the Adam step
    behind it when asked.  Against the two-stage path with the exchange done by the caller, on the same one-rank models:
    ELBO / gradient / three Adam steps bit for bit (the same kernels in the same order; a one-rank sum changes nothing)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_one_call_child.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-2500:])
    assert "ONE-CALL OK" in r.stdout, r.stdout[-2500:]
