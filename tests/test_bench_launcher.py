"""CPU: `bench.py --gpus N` launches its own ranks.  With N > 1 and no WORLD_SIZE the parent must start
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD before anything touches a GPU (the driver
runs `python3 bench.py --gpus 8`; the reference does `mpirun -np {num_gpus}`, examples/benchmark/run_one.py:48), relay
the JSON line and pass the exit code on.  In this container there is no GPU, so the ranks must get as far as the
device check and fail THERE — each rank with the no-device message — and the parent must report failure."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)


def test_parent_launches_two_ranks_that_reach_the_device_check():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the launcher is exercised for real by the multi-rank GPU tests")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"ANI_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""                       # no JSON line from a failed run
    # the ranks were started and got to the device check: both normally, but the launcher ends the second rank as soon as
    # the first has failed, so on a slow start only one message may make it out
    assert 1 <= r.stderr.count("no HIP device visible") <= 2
    assert "SystemExit: --gpus" not in r.stderr and "must be launched with" not in r.stderr


def test_relay_prints_only_the_json_line_and_passes_the_exit_code(tmp_path):
    """The relay logic alone, with a stand-in for torch.distributed.run: a fake `torch.distributed.run` module earlier on
    PYTHONPATH that prints noise plus a JSON line and exits with a chosen code."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import sys, json\n"
        "print('banner noise')\n"
        "print(json.dumps({'metric': 'x', 'argv': sys.argv[1:]}))\n"
        "sys.exit(int(__import__('os').environ.get('FAKE_RC', '0')))\n")
    for rc in (0, 3):
        r = _run(["--gpus", "4", "--steps", "7"], {"PYTHONPATH": str(tmp_path), "FAKE_RC": str(rc)})
        assert r.returncode == rc
        lines = r.stdout.splitlines()
        assert len(lines) == 1
        d = json.loads(lines[0])
        argv = d["argv"]
        assert "--nproc-per-node=4" in argv and "--nnodes=1" in argv
        assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
        assert argv[-4:] == ["--gpus", "4", "--steps", "7"] and argv[-5].endswith("bench.py")


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_bench_on_one_card_takes_the_fallback_transport():
    """First-contact rehearsal of `bench.py --gpus N` on the one card a GPU box has: the parent launches two ranks (before it
    touches the GPU), they meet over gloo (RCCL refuses two ranks on one device), try the native communicator, find RCCL
    disabled, agree on that, and fall back to torch.distributed for the ghost exchange -- the sequence a node with a broken
    librccl would take.  The JSON line must keep the driver's contract for N ranks."""
    r = _run(["--gpus", "2", "--atoms", "6000", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-dense-pass", "--no-extra"],
             {"ANI_BENCH_BACKEND": "gloo", "ANI_BENCH_NATIVE_COMM": "try", "ANI_COMM_DISABLE_RCCL": "1",
              "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 12 and d["warmup"] == 3
    assert d["scaling"] == "strong" and d["unit"] == "ns/day" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["grid"] == [2, 1, 1] and d["config"]["backend"] == "gloo"
    assert d["config"]["nlocal_rank0"] < 6000 and d["config"]["nghost_rank0"] > 0      # rank 0 holds a brick, not the box
    ex = d["md_loop"]["exchange"]
    assert "ani_comm unavailable" in ex and "ANI_COMM_DISABLE_RCCL" in ex and "torch.distributed all_to_all_single (gloo)" in ex
    assert d["md_loop"]["energy_finite"] and d["md_loop"]["steps"] == 12
    assert abs(d["value"] - 0.0432 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in d["roofline"], k
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_committed_counter_summary_belongs_to_the_committed_kernel_sources():
    """bench.py fills `roofline.traffic` from profiles/<PMC_SUMMARY> only while the sha256 over lammps-ani_amd/csrc/* still is the
    one the counters were collected on (tools/profile_round.sh): a source change without a new counter run would silently turn
    the driver's bench line back to `traffic: null`.  CPU check of exactly that equality."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    path = os.path.join(root, "profiles", bench.PMC_SUMMARY)
    assert os.path.exists(path), f"profiles/{bench.PMC_SUMMARY} is missing"
    summary = json.load(open(path))
    assert summary["source_digest"] == bench.source_digest(), \
        "lammps-ani_amd/csrc changed after the counters were collected: run tools/r4_evidence.sh on a GPU box and commit profiles/"
    for kernel in ("ani::aev_backward_fast", "ani::aev_forward_fused", "ani::mlp_fused16"):
        assert any(k.startswith(kernel) for k in summary["kernels"]), kernel
