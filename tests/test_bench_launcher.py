"""bench.py as its own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts the N ranks
as a CHILD (python -m torch.distributed.run ...) before anything touches the GPU, relays the JSON line and the exit code.
CPU-only: argv / env construction and the relay are checked with a stand-in child."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (importing bench.py must not import torch)


def test_importing_bench_does_not_import_torch():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print('torch' in sys.modules)" % ROOT],
                       capture_output=True, text=True, check=True)
    assert r.stdout.strip() == "False"


def test_launcher_command_argv_and_env(monkeypatch):
    monkeypatch.setenv("RANK", "3")            # leftovers of an outer launcher must not leak into the child launcher
    monkeypatch.setenv("WORLD_SIZE", "9")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("OI_BENCH_BACKEND", "gloo")     # rehearsal switches travel to the ranks
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    cmd, env = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29517, python="/usr/bin/python3")
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]          # the user's arguments, unchanged
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert env["OI_BENCH_BACKEND"] == "gloo"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert k not in env


def test_free_port_is_bindable():
    import socket
    p = bench.free_port()
    assert 1024 < p < 65536
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))


def test_self_launch_relays_output_and_exit_code(monkeypatch, capfd):
    child = "import sys; print('{\"metric\": \"x\", \"n_gpus\": 2}'); sys.stdout.flush(); sys.exit(7)"
    monkeypatch.setattr(bench, "launcher_command", lambda argv, n, port, python=None: ([sys.executable, "-c", child], dict(os.environ)))
    rc = bench.self_launch(["--gpus", "2"], 2)
    assert rc == 7
    assert '"n_gpus": 2' in capfd.readouterr().out


def test_parent_of_gpus_2_launches_before_touching_torch():
    # the real main(): --gpus 2, no WORLD_SIZE -> the child is started (here a stand-in) and main exits with ITS code,
    # without torch ever being imported in the parent
    code = (
        "import sys, os; sys.path.insert(0, %r); os.environ.pop('WORLD_SIZE', None)\n"
        "import bench\n"
        "bench.launcher_command = lambda argv, n, port, python=None: ([sys.executable, '-c', 'print(\"child\", %%r)' %% (argv,)], dict(os.environ))\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '5']\n"
        "try:\n    bench.main()\nexcept SystemExit as e:\n    print('exit', e.code, 'torch' in sys.modules)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert "child ['--gpus', '2', '--steps', '5']" in r.stdout
    assert "exit 0 False" in r.stdout


def test_a_rank_does_not_launch_again(monkeypatch):
    # under torch.distributed.run WORLD_SIZE is set: main() must go on as a rank (here it fails later, on the missing GPU,
    # NOT in the launcher)
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda *a: called.append(a) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        bench.main()
    except BaseException:
        pass
    assert called == []


def test_other_configs_keeps_the_childs_headline_fields_and_never_raises(monkeypatch):
    """bench.py's `other_configs` block (round 5): configs[1] and one configs[4] shard run as CHILD processes of the default N = 1
    run.  With stand-in children: the headline fields are carried over, a failing child leaves {"error": ...}, and nothing is
    started at all under rocprofv3 (the children would be profiled into the run's statistics)."""
    import json
    import subprocess as sp
    for k in [k for k in os.environ if k.startswith("ROCPROF")] + ["LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH"]:
        monkeypatch.delenv(k, raising=False)
    calls = []

    def fake_run(cmd, capture_output=True, text=True, timeout=None):
        calls.append(cmd)
        assert "--no-other-configs" in cmd and "--no-cpu-baseline" in cmd and cmd[1].endswith("bench.py")
        if "--corpus" in cmd:            # the configs[4] shard: a child that dies
            return sp.CompletedProcess(cmd, 3, stdout="", stderr="boom")
        line = {"value": 2900.0, "ms_per_step": 0.34, "p50_ms": 0.33, "p95_ms": 0.35, "steps": 300, "dtype": "f32",
                "roofline": {"bound": "hbm", "achieved": 6100.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.76, "kernel": "k", "x": 1},
                "f32_stream_scorer": {"ms_per_step": 0.55, "queries_per_s": 1800.0, "hbm_frac_on_its_bytes": 0.8, "note": "n"}}
        return sp.CompletedProcess(cmd, 0, stdout="noise\n" + json.dumps(line) + "\n", stderr="")

    monkeypatch.setattr(sp, "run", fake_run)
    out = bench.other_configs()
    assert len(calls) == 2
    c1 = out["config1_single_query"]
    assert c1["queries_per_s"] == 2900.0 and c1["ms_per_step"] == 0.34 and c1["roofline"]["frac"] == 0.76 and "x" not in c1["roofline"]
    assert c1["without_screening_copy"] == {"ms_per_step": 0.55, "queries_per_s": 1800.0, "hbm_frac_on_its_bytes": 0.8}
    assert "rc 3" in out["config4_one_shard"]["error"] and "boom" in out["config4_one_shard"]["error"]
    json.dumps(out)
    monkeypatch.setenv("ROCPROFILER_LIBRARY_PATH", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    calls.clear()
    assert "skipped" in bench.other_configs() and not calls
