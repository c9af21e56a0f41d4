"""The entry points start their own ranks (sar_ssl_amd/launch.py): `python bench.py --gpus N` and `python run_pretrain.py --gpu-id a,b,..`
are ONE command like the reference's multi-GPU form (code/run_pretrain.py:204-205 -> code/learner.py:25-31), the parent never touches
the GPU, rank 0's single line is the command's single line, a dead rank fails the command, and a `--gpus N` that does not match the
process group is refused instead of printing an n_gpus line for another world.  CPU only (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa: E402,F401


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "SARSSL_DIST_FORCE", "SARSSL_SELF_LAUNCHED")}
    env.update(extra)
    return env


def test_bench_gpus_2_starts_its_own_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-probe"], capture_output=True, text=True,
                       timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["world"] == 2 and out["dist"]["backend"] == "gloo" and out["dist"]["rank_sum"] == 1.0
    assert out["self_launched"] is True


def test_bench_fails_when_a_rank_dies():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-probe"], capture_output=True, text=True,
                       timeout=300, env=_clean_env(SARSSL_PROBE_FAIL_RANK="1"))
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])


def test_bench_refuses_a_world_that_is_not_the_one_asked_for():
    """`--gpus 8` inside a one-rank environment (a launcher that started one process) must not print an n_gpus: 1 line."""
    env = _clean_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-probe"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and not r.stdout.strip() and "--gpus 8" in r.stderr


def test_rank_environments_and_gpu_id_parsing():
    from sar_ssl_amd import launch
    assert launch.parse_gpu_ids("0,") == ["0"] and launch.parse_gpu_ids("0,1,2,3") == ["0", "1", "2", "3"] and launch.parse_gpu_ids("7") == ["7"]
    envs = launch.rank_envs(3, gpu_ids=["4", "6", "7"], base={"X": "1"}, port=1234)
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["HIP_VISIBLE_DEVICES"] == "4,6,7" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "1234"
               and e["X"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    shared = launch.rank_envs(2, gpu_ids=["0", "0"], base={}, port=1)          # two ranks on one GPU (functional tests over gloo)
    assert [e["LOCAL_RANK"] for e in shared] == ["0", "0"] and shared[0]["HIP_VISIBLE_DEVICES"] == "0"
    plain = launch.rank_envs(2, base={}, port=1)
    assert "HIP_VISIBLE_DEVICES" not in plain[0] and [e["LOCAL_RANK"] for e in plain] == ["0", "1"]


def test_run_pretrain_with_several_gpu_ids_spawns_one_rank_per_id(monkeypatch):
    """`python run_pretrain.py --pretrain --simu-exp --gpu-id 0,1,2` (no launcher in the environment): three ranks, started before
    torch / the GPU is touched; with one id, or under a launcher, nothing is spawned."""
    from sar_ssl_amd import launch, run_pretrain
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    calls = []
    monkeypatch.setattr(launch, "spawn_ranks", lambda script, argv, nproc, gpu_ids=None: (calls.append((script, list(argv), nproc, gpu_ids)), 0)[1])
    argv = ["--pretrain", "--simu-exp", "--gpu-id", "0,1,2", "--work-dir", "/tmp/x"]
    try:
        run_pretrain.main(argv)
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert e.code == 0
    assert len(calls) == 1 and calls[0][0].endswith("run_pretrain.py") and calls[0][1] == argv and calls[0][2] == 3 and calls[0][3] == ["0", "1", "2"]
    # under a launcher (torchrun or our own children) the same command line does not spawn again: it reaches the GPU check
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "3"); monkeypatch.setenv("LOCAL_RANK", "0")
    try:
        run_pretrain.main(argv + ["--no-cuda"])
    except SystemExit as e:
        assert "needs an MI355X GPU" in str(e.code)
    assert len(calls) == 1


def test_sigterm_to_the_parent_stops_every_rank(tmp_path):
    """Advisor (round 5): a SIGTERM / SIGHUP to the parent (scheduler kill, timeout(1)) left the ranks running - holding GPUs, possibly
    inside a collective.  The parent now forwards the signal and reaps its children whatever ends it."""
    import signal
    import time
    child = tmp_path / "rank.py"
    child.write_text("import os, sys, time\nopen(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n" % str(tmp_path))
    parent = tmp_path / "parent.py"
    parent.write_text("import sys\nsys.path.insert(0, %r)\nimport sarssl_boot\nfrom sar_ssl_amd import launch\nsys.exit(launch.spawn_ranks(%r, [], 2, grace_s=5.0))\n"
                      % (ROOT, str(child)))
    p = subprocess.Popen([sys.executable, str(parent)], env=_clean_env(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / ("pid%d" % r)).exists() and (tmp_path / ("pid%d" % r)).read_text() for r in (0, 1)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc == 143, rc
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive, "rank process %d survived its parent" % pid
