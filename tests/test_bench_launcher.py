"""bench.py's rank launcher (-m "not gpu"): `python bench.py --gpus N` must start N ranks by itself,
refuse to claim more GPUs than the box has, and keep stdout to rank 0's one JSON line.  The
torchrun command it builds is run for real here on a stand-in script (gloo, two ranks) -- the
benchmark itself needs a GPU and is covered by tests/test_gpu_fullsize.py."""
import io
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parse(*argv):
    return bench.build_parser().parse_args(list(argv))


def test_one_rank_or_existing_ranks_run_in_process():
    assert bench.plan_launch(parse(), [], {}, 1) is None
    assert bench.plan_launch(parse("--gpus", "1"), ["--gpus", "1"], {}, 8) is None
    # under torchrun (the driver's multi-GPU form) the environment's ranks are used as they are
    env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}
    assert bench.plan_launch(parse("--gpus", "8"), ["--gpus", "8"], env, 8) is None


def test_several_gpus_without_ranks_builds_the_torchrun_command():
    argv = ["--gpus", "4", "--steps", "8", "--warmup", "2"]
    cmd = bench.plan_launch(parse(*argv), argv, {}, 8, port=29611, python="py", script="/x/bench.py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                   "--master-addr", "127.0.0.1", "--master-port", "29611", "/x/bench.py"] + argv
    # no port given: a free one is picked
    cmd = bench.plan_launch(parse(*argv), argv, {}, 4)
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert bench.plan_launch(parse("--gpus", "2", "--master-port", "4242"), [], {}, 2)[9] == "4242"


def test_more_ranks_than_devices_is_refused_not_measured():
    with pytest.raises(bench.LaunchError, match="8 ranks asked for, 1 HIP device"):
        bench.plan_launch(parse("--gpus", "8"), ["--gpus", "8"], {}, 1)
    # ... also when torchrun made the ranks (the check each rank runs)
    with pytest.raises(bench.LaunchError):
        bench.check_rank_request(parse("--gpus", "8"), 8, 4)
    bench.check_rank_request(parse("--gpus", "8"), 8, 8)
    with pytest.raises(bench.LaunchError, match="no HIP device"):
        bench.check_rank_request(parse(), 1, 0)


def test_rehearsal_needs_gloo_and_lifts_the_device_check():
    with pytest.raises(bench.LaunchError, match="needs --dist-backend gloo"):
        bench.plan_launch(parse("--gpus", "2", "--share-device"), [], {}, 1)
    argv = ["--gpus", "8", "--dist-backend", "gloo", "--share-device"]
    cmd = bench.plan_launch(parse(*argv), argv, {}, 1, port=1)
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[-len(argv):] == argv


STAND_IN = textwrap.dedent('''
    import json, os, sys
    import torch, torch.distributed as dist
    print("noise on stdout from rank %s" % os.environ["RANK"])
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"metric": "stand-in", "n_gpus": int(t.item()), "argv": sys.argv[1:]}), flush=True)
    dist.destroy_process_group()
    sys.exit(int(os.environ.get("STAND_IN_RC", "0")))
''')


def test_self_launch_runs_the_ranks_and_passes_one_line_through(tmp_path, capfd):
    script = tmp_path / "stand_in.py"
    script.write_text(STAND_IN)
    argv = ["--gpus", "2", "--steps", "3"]
    cmd = bench.plan_launch(parse(*argv), argv, {}, 2, script=str(script))
    out = io.StringIO()
    rc = bench.self_launch(cmd, out)
    assert rc == 0
    lines = out.getvalue().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["argv"] == argv
    # what else the ranks wrote on stdout went to stderr
    assert "noise on stdout from rank 1" in capfd.readouterr().err


def test_self_launch_reports_failing_ranks(tmp_path):
    script = tmp_path / "stand_in.py"
    script.write_text(STAND_IN)
    cmd = bench.plan_launch(parse("--gpus", "2"), [], {}, 2, script=str(script))
    os.environ["STAND_IN_RC"] = "3"
    try:
        assert bench.self_launch(cmd, io.StringIO()) != 0
    finally:
        del os.environ["STAND_IN_RC"]
    # ranks that exit 0 without a line are a failure too
    quiet = tmp_path / "quiet.py"
    quiet.write_text("print('nothing useful')\n")
    cmd = bench.plan_launch(parse("--gpus", "2"), [], {}, 2, script=str(quiet))
    assert bench.self_launch(cmd, io.StringIO()) == 1


def test_scans_per_launch_and_the_block_layout():
    """Config C5 (round 6): a rank's hops of k consecutive scans share one persistent launch.  k divides the scans
    of a reduce block and k x (the most shards any rank owns) fits the kernel's hop table (16); the block's rows are
    ordered so that whatever ONE launch writes -- a rank's consecutive hops x the k scans of a batch, or one hop x the
    scans of a region's ragged tail -- is a run of consecutive rows (rpf_device_reduce writes n_hops x N contiguously)."""
    import rtl_power_fftw_amd as rpf
    hops, frames, per_block, table = 8, 5000, 4, 16
    for world, want in ((1, 2), (2, 4), (3, 4), (4, 4), (8, 4)):
        shards = [rpf.sharding.shard_hops(hops, frames, world, r) for r in range(world)]
        k = bench.scans_per_launch(per_block, max(len(m) for m in shards), table)
        assert k == want, (world, k)
        assert bench.scans_per_launch(per_block, max(len(m) for m in shards), table, asked=1) == 1
        # every (hop, scan) has a row of its own
        rows = sorted(bench.block_row(h, sub, hops, k) for h in range(hops) for sub in range(per_block))
        assert rows == list(range(hops * per_block))
        for mine in shards:
            hop_ids = [m[0] for m in mine]
            assert hop_ids == list(range(hop_ids[0], hop_ids[0] + len(mine)))      # hop-major: consecutive hops
            for first in range(0, per_block, k):
                # a whole batch: hops outer, scans inner -- the order bench.py lists them in
                got = [bench.block_row(h, first + j, hops, k) for h in hop_ids for j in range(k)]
                assert got == list(range(got[0], got[0] + len(got)))
                # a ragged tail of n < k scans goes hop by hop
                for n in range(1, k):
                    for h in hop_ids:
                        got = [bench.block_row(h, first + j, hops, k) for j in range(n)]
                        assert got == list(range(got[0], got[0] + n))
    assert bench.scans_per_launch(1, 1, 16) == 1 and bench.scans_per_launch(6, 2, 16) == 6
    assert bench.scans_per_launch(4, 20, 16) == 1          # (never fits: a launch per scan, hop by hop in the engine)
