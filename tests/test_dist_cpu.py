"""N > 1 path on CPU: world_size-2 `gloo` run of the sharded registration.

The source cloud is sharded over ranks (lidar_slam_amd.dist.shard_bounds), every rank holds the full target, the
32-double normal-equation block is all-reduced once per pass, and each rank runs the product's host Newton /
More-Thuente driver (lsa_gn_solve) on the reduced block.  Without a GPU the per-shard pass sums come from the
oracle (the checker); what is under test is the sharding, the collective wiring and that all ranks take the
identical solve.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tgt, src, guess, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401  (registers the package)
    import torch.distributed as dist
    import oracle_py
    from lidar_slam_amd import dist as lsa_dist
    from lidar_slam_amd import registration

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ndt = oracle_py.OracleNDT()
    ndt.set_input_target(tgt)
    lo, hi = lsa_dist.shard_bounds(src.shape[0], rank, world)
    shard = src[lo:hi]
    n_calls = [0]

    def evaluator(T, p, kind):
        n_calls[0] += 1
        if kind == 2:
            ndt.derivatives(shard, T, p, True)
            out = registration.pack_out32(0.0, np.zeros(6), ndt.hessian(shard, T, p), 0, shard.shape[0])
        else:
            s, g, H, pairs = ndt.derivatives(shard, T, p, compute_hessian=(kind == 0))
            out = registration.pack_out32(s, g, H, pairs, shard.shape[0])
        return lsa_dist.allreduce_out32_cpu(out)  # ONE collective per pass

    T, st = registration.gn_solve(evaluator, 1.0, 0.1, 0.01, 30, 0.55, guess)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=T, p=st["final_p"], it=st["iterations"], passes=st["passes"],
             calls=n_calls[0], tp=st["trans_probability"], lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_registration_two_ranks_gloo(scan_pair, oracle, tmp_path):
    import torch.multiprocessing as mp
    import oracle_py
    from lidar_slam_amd import dist as lsa_dist
    from lidar_slam_amd import synth

    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    guess = np.eye(4, dtype=np.float32)

    # shards are contiguous, disjoint and cover the cloud
    b = [lsa_dist.shard_bounds(src.shape[0], r, 2) for r in range(2)]
    assert b[0][0] == 0 and b[0][1] == b[1][0] and b[1][1] == src.shape[0]
    b8 = [lsa_dist.shard_bounds(1001, r, 8) for r in range(8)]
    assert b8[0][0] == 0 and b8[-1][1] == 1001 and all(b8[i][1] == b8[i + 1][0] for i in range(7))
    assert max(h - l for l, h in b8) - min(h - l for l, h in b8) <= 1

    ref = oracle_py.OracleNDT()
    ref.set_input_target(tgt)
    T_ref, st_ref = ref.scan_match(src, guess)

    port = _free_port()
    mp.spawn(_worker, args=(2, port, tgt, src, guess, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank takes the identical solve: bit-identical poses, same iteration / pass counts, one collective per pass
    assert np.array_equal(r0["T"], r1["T"]) and np.array_equal(r0["p"], r1["p"])
    assert int(r0["it"]) == int(r1["it"]) == st_ref["iterations"]
    assert int(r0["passes"]) == int(r1["passes"]) == st_ref["passes"] == int(r0["calls"]) == int(r1["calls"])
    # and the sharded sum reproduces the single-process registration (only the summation grouping differs)
    assert np.max(np.abs(r0["T"] - T_ref)) <= 1e-6
    assert np.max(np.abs(r0["p"] - st_ref["final_p"])) <= 1e-8
    assert abs(float(r0["tp"]) - st_ref["trans_probability"]) <= 1e-9 * abs(st_ref["trans_probability"])


def _icp_worker(rank, world, port, tgt, src, guess, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    import torch.distributed as dist
    import oracle_py
    from lidar_slam_amd import dist as lsa_dist
    from lidar_slam_amd import registration

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = lsa_dist.shard_bounds(src.shape[0], rank, world)
    shard = src[lo:hi]
    n_calls = [0]

    def evaluator(T16):
        n_calls[0] += 1
        return lsa_dist.allreduce_out32_cpu(oracle_py.icp_sums(tgt, shard, T16, 2.0))   # ONE collective per iteration

    T = registration.icp_gn_solve(evaluator, 12, guess)
    np.savez(os.path.join(out_dir, "icp_rank%d.npz" % rank), T=T, calls=n_calls[0])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_icp_two_ranks_gloo(scan_pair, oracle, tmp_path):
    """ICP over the exchange step ("ICP shards identically"): the source sharded over two ranks, the target replicated,
    the iteration's 32 sums all-reduced over gloo, the product's host step (lsa_icp_gn_solve: the code icp.hip runs between two
    iterations) on every rank.  Without a GPU the per-shard sums come from the oracle; under test are the sharding, the wiring and
    that both ranks take the identical step."""
    import torch.multiprocessing as mp
    import oracle_py
    from lidar_slam_amd import registration, synth

    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0[::3])
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    guess = synth.pose_matrix(0.8, 0.03, 0.0, 0.0, 0.0, 0.015).astype(np.float32)
    # one process, the whole cloud, through the same driver
    T_one = registration.icp_gn_solve(lambda T16: oracle_py.icp_sums(tgt, src, T16, 2.0), 12, guess)
    # ... which is ICPRegistrationManual up to the f64 accumulation of H and B (the reference sums in f32)
    T_ref = oracle_py.icp_manual(tgt, src, guess, 2.0, 12)
    assert np.max(np.abs(T_one - T_ref)) <= 2e-4
    port = _free_port()
    mp.spawn(_icp_worker, args=(2, port, tgt, src, guess, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "icp_rank0.npz"), np.load(tmp_path / "icp_rank1.npz")
    assert np.array_equal(r0["T"], r1["T"])                        # every rank took the identical steps: no broadcast needed
    assert int(r0["calls"]) == int(r1["calls"]) == 12              # one collective per iteration
    assert np.max(np.abs(r0["T"] - T_one)) <= 1e-6                 # only the grouping of the f64 sums differs


def test_shard_pays_break_even():
    """The shard-or-not rule every rank evaluates (dist.shard_pays): KITTI-sized filtered scans stay unsharded, cfg5-sized
    sources are sharded; the break-even shrinks with the rank count and the rule is monotone in the point count."""
    from lidar_slam_amd import dist as lsa_dist
    assert not lsa_dist.shard_pays(7512, 1) and not lsa_dist.shard_pays(10 ** 7, 1)
    for world in (2, 4, 8):
        assert not lsa_dist.shard_pays(7512, world)          # bench.py's cfg2 source
        assert not lsa_dist.shard_pays(132465, world)        # cfg4's 132 k-point source: 22 us of kernel vs 28 + 10 us of exchange
        assert lsa_dist.shard_pays(2097152, world)           # cfg5
    be = [min(m for m in range(10000, 1000000, 1000) if lsa_dist.shard_pays(m, w)) for w in (2, 4, 8)]
    assert be[0] > be[1] > be[2] and 350000 < be[2] < 450000
