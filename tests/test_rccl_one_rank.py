"""First contact with RCCL on one card: a single rank drives the SEVERAL-rank code paths.

RCCL refuses two ranks on one device, and the GPU box has one; what one card can prove is that librccl loads beside the
library, that collectives and grouped ncclSend / ncclRecv enqueue on the compute stream in order with the kernels around
them, and that `HSA_ENABLE_IPC_MODE_LEGACY=0` is harmless.  Two transports, both with one participant:

  * torch.distributed backend "nccl", world_size 1, `force_collectives=True`: DomainComm / VerletRun call
    all_to_all_single with device tensors and uneven splits, the device all-reduce (MAX) of the displacement check and
    the barrier — the branch a multi-GPU run takes (lammps-ani_amd/comm.py, md.py);
  * the native exchange of include/ani_comm.h (`ani_hip.NativeComm`): ncclCommInitRank with one rank, the rank's own
    chunk sent THROUGH ncclSend / ncclRecv (option self_through_rccl) and as a device copy.

In every case the MD trajectory (with re-neighbourings) must equal the plain single-rank loop's, whose ghost exchange is
two copy kernels.  Reference counterpart: comm->reverse_comm(this), src/pair_ani.cpp:197-201,461-484.
"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NATOMS, STEPS, DT, VSIGMA = 1536, 60, 0.25, 0.03


def _worker(rank, port, model_path, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    import torch
    import torch.distributed as dist
    from lammps_ani_amd import ani_hip, comm, md, harness as hx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    dist.barrier()
    assert float(one) == 1.0
    sysm = hx.spatial_sort(hx.water_box(NATOMS))
    L = sysm.boxhi - sysm.boxlo
    inp = hx.decompose(sysm, comm.grid_for(1), 0)
    table = np.random.default_rng(99).normal(0.0, VSIGMA, size=(sysm.natoms, 3))
    res = {}

    def trajectory(name, **kw):
        ani = ani_hip.ANI(model_path, 0)
        run = md.VerletRun(ani, inp, L, dev, dt=DT, box_lo=sysm.boxlo, **kw)
        run.v = torch.as_tensor(table[run.tag.cpu().numpy()], dtype=torch.float64, device=dev)
        e = [run.potential_energy() + run.kinetic_energy()]
        for _ in range(STEPS):
            run.step()
            e.append(run.potential_energy() + run.kinetic_energy())
        x = np.zeros((NATOMS, 3))
        x[run.tag.cpu().numpy()] = run.x[: run.nlocal].cpu().numpy()
        res[name + "_e"], res[name + "_x"], res[name + "_builds"] = np.array(e), x, run.nbuilds
        ani.close()

    trajectory("plain")
    trajectory("torch_nccl", force_collectives=True)
    trajectory("torch_nccl_overlap", force_collectives=True, overlap=True)
    nat = ani_hip.NativeComm.from_torch(0)
    assert (nat.world, nat.rank) == (1, 0)
    trajectory("native_copy", force_collectives=True, native_comm=nat)
    nat.set_option("self_through_rccl", 1)
    trajectory("native_rccl", force_collectives=True, native_comm=nat)
    trajectory("native_rccl_overlap", force_collectives=True, native_comm=nat, overlap=True)

    # the pieces on their own, own chunk through RCCL: counts, a byte all-to-all, forward / reverse against torch indexing
    st = torch.cuda.current_stream().cuda_stream
    assert nat.exchange_counts([37], stream=st) == [37]
    a = torch.arange(37 * 5, dtype=torch.int32, device=dev).reshape(37, 5)
    b = torch.zeros_like(a)
    nat.alltoallv(a.data_ptr(), [37], b.data_ptr(), [37], 20, stream=st)
    torch.cuda.synchronize()
    res["a2a_equal"] = bool(torch.equal(a, b))
    nl, ng = 500, 120
    g = torch.Generator(device="cpu").manual_seed(3)
    idx = torch.randint(0, nl, (ng,), generator=g).to(dev)
    shift = torch.randn((ng, 3), generator=g, dtype=torch.float64).to(dev)
    x = torch.randn((nl + ng, 3), generator=g, dtype=torch.float64).to(dev)
    f = torch.randn((nl + ng, 3), generator=g, dtype=torch.float64).to(dev)
    nat.set_epoch([ng], [ng], idx, shift)
    x_ref = x.clone()
    x_ref[nl:] = x[:nl][idx] + shift
    f_ref = f.clone()
    f_ref[:nl].index_add_(0, idx, f[nl:])
    nat.forward(x.data_ptr(), nl, stream=st)
    nat.reverse(f.data_ptr(), nl, stream=st)
    t = torch.tensor([1.5, -2.0, 7.25], dtype=torch.float64, device=dev)
    nat.allreduce(t.data_ptr(), 3, "max", stream=st)
    torch.cuda.synchronize()
    res["fwd_err"] = (x - x_ref).abs().max().item()
    res["rev_err"] = (f[:nl] - f_ref[:nl]).abs().max().item()
    res["allreduce"] = t.cpu().numpy()
    # ghosts in another order than the messages (LAMMPS' swap order): host maps, ani_comm_set_epoch_host + ghost order
    import ctypes as C
    perm = torch.randperm(ng, generator=g)                     # message slot k stands for ghost perm[k]
    idx_h = idx.cpu().numpy().astype(np.int64)
    shift_h = np.ascontiguousarray(shift.cpu().numpy())
    cnt = np.array([ng], dtype=np.int64)
    perm_h = perm.numpy().astype(np.int64)
    lib = ani_hip.lib()
    assert lib.ani_comm_set_epoch_host(nat._h, cnt.ctypes.data, cnt.ctypes.data, idx_h.ctypes.data, shift_h.ctypes.data,
                                       perm_h.ctypes.data) == 0
    x2 = torch.randn((nl + ng, 3), generator=g, dtype=torch.float64).to(dev)
    f2 = torch.randn((nl + ng, 3), generator=g, dtype=torch.float64).to(dev)
    x2_ref = x2.clone()
    x2_ref[nl + perm.to(dev)] = x2[:nl][idx] + shift
    f2_ref = f2.clone()
    f2_ref[:nl].index_add_(0, idx, f2[nl + perm.to(dev)])
    nat.forward(x2.data_ptr(), nl, stream=st)
    nat.reverse(f2.data_ptr(), nl, stream=st)
    torch.cuda.synchronize()
    res["fwd_err_perm"] = (x2 - x2_ref).abs().max().item()
    res["rev_err_perm"] = (f2[:nl] - f2_ref[:nl]).abs().max().item()
    nat.close()
    np.savez(os.path.join(out_dir, "out.npz"), **res)
    dist.destroy_process_group()


def test_one_rank_drives_the_multi_rank_paths_over_rccl(tmp_path):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import model_file as mf
    path = str(tmp_path / "gentle.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    port = 29500 + (os.getpid() % 2000) + 91
    mp.spawn(_worker, args=(port, path, str(tmp_path)), nprocs=1, join=True)
    d = np.load(tmp_path / "out.npz")
    assert int(d["plain_builds"]) >= 2          # the loop re-neighboured (exchange + borders ran through the collectives)
    for name in ("torch_nccl", "torch_nccl_overlap", "native_copy", "native_rccl", "native_rccl_overlap"):
        assert int(d[name + "_builds"]) == int(d["plain_builds"]), name
        # the same atoms, forces differing by the order of fp32 atomics only
        assert np.abs(d[name + "_e"] - d["plain_e"]).max() < 5e-3, name
        assert np.abs(d[name + "_x"] - d["plain_x"]).max() < 1e-4, name
    assert bool(d["a2a_equal"])
    assert float(d["fwd_err"]) == 0.0
    assert float(d["rev_err"]) < 1e-12
    assert d["allreduce"].tolist() == [1.5, -2.0, 7.25]
    assert float(d["fwd_err_perm"]) == 0.0
    assert float(d["rev_err_perm"]) < 1e-12
