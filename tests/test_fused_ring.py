"""CPU: the weight slots of the fused MLP kernel, replayed on the host.

tests/ring_sim.cpp includes lammps-ani_amd/csrc/ani_fused_ring.h -- the very bookkeeping functions the HIP kernel runs --
and walks the kernel's boundary sequence for every compiled shape, both arithmetics, AEV widths from 16 to 1024 columns
and one to three ensemble members: every boundary must find the slab it expects (size and slot) already issued, and no
load may write a slot whose slab a wave may still be reading (a boundary can stand in front of the last block of the slab
before)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ring_schedule_never_reads_early_and_never_overwrites_live_slabs(tmp_path):
    exe = str(tmp_path / "ring_sim")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "lammps-ani_amd", "csrc"),
                    os.path.join(ROOT, "tests", "ring_sim.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 failing configurations" in r.stdout
