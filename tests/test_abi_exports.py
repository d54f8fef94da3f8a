"""CPU: the C-ABI library loads and exports every function declared in include/ani_hip.h (no compute calls)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    from lammps_ani_amd import ani_hip
    ani_hip.build()
    hdr = open(os.path.join(ROOT, "include", "ani_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ani_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = C.CDLL(ani_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ani_hip.h but not exported"
    assert set(ani_hip.EXPORTS) == set(declared)


def test_every_header_under_include_is_exported():
    """include/ani_md.h (the timestep loop's own kernels: not the drop-in boundary, but a C ABI of the same library)."""
    from lammps_ani_amd import ani_hip
    ani_hip.build()
    lib = C.CDLL(ani_hip.LIB_PATH)
    inc = os.path.join(ROOT, "include")
    seen = 0
    for name in sorted(os.listdir(inc)):
        if not name.endswith(".h"):
            continue
        hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, name)).read(), flags=re.S)
        for sym in sorted(set(re.findall(r"\b(ani_[a-z_0-9]+)\s*\(", hdr))):
            assert hasattr(lib, sym), f"{sym} declared in include/{name} but not exported"
            seen += 1
    assert seen >= 28


def test_create_fails_loudly_without_gpu_or_for_cpu_device(tmp_path):
    """No GPU in the build container: ani_create must return an error, never fall back."""
    import torch
    from lammps_ani_amd import ani_hip, model_file as mf
    p = str(tmp_path / "t.anim")
    mf.write_model(p, mf.synthetic_model("tiny", 1, 1))
    lib = ani_hip.lib()
    h = C.c_void_p()
    rc = lib.ani_create(p.encode(), -1, -1, 1, 1, 1, C.byref(h))
    assert rc != 0 and not h.value and b"cpu" in lib.ani_last_error(None)
    if not torch.cuda.is_available():
        rc = lib.ani_create(p.encode(), 0, -1, 1, 1, 1, C.byref(h))
        assert rc != 0 and not h.value and b"HIP device" in lib.ani_last_error(None)
