"""The Julia shim's view of libnlls_amd.so, pinned from outside Python: tests/abi/abi_replay.c (plain C, gcc) static_asserts the
struct layouts NLLSsolverAMD.jl mirrors by hand, resolves every symbol it ccalls and -- on the GPU -- replays the exact call
sequence and argument types of its device-resident Levenberg-Marquardt loop on a small noise-free bundle adjustment."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi", "abi_replay.c")
EXE = os.path.join(ROOT, "tests", "abi", "abi_replay.out")
LIB = os.path.join(ROOT, "nllssolver.jl_amd", "csrc", "libnlls_amd.so")


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "nlls_amd.h"))):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-o", EXE, SRC, "-ldl", "-lm"])
    return EXE


def test_shim_struct_layouts_and_symbols():
    out = subprocess.run([_build(), "--layout", LIB], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "sizeof(nlls_cost_group)=64 sizeof(nlls_info)=112 offsetof(ndof)=24" in out.stdout


@pytest.mark.gpu
def test_shim_call_sequence_replay():
    out = subprocess.run([_build(), "--replay", LIB], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "replay: start" in out.stdout
    # the generic loop of the non-LM iterators: host vectors advance through the shim's updatefromnext! (advisor, round 4)
    assert "replay (Newton through the generic loop" in out.stdout


@pytest.mark.gpu
def test_collectives_from_plain_c():
    """nlls_comm_unique_id / nlls_comm_init_rccl / nlls_lm_iterations from a C program: no Python, no PyTorch in the process -- the library loads
    librccl.so.1 itself, and one rank through the collective route takes the same iterations as the single-GPU loop."""
    out = subprocess.run([_build(), "--replay", LIB, "--rccl"], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "collective route over the library's own RCCL communicator" in out.stdout
