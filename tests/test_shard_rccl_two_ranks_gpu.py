"""The C transport (csrc/shard_rccl.cpp) with TWO communicator ranks.  A lease has one GPU and RCCL refuses two ranks on
one device, so the ranks are two threads of one process and the nine nccl* entry points come from an in-process stand-in
(tests/stubs/nccl_stub.cpp, built here, loaded through lrzgpu_rccl_use_library): what runs is the transport's own code --
its staging pieces and events, the pairing of sends and receives, the ragged last piece, an all-reduce between two ranks,
the whole sharded entry point with rank 0 laying the file out from what rank 1 hands over, and a peer's failure ending
the other rank's pending call.  Each scenario in a fresh process (the library picks its RCCL once)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stub") / "libnccl_stub.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", out,
                    os.path.join(ROOT, "tests", "stubs", "nccl_stub.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"], check=True)
    return out


@pytest.mark.parametrize("scenario", ["exchange", "sharded", "peer_fails"])
def test_two_ranks(B, stub, scenario):
    # (the child shares the GPU with this process: what earlier tests left parked here goes back first)
    B.lib().lrzgpu_trim()
    try:
        import torch
        torch.cuda.empty_cache()
    except Exception:
        pass
    env = dict(os.environ)
    if scenario == "peer_fails":
        env["NCCL_STUB_FAIL_SEND"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_two_ranks_script.py"), stub, scenario], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and ("TWO-RANKS-OK " + scenario) in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
