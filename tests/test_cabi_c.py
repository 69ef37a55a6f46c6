"""The C ABI from plain C: tests/cabi/cabi_consumer.c is compiled with gcc against include/emx.h and libemx.so -- no torch,
no C++ -- and run as its own process.  CPU: the symbols resolve, version / target answer, argument validation rejects with a
message.  GPU: known answers of the reference's tests (tests/test_pairwise_distances.py:140-153, tests/test_losses.py:815-907)
through the ABI with device memory from the HIP runtime API, and emx_mlp_infer (encode(): autoencoder.py:1110-1136) on a narrow and
on a wide-input encoder against the same layers in double."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_consumer.c")


def _build(tmp_path):
    import __graft_entry__ as g

    g.build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if shutil.which("gcc") is None or not os.path.isdir(os.path.join(rocm, "include", "hip")):
        pytest.skip("gcc or the HIP headers are missing")
    exe = str(tmp_path / "cabi_consumer")
    libdir = os.path.join(ROOT, "encodermap_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), SRC, "-o", exe,
           "-L" + libdir, "-l:libemx.so", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_consumer_links_and_validates(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "target gfx950" in r.stdout and "ok" in r.stdout


@pytest.mark.gpu
def test_c_consumer_known_answers_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
