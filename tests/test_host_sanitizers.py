"""Host code of the C ABI (vqs_api.cpp, vqs_qwen.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU
needed): a C++ driver (tests/cabi/host_abi_exercise.cpp) walks everything that runs before the first device access --
handle creation and validation, workspace / packed-buffer sizing, the option setter, tap registration, named workspace
offsets, the argument checks of the pass entry points, the integer helpers -- linked against the sanitised host objects
and the ordinary device objects.  The reference has no sanitizer or race tooling; the device side is
covered by the bitwise-repeatability and stage-locked tests on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OBJ = os.path.join(ROOT, "build", "obj")

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs the ROCm compilers")


def test_host_abi_under_asan_and_ubsan():
    dev_objs = [os.path.join(OBJ, f) for f in ("gemm.hip.o", "attn.hip.o", "elementwise.hip.o", "qwen_decode.hip.o")]
    if not all(os.path.exists(o) for o in dev_objs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "t2v_metrics_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    out = os.path.join(ROOT, "build", "asan")
    os.makedirs(out, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17"]
    csrc = os.path.join(ROOT, "t2v_metrics_amd", "csrc")
    for src in ("vqs_api.cpp", "vqs_qwen.cpp"):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-fPIC", *san, "-c", os.path.join(csrc, src), "-o",
                               os.path.join(out, src + ".o")], stderr=subprocess.DEVNULL)
    subprocess.check_call([CLANG, *san, "-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "cabi", "host_abi_exercise.cpp"), "-o", os.path.join(out, "driver.o")])
    exe = os.path.join(out, "host_abi_exercise")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", os.path.join(out, "driver.o"),
                           os.path.join(out, "vqs_api.cpp.o"), os.path.join(out, "vqs_qwen.cpp.o"), *dev_objs, "-o", exe],
                          stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "ok" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
