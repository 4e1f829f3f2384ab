"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only -- GPU ASan is
not available on this pool).  `make -C oracle asan` builds oracle/_build/libnerf_oracle_asan.so; a child process with the
ASan runtime preloaded runs the whole oracle-vs-golden suite against it.  Any out-of-bounds access, use of an
uninitialised shift / overflow in the oracle ends the child with a sanitizer report."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "_build", "libnerf_oracle_asan.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(asan_rt) and os.path.exists(asan_rt), "no ASan runtime next to gcc"
    env = dict(os.environ, NERF_ORACLE_SO=so, LD_PRELOAD=asan_rt,
               # python itself leaks by design; the interpreter's own allocator tricks are not the subject
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider",
                          # (the 20 000-row gradient case runs the same functions as test_end_to_end / test_mlp_forward_backward
                          # on more rows: 110 s under the sanitizers, and the CPU suite is to stay within minutes)
                          "--deselect", "tests/test_oracle_golden.py::test_end_to_end_gradients_with_the_references_relu_decisions"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
