"""CPU: the product's host-only code (Newton / More-Thuente driver, the scalar math it shares with the device loop, point
packing, the file readers) and the whole CPU oracle built with -fsanitize=address,undefined and driven through the same
control flow the GPU tests walk (tests/cpp/test_sanitize.cpp).  GPU sanitizers are not available; this is the
reference's "race detection / sanitizers" row for everything that runs on the host."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_and_oracle_under_asan_ubsan():
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-s", "-C", d, "asan"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(d, "build", "test_sanitize")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "sanitize ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
