"""CPU: the host side of libsbtv under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C csrc sanitize` compiles every translation unit with -fsanitize=address,undefined on the HOST side (the device
code is compiled as usual: GPU sanitizers are not available on the target pool) and links tests/host_sanitize.cpp, which
drives the host-only entry points: PSF tap builders and err_psf at the smallest / largest sizes, option defaults, argument
validation without a context, the no-device error paths of sbtv_ctx_create / sbtv_group_create (every half-built piece is
released: LeakSanitizer is part of ASan), the switches report into tiny buffers.  Never run on the GPU box."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd", "csrc")


def test_host_entries_are_clean_under_asan_ubsan():
    subprocess.run(["make", "-C", CSRC, "-j", "4", "sanitize"], check=True, capture_output=True, timeout=1500)
    exe = os.path.join(os.path.dirname(CSRC), "lib", "host_sanitize")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "all checks passed" in r.stdout
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]
