"""The checker itself under AddressSanitizer + UndefinedBehaviorSanitizer ("run host tests under
ASan/UBSan"): every parity claim hangs on oracle/*.c -- 1 500 lines of C with hand-rolled realloc trees and index
arithmetic -- so the CPU tests that exercise it are run once more against `make -C oracle asan`
(-fsanitize=address,undefined -fno-sanitize-recover=undefined) in a child interpreter with libasan preloaded.  CPU
container only: never on the GPU box (its `-m gpu` run does not collect this file's test)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITE = ["tests/test_oracle_icp.py", "tests/test_oracle_grid.py", "tests/test_oracle_solves.py", "tests/test_gseg.py",
         "tests/test_ccicp.py", "tests/test_multi_rank.py"]


def test_cpu_suite_against_the_sanitized_oracle():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "no libasan in this toolchain: %r" % asan
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update({"SLAM_ORACLE_SANITIZED": "1", "LD_PRELOAD": asan, "OMP_NUM_THREADS": "1",
                # python itself leaks by design and numpy's allocator is not instrumented: leaks off, everything else fatal
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + SUITE,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout[-3000:] + "\n" + p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, tail
    assert p.returncode == 0, tail
    assert " passed" in p.stdout, tail
