"""Host side of the C ABI under AddressSanitizer / UndefinedBehaviorSanitizer / LeakSanitizer: a CPU-only test (no kernel
is launched).  Kept in its own file and listed in .gpurunignore: the GPU pool refuses any call whose snapshot contains a
sanitizer build line (GPU sanitizers are not available there), and the GPU run does not need this file."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_host_glue_under_address_and_ub_sanitizers(tmp_path):
    """The host side of every csrc/*.hip -- argument validation, handle creation / destruction, error reporting --
    compiled with AddressSanitizer + UndefinedBehaviorSanitizer (+ LeakSanitizer at exit) and driven by
    tests/abi_probe_sanitized.c.  Device code is compiled as usual but not instrumented (-fno-gpu-sanitize: GPU
    sanitizers are not available on this pool); the probe launches no kernel and needs no GPU."""
    from pita_amd import build as _b

    flags = ["--offload-arch=gfx950", "-O1", "-g0", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize"]
    procs, objs = [], []
    for s in _b.SOURCES:
        obj = tmp_path / s.replace(".hip", ".o")
        procs.append((s, subprocess.Popen([_b.HIPCC, *flags, "-c", os.path.join(_b.CSRC, s), "-o", str(obj)],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    for s, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, f"{s}: {out[-2000:]}"
    lib = tmp_path / "libpita_hip_asan.so"
    subprocess.check_call([_b.HIPCC, "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", str(lib), *objs])
    nm = subprocess.check_output(["nm", "-D", str(lib)], text=True)
    assert "__asan_init" in nm and "__ubsan_handle" in nm, "the host objects are not instrumented"
    exe = tmp_path / "abi_probe_sanitized"
    clang = os.path.join(os.path.dirname(os.path.realpath(_b.HIPCC)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    subprocess.check_call([clang, "-std=c99", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_probe_sanitized.c"),
                           "-o", str(exe), "-L", str(tmp_path), "-lpita_hip_asan", f"-Wl,-rpath,{tmp_path}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "sanitized abi ok" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr and \
        "LeakSanitizer" not in out.stderr, out.stderr[-4000:]
