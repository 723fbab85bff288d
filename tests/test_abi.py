"""The C-ABI library loads without a GPU and exports every symbol include/unet_hip.h declares, and the ctypes
table of unet_amd/_lib.py lists exactly those symbols with matching argument counts.  No kernel is launched."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(uh_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_symbols_are_exported_and_bound():
    from unet_amd import _lib
    decl = declared()
    assert len(decl) >= 50
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in unet_hip.h but not exported by libunet_hip.so"
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, n in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} arguments, ctypes table {len(_lib.SIGNATURES[name][1])}"


def test_host_only_entry_points():
    from unet_amd import _lib
    lib = _lib.lib
    assert lib.uh_version() >= 1
    assert lib.uh_guard_rows(256) >= 256 + 2 + 512 + 16
    assert lib.uh_conv_gemm_mtiles(32, 256, 256, 64) == (32 * 257 * 257 + 511) // 512
    assert lib.uh_conv_gemm_mtiles(32, 256, 256, 128) == (32 * 257 * 257 + 255) // 256
    s = lib.uh_conv_wgrad_splits(32, 256, 256, 64, 64, 9)
    assert 1 <= s <= 512 and lib.uh_conv_wgrad_workspace(32, 256, 256, 64, 64, 9) == s * 64 * 9 * 64 * 4
    n = 32 * 256 * 256
    assert lib.uh_cldice_workspace(32, 256, 256) >= 27 * n * 4
    # argument checking happens on the host before any launch
    assert lib.uh_conv_gemm(1, 4, 4, None, 64, 0, 48, None, 64, 9, 0, None, 64, 0, None, None, None, None, None, None) == -1
    assert b"multiple of 64" in lib.uh_last_error()


def test_product_path_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under unet_amd/ may import it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "unet_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_host_side_of_the_c_abi_under_address_sanitizer():
    """`make asan` builds the library with its HOST code instrumented by AddressSanitizer (device code as usual) and a
    plain C++ driver (tests/asan_driver.cpp) that walks the host paths no GPU is needed for: size and count functions,
    argument validation of the launching entry points, the thread-local error buffer from several threads."""
    import subprocess
    csrc = os.path.join(ROOT, "unet_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan", "-j8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([os.path.join(ROOT, "tests", "asan_driver")], capture_output=True, text=True, timeout=300, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN WALK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
