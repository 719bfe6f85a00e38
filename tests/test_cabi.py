"""CPU: the C-ABI library loads and exports every symbol include/stegan_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "stegan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(stegan_\w+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from ste_gan_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    for kind, path in _lib.LIB_PATHS.items():                  # the bf16 and the fp16 build of the same sources
        handle = ctypes.CDLL(path)
        for s in syms:
            assert hasattr(handle, s), f"{s} declared in include/stegan_hip.h but not exported by {os.path.basename(path)}"
        handle.stegan_operand_type.restype = ctypes.c_char_p
        assert handle.stegan_operand_type().decode() == kind
    bound = set(_lib.SIGNATURES) | set(_lib.INT64_RETURN)
    assert set(syms) == bound, f"ctypes binding out of sync with the header: {set(syms) ^ bound}"


def test_every_exported_entry_point_is_declared():
    """The other direction: nothing is exported (extern "C" stegan_*) that include/stegan_hip.h does not declare."""
    import shutil
    import subprocess
    from ste_gan_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    for path in _lib.LIB_PATHS.values():
        out = subprocess.run([nm, "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("stegan_")}
        assert exported, "no stegan_* symbols found"
        undeclared = exported - set(declared_symbols())
        assert not undeclared, f"{os.path.basename(path)} exports what include/stegan_hip.h does not declare: {sorted(undeclared)}"


def test_library_loads_and_selftest():
    from ste_gan_amd import _lib
    h = _lib.lib()
    assert h.stegan_abi_version() == _lib.ABI_VERSION
    assert h.stegan_selftest_host() == 0          # host-side index helpers (magic division)


def test_invalid_descriptor_is_rejected_without_gpu():
    """Argument validation happens before anything is enqueued, so it can be exercised on the CPU."""
    from ste_gan_amd import _lib
    d = _lib.ConvDesc(1, 8, 8, 3, 3, 1, 1, 1, 16, 16, 1, 0, 0, 0)      # groups=3 does not divide 8 channels
    assert _lib.lib().stegan_conv1d_fwd(ctypes.byref(d), None, None, None, None, None, None, None) == -1
    tile, splits, macs = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    d = _lib.ConvDesc(64, 192, 192, 1, 3, 1, 27, 27, 192, 192, 1, 1, 0, 0)
    assert _lib.lib().stegan_conv1d_plan(ctypes.byref(d), 0, ctypes.byref(tile), ctypes.byref(splits), ctypes.byref(macs)) == 0
    assert macs.value == 64 * 192 * 192 * 192 * 3
    d = _lib.ConvDesc(64, 768, 768, 1, 3, 1, 27, 27, 12, 12, 1, 1, 0, 0)  # d=27 at T=12: only the centre tap is live
    assert _lib.lib().stegan_conv1d_plan(ctypes.byref(d), 0, ctypes.byref(tile), ctypes.byref(splits), ctypes.byref(macs)) == 0
    assert macs.value == 64 * 12 * 768 * 768 * 1
