"""C-ABI checks that need no GPU: the library loads, exports every symbol include/cjs_hip.h declares,
and fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import importlib
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "compressjs-flattened_amd")
LIB = os.path.join(PKG, "libcjs_hip.so")


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cjs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cjs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run `python __graft_entry__.py` first"
    lib = ctypes.CDLL(LIB)
    names = _declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "symbols declared in include/cjs_hip.h but not exported: %s" % missing


def test_strerror_and_version():
    lib = ctypes.CDLL(LIB)
    lib.cjs_strerror.restype = ctypes.c_char_p
    lib.cjs_version.restype = ctypes.c_char_p
    assert lib.cjs_strerror(-2) == b"Not bzip data"
    assert lib.cjs_strerror(-5) == b"Data error"
    assert lib.cjs_strerror(-20) == b"Invalid block size multiplier"
    assert lib.cjs_strerror(-21) == b"Bad magic"
    assert b"gfx950" in lib.cjs_version()


def test_python_front_has_reference_surface():
    import sys
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("compressjs-flattened_amd")
    for obj in (pkg.Bzip2, pkg.BWTC):
        assert callable(obj.compressFile) and callable(obj.decompressFile)
    assert pkg.BWTC.MAGIC == "bwtc"
    with pytest.raises(pkg.CjsError) as e:            # Q17: outside 1..9 throws before touching the device
        pkg.Bzip2.compressFile(b"abc", None, 10)
    assert e.value.errorCode == -20


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sys
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("compressjs-flattened_amd")
    with pytest.raises(pkg.CjsError) as e:
        pkg.Bzip2.compressFile(np.arange(100, dtype=np.uint8), None, 9)
    assert e.value.errorCode == -30          # CJS_E_NO_DEVICE: the product never routes through a CPU path


def test_free_and_trim_need_no_device():
    # cjs_free takes what the library handed out (plain malloc or a pinned result buffer) and, like free(), a null pointer;
    # memory it does not know is plain malloc'd memory.  cjs_trim with nothing cached is a no-op.  Neither needs a GPU.
    lib = ctypes.CDLL(LIB)
    lib.cjs_free.argtypes = [ctypes.c_void_p]
    lib.cjs_free.restype = None
    lib.cjs_free(None)
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    p = libc.malloc(4096)
    assert p
    lib.cjs_free(ctypes.c_void_p(p))
    lib.cjs_trim.restype = None
    lib.cjs_trim()


def test_product_does_not_link_the_oracle():
    out = subprocess.run(["nm", "-D", LIB], capture_output=True, text=True).stdout
    assert "cjs_oracle" not in out
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".hip", ".h", ".hpp", ".cc", ".js", ".py")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "cjs_oracle" not in text and "libcjs_oracle" not in text, "%s references the oracle" % f


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_addon_loads_and_mirrors_the_reference_api():
    addon = os.path.join(PKG, "js", "cjs_napi.node")
    assert os.path.exists(addon), "N-API addon not built"
    script = r"""
      const m = require(process.argv[1]);
      const r = { version: m.native().version(), devices: m.native().deviceCount(),
                  bz: Object.keys(m.Bzip2).sort(), bw: Object.keys(m.BWTC).sort(), magic: m.BWTC.MAGIC };
      try { m.Bzip2.compressFile(new Uint8Array(4), null, 0); } catch (e) { r.level = e.message; }
      try { m.Bzip2.compressFile(new Uint8Array(4)); r.compress = 'ok'; } catch (e) { r.compress = e.message; r.code = e.errorCode; }
      console.log(JSON.stringify(r));
    """
    out = subprocess.run(["node", "-e", script, os.path.join(PKG, "js", "index.js")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert "compressFile" in r["bz"] and "decompressFile" in r["bz"]
    assert "compressFile" in r["bw"] and "decompressFile" in r["bw"] and r["magic"] == "bwtc"
    assert r["level"] == "Invalid block size multiplier"
    if r["devices"] == 0:
        assert r["code"] == -30, r


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_cli_checks_and_messages_of_the_reference():
    # the argument checks of NPM/bin/compressjs:31-58 with their texts; without a GPU a real call fails loudly (no fallback)
    cli = os.path.join(PKG, "js", "cli.js")

    def run(*args, data=b""):
        return subprocess.run(["node", cli] + list(args), input=data, capture_output=True, timeout=60)
    assert b"Can't specify both -9 and -1" in run("-z", "-9", "-1", "-t", "bzip2").stderr
    assert b"Compression level has no effect when decompressing." in run("-d", "-3", "-t", "bwtc").stderr
    assert b"Must specify either -d or -z." in run("-d", "-z", "-t", "bzip2").stderr
    assert b"--block can only be used with decompression" in run("-z", "-b", "32", "-t", "bzip2").stderr
    assert b"Unknown compressor: lzp3" in run("-z", "-t", "lzp3").stderr
    assert run("-z", "-9", "-1", "-t", "bzip2").returncode == 1
    import torch
    if not torch.cuda.is_available():
        out = run("-z", "-t", "bzip2", data=b"abc")
        assert out.returncode == 1 and b"no HIP device" in out.stderr
