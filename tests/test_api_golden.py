"""Boundary behaviour pinned by tests/golden/golden_api.json (cut from the reference JS by make_golden.js api):
error class / message / errorCode for damaged .bz2 input (J/Bzip2_joined_.js:1385-1391 and the _throw call sites),
the size field BWTC writes for a stream input without .size (SURVEY W1, J/BWTC_joined_.js:529-543), and a multistream
file whose members change level (J/Bzip2_joined_.js:1787-1792).

CPU part: the oracle agrees with the reference on every code / stream.  GPU part: the HIP library through the C ABI
(code + cjs_last_error_detail) and the JS fronts under Node reproduce the reference's messages verbatim.
"""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import recipes
import support

G = support.load_golden("golden_api.json")
MSG = {-2: "Not bzip data", -5: "Data error", -7: "Obsolete (pre 0.9.5) bzip format not supported."}


def _hex(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint8)


def _mixed_stream(compress):
    parts = G["mixed_level_multistream"]["parts"]
    datas = [recipes.build(p["recipe"]) for p in parts]
    streams = []
    for d, p in zip(datas, parts):
        rc, s = compress(d, p["level"])
        assert rc == 0
        streams.append(s)
    return np.concatenate(streams), np.concatenate(datas), datas


# ---------------------------------------------------------------- CPU: the oracle is pinned on these too
@pytest.mark.parametrize("case", G["bzip2_decode_errors"], ids=lambda c: c["name"])
def test_oracle_error_codes(oracle, case):
    rc, out = oracle.bzip2_decompress(_hex(case["input_hex"]), 1 if case["multistream"] else 0)
    if case["ok"]:
        assert rc == 0 and out.tobytes().hex() == case["out_hex"]
    else:
        assert rc == case["errorCode"]


def test_oracle_mixed_level_multistream(oracle):
    m = G["mixed_level_multistream"]
    cat, whole, datas = _mixed_stream(oracle.bzip2_compress)
    assert cat.size == m["stream_len"] and support.sha256(cat) == m["stream_sha256"]
    rc, out = oracle.bzip2_decompress(cat, 1)
    assert rc == 0 and out.size == m["multistream_out_len"] and support.sha256(out) == m["multistream_out_sha256"]
    assert np.array_equal(out, whole)
    rc, out = oracle.bzip2_decompress(cat, 0)
    assert rc == 0 and out.size == m["single_out_len"] and support.sha256(out) == m["single_out_sha256"]


# ---------------------------------------------------------------- GPU: the product
@pytest.mark.gpu
@pytest.mark.parametrize("case", G["bzip2_decode_errors"], ids=lambda c: c["name"])
def test_hip_error_code_and_detail(hip, case):
    rc, out = hip.bzip2_decompress(_hex(case["input_hex"]), 1 if case["multistream"] else 0)
    if case["ok"]:
        assert rc == 0 and out.tobytes().hex() == case["out_hex"]
        return
    assert rc == case["errorCode"]
    detail = hip.last_error_detail()
    msg = hip.L.cjs_strerror(rc).decode() + (": " + detail if detail else "")
    assert msg == case["message"]


@pytest.mark.gpu
def test_hip_mixed_level_multistream(hip):
    m = G["mixed_level_multistream"]
    cat, whole, datas = _mixed_stream(hip.bzip2_compress)
    assert cat.size == m["stream_len"] and support.sha256(cat) == m["stream_sha256"]
    rc, out = hip.bzip2_decompress(cat, 1)
    assert rc == 0 and out.size == m["multistream_out_len"] and support.sha256(out) == m["multistream_out_sha256"]
    rc, out = hip.bzip2_decompress(cat, 0)
    assert rc == 0 and out.size == m["single_out_len"] and support.sha256(out) == m["single_out_sha256"]
    # Bzip2.table walks every member as well
    rc, tab = hip.bzip2_table(cat, 1)
    assert rc == 0 and sum(sz for _, sz in tab) == whole.size


@pytest.mark.gpu
def test_hip_bwtc_size_unknown_flag(hip):
    import ctypes

    class Opts(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("n_devices", ctypes.c_uint32),
                    ("flags", ctypes.c_uint32), ("stats", ctypes.c_void_p)]
    for c in G["bwtc_stream_input"]:
        data = recipes.textgen(3000, 5) if c["input_hex"] is None else _hex(c["input_hex"])
        o = Opts(ctypes.sizeof(Opts), -1, 0, 1, None)          # CJS_FLAG_SIZE_UNKNOWN
        rc, out = hip._call_stream(hip.L.cjs_bwtc_compress, hip._free, data, c["level"], tail=(ctypes.byref(o),))
        assert rc == 0 and out.tobytes().hex() == c["no_size_hex"], (c["name"], c["level"])
        rc, out = hip.bwtc_compress(data, c["level"])
        assert rc == 0 and out.tobytes().hex() == c["array_hex"]
        rc, back = hip.bwtc_decompress(_hex(c["no_size_hex"]))
        assert rc == 0 and np.array_equal(back, data)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_fronts_reproduce_reference_messages():
    tmp = tempfile.mkdtemp()
    jf = os.path.join(tmp, "api.json")
    text = recipes.textgen(3000, 5)
    text.tofile(os.path.join(tmp, "textgen_3000_s5.bin"))
    json.dump({"golden": G, "dir": tmp}, open(jf, "w"))
    script = r"""
      const fs = require('fs'), path = require('path');
      const m = require(process.argv[1]);
      const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8')), g = job.golden;
      const rep = { errors: [], levels: [], sizeless: [], bwtc_errors: [] };
      g.bzip2_decode_errors.forEach(c => {
        const r = { name: c.name };
        try { const o = m.Bzip2.decompressFile(Buffer.from(c.input_hex, 'hex'), null, c.multistream); r.ok = true; r.out_hex = Buffer.from(o).toString('hex'); }
        catch (e) { r.ok = false; r.error_class = e.constructor.name; r.message = e.message; r.errorCode = e.errorCode === undefined ? null : e.errorCode; }
        rep.errors.push(r);
      });
      g.bzip2_level_errors.filter(c => !c.ok).forEach(c => {
        try { m.Bzip2.compressFile(Buffer.from('banana'), null, c.level); rep.levels.push({ level: c.level, ok: true }); }
        catch (e) { rep.levels.push({ level: c.level, ok: false, error_class: e.constructor.name, message: e.message }); }
      });
      const asStream = (buf, withSize) => { let pos = 0; const st = { readByte: function () { return pos < buf.length ? buf[pos++] : -1; } }; if (withSize) st.size = buf.length; return st; };
      g.bwtc_stream_input.forEach(c => {
        const data = c.input_hex === null ? fs.readFileSync(path.join(job.dir, 'textgen_3000_s5.bin')) : Buffer.from(c.input_hex, 'hex');
        const hex = b => Buffer.from(b).toString('hex');
        rep.sizeless.push({ name: c.name, level: c.level, no_size_hex: hex(m.BWTC.compressFile(asStream(data, false), null, c.level)),
                            with_size_hex: hex(m.BWTC.compressFile(asStream(data, true), null, c.level)), array_hex: hex(m.BWTC.compressFile(data, null, c.level)),
                            bzip2_no_size_hex: hex(m.Bzip2.compressFile(asStream(data, false), null, c.level)) });
      });
      g.bwtc_decode_errors.forEach(c => {
        try { m.BWTC.decompressFile(Buffer.from(c.input_hex, 'hex')); rep.bwtc_errors.push({ ok: true }); }
        catch (e) { rep.bwtc_errors.push({ ok: false, error_class: e.constructor.name, message: e.message }); }
      });
      console.log(JSON.stringify(rep));
    """
    out = subprocess.run(["node", "-e", script, os.path.join(support.PKG, "js", "index.js"), jf], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    for got, want in zip(rep["errors"], G["bzip2_decode_errors"]):
        for k in ("ok", "error_class", "message", "errorCode", "out_hex"):
            assert got.get(k) == want.get(k), (want["name"], k, got.get(k), want.get(k))
    for got, want in zip(rep["levels"], [c for c in G["bzip2_level_errors"] if not c["ok"]]):
        assert got == want
    for got, want in zip(rep["sizeless"], G["bwtc_stream_input"]):
        for k in ("no_size_hex", "with_size_hex", "array_hex", "bzip2_no_size_hex"):
            assert got[k] == want[k], (want["name"], want["level"], k)
    for got, want in zip(rep["bwtc_errors"], G["bwtc_decode_errors"]):
        assert got["ok"] == want["ok"] and got.get("error_class") == want.get("error_class") and got.get("message") == want.get("message")
