// js_front_check.js — exercises the JS fronts (Node + N-API + HIP) and prints a JSON report;
// driven by tests/test_js_front.py.  argv[2] = JSON file listing [{path, level, algo}] inputs.
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const path = require('path');
const m = require(path.join(__dirname, '..', 'compressjs-flattened_amd', 'js', 'index.js'));
const jobs = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const report = { version: m.native().version(), devices: m.native().deviceCount(), results: [], api: {} };
jobs.forEach(j => {
  const input = fs.readFileSync(j.path);
  const out = m[j.algo].compressFile(input, null, j.level);
  report.results.push({ name: j.name, algo: j.algo, level: j.level, len: out.length,
                        sha256: crypto.createHash('sha256').update(out).digest('hex'), isU8: out instanceof Uint8Array });
});
// API semantics of the reference (SURVEY.md §8b)
const small = Buffer.from('banana');
report.api.array_input = Buffer.from(m.Bzip2.compressFile([98, 97, 110, 97, 110, 97])).toString('hex');
report.api.buffer_input = Buffer.from(m.Bzip2.compressFile(small)).toString('hex');
let pos = 0;
const stream = { readByte: function () { return pos < small.length ? small[pos++] : -1; } };
report.api.stream_input = Buffer.from(m.Bzip2.compressFile(stream)).toString('hex');
const sink = { bytes: [], writeByte: function (b) { this.bytes.push(b); } };
const ret = m.Bzip2.compressFile(small, sink, 9);
report.api.sink_returned = ret === sink;
report.api.sink_hex = Buffer.from(sink.bytes).toString('hex');
try { m.Bzip2.compressFile(small, null, 0); report.api.level0 = 'no throw'; } catch (e) { report.api.level0 = e.message; }
try { m.Bzip2.compressFile(small, new Uint8Array(3)); report.api.short_out = 'no throw'; } catch (e) { report.api.short_out = e.constructor.name + ':' + e.message; }
report.api.default_level = Buffer.from(m.Bzip2.compressFile(small, null, 'x')).toString('hex');
console.log(JSON.stringify(report));
