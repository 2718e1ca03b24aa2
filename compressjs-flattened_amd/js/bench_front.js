'use strict';
/* bench_front.js — times the real boundary: Bzip2.compressFile of this package's front on a Uint8Array, under Node.
 * usage: node bench_front.js <input file> <level> <reps>   -> one JSON line {median_ms, out_len, out_sha256, node, reps}
 * (bench.py's `e2e_js_front` leg; SURVEY.md §8(d) "end-to-end from the JS Uint8Array") */
var fs = require('fs');
var crypto = require('crypto');
var cjs = require('./index.js');

var file = process.argv[2], level = parseInt(process.argv[3] || '9', 10), reps = parseInt(process.argv[4] || '5', 10);
var buf = fs.readFileSync(file);
var input = new Uint8Array(buf.buffer, buf.byteOffset, buf.length);
var out = cjs.Bzip2.compressFile(input, null, level);          // warm-up: workspace, pinned result buffers
function median(a) { a.sort(function (x, y) { return x - y; }); return a.length % 2 ? a[(a.length - 1) / 2] : (a[a.length / 2 - 1] + a[a.length / 2]) / 2; }
function series(collect) {
  var times = [];
  for (var i = 0; i < reps; i++) {
    if (collect) { out = null; global.gc(); }                  // (outside the timed call)
    var t0 = process.hrtime();
    out = cjs.Bzip2.compressFile(input, null, level);
    var dt = process.hrtime(t0);
    times.push(dt[0] * 1e3 + dt[1] / 1e6);
  }
  return median(times);
}
// 1. calls back to back in one synchronous stretch of JavaScript: V8 finalises the dropped results late, so every call pins a fresh
//    result buffer (hipHostMalloc of ~1/3 of the input) -- the worst case for the library's pool of pinned result buffers
var sync_ms = series(false);
// 2. the dropped result collected between the calls (node --expose-gc): its pinned buffer is back in the pool for the next result
var gc_ms = typeof global.gc === 'function' ? series(true) : null;
var sha = crypto.createHash('sha256').update(Buffer.from(out.buffer, out.byteOffset, out.length)).digest('hex');
console.log(JSON.stringify({ median_ms: sync_ms, median_ms_results_collected: gc_ms, out_len: out.length, out_sha256: sha, node: process.version, reps: reps }));
