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
function timed() {
  var t0 = process.hrtime();
  out = cjs.Bzip2.compressFile(input, null, level);
  var dt = process.hrtime(t0);
  return dt[0] * 1e3 + dt[1] / 1e6;
}
// 1. calls back to back in one synchronous stretch of JavaScript: N-API finalizers of dropped results only run when control is back
//    in the event loop, so every call pins a fresh result buffer (hipHostMalloc of ~1/3 of the input) -- the worst case for the
//    library's pool of pinned result buffers
var sync = [];
for (var i = 0; i < reps; i++) { sync.push(timed()); }
var sync_ms = median(sync);
// 2. one call per turn of the event loop (how a server calls it), the dropped result collected in between (node --expose-gc): its
//    pinned buffer is back in the pool for the next result
var turns = [];
function turn(k) {
  if (k === reps + 1) { return finish(); }
  out = null;
  if (typeof global.gc === 'function') { global.gc(); }
  setImmediate(function () { var ms = timed(); if (k > 0) { turns.push(ms); } setImmediate(function () { turn(k + 1); }); });
}
function finish() {
  var sha = crypto.createHash('sha256').update(Buffer.from(out.buffer, out.byteOffset, out.length)).digest('hex');
  console.log(JSON.stringify({ median_ms: sync_ms, median_ms_results_collected: median(turns), out_len: out.length, out_sha256: sha, node: process.version, reps: reps }));
}
turn(0);
